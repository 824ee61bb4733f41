#!/usr/bin/env python3
"""Per-layer time breakdown of one training step (HIP events per conv launch, tagged with the layer name).
  python tools/profile_step.py [--batch 8] [--arch resnet50]"""
import argparse
import collections
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import modules, prof, synth, conv as cv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    net = getattr(modules, args.arch)(num_classes=8)
    net.load_state_dict(synth.state_dict(args.arch, 8, 12, seed=2))
    net = net.to(dev).train()
    img = torch.randn(args.batch, 3, args.height, args.width, device=dev)
    ann = synth.labels_dir(args.batch, 10, args.height, args.width, 8, seed=1).to(dev)
    params = list(net.parameters())
    opt = torch.optim.Adam(params, lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        l = net([img, ann])
        sum(x.mean() for x in l).backward()
        torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
    step()
    torch.cuda.synchronize()
    # tag launches with the layer: wrap Layer methods
    from retinanet_mi355x import engine
    cur = {"tag": ""}
    orig_timed = prof.timed

    def timed(kind, work, fn):
        return orig_timed(kind + "|" + cur["tag"], work, fn)
    prof.timed = timed
    cv.prof = prof
    for name in ("fwd", "bwd_params", "bwd_data", "fwd_group", "bwd_data_group", "bwd_data_compact"):
        orig = getattr(engine.Layer, name)

        def wrap(orig=orig, name=name):
            def f(self, *a, **k):
                cur["tag"] = self.spec.name + ":" + name
                return orig(self, *a, **k)
            return f
        setattr(engine.Layer, name, wrap())
    t = prof.ACTIVE = prof.KernelTimer()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    step()
    e1.record()
    summ = t.summary()
    prof.ACTIVE = None
    total = e0.elapsed_time(e1)
    conv_total = sum(a["ms_total"] for a in summ.values())
    print("step %.1f ms, conv kernels %.1f ms" % (total, conv_total))
    rows = sorted(summ.items(), key=lambda kv: -kv[1]["ms_total"])
    print("%-58s %4s %8s %7s" % ("kernel|layer:phase", "n", "ms", "TF/s"))
    for k, a in rows:
        tf = a["work_total"] / (a["ms_total"] * 1e-3) / 1e12
        print("%-58s %4d %8.3f %7.1f" % (k, a["launches"], a["ms_total"], tf))
    # by phase
    by = collections.OrderedDict()
    for k, a in summ.items():
        ph = k.split(":")[-1] + " " + k.split("|")[0]
        b = by.setdefault(ph, [0.0, 0.0])
        b[0] += a["ms_total"]
        b[1] += a["work_total"]
    for ph, (ms, w) in by.items():
        print("%-30s %8.2f ms %7.1f TF/s" % (ph, ms, w / (ms * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
