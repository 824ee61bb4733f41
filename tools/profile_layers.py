#!/usr/bin/env python3
"""Per-layer-shape timing of one bf16 training step (HIP events around every conv launch, keyed by shape):
where the step's convolution time goes.   python tools/profile_layers.py [--steps 3]"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import modules, optim, prof, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, H, W = 8, 1080, 1920
    net = modules.resnet50(num_classes=8)
    net.load_state_dict(synth.state_dict("resnet50", 8, 12, seed=2))
    net = net.to(dev)
    if args.dtype != "fp32":
        net.set_compute_dtype(args.dtype)
    net.train()
    net.freeze_bn()
    net.use_flat_gradients()
    opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
    img = torch.randn(B, 3, H, W, device=dev)
    ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = sum(l.mean() for l in net([img, ann]))
        loss.backward()
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    prof.BY_SHAPE = True
    t = prof.ACTIVE = prof.KernelTimer()
    for _ in range(args.steps):
        step()
    rows = sorted(t.summary().items(), key=lambda kv: -kv[1]["ms_total"])
    prof.ACTIVE = None
    tot = sum(a["ms_total"] for _, a in rows) / args.steps
    print("%-64s %6s %9s %9s" % ("kernel / shape (N x Ho x Wo Cin->Cout)", "n/step", "ms/step", "TFLOP/s"))
    for k, a in rows[:60]:
        print("%-64s %6d %9.3f %9.1f" % (k, a["launches"] // args.steps, a["ms_total"] / args.steps,
                                         a["work_total"] / (a["ms_total"] * 1e-3) / 1e12 if a["ms_total"] else 0))
    print("total timed %.2f ms/step" % tot)


if __name__ == "__main__":
    main()
