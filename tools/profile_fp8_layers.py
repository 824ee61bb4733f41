#!/usr/bin/env python3
"""Per-layer-shape timing of the fp8 inference forward (BASELINE configs[4]: ResNet-101, batch 16, 1080p): HIP events around every conv
launch, keyed by shape.   RN_FP8_P8=0|1|2 python tools/profile_fp8_layers.py [--steps 3] [--arch resnet101] [--batch 16]"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import modules, prof, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, H, W = args.batch, 1080, 1920
    net = getattr(modules, args.arch)(num_classes=8)
    net.load_state_dict(synth.state_dict(args.arch, 8, 12, seed=2))
    net = net.to(dev).eval()
    img = torch.randn(B, 3, H, W, device=dev)
    net.calibrate_fp8(img[:2])
    P = net._tensor_dict()

    def step():
        with torch.no_grad():
            return net._engine.forward(P, img, save=False)
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
