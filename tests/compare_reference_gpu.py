#!/usr/bin/env python3
"""NOT a pytest test.  Context measurement: the reference's algorithm (oracle/, plain torch ops exactly as the
reference issues them: nn.Conv2d / BatchNorm2d(eval) / ReLU / per-image FocalLoss loop / numpy anchors) executed on
THIS GPU through torch + MIOpen, same workload and step as bench.py (BASELINE configs[1]).  It answers "what would
the unmodified reference get on an MI355X" -- the reference itself cannot travel to the GPU box.
  python tests/compare_reference_gpu.py [--batch 8] [--steps 3]"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
sys.path.insert(0, REPO)
from oracle import model as omodel  # noqa: E402
from retinanet_mi355x import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--channels-last", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = synth.state_dict("resnet50", 8, 12, seed=2)
    params = {}
    for k, v in sd.items():
        v = v.to(dev)
        if args.channels_last and v.dim() == 4:
            v = v.contiguous(memory_format=torch.channels_last)
        params[k] = v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v
    leaves = [p for p in params.values() if p.requires_grad]
    opt = torch.optim.Adam(leaves, lr=1e-4)
    img = torch.randn(args.batch, 3, 1080, 1920, device=dev)
    if args.channels_last:
        img = img.contiguous(memory_format=torch.channels_last)
    ann = synth.labels_dir(args.batch, 10, 1080, 1920, 8, seed=1).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        losses = omodel.train_forward(img, ann, params, "resnet50")
        loss = sum(l.mean() for l in losses)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 0.1)
        opt.step()
        return loss
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
    print("torch+MIOpen reference path%s: %.1f ms/step, %.2f images/s (batch %d, loss %.4f, peak mem %.1f GB)" % (
        " (channels_last)" if args.channels_last else "", dt * 1e3, args.batch / dt, args.batch, float(loss),
        torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)


if __name__ == "__main__":
    main()
