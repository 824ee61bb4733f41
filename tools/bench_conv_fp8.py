#!/usr/bin/env python3
"""fp8 forward convolution kernels (rn_conv_igemm_fp8: csrc/conv_fp8.hip / conv_fp8_p8.hip) on the layer shapes of BASELINE configs[4]
(ResNet-101, batch 16, 1080p): ms, TFLOP/s against the 5 PF dense fp8 peak, algorithmic GB/s (each operand once) against 8 TB/s.
  RN_FP8_P8=0|1|2 python tools/bench_conv_fp8.py [--only substring]"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import conv as cv  # noqa: E402

LAYERS = [  # name, cin, cout, k, H, W, residual add + ReLU
    ("head tower 3x3 256->256 @135x240", 256, 256, 3, 135, 240, False),
    ("layer3 conv2 3x3 256->256 @68x120", 256, 256, 3, 68, 120, False),
    ("layer3 conv3 1x1 256->1024 @68x120 +res", 256, 1024, 1, 68, 120, True),
    ("layer3 conv1 1x1 1024->256 @68x120", 1024, 256, 1, 68, 120, False),
    ("layer1 conv3 1x1 64->256 @270x480 +res", 64, 256, 1, 270, 480, True),
    ("layer2 conv3 1x1 128->512 @135x240 +res", 128, 512, 1, 135, 240, True),
    ("layer1 conv2 3x3 64->64 @270x480", 64, 64, 3, 270, 480, False),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    print("%-44s %9s %8s %7s %9s %7s" % ("layer", "ms", "TFLOP/s", "of 5P", "GB/s", "of 8T"))
    for name, cin, cout, k, H, W, res in LAYERS:
        if args.only and args.only not in name:
            continue
        x = cv.fp8_quantize(torch.randn(B, H, W, cin, device=dev), 0.01)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.02
        wq, sw = cv.fp8_quantize_weights(cv.pack_weights(w, 0, presplit=False))
        scale = (sw * 0.01).contiguous()
        y = torch.empty((B, H, W, cout), dtype=torch.uint8, device=dev)
        add = cv.fp8_quantize(torch.randn(B, H, W, cout, device=dev), 0.01) if res else None
        geom = (H, W, cout, k, k, 1, 1, -(k // 2), 0)
        fn = lambda: cv.conv_igemm_fp8(x, wq, y, geom, scale, add=add, add_mode=1 if res else 0, act=cv.ACT_RELU, out_scale=0.02)
        ms = timeit(fn)
        flops = 2.0 * B * H * W * cout * cin * k * k
        nbytes = x.numel() + y.numel() * (2 if res else 1) + wq.numel()
        print("%-44s %9.3f %8.1f %6.1f%% %9.0f %6.1f%%" % (name, ms, flops / ms / 1e9, flops / ms / 1e9 / 50.0, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80.0), flush=True)


if __name__ == "__main__":
    main()
