#!/usr/bin/env python3
"""bf16 convolution kernels (rn_conv_igemm_bf16 / rn_conv_wgrad_bf16) on the benchmark's layer shapes (ResNet-50, batch 8,
1080x1920): forward, data gradient and weight gradient -- TFLOP/s against the 2.5 PF dense bf16 MFMA peak, the algorithmic
HBM bytes (each operand once) against 8 TB/s, and the fp32 kernel on the same layer for scale.
  python tools/bench_conv_bf16.py"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import conv as cv  # noqa: E402

PEAK_BF16, PEAK_F32, PEAK_HBM = 2500.0, 157.3, 8000.0

LAYERS = [  # name, cin, cout, k, stride, pad, H, W  (input size), batch 8
    ("head tower 3x3 256->256 @135x240", 256, 256, 3, 1, 1, 135, 240),
    ("layer2 conv2 3x3 128->128 @135x240", 128, 128, 3, 1, 1, 135, 240),
    ("layer3 conv2 3x3 256->256 @68x120", 256, 256, 3, 1, 1, 68, 120),
    ("layer4 conv2 3x3 512->512 @34x60", 512, 512, 3, 1, 1, 34, 60),
    ("layer1 conv3 1x1 64->256 @270x480", 64, 256, 1, 1, 0, 270, 480),
    ("layer2 conv3 1x1 128->512 @135x240", 128, 512, 1, 1, 0, 135, 240),
    ("layer3 conv1 1x1 1024->256 @68x120", 1024, 256, 1, 1, 0, 68, 120),
    ("fpn P3_1 1x1 512->256 @135x240", 512, 256, 1, 1, 0, 135, 240),
]


def timeit(fn, iters=10):
    for _ in range(2):
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
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="substring of a layer name")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 kernel beside each row (PMC runs)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = 8
    print("%-38s %-6s %9s %8s %7s %9s %7s   %s" % ("layer", "op", "ms", "TFLOP/s", "of 2.5P", "GB/s", "of 8T", "fp32 kernel ms (TF)"))
    for name, cin, cout, k, st, pad, H, W in LAYERS:
        if args.only and args.only not in name:
            continue
        Ho, Wo = cv.out_size(H, k, st, pad), cv.out_size(W, k, st, pad)
        x32 = torch.randn(B, H, W, cin, device=dev)
        g32 = torch.randn(B, Ho, Wo, cout, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.02
        x, g = cv.to_bf16(x32), cv.to_bf16(g32)
        wf, wd = cv.pack_weights_bf16(w, 0), cv.pack_weights_bf16(w, 1)
        wf32, wd32 = cv.pack_weights(w, 0), cv.pack_weights(w, 1)
        flops = 2.0 * B * Ho * Wo * cout * cin * k * k
        y = torch.empty(B, Ho, Wo, cout, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(B, H, W, cin, dtype=torch.bfloat16, device=dev)
        y32, dx32 = torch.empty(B, Ho, Wo, cout, device=dev), torch.empty(B, H, W, cin, device=dev)
        kp = (k * k * cin + 31) // 32 * 32
        dw = torch.zeros(cout, kp, device=dev)
        rows = [
            ("fprop", lambda: cv.conv_igemm_bf16(x, wf, y, (Ho, Wo, cout, k, k, st, 1, -pad, 0)),
             2 * (x.numel() + y.numel() + wf.numel()),
             lambda: cv.conv_igemm(x32, wf32, y32, (Ho, Wo, cout, k, k, st, 1, -pad, 0))),
            ("dgrad", lambda: cv.conv_igemm_bf16(g, wd, dx, (H, W, cin, k, k, 1, -1, pad, 0)),
             2 * (g.numel() + dx.numel() + wd.numel()),
             lambda: cv.conv_igemm(g32, wd32, dx32, (H, W, cin, k, k, 1, -1, pad, 0))),
            ("wgrad", lambda: cv.wgrad_bf16(g, x, dw, cout, k, st, pad), 2 * (g.numel() + x.numel()) + 4 * dw.numel(),
             lambda: cv.wgrad(g32, x32, dw, cout, k, st, pad)),
        ]
        for op, fn, nbytes, fn32 in rows:
            ms, ms32 = timeit(fn), (float("nan") if args.no_fp32 else timeit(fn32, 5))
            tf, gbs = flops / ms / 1e9, nbytes / ms / 1e6
            print("%-38s %-6s %9.3f %8.1f %6.1f%% %9.0f %6.1f%%   %.3f (%.0f)"
                  % (name, op, ms, tf, 100 * tf / PEAK_BF16, gbs, 100 * gbs / PEAK_HBM, ms32, flops / ms32 / 1e9), flush=True)


if __name__ == "__main__":
    main()
