#!/usr/bin/env python3
"""Micro-benchmark of the conv engine on the layer shapes of BASELINE cfg2 (ResNet-50, B=8, 1080x1920).

Prints achieved TFLOP/s (algorithmic 2*M*N*K) per kernel against the 157.3 TF fp32 MFMA peak.
Run on the GPU box:  python tools/bench_conv.py [--batch 8]
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import conv as cv  # noqa: E402

PEAK = 157.3

SHAPES = [  # name, cin, cout, k, stride, pad, H, W
    ("head 3x3 256->256 P3", 256, 256, 3, 1, 1, 135, 240),
    ("head 3x3 256->256 P4", 256, 256, 3, 1, 1, 68, 120),
    ("head out 256->108 P3", 256, 108, 3, 1, 1, 135, 240),
    ("l1 1x1 64->256", 64, 256, 1, 1, 0, 270, 480),
    ("l1 1x1 256->64", 256, 64, 1, 1, 0, 270, 480),
    ("l1 3x3 64->64", 64, 64, 3, 1, 1, 270, 480),
    ("l2 3x3 128->128", 128, 128, 3, 1, 1, 135, 240),
    ("l2 1x1 128->512", 128, 512, 1, 1, 0, 135, 240),
    ("l3 3x3 256->256", 256, 256, 3, 1, 1, 68, 120),
    ("l3 1x1 1024->256", 1024, 256, 1, 1, 0, 68, 120),
    ("l4 3x3 512->512", 512, 512, 3, 1, 1, 34, 60),
    ("l4 1x1 512->2048", 512, 2048, 1, 1, 0, 34, 60),
    ("l4 1x1 2048->512", 2048, 512, 1, 1, 0, 34, 60),
    ("l3 1x1 256->1024", 256, 1024, 1, 1, 0, 68, 120),
    ("l2.0 3x3 s2 128->128", 128, 128, 3, 2, 1, 270, 480),
    ("gemm-like 1x1 256->256 P3", 256, 256, 1, 1, 0, 135, 240),      # K = 256: the Winograd stage's GEMM shape per position
    ("ksweep 1x1 64->256 P3", 64, 256, 1, 1, 0, 135, 240),
    ("ksweep 1x1 128->256 P3", 128, 256, 1, 1, 0, 135, 240),
    ("ksweep 1x1 512->256 P3", 512, 256, 1, 1, 0, 135, 240),
    ("ksweep 1x1 1024->256 P3", 1024, 256, 1, 1, 0, 135, 240),
    ("ksweep 1x1 2048->256 P3", 2048, 256, 1, 1, 0, 135, 240),
    # occupancy probes (run with --batch 1): 256 / 512 / 1024 / 2048 workgroups of the 128x128 tile, K = 2304
    ("probe 256wg", 256, 128, 3, 1, 1, 128, 256),
    ("probe 512wg", 256, 128, 3, 1, 1, 256, 256),
    ("probe 1024wg", 256, 128, 3, 1, 1, 512, 256),
    ("probe 2048wg", 256, 128, 3, 1, 1, 512, 512),
]


def timeit(fn, iters):
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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--mfma", default="", choices=["", "native", "split", "split3"], help="fp32 product mode (default: the library's)")
    args = ap.parse_args()
    if args.mfma:
        cv.set_fp32_mfma(args.mfma)
    print("fp32 products:", cv.get_fp32_mfma())
    dev = torch.device("cuda:0")
    B = args.batch
    print("%-26s %9s %9s %9s   (TFLOP/s; %% of %.1f TF fp32 MFMA peak)" % ("layer", "fprop", "dgrad", "wgrad", PEAK))
    # The Winograd stage's batched GEMM as the training step launches it (conv.wino_conv_group): 36 positions x T tiles of the five
    # pyramid levels of a tower layer at batch B, K = N = 256, raw epilogue.  fprop column only.
    for name, c, co in (("wino gemm 36 x T 256->256", 256, 256), ("wino gemm 36 x T 512->512 l4", 512, 512), ("wino gemm 36 x T 128->128 l2", 128, 128),
                        ("wino gemm 36 x T 256->256 l3", 256, 256)):
        if args.only and args.only not in name:
            continue
        hw = {"256->256": [(135, 240), (68, 120), (34, 60), (17, 30), (9, 15)], "512 l4": [(34, 60)], "128 l2": [(135, 240)],
              "256 l3": [(68, 120)]}["256->256" if name.endswith("256->256") else name[-6:]]
        T = sum(B * ((h + 3) // 4) * ((w_ + 3) // 4) for h, w_ in hw)
        Tpad = (T + 255) // 256 * 256
        V = torch.randn(36, 1, Tpad, c, device=dev)
        Mo = torch.empty(36, 1, Tpad, co, device=dev)
        U = torch.randn(36, co, c, device=dev) * 0.05
        Uv = U.view(36 * co, c)
        if cv.PRESPLIT and cv.get_fp32_mfma() == "split":
            Uv = cv.split_weights(Uv)
        elif cv.PRESPLIT and cv.get_fp32_mfma() == "split3":
            Uv = cv.split_weights_f16(Uv)
        t = timeit(lambda: cv.conv_igemm(V, Uv, Mo, (1, Tpad, co, 1, 1, 1, 1, 0, 0), w_batch_stride=co * c), args.iters)
        tf = 2.0 * 36 * Tpad * co * c / (t * 1e-3) / 1e12
        print("%-26s %5.1f %3.0f%%   ms %.3f   (T %d, %.0f MB in + %.0f MB out)" % (name[:26], tf, 100 * tf / PEAK, t, Tpad, V.numel() * 4e-6, Mo.numel() * 4e-6))
        del V, Mo, U, Uv
    for name, cin, cout, k, stride, pad, H, W in SHAPES:
        if args.only and args.only not in name:
            continue
        x = torch.randn(B, H, W, cin, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        wp = cv.pack_weights(w, 0)
        cpad = (cout + 3) // 4 * 4
        wd = cv.pack_weights(w, 1, c_pad=cpad)
        Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
        dy = torch.randn(B, Ho, Wo, cpad, device=dev)
        dw = torch.zeros_like(wp)
        flops = 2.0 * B * Ho * Wo * cout * cin * k * k
        bias = torch.randn(cout, device=dev)
        t_f = timeit(lambda: cv.fprop(x, wp, cout, k, stride, pad, shift=bias, act=cv.ACT_RELU), args.iters)
        if stride == 2 and k > 1:                       # what the engine runs: one launch per output-parity class, no wasted taps
            wcls = [cv.pack_weights(w, 1, c_pad=cpad, taps=c[2]) for c in cv.s2_classes(k, pad)]
            t_d = timeit(lambda: cv.dgrad_s2_classes(dy, wcls, (H, W), cin, k, pad), args.iters)
        else:
            t_d = timeit(lambda: cv.dgrad(dy, wd, (H, W), cin, k, stride, pad), args.iters)
        t_w = timeit(lambda: cv.wgrad(dy, x, dw, cout, k, stride, pad), args.iters)
        tf = [flops / (t * 1e-3) / 1e12 for t in (t_f, t_d, t_w)]
        print("%-26s %5.1f %3.0f%% %5.1f %3.0f%% %5.1f %3.0f%%   ms %.3f %.3f %.3f" % (
            name, tf[0], 100 * tf[0] / PEAK, tf[1], 100 * tf[1] / PEAK, tf[2], 100 * tf[2] / PEAK, t_f, t_d, t_w))
        del x, w, wp, wd, dy, dw
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
