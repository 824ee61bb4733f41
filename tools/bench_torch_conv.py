#!/usr/bin/env python3
"""What torch (MIOpen) achieves on the same layer shapes on this GPU -- the arithmetic the reference's own
nn.Conv2d calls would run on an MI355X.  Comparison material only: nothing in the product path uses it.
  python tools/bench_torch_conv.py [--batch 8]"""
import argparse

import torch
import torch.nn.functional as F

from bench_conv import SHAPES, timeit, PEAK   # noqa: E402  (same shapes / timer)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False        # MIOpen immediate mode: the exhaustive find takes minutes per shape
    B = args.batch
    print("%-26s %22s %22s   (TFLOP/s, torch %s fp32)" % ("layer", "NCHW  fwd / bwd(d+w)", "NHWC  fwd / bwd(d+w)", torch.__version__), flush=True)
    for name, cin, cout, k, stride, pad, H, W in SHAPES:
        if args.only and not any(t in name for t in args.only.split(",")):
            continue
        row = []
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
            w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=fmt).requires_grad_(True)
            y = F.conv2d(x, w, None, stride, pad)
            gy = torch.randn_like(y)
            flops = 2.0 * y.numel() * cin * k * k
            t_f = timeit(lambda: F.conv2d(x, w, None, stride, pad), args.iters)

            def bwd():
                x.grad = w.grad = None
                torch.autograd.grad(F.conv2d(x, w, None, stride, pad), (x, w), gy)
            t_fb = timeit(bwd, args.iters)
            t_b = max(t_fb - t_f, 1e-6)
            row.append("%5.1f / %5.1f" % (flops / t_f / 1e9, 2 * flops / t_b / 1e9))
        print("%-26s %22s %22s" % (name, row[0], row[1]), flush=True)


if __name__ == "__main__":
    main()
