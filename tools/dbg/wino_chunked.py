#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache pay for a CHUNKED Winograd pipeline?  The head-tower layer (8 images x five pyramid levels x 256 channels)
as it runs today -- input transform of everything (V: 812 MB), one GEMM (M: 812 MB), output transform -- against the same three stages per
group of images (per image: V 99 MB + M 99 MB, the scratch reused chunk after chunk, so that the GEMM reads V and the output transform
reads M while they are still in the memory-side cache)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import conv as cv  # noqa: E402


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    B, C = 8, 256
    hw = [(135, 240), (68, 120), (34, 60), (17, 30), (9, 15)]
    xs = [torch.randn(B, h, w, C, device=dev) for h, w in hw]
    outs = [torch.empty_like(x) for x in xs]
    w = torch.randn(C, C, 3, 3, device=dev) * 0.02
    U = cv.wino_weights(w, 0)
    bias = torch.randn(C, device=dev)
    full = timeit(lambda: cv.wino_conv_group(xs, U, outs=outs, shift=bias, act=cv.ACT_RELU))
    print("all 8 images at once:                 %.3f ms" % full)
    ref = [o.clone() for o in outs]
    for per in (4, 2, 1):
        def chunked():
            for n in range(0, B, per):
                cv.wino_conv_group([x[n:n + per] for x in xs], U, outs=[o[n:n + per] for o in outs], shift=bias, act=cv.ACT_RELU)
        t = timeit(chunked)
        same = all(torch.equal(a, b) for a, b in zip(outs, ref))
        print("%d image(s) per pass (%d passes):        %.3f ms   results identical: %s" % (per, B // per, t, same))
    # the P3 level alone (75 % of the tiles), same question
    x3, o3 = [xs[0]], [outs[0]]
    print("P3 alone, all images: %.3f ms" % timeit(lambda: cv.wino_conv_group(x3, U, outs=o3, shift=bias, act=cv.ACT_RELU)))
    for per in (2, 1):
        def chunked3():
            for n in range(0, B, per):
                cv.wino_conv_group([xs[0][n:n + per]], U, outs=[outs[0][n:n + per]], shift=bias, act=cv.ACT_RELU)
        print("P3 alone, %d image(s) per pass: %.3f ms" % (per, timeit(chunked3)))


if __name__ == "__main__":
    main()
