#!/usr/bin/env python3
"""The Winograd output transform by epilogue form on the head-tower group (8 images, five pyramid levels, 256 channels): plain / ReLU + sign
bits written (forward) / ReLU-mask bits read (data gradient) / mask bits + accumulated addend."""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import _hip, conv as cv  # noqa: E402


def timeit(fn, iters=20):
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
    lib = _hip.load()
    B, C = 8, 256
    hw = [(135, 240), (68, 120), (34, 60), (17, 30), (9, 15)]
    xs = [torch.randn(B, h, w, C, device=dev) for h, w in hw]
    outs = [torch.empty_like(x) for x in xs]
    adds = [torch.randn_like(x) for x in xs]
    T = sum(B * ((h + 3) // 4) * ((w + 3) // 4) for h, w in hw)
    Tpad = cv.wino_tpad(T)
    M = torch.randn(36 * Tpad * C, device=dev)
    scale, shift = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    nb = 4.0 * (36 * T * C + sum(x.numel() for x in xs))

    def run(label, g, mm, act, sc=None, sh=None, extra=0.0):
        t = timeit(lambda: _hip.check(lib.rn_wino_output_group(ctypes.byref(g), M.data_ptr(), C, 0, Tpad, _hip.ptr(sc), _hip.ptr(sh), mm, act, 0,
                                                               _hip.stream()), "out"))
        print("%-58s %.3f ms   %.2f TB/s" % (label, t, (nb + extra) / t / 1e9), flush=True)

    run("plain", cv._wino_group(xs, dsts=outs), 0, 0)
    signs = [cv._sign_words(o, True) for o in outs]
    run("scale + shift + ReLU, sign bits written (forward)", cv._wino_group(xs, dsts=outs, signs=signs), 0, 1, scale, shift)
    masks = [torch.randn_like(x) for x in xs]
    for m in masks:
        cv._sign_words(m, True).random_(-2 ** 31, 2 ** 31 - 1)           # (timing: any bits)
    outs2 = [torch.empty_like(x) for x in xs]
    have_bits = all(getattr(m, "_rn_sign", None) is not None for m in masks)
    if have_bits:
        run("mask bits read (data gradient)", cv._wino_group(xs, dsts=outs2, masks=masks, mask_bits=True), 2 | cv.MASK_BITS, 0)
        run("mask bits + addend", cv._wino_group(xs, dsts=outs2, masks=masks, mask_bits=True, adds=adds), 2 | cv.MASK_BITS, 0,
            extra=4.0 * sum(x.numel() for x in xs))
    run("fp32 mask read", cv._wino_group(xs, dsts=outs2, masks=masks), 2, 0, extra=4.0 * sum(x.numel() for x in xs))
    run("fp32 mask + addend", cv._wino_group(xs, dsts=outs2, masks=masks, adds=adds), 2, 0, extra=8.0 * sum(x.numel() for x in xs))


if __name__ == "__main__":
    main()
