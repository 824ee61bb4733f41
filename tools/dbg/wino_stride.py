#!/usr/bin/env python3
"""Does the plane stride of the Winograd-domain tensors matter?  V / M are [36][Tpad][C] with Tpad a multiple of 256: the 36 values a thread
combines lie Tpad * C * 4 bytes apart (21.5 MiB for the head towers: low 19 address bits equal).  Times rn_wino_output_group and
rn_wino_input_group on the head-tower group with the stride as a free parameter (the kernels take it as one)."""
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
    T = sum(B * ((h + 3) // 4) * ((w + 3) // 4) for h, w in hw)
    base = (T + 255) // 256 * 256
    print("T %d, Tpad %d: plane stride %.3f MiB" % (T, base, base * C * 4 / 2 ** 20))
    for extra in (0, 1, 2, 4, 5, 8, 16, 17, 32, 33, 64, 128, 129):
        Tpad = base + extra
        M = torch.randn(36 * Tpad * C, device=dev)
        V = torch.empty(36 * Tpad * C, device=dev)
        g_out = cv._wino_group(xs, dsts=outs)
        g_in = cv._wino_group(xs, srcs=xs)
        t_out = timeit(lambda: _hip.check(lib.rn_wino_output_group(ctypes.byref(g_out), M.data_ptr(), C, 0, Tpad, None, None, 0, 0, 0, _hip.stream()), "out"))
        t_in = timeit(lambda: _hip.check(lib.rn_wino_input_group(ctypes.byref(g_in), V.data_ptr(), C, 0, Tpad, 0, None, None, _hip.stream()), "in"))
        nb_out = 4.0 * (36 * T * C + sum(x.numel() for x in xs))
        print("Tpad + %3d (stride %% 4 KiB = %4d B): output %.3f ms  %.2f TB/s | input %.3f ms  %.2f TB/s"
              % (extra, (Tpad * C * 4) % 4096, t_out, nb_out / t_out / 1e9, t_in, nb_out / t_in / 1e9), flush=True)
        del M, V


if __name__ == "__main__":
    main()
