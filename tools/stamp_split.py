#!/usr/bin/env python3
"""Where a K-step of the SPLIT 3 implicit-GEMM kernel spends its cycles: runs one layer shape on the STAMP build
(tools/build_variant.sh STAMP -DRN_STAMP=1) and prints the per-segment averages of the in-kernel s_memtime stamps
(conv_igemm_tile.h, RN_STAMP).  The stamped build is slower than the product one (its fences forbid overlap): read the shares.

  RN_LIB_PATH=3d-playground_amd/retinanet_mi355x/lib/ab/libSTAMP.so python tools/stamp_split.py [--only "head 3x3 256->256 P3"]
"""
import argparse
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
sys.path.insert(0, os.path.join(REPO, "tools"))
from retinanet_mi355x import _hip, conv as cv  # noqa: E402
from bench_conv import SHAPES, timeit  # noqa: E402

SEG = ["issue loads (3 DMA + 2 register loads)", "operand reads landed (12 ds_read_b128)", "split + plane stores landed",
       "24 MFMAs issued", "wait vmcnt(0)", "barrier"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="head 3x3 256->256 P3")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    seg = SEG
    os.environ["RN_MF16"] = "0"                              # the stamps are in the 32x32x16 kernel (conv_igemm_tile.h, SPLIT 3)
    cv.set_fp32_mfma("split")
    lib = ctypes.CDLL(_hip.LIB_PATH)
    buf = (ctypes.c_ulonglong * 16)()
    dev = torch.device("cuda:0")
    for name, cin, cout, k, stride, pad, H, W in SHAPES:
        if args.only not in name:
            continue
        x = torch.randn(args.batch, H, W, cin, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        wp = cv.pack_weights(w, 0)
        bias = torch.randn(cout, device=dev)
        read = lib.rn_debug_stamps
        read(buf)                                                 # clear
        t = timeit(lambda: cv.fprop(x, wp, cout, k, stride, pad, shift=bias, act=cv.ACT_RELU), args.iters)
        assert read(buf) == 0
        v = list(buf)
        steps, waves = v[8], v[9]
        print("%s: fprop %.3f ms (stamped build), %d waves, %.1f K-steps per wave; s_memtime = shader clock cycles"
              % (name, t, waves, steps / max(waves, 1)))
        tot = sum(v[:len(seg)])
        for i in range(len(seg)):
            print("   %-48s %8.1f cycles per K-step   %5.1f %%" % (seg[i], v[i] / max(steps, 1), 100.0 * v[i] / max(tot, 1)))
        print("   %-44s %8.1f" % ("sum", tot / max(steps, 1)))
        print("   shader clock over the K loops: %.0f MHz (s_memtime / s_memrealtime x 100 MHz); loop cycles per wave %.0f"
              % (100.0 * v[10] / max(v[11], 1), v[10] / max(waves, 1)))


if __name__ == "__main__":
    main()
