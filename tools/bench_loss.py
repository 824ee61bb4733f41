#!/usr/bin/env python3
"""Fused IoU + assign + focal/smooth-L1/VP loss at BASELINE cfg2 shape (B=8, A=389 205, C=8, N=10) against the HBM
roof, cold (buffers rotated so the 105.9 MB working set cannot sit in the 256 MB Infinity Cache) and warm."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import _hip, ops, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _hip.load()
    B, H, W, C, N = 8, 1080, 1920, 8, 10
    A = ops.anchor_count(H, W)
    cls1, reg1 = synth.head_outputs(1, A, C, 12, seed=3)
    nset = 6                                                   # 6 x 250 MB of cls+reg: rotates through > 256 MiB
    sets = []
    for i in range(nset):
        cls = cls1.to(dev).expand(B, A, C).contiguous()
        reg = reg1.to(dev).expand(B, A, 12).contiguous()
        sets.append((cls, reg))
    ann = synth.labels_dir(B, N, H, W, C, seed=1).to(dev)
    anc = ops.anchors(H, W, dev)
    ws = torch.zeros(lib.rn_focal_workspace_bytes(B, A), dtype=torch.uint8, device=dev)   # counter zero on entry
    out = torch.empty(3, device=dev)
    g = torch.ones(3, device=dev)
    dcls, dreg = torch.empty_like(sets[0][0]), torch.empty_like(sets[0][1])
    s = _hip.stream()

    def fwd(i):
        cls, reg = sets[i % nset]
        _hip.check(lib.rn_focal_loss_fwd(cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), ann.data_ptr(), B, A, C, N, 1,
                                         ws.data_ptr(), out.data_ptr(), s), "fwd")

    def bwd(i):
        cls, reg = sets[i % nset]
        _hip.check(lib.rn_focal_loss_bwd(cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), ann.data_ptr(), B, A, C, N, 1,
                                         ws.data_ptr(), g.data_ptr(), dcls.data_ptr(), dreg.data_ptr(), s), "bwd")

    def timeit(fn, iters, rotate):
        for i in range(3):
            fn(i if rotate else 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i if rotate else 0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    fbytes = B * A * C * 4 + A * 16 + B * N * 27 * 4
    bbytes = fbytes + B * A * C * 4 + B * A * 12 * 4
    for name, fn, nb in (("fwd", fwd, fbytes), ("bwd", bwd, bbytes)):
        for rotate in (True, False):
            us = timeit(fn, 30, rotate)
            print("%s %-5s %8.1f us  %7.1f GB/s  %5.1f%% of 8 TB/s   (algorithmic %.1f MB)" % (
                name, "cold" if rotate else "warm", us, nb / us / 1e3, 100 * nb / us / 1e3 / 8000, nb / 1e6))
    print("losses", out.tolist())
    # phase-1 cost in isolation: the assignment-only entry point (same IoU loop, no classification stream)
    iou = torch.empty(B, A, device=dev)
    arg = torch.empty(B, A, dtype=torch.int32, device=dev)
    st = torch.empty(B, A, dtype=torch.int32, device=dev)

    def assign(i):
        _hip.check(lib.rn_assign(anc.data_ptr(), ann.data_ptr(), B, A, N, 1, iou.data_ptr(), arg.data_ptr(), st.data_ptr(), s), "assign")
    print("assign-only %.1f us" % timeit(assign, 30, False))
    # pure streaming reference: torch sum over cls (reads 99.6 MB)
    c0 = sets[0][0]
    print("torch.sum(cls) %.1f us" % timeit(lambda i: sets[i % nset][0].sum(), 30, True))


if __name__ == "__main__":
    main()
