#!/usr/bin/env python3
"""Errors of the fp32 convolution kernels against fp64, in both product modes (include/retinanet_mi355x.h: RN_FP32_NATIVE =
v_mfma_f32_32x32x2_f32, RN_FP32_SPLIT = three-term bf16 splits on v_mfma_f32_32x32x16_bf16), and of torch's own fp32
convolution on the same GPU beside them.  Same inputs for all; reference = torch fp64 convolution on the GPU.

  python tools/fp32_mode_errors.py
"""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import conv as cv  # noqa: E402

SHAPES = [  # name, cin, cout, k, stride, pad, N, H, W
    ("3x3 256->256 (K 2304)", 256, 256, 3, 1, 1, 1, 68, 120),
    ("3x3 512->512 (K 4608)", 512, 512, 3, 1, 1, 1, 34, 60),
    ("1x1 1024->256", 1024, 256, 1, 1, 0, 1, 68, 120),
    ("1x1 64->256", 64, 256, 1, 1, 0, 1, 135, 240),
    ("3x3 64->64", 64, 64, 3, 1, 1, 1, 135, 240),
    ("3x3 s2 128->128", 128, 128, 3, 2, 1, 1, 136, 240),
]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def err(got, want):
    d = (got.double() - want).abs()
    return float(d.max() / want.abs().max()), float(d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.allow_tf32 = False
    g = torch.Generator(device="cpu").manual_seed(11)
    print("error against fp64: max |err| / max |ref|, rms err / rms ref")
    print("%-24s %-6s %-23s %-23s %-23s" % ("layer", "", "native fp32 MFMA", "split bf16x3 MFMA", "torch fp32 (MIOpen)"))
    for name, cin, cout, k, stride, pad, N, H, W in SHAPES:
        x = torch.randn(N, cin, H, W, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev)
        Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
        gy = torch.randn(N, cout, Ho, Wo, generator=g).to(dev)
        xd, wd_ = x.double().requires_grad_(True), w.double().requires_grad_(True)
        yd = F.conv2d(xd, wd_, stride=stride, padding=pad)
        yd.backward(gy.double())
        xt, wt = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yt = F.conv2d(xt, wt, stride=stride, padding=pad)
        yt.backward(gy)
        res = {}
        for mode in ("native", "split"):
            cv.set_fp32_mfma(mode)
            wp, wdg = cv.pack_weights(w, 0), cv.pack_weights(w, 1)
            y = cv.fprop(nhwc(x), wp, cout, k, stride, pad)
            if stride == 2:
                wcls = [cv.pack_weights(w, 1, taps=c[2]) for c in cv.s2_classes(k, pad)]
                dx = cv.dgrad_s2_classes(nhwc(gy), wcls, (H, W), cin, k, pad)
            else:
                dx = cv.dgrad(nhwc(gy), wdg, (H, W), cin, k, stride, pad)
            dw = torch.zeros_like(wp)
            cv.wgrad(nhwc(gy), nhwc(x), dw, cout, k, stride, pad)
            dwu = cv.unpack_wgrad(dw, wp, tuple(w.shape))[0]
            res[mode] = (err(nchw(y), yd.detach()), err(nchw(dx), xd.grad), err(dwu, wd_.grad))
        res["torch"] = (err(yt.detach(), yd.detach()), err(xt.grad, xd.grad), err(wt.grad, wd_.grad))
        for i, what in enumerate(("fprop", "dgrad", "wgrad")):
            print("%-24s %-6s %s" % (name if i == 0 else "", what, "   ".join(
                "%.2e / %.2e" % res[m][i] for m in ("native", "split", "torch"))))
    cv.set_fp32_mfma("native")


if __name__ == "__main__":
    main()
