#!/usr/bin/env python3
"""Errors of the fp32 convolution kernels against fp64, in the three product modes (include/retinanet_mi355x.h: RN_FP32_NATIVE =
v_mfma_f32_32x32x2_f32, RN_FP32_SPLIT = three-term bf16 splits on v_mfma_f32_32x32x16_bf16, RN_FP32_SPLIT3 = two-term fp16 splits of
power-of-two-scaled operands on v_mfma_f32_16x16x32_f16 / 32x32x16_f16), and of torch's own fp32 convolution on the same GPU beside
them.  Same inputs for all; reference = torch fp64 convolution on the GPU.

  python tools/fp32_mode_errors.py [--data normal|relu|wide|widepix]
      normal   N(0,1) activations and gradients (default)
      relu     activations as the network has them: non-negative, half of them zero (RN_ERR_RELU=1 is the old spelling)
      wide     every element of x and dY multiplied by its own 2^U(-40, 0): 40 binades inside one tensor, element by element
      widepix  the same factor per PIXEL (all channels of a pixel share it): whole regions of a gradient map 2^-40 below others,
               the shape real loss gradients have (a few positive anchors, a sea of easy negatives)
The last column is split3's rms error over the native kernel's (the gate of VERDICT r4 item 1: <= 1.25).
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import conv as cv  # noqa: E402

SHAPES = [  # name, cin, cout, k, stride, pad, N, H, W   (large enough that none takes the split-K path, which stays on the fp32 MFMA)
    ("3x3 256->256 (K 2304)", 256, 256, 3, 1, 1, 4, 68, 120),
    ("3x3 512->512 (K 4608)", 512, 512, 3, 1, 1, 8, 34, 60),
    ("1x1 1024->256", 1024, 256, 1, 1, 0, 4, 68, 120),
    ("1x1 256->1024", 256, 1024, 1, 1, 0, 4, 68, 120),
    ("3x3 64->64", 64, 64, 3, 1, 1, 2, 135, 240),
    ("3x3 s2 128->128", 128, 128, 3, 2, 1, 4, 136, 240),
    ("7x7 s2 4->64 (stem)", 4, 64, 7, 2, 3, 2, 270, 480),
]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def err(got, want):
    d = (got.double() - want).abs()
    return float(d.max() / want.abs().max()), float(d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())


def conv64(x, w, gy, stride, pad):
    """fp64 convolution, data gradient and weight gradient as unfold + matmul + fold (rocBLAS dgemm; MIOpen's fp64 convolution
    takes minutes at these sizes)."""
    N, cin, H, W = x.shape
    cout, _, k, _ = w.shape
    cols = F.unfold(x.double(), k, padding=pad, stride=stride)              # [N, cin*k*k, L]
    wm = w.double().view(cout, -1)
    y = (wm @ cols).view(N, cout, gy.shape[2], gy.shape[3])
    g = gy.double().view(N, cout, -1)
    dx = F.fold(wm.t() @ g, (H, W), k, padding=pad, stride=stride)
    dw = torch.einsum("ncl,nkl->ck", g, cols).view_as(w)
    return y, dx, dw


MODES = ("native", "split", "split3")


def widen(t, g, per_pixel):
    """t [N,C,H,W] times 2^U(-40,0): per element, or one factor per pixel."""
    shape = (t.shape[0], 1, t.shape[2], t.shape[3]) if per_pixel else tuple(t.shape)
    return t * torch.exp2(-40.0 * torch.rand(shape, generator=g)).to(t.device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="relu" if os.environ.get("RN_ERR_RELU") == "1" else "normal",
                    choices=["normal", "relu", "wide", "widepix"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.allow_tf32 = False
    g = torch.Generator(device="cpu").manual_seed(11)
    print("data: %s;  error against fp64: max |err| / max |ref|, rms err / rms ref" % args.data)
    print("%-24s %-6s %-23s %-23s %-23s %-23s %s" % ("layer", "", "native fp32 MFMA", "split bf16x3 MFMA", "split3 fp16x2 MFMA",
                                                      "torch fp32 (MIOpen)", "split3/native rms"))
    relu_data = args.data == "relu"
    only = os.environ.get("RN_ERR_ONLY", "")
    worst = 0.0
    for name, cin, cout, k, stride, pad, N, H, W in SHAPES:
        if only and only not in name:
            continue
        x = torch.randn(N, cin, H, W, generator=g).to(dev)
        if relu_data:
            x = torch.relu(x)
        w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(dev)
        Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
        gy = torch.randn(N, cout, Ho, Wo, generator=g).to(dev)
        if args.data in ("wide", "widepix"):
            x, gy = widen(x, g, args.data == "widepix"), widen(gy, g, args.data == "widepix")
        yd, dxd, dwd = conv64(x, w, gy, stride, pad)
        xt, wt = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yt = F.conv2d(xt, wt, stride=stride, padding=pad)
        yt.backward(gy)
        res = {}
        for mode in MODES:
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
            res[mode] = (err(nchw(y), yd), err(nchw(dx), dxd), err(dwu, dwd))
        res["torch"] = (err(yt.detach(), yd), err(xt.grad, dxd), err(wt.grad, dwd))
        for i, what in enumerate(("fprop", "dgrad", "wgrad")):
            ratio = res["split3"][i][1] / res["native"][i][1]
            worst = max(worst, ratio)
            print("%-24s %-6s %s   %.2f" % (name if i == 0 else "", what, "   ".join(
                "%.2e / %.2e" % res[m][i] for m in MODES + ("torch",)), ratio))
    if only:
        cv.set_fp32_mfma("split")
        return
    # the Winograd path (transforms in fp32, its 36 GEMMs in the mode under test)
    x = torch.randn(2, 256, 68, 120, generator=g).to(dev)
    w = (torch.randn(256, 256, 3, 3, generator=g) * (2.0 / 2304) ** 0.5).to(dev)
    if args.data == "relu":
        x = torch.relu(x)
    elif args.data != "normal":
        x = widen(x, g, args.data == "widepix")
    yd = conv64(x, w, torch.zeros(2, 256, 68, 120, device=dev), 1, 1)[0]
    yt = F.conv2d(x, w, padding=1)
    row = []
    for mode in MODES:
        cv.set_fp32_mfma(mode)
        y = cv.wino_conv_group([nhwc(x)], cv.wino_weights(w, 0))[0]
        row.append(err(nchw(y), yd))
    row.append(err(yt, yd))
    ratio = row[2][1] / row[0][1]
    worst = max(worst, ratio)
    print("%-24s %-6s %s   %.2f" % ("Winograd F(4x4,3x3) 256->256", "fprop", "   ".join("%.2e / %.2e" % r for r in row), ratio))
    print("largest split3 / native rms-error ratio: %.2f" % worst)
    cv.set_fp32_mfma("split")


if __name__ == "__main__":
    main()
