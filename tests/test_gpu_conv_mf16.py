"""GPU parity of the split-operand implicit GEMM on 16x16x32 MFMAs (csrc/conv_igemm_mf16.hip: 128 x 128 tiles, a wave owns 32 rows x 128
columns, the activation operand split in registers) against torch's CPU conv2d in fp64, in BOTH of its arithmetic forms: "mf16" =
RN_FP32_SPLIT (three bf16 terms, six MFMAs per product) and "mf16h" = RN_FP32_SPLIT3 (two fp16 terms of operands scaled by powers of
two, three MFMAs; round 5).  The kernel is the product default for wide layers; here RN_OPT_MF16_MIN = 1 forces it onto small shapes:
ragged last row tile, Cout not a multiple of 128 (192, 320: zero-filled weight rows, masked columns), stride 2, 1x1 and 3x3, every
epilogue form (bias + ReLU, residual add, ReLU mask before / after, strided head-slice output), the grouped launch over pyramid levels
and the plain-GEMM (Winograd) form; every case is also compared with the 32x32x16 kernels (RN_OPT_MF16 = 0).  Tolerance 1e-5 of the
output's max magnitude -- the same in both forms.  (Round 5 moved the 256 x 256 tiles and the persistent form this file also covered to
tools/probes/quarantine_r05/, with their cases.)"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


OPTS = ("OPT_MF16", "OPT_MF16_MIN")


def _set(cv, opts):
    for k, v in opts.items():
        cv.set_option(getattr(cv, k), v)


@pytest.fixture(scope="module", params=["mf16", "mf16h"])
def cv(dev, request):
    from retinanet_mi355x import conv
    before = (conv.get_fp32_mfma(), conv.PRESPLIT) + tuple(conv.get_option(getattr(conv, k)) for k in OPTS)
    conv.set_fp32_mfma("split3" if request.param == "mf16h" else "split")
    conv.PRESPLIT = True
    conv._big_on = {"OPT_MF16": 1, "OPT_MF16_MIN": 1}
    conv._half = request.param == "mf16h"
    _set(conv, conv._big_on)
    yield conv
    conv.set_fp32_mfma(before[0])
    conv.PRESPLIT = before[1]
    _set(conv, dict(zip(OPTS, before[2:])))


def rnd(shape, seed, std=1.0):
    from retinanet_mi355x import synth
    return torch.from_numpy(synth.normal(shape, seed, std))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def close(got, want, tol=1e-5):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = float((got - want).abs().max())
    assert err <= tol * (float(want.abs().max()) + 1e-12), "max err %.3e vs max |ref| %.3e" % (err, float(want.abs().max()))


def _both_kernels(cv, fn):
    """fn() with the 16x16x32 kernel on, then off: the two kernels must agree with each other too."""
    a = fn()
    _set(cv, {"OPT_MF16": 0})
    try:
        b = fn()
    finally:
        _set(cv, cv._big_on)
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    return a


CASES = [  # cin, cout, k, stride, pad, N, H, W
    (64, 256, 3, 1, 1, 2, 19, 23),       # M = 874: three full row tiles + a ragged one
    (256, 512, 1, 2, 0, 1, 33, 31),      # 1x1 stride 2
    (128, 320, 3, 1, 1, 1, 30, 34),      # Cout = 320: second column tile 64 wide
    (32, 192, 3, 1, 1, 3, 9, 11),        # Cout = 192 < 256, several images inside one row tile
    (1024, 256, 1, 1, 0, 2, 17, 15),     # long K
    (128, 256, 3, 2, 1, 2, 37, 29),      # 3x3 stride 2
    (64, 64, 3, 1, 1, 2, 19, 23),        # at most 64 output channels: not this kernel's (the 256 x 64 tile of conv_igemm_tile.h)
    (256, 64, 1, 1, 0, 1, 33, 31),
    (128, 32, 3, 2, 1, 2, 21, 17),       # 32 of the 64 columns masked
]


@pytest.mark.parametrize("case", CASES)
def test_fprop_bias_relu(cv, dev, case):
    cin, cout, k, stride, pad, N, H, W = case
    x, w, b = rnd((N, cin, H, W), 1), rnd((cout, cin, k, k), 2, (2.0 / (k * k * cin)) ** 0.5), rnd((cout,), 3)
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride, pad))
    xg, wp = nhwc(x).to(dev), cv.pack_weights(w.to(dev), 0)
    twin = "_rn_split16" if cv._half and cv.f16_shape_ok(cout, cin, k * k) else "_rn_split"
    assert getattr(wp, twin, None) is not None
    y = _both_kernels(cv, lambda: cv.fprop(xg, wp, cout, k, stride, pad, shift=b.to(dev), act=cv.ACT_RELU))
    close(y.permute(0, 3, 1, 2), want)


def test_dgrad_with_mask_and_residual(cv, dev):
    """The data gradient of a 3x3 256->256 layer with the ReLU mask of its input and an accumulated addend (the backbone's
    conv2 -> conv1 hand-over): mask after the add (mode 2) and before (mode 1)."""
    cin, cout, N, H, W = 256, 256, 2, 21, 27
    w, gy = rnd((cout, cin, 3, 3), 5, 0.03), rnd((N, cout, H, W), 6)
    z, addend = rnd((N, cin, H, W), 7), rnd((N, cin, H, W), 8)
    dx = F.conv_transpose2d(gy.double(), w.double(), None, 1, 1)
    wd = cv.pack_weights(w.to(dev), 1)
    for mode, want in ((2, torch.where(z.double() > 0, dx + addend.double(), torch.zeros_like(dx))),
                       (1, torch.where(z.double() > 0, dx, torch.zeros_like(dx)) + addend.double())):
        got = _both_kernels(cv, lambda: cv.dgrad(nhwc(gy).to(dev), wd, (H, W), cin, 3, 1, 1, add=nhwc(addend).to(dev), add_mode=1,
                                                 mask=nhwc(z).to(dev), mask_mode=mode))
        close(got.permute(0, 3, 1, 2), want)


def test_strided_head_slice_output_and_grouped_levels(cv, dev):
    """Three pyramid levels through ONE grouped launch, each writing its slice of a concatenated [B, A, n] tensor (the heads'
    NHWC flatten, D/model.py:155-157) -- with 256 output channels so that this kernel takes it."""
    cin, cout, B = 64, 256, 2
    w, b = rnd((cout, cin, 3, 3), 9, 0.05), rnd((cout,), 10)
    levels = [(17, 23), (9, 12), (5, 6)]
    xs = [rnd((B, cin, h, w_), 20 + i) for i, (h, w_) in enumerate(levels)]
    total = sum(h * w_ for h, w_ in levels) * cout
    out = torch.zeros((B, total), device=dev)
    wp = cv.pack_weights(w.to(dev), 0)
    probs, off = [], 0
    for x, (h, w_) in zip(xs, levels):
        probs.append({"x": nhwc(x).to(dev), "y": out[:, off:], "geom": (h, w_, cout, 3, 3, 1, 1, -1, 0), "y_batch_stride": total})
        off += h * w_ * cout
    cv.conv_igemm_grouped(probs, wp, shift=b.to(dev), act=cv.ACT_RELU)
    off = 0
    for x, (h, w_) in zip(xs, levels):
        want = F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1)).permute(0, 2, 3, 1).reshape(B, -1)
        close(out[:, off:off + h * w_ * cout], want)
        off += h * w_ * cout


def test_winograd_group_takes_the_plain_gemm(cv, dev):
    """The Winograd path's batched GEMM (36 positions, per-position weights, T padded to 256) through this kernel's plain-GEMM
    form, against the direct convolution in fp64 (Winograd F(4x4,3x3): 1e-4 of the max)."""
    C, B = 256, 2
    w, b = rnd((C, C, 3, 3), 11, 0.03), rnd((C,), 12)
    xs = [rnd((B, C, 35, 41), 13), rnd((B, C, 18, 21), 14)]
    U = cv.wino_weights(w.to(dev), 0)
    ys = cv.wino_conv_group([nhwc(x).to(dev) for x in xs], U, shift=b.to(dev), act=cv.ACT_RELU)
    for x, y in zip(xs, ys):
        close(y.permute(0, 3, 1, 2), F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1)), tol=1e-4)


PLAIN = [  # cin, cout, N, H, W: 1x1, no epilogue -> the plain-GEMM instance
    (64, 256, 2, 19, 23),        # two K-steps: eight stores per step; M = 874, ragged last row tile
    (128, 192, 1, 30, 34),       # four K-steps; Cout = 192: the second column tile's upper half is past Cout
    (192, 256, 3, 17, 15),       # six K-steps (not a power of two): the slices leave a remainder for the end of the tile
    (512, 320, 1, 33, 31),       # sixteen K-steps: one store per step; three column tiles, the last 64 wide
    (256, 128, 4, 16, 16),       # one column tile, M a multiple of 128
]


@pytest.mark.parametrize("case", PLAIN)
def test_plain_gemm_short_reductions(cv, dev, case):
    cin, cout, N, H, W = case
    x, w = rnd((N, cin, H, W), 31), rnd((cout, cin, 1, 1), 32, (2.0 / cin) ** 0.5)
    want = F.conv2d(x.double(), w.double())
    xg, wp = nhwc(x).to(dev), cv.pack_weights(w.to(dev), 0)
    y = _both_kernels(cv, lambda: cv.fprop(xg, wp, cout, 1, 1, 0))
    close(y.permute(0, 3, 1, 2), want)


@pytest.mark.parametrize("C", [128, 256, 512])
def test_plain_gemm_per_position_weights(cv, dev, C):
    """The Winograd stage's launch shape: `images` of 256 rows with a weight matrix of their own each (rn_conv_desc.w_batch_stride), the
    tiles of one workgroup crossing from image to image.  K = 128: activations straight into registers; K = 256 / 512 (round 5): staged
    through the wave-private LDS strips (two column tiles share every row tile)."""
    P, T, Co = 9, 256, 256
    V = rnd((P, 1, T, C), 41).to(dev)
    U = rnd((P, Co, C), 42, (2.0 / C) ** 0.5).to(dev)
    Uv = (cv.split_weights_f16 if cv._half else cv.split_weights)(U.view(P * Co, C).contiguous())
    M = torch.empty((P, 1, T, Co), device=dev)
    cv.conv_igemm(V, Uv, M, (1, T, Co, 1, 1, 1, 1, 0, 0), w_batch_stride=Co * C)
    want = torch.einsum("ptc,poc->pto", V[:, 0].double().cpu(), U.double().cpu())
    close(M[:, 0], want)
