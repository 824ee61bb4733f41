"""GPU parity of the fp8 (e4m3fn) forward convolution (csrc/conv_fp8.hip: v_mfma_scale_f32_32x32x64_f8f6f4, BASELINE configs[4],
first cut).  The reference has no fp8 mode, so these pin that the kernel computes the same FUNCTION:

* with operands that ARE fp8 values (small integers and halves) every product and sum is exact in fp32: the kernel must equal
  torch's CPU conv2d bit for bit -- that checks the operand lane map of the 32x32x64 MFMA, the staging, the swizzle and the
  epilogue without any tolerance (an A = I style check with asymmetric data);
* with random operands, against the fp32 convolution of the DEQUANTISED operands (same rounded inputs): 1e-4 of the max for an
  fp32 result (measured 1.1e-5), one e4m3 rounding (2^-4 relative) for an fp8 result;
* against the fp32 convolution of the ORIGINAL operands: the quantisation error itself, ~2-3 % rms for e4m3 at K ~ 10^3
  (stated, not hidden: this is what fp8 costs).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(shape, seed, std=1.0):
    from retinanet_mi355x import synth
    return torch.from_numpy(synth.normal(shape, seed, std))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


CASES = [  # cin, cout, k, stride, pad, N, H, W
    (64, 128, 3, 1, 1, 2, 13, 17),
    (256, 256, 3, 1, 1, 1, 19, 23),
    (1024, 256, 1, 1, 0, 2, 9, 11),
    (128, 144, 3, 2, 1, 2, 21, 18),       # Cout not a multiple of the tile, stride 2
    (48, 64, 1, 1, 0, 3, 7, 9),           # Cin % 64 != 0: chunk-wise addressing; K = 48 padded to 64
]


@pytest.mark.parametrize("case", CASES)
def test_exact_on_fp8_valued_operands(dev, case):
    from retinanet_mi355x import conv as cv
    cin, cout, k, stride, pad, N, H, W = case
    # integers in [-4, 4] (activations) and multiples of 1/2 in [-2, 2] (weights): all exactly representable in e4m3, sums < 2^24
    x = torch.round(rnd((N, cin, H, W), 1, 2.0)).clamp(-4, 4)
    w = (torch.round(rnd((cout, cin, k, k), 2, 2.0)) / 2).clamp(-2, 2)
    w[:, 0, 0, 0] = 2.0                                        # every row's max is 2 -> row scale 2 / 448: a power of two times 7/..., see below
    want = F.conv2d(x.double(), w.double(), None, stride, pad)
    xq = cv.fp8_quantize(nhwc(x).to(dev), 1.0)
    wp = cv.pack_weights(w.to(dev), 0, presplit=False)
    # exactness needs power-of-two weight scales: quantise the rows by hand with scale 1 (values are fp8 already)
    kp = wp.shape[1]
    wq = torch.zeros((cout, (kp + 63) // 64 * 64), dtype=torch.uint8, device=dev)
    wq[:, :kp] = cv.fp8_quantize(wp, 1.0)
    Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
    y = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=dev)
    ones = torch.ones(cout, device=dev)
    cv.conv_igemm_fp8(xq, wq, y, (Ho, Wo, cout, k, k, stride, 1, -pad, 0), ones)
    assert torch.equal(y.permute(0, 3, 1, 2).cpu().double(), want), float((y.permute(0, 3, 1, 2).cpu().double() - want).abs().max())


@pytest.mark.parametrize("case", CASES[:4])
def test_random_operands_bias_relu_residual_and_fp8_output(dev, case):
    from retinanet_mi355x import conv as cv
    cin, cout, k, stride, pad, N, H, W = case
    x = F.relu(rnd((N, cin, H, W), 3))
    w = rnd((cout, cin, k, k), 4, (2.0 / (k * k * cin)) ** 0.5)
    b = rnd((cout,), 5, 0.1)
    sx = float(x.abs().max()) / cv.FP8_MAX
    xq = cv.fp8_quantize(nhwc(x).to(dev), sx)
    wq, sw = cv.fp8_quantize_weights(cv.pack_weights(w.to(dev), 0, presplit=False))
    x_dq = cv.fp8_dequantize(xq).cpu().permute(0, 3, 1, 2)
    kp = k * k * cin
    w_dq = (cv.fp8_dequantize(wq, 1.0)[:, :kp] * sw[:, None]).cpu().reshape(cout, k, k, cin).permute(0, 3, 1, 2)
    Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
    res = F.relu(rnd((N, cout, Ho, Wo), 6))
    sr = float(res.abs().max()) / cv.FP8_MAX
    rq = cv.fp8_quantize(nhwc(res).to(dev), sr)
    res_dq = cv.fp8_dequantize(rq).cpu().permute(0, 3, 1, 2)
    want_dq = F.relu(F.conv2d(x_dq.double(), w_dq.double(), b.double(), stride, pad) + res_dq.double())
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride, pad) + res.double())
    scale = (sw * sx).contiguous()
    geom = (Ho, Wo, cout, k, k, stride, 1, -pad, 0)
    # fp32 result: the kernel against the same rounded operands
    y = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=dev)
    cv.conv_igemm_fp8(xq, wq, y, geom, scale, shift=b.to(dev), add=rq, add_mode=1, act=cv.ACT_RELU)
    got = y.permute(0, 3, 1, 2).cpu().double()
    # (measured 1.1e-5: the f8f6f4 MFMA adds its 64 products in its own internal order / precision before the fp32 accumulate)
    assert float((got - want_dq).abs().max()) <= 1e-4 * float(want_dq.abs().max())
    rms = float(((got - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
    assert rms <= 0.06, rms                                    # e4m3 quantisation of both operands (measured 2-4 %)
    # fp8 result (Cout % 16 == 0): one more e4m3 rounding
    if cout % 16 == 0:
        sy = float(want_dq.abs().max()) / cv.FP8_MAX
        yq = torch.empty((N, Ho, Wo, cout), dtype=torch.uint8, device=dev)
        cv.conv_igemm_fp8(xq, wq, yq, geom, scale, shift=b.to(dev), add=rq, add_mode=1, act=cv.ACT_RELU, out_scale=sy)
        got_q = cv.fp8_dequantize(yq).permute(0, 3, 1, 2).cpu().double()
        err = (got_q - want_dq).abs()
        assert float((err - (2.0 ** -4) * want_dq.abs()).max()) <= 2.0 ** -9 * 448 * sy + 1e-4 * float(want_dq.abs().max())   # half an ulp + the subnormal step


def test_maxpool_straight_to_fp8(dev):
    """rn_maxpool_fwd_fp8out (the fp8 engine's stem boundary, D/model.py:232): max-pool then quantisation in one pass = the two-step path,
    byte for byte; odd sizes (the clipped windows at the borders)."""
    from retinanet_mi355x import conv as cv
    x = F.relu(rnd((2, 64, 37, 45), 91)).permute(0, 2, 3, 1).contiguous().to(dev)
    scale = float(x.max()) / cv.FP8_MAX
    got = cv.maxpool_fwd_fp8(x, scale)
    want = cv.fp8_quantize(cv.maxpool_fwd(x), scale)
    assert got.shape == want.shape and torch.equal(got, want)
    assert got._rn_scale == want._rn_scale
