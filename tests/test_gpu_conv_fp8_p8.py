"""The fp8 engine's eight-wave 256 x 256 x 128 kernel (csrc/conv_fp8_p8.hip, RN_OPT_FP8_P8; BASELINE configs[4]) against the kernel it
replaces (csrc/conv_fp8.hip, which tests/test_gpu_conv_fp8.py pins against torch's CPU convolution bit for bit):

* operands that ARE e4m3 values (small integers and halves): every product and sum is exact in fp32, so both kernels hand the same
  fp32 value to the same epilogue -- the e4m3 results must be IDENTICAL, bytes for bytes (the lane map of the 16x16x128 MFMA, the
  staging, the new chunk permutation, the two-level lane exchange of the epilogue, the padding bits: no tolerance);
* random operands, bias + residual + ReLU, e4m3 result: against the fp64 convolution of the dequantised operands within one e4m3
  rounding (the bound of tests/test_gpu_conv_fp8.py), and against the other kernel within one e4m3 step;
* the grouped (pyramid) launch and the launcher's choice."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def cv(dev):
    from retinanet_mi355x import conv
    old = conv.get_option(conv.OPT_FP8_P8)
    conv.set_option(conv.OPT_FP8_P8, 2)              # wherever the kernel is legal
    yield conv
    conv.set_option(conv.OPT_FP8_P8, old)


def rnd(shape, seed, std=1.0):
    from retinanet_mi355x import synth
    return torch.from_numpy(synth.normal(shape, seed, std))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def launcher_tile(cv, x_shape, cout, k, stride, pad):
    from retinanet_mi355x import _hip
    N, H, W, cin = x_shape
    Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
    d = cv._make_desc(torch.empty(x_shape, device="meta"), (Ho, Wo, cout, k, k, stride, 1, -pad, 0), 0, 0, (0, 0), 0, False, None, None, None, None)
    return _hip.load().rn_conv_igemm_fp8_tile(ctypes.byref(d), 0)


CASES = [  # cin, cout, k, pad, N, H, W
    (256, 256, 3, 1, 1, 19, 23),         # 437 rows: a full and a ragged row tile
    (128, 320, 3, 1, 2, 9, 15),          # one K-tile per tap; a ragged second channel tile; an image boundary inside the tile
    (1024, 256, 1, 0, 2, 9, 11),         # 1x1, eight K-tiles
    (384, 64, 3, 1, 1, 16, 16),          # three K-tiles per tap (not a power of two); a quarter of a channel tile
    (64, 256, 1, 0, 2, 15, 17),          # Cin = 64: ONE K-tile whose second half lies past the reduction (zero-filled on both sides)
    (64, 64, 3, 1, 2, 13, 17),           # Cin = 64, 3x3: the halves of a K-tile are two different taps; 4.5 K-tiles
    (192, 128, 3, 1, 1, 9, 15),          # Cin = 192: the K-tiles straddle the taps at alternating positions
]


@pytest.mark.parametrize("case", CASES)
def test_identical_to_the_other_kernel_on_exact_operands(cv, dev, case):
    cin, cout, k, pad, N, H, W = case
    x = torch.round(rnd((N, cin, H, W), 1, 2.0)).clamp(-4, 4)
    w = (torch.round(rnd((cout, cin, k, k), 2, 2.0)) / 2).clamp(-2, 2)
    res = torch.round(rnd((N, cout, H, W), 3, 2.0)).clamp(-4, 4)
    b = torch.round(rnd((cout,), 4, 2.0))
    xq = cv.fp8_quantize(nhwc(x).to(dev), 1.0)
    rq = cv.fp8_quantize(nhwc(res).to(dev), 1.0)
    wp = cv.pack_weights(w.to(dev), 0, presplit=False)
    wq = cv.fp8_quantize(wp, 1.0)                      # values are e4m3 already; K % 128 == 0: no row padding
    geom = (H, W, cout, k, k, 1, 1, -pad, 0)
    scale = torch.full((cout,), 2.0 ** -6, device=dev)  # |sum| < 2^12: the scaled result stays in e4m3's range, rounding happens
    outs = []
    for mode in (2, 0):
        cv.set_option(cv.OPT_FP8_P8, mode)
        assert launcher_tile(cv, xq.shape, cout, k, 1, pad) == (256256 if mode else 128128)
        y1 = torch.zeros((N, H, W, cout), dtype=torch.uint8, device=dev)
        cv.conv_igemm_fp8(xq, wq, y1, geom, scale, out_scale=1.0)
        y2 = torch.zeros((N, H, W, cout), dtype=torch.uint8, device=dev)
        cv.conv_igemm_fp8(xq, wq, y2, geom, scale, shift=b.to(dev), add=rq, add_mode=1, act=cv.ACT_RELU, out_scale=0.5)
        outs.append((y1, y2))
    cv.set_option(cv.OPT_FP8_P8, 2)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    # and the function itself, on the plain result: e4m3(conv * 2^-6) of exact sums
    want = F.conv2d(x.double(), w.double(), None, 1, pad) * 2.0 ** -6
    got = cv.fp8_dequantize(outs[0][0], 1.0).permute(0, 3, 1, 2).cpu().double()
    assert float((got - want).abs().max()) <= 2.0 ** -4 * float(want.abs().max())
    assert float(got.abs().max()) > 0


@pytest.mark.parametrize("case", CASES[:3] + CASES[4:6])
def test_random_operands_bias_relu_residual(cv, dev, case):
    cin, cout, k, pad, N, H, W = case
    x = F.relu(rnd((N, cin, H, W), 5))
    w = rnd((cout, cin, k, k), 6, (2.0 / (k * k * cin)) ** 0.5)
    b = rnd((cout,), 7, 0.1)
    sx = float(x.abs().max()) / cv.FP8_MAX
    xq = cv.fp8_quantize(nhwc(x).to(dev), sx)
    wq, sw = cv.fp8_quantize_weights(cv.pack_weights(w.to(dev), 0, presplit=False))
    x_dq = cv.fp8_dequantize(xq).cpu().permute(0, 3, 1, 2)
    w_dq = (cv.fp8_dequantize(wq, 1.0)[:, :k * k * cin] * sw[:, None]).cpu().reshape(cout, k, k, cin).permute(0, 3, 1, 2)
    res = F.relu(rnd((N, cout, H, W), 8))
    sr = float(res.abs().max()) / cv.FP8_MAX
    rq = cv.fp8_quantize(nhwc(res).to(dev), sr)
    res_dq = cv.fp8_dequantize(rq).cpu().permute(0, 3, 1, 2)
    want_dq = F.relu(F.conv2d(x_dq.double(), w_dq.double(), b.double(), 1, pad) + res_dq.double())
    scale = (sw * sx).contiguous()
    geom = (H, W, cout, k, k, 1, 1, -pad, 0)
    sy = float(want_dq.abs().max()) / cv.FP8_MAX
    yq = torch.empty((N, H, W, cout), dtype=torch.uint8, device=dev)
    cv.conv_igemm_fp8(xq, wq, yq, geom, scale, shift=b.to(dev), add=rq, add_mode=1, act=cv.ACT_RELU, out_scale=sy)
    got = cv.fp8_dequantize(yq).permute(0, 3, 1, 2).cpu().double()
    err = (got - want_dq).abs()
    assert float((err - (2.0 ** -4) * want_dq.abs()).max()) <= 2.0 ** -9 * 448 * sy + 1e-4 * float(want_dq.abs().max())
    cv.set_option(cv.OPT_FP8_P8, 0)
    y0 = torch.empty((N, H, W, cout), dtype=torch.uint8, device=dev)
    cv.conv_igemm_fp8(xq, wq, y0, geom, scale, shift=b.to(dev), add=rq, add_mode=1, act=cv.ACT_RELU, out_scale=sy)
    cv.set_option(cv.OPT_FP8_P8, 2)
    got0 = cv.fp8_dequantize(y0).permute(0, 3, 1, 2).cpu().double()
    # the two kernels add the same products in different orders: results differ by at most one e4m3 step, in a small share of the elements
    assert float(((got - got0).abs() - (2.0 ** -3) * got0.abs()).max()) <= 2.0 ** -9 * 448 * sy
    assert float((got != got0).double().mean()) < 0.02


def test_grouped_pyramid_launch_and_the_default_rule(cv, dev):
    cin = cout = 256
    w = (torch.round(rnd((cout, cin, 3, 3), 11, 2.0)) / 2).clamp(-2, 2)
    wq = cv.fp8_quantize(cv.pack_weights(w.to(dev), 0, presplit=False), 1.0)
    scale = torch.full((cout,), 2.0 ** -6, device=dev)
    sizes = [(2, 34, 60), (2, 17, 30), (2, 9, 15), (2, 5, 8), (2, 3, 4)]
    xs = [cv.fp8_quantize(nhwc(torch.round(rnd((n, cin, h, ww), 12 + i, 2.0)).clamp(-4, 4)).to(dev), 1.0) for i, (n, h, ww) in enumerate(sizes)]
    outs = []
    for mode in (2, 0):
        cv.set_option(cv.OPT_FP8_P8, mode)
        ys = [torch.zeros((n, h, ww, cout), dtype=torch.uint8, device=dev) for (n, h, ww) in sizes]
        problems = [dict(x=x, y=y, geom=(x.shape[1], x.shape[2], cout, 3, 3, 1, 1, -1, 0)) for x, y in zip(xs, ys)]
        cv.conv_igemm_fp8_grouped(problems, wq, scale, act=cv.ACT_RELU, out_scale=1.0)
        outs.append(ys)
    for a, b in zip(*outs):
        assert torch.equal(a, b) and int(a.max()) > 0
    cv.set_option(cv.OPT_FP8_P8, 1)                  # the default rule: wherever the kernel is legal
    assert launcher_tile(cv, (16, 135, 240, 256), 256, 3, 1, 1) == 256256
    assert launcher_tile(cv, (16, 68, 120, 1024), 256, 1, 1, 0) == 256256
    assert launcher_tile(cv, (16, 135, 240, 256), 80, 3, 1, 1) == 256256
    assert launcher_tile(cv, (16, 68, 120, 256), 256, 3, 2, 1) == 256256      # strided: the general-geometry instances
    assert launcher_tile(cv, (16, 270, 480, 64), 256, 1, 1, 0) == 256256       # Cin = 64: half K-tiles
    assert launcher_tile(cv, (16, 68, 120, 32), 256, 3, 1, 1) == 128128       # Cin not a multiple of 64
    cv.set_option(cv.OPT_FP8_P8, 2)


@pytest.mark.parametrize("case", [(64, 256, 1, 0, 2, 270, 261), (128, 320, 3, 1, 1, 259, 257), (256, 256, 3, 1, 3, 150, 301)])
def test_persistent_form_identical_on_exact_operands(cv, dev, case):
    """Launches with at least two tiles per CU run as persistent workgroups that stage the next tile's first K-tiles before the current
    tile's epilogue (conv_fp8_p8.hip: PERSIST): 550 / 522 / 530 tiles here, ragged last row tiles, a ragged channel tile -- byte-identical
    to the 128 x 128 kernel on exact operands, residual + ReLU epilogue included."""
    cin, cout, k, pad, N, H, W = case
    g = torch.Generator().manual_seed(77)
    x = torch.randint(-4, 5, (N, H, W, cin), generator=g).float()
    w = (torch.randint(-4, 5, (cout, cin, k, k), generator=g).float() / 2)
    res = torch.randint(-4, 5, (N, H, W, cout), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    xq, rq = cv.fp8_quantize(x.to(dev), 1.0), cv.fp8_quantize(res.to(dev), 1.0)
    wq = cv.fp8_quantize(cv.pack_weights(w.to(dev), 0, presplit=False), 1.0)
    geom = (H, W, cout, k, k, 1, 1, -pad, 0)
    scale = torch.full((cout,), 2.0 ** -6, device=dev)
    outs = []
    for mode in (2, 0):
        cv.set_option(cv.OPT_FP8_P8, mode)
        y = torch.zeros((N, H, W, cout), dtype=torch.uint8, device=dev)
        cv.conv_igemm_fp8(xq, wq, y, geom, scale, shift=b.to(dev), add=rq, add_mode=1, act=cv.ACT_RELU, out_scale=0.5)
        outs.append(y)
    cv.set_option(cv.OPT_FP8_P8, 2)
    assert torch.equal(outs[0], outs[1])
    assert 0.2 < float((outs[0] > 0).float().mean()) < 0.95


@pytest.mark.parametrize("case", [(256, 512, 1, 2, 0, 2, 18, 22), (128, 128, 3, 2, 1, 2, 21, 18), (64, 144, 1, 2, 0, 3, 15, 17),
                                  (64, 64, 3, 2, 1, 1, 17, 19), (128, 256, 3, 1, 0, 2, 12, 14)])
def test_strided_and_valid_geometries_identical_on_exact_operands(cv, dev, case):
    """The general geometry (a per-lane byte offset for each of the lane's four pixel rows instead of one flat sequence): stride-2 1x1
    shortcuts and 3x3 layers, Cin = 64 halves, an unpadded 3x3 -- byte-identical to the 128 x 128 kernel, and the function itself against
    torch's fp64 convolution of the same exact operands."""
    cin, cout, k, stride, pad, N, H, W = case
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-4, 5, (N, cin, H, W), generator=g).float()
    w = torch.randint(-4, 5, (cout, cin, k, k), generator=g).float() / 2
    Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
    xq = cv.fp8_quantize(nhwc(x).to(dev), 1.0)
    wq = cv.fp8_quantize(cv.pack_weights(w.to(dev), 0, presplit=False), 1.0)
    geom = (Ho, Wo, cout, k, k, stride, 1, -pad, 0)
    scale = torch.full((cout,), 2.0 ** -6, device=dev)
    outs = []
    for mode in (2, 0):
        cv.set_option(cv.OPT_FP8_P8, mode)
        assert launcher_tile(cv, xq.shape, cout, k, stride, pad) == (256256 if mode else 128128)
        y = torch.zeros((N, Ho, Wo, cout), dtype=torch.uint8, device=dev)
        cv.conv_igemm_fp8(xq, wq, y, geom, scale, act=cv.ACT_RELU, out_scale=1.0)
        outs.append(y)
    cv.set_option(cv.OPT_FP8_P8, 2)
    assert torch.equal(outs[0], outs[1])
    want = F.relu(F.conv2d(x.double(), w.double(), None, stride, pad)) * 2.0 ** -6
    got = cv.fp8_dequantize(outs[0], 1.0).permute(0, 3, 1, 2).cpu().double()
    assert float((got - want).abs().max()) <= 2.0 ** -4 * float(want.abs().max())
    assert float(got.max()) > 0
