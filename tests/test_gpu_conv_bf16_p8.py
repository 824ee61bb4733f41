"""The bf16 engine's eight-wave 256 x 256 x 64 kernel (csrc/conv_bf16_p8.hip, RN_OPT_BF16_P8) against torch's fp32 convolution of the
same bf16-rounded operands (the oracle of tests/test_gpu_conv_bf16.py) and against the 128 x 128 kernel it replaces: forward with the
fused epilogue (D/model.py:59-205: folded batch-norm, residual add, ReLU), the stride-1 data gradient with its ReLU mask (tensor and
sign bits), the sign bits it writes, ragged row / channel tiles, images inside and across tiles, the grouped (pyramid) launch.
Tolerance: a bf16 rounding of the result (2^-8 relative) + 1e-3 of the tensor's largest value, as for the kernel it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture()
def cv(dev):
    from retinanet_mi355x import conv
    old = conv.get_option(conv.OPT_BF16_P8)
    conv.set_option(conv.OPT_BF16_P8, 2)             # wherever the kernel is legal
    yield conv
    conv.set_option(conv.OPT_BF16_P8, old)


def rnd(shape, seed, std=1.0):
    from retinanet_mi355x import synth
    return torch.from_numpy(synth.normal(shape, seed, std))


def r16(t):
    return t.bfloat16().float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def packed_sign(y):
    """(y > 0) of a dense tensor packed as the kernels pack it: bit (e & 31) of word (e >> 5)."""
    b = (y.reshape(-1) > 0).to(torch.int64).reshape(-1, 32)
    w = (b << torch.arange(32, device=y.device)).sum(1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def close_bf16(got, want):
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    assert got.shape == want.shape
    bound = want.abs() * 2.0 ** -8 + 1e-3 * float(want.abs().max())
    bad = (got - want).abs() > bound
    assert not bool(bad.any()), "%d elements off by more than a bf16 rounding, worst %.3e" % (int(bad.sum()), float((got - want).abs().max()))


CASES = [  # cin, cout, k, pad, N, H, W
    (64, 256, 3, 1, 2, 19, 23),          # 874 rows: three full row tiles + a ragged one; an image boundary inside a tile
    (128, 64, 3, 1, 1, 16, 16),          # exactly one row tile, a quarter of a channel tile
    (256, 512, 1, 0, 2, 15, 17),         # 1x1, two channel tiles
    (64, 72, 3, 1, 3, 5, 7),             # Cout % 8 == 0 only; three images inside one tile
    (192, 264, 3, 1, 1, 9, 15),          # Cin = 3 K-tiles per tap (not a power of two); a ragged second channel tile
    (64, 256, 3, 2, 1, 12, 14),          # pad 2 > (k - 1) / 2 is not same-size: must fall back to the other kernel and still be right
]


@pytest.mark.parametrize("case", CASES)
def test_forward_with_epilogue(cv, dev, case):
    cin, cout, k, pad, N, H, W = case
    x, w = rnd((N, cin, H, W), 1), rnd((cout, cin, k, k), 2, (2.0 / (k * k * cin)) ** 0.5)
    scale, shift = rnd((cout,), 3, 0.3) + 1.0, rnd((cout,), 4, 0.2)
    xb = cv.to_bf16(nhwc(x).to(dev))
    wp = cv.pack_weights_bf16(w.to(dev), 0)
    want = F.conv2d(r16(x), r16(w), None, 1, pad)
    y = cv.fprop_bf16(xb, wp, cout, k, 1, pad)
    close_bf16(nchw(y.float()), want)
    res = rnd(tuple(want.shape), 5)
    resb = cv.to_bf16(nhwc(res).to(dev))
    want2 = F.relu(want * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + r16(res))
    y2 = cv.fprop_bf16(xb, wp, cout, k, 1, pad, scale=scale.to(dev), shift=shift.to(dev), add=resb, add_mode=1, act=cv.ACT_RELU)
    close_bf16(nchw(y2.float()), want2)
    # the kernel it replaces, on the same operands: both round one fp32 value per element, whose sums differ in order only
    cv.set_option(cv.OPT_BF16_P8, 0)
    y2_old = cv.fprop_bf16(xb, wp, cout, k, 1, pad, scale=scale.to(dev), shift=shift.to(dev), add=resb, add_mode=1, act=cv.ACT_RELU)
    cv.set_option(cv.OPT_BF16_P8, 2)
    close_bf16(y2.float(), y2_old.float())


def launcher_tile(cv, x_shape, cout, k, stride, pad):
    """The tile the launcher takes for this layer (rn_conv_igemm_bf16_tile_rows on a group of one): 256256 = this kernel (the other
    256 x 256 tile left the library in round 5), 128128 / 256128 = conv_bf16.hip's."""
    import ctypes
    from retinanet_mi355x import _hip
    N, H, W, cin = x_shape
    Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
    d = cv._make_desc(torch.empty(x_shape, device="meta"), (Ho, Wo, cout, k, k, stride, 1, -pad, 0), 0, 0, (0, 0), 0, False, None, None, None, None)
    return _hip.load().rn_conv_igemm_bf16_tile(ctypes.byref(d), 0) % 1000000


def test_kernel_is_the_one_that_ran(cv, dev):
    """Forced and never are two different kernels on a legal shape (the launcher's tile says which; the results agree to a rounding),
    and a shape the kernel refuses gives bit-identical results in both settings."""
    def both(cin, cout, k, pad, stride=1):
        x, w = rnd((2, cin, 20, 24), 6), rnd((cout, cin, k, k), 7, (2.0 / (k * k * cin)) ** 0.5)
        xb, wp = cv.to_bf16(nhwc(x).to(dev)), cv.pack_weights_bf16(w.to(dev), 0)
        a, ta = cv.fprop_bf16(xb, wp, cout, k, stride, pad), launcher_tile(cv, xb.shape, cout, k, stride, pad)
        cv.set_option(cv.OPT_BF16_P8, 0)
        b, tb = cv.fprop_bf16(xb, wp, cout, k, stride, pad), launcher_tile(cv, xb.shape, cout, k, stride, pad)
        cv.set_option(cv.OPT_BF16_P8, 2)
        return a, b, ta, tb
    a, b, ta, tb = both(256, 256, 3, 1)
    assert (ta, tb) == (256256, 128128)
    close_bf16(a.float(), b.float())
    a, b, ta, tb = both(256, 256, 3, 1, stride=2)    # strided: not this kernel's
    assert ta == tb == 128128 and torch.equal(a, b)
    a, b, ta, tb = both(40, 256, 3, 1)               # Cin not a multiple of 64
    assert ta == tb == 128128 and torch.equal(a, b)
    cv.set_option(cv.OPT_BF16_P8, 1)                 # the default rule: full 256-channel tiles, enough of them
    assert launcher_tile(cv, (8, 135, 240, 256), 256, 3, 1, 1) == 256256
    assert launcher_tile(cv, (8, 135, 240, 256), 72, 3, 1, 1) != 256256
    assert launcher_tile(cv, (1, 20, 24, 256), 256, 3, 1, 1) != 256256


@pytest.mark.parametrize("bits", [False, True])
@pytest.mark.parametrize("case", [(64, 256, 3, 1, 2, 19, 23), (256, 64, 1, 0, 2, 15, 17), (128, 128, 3, 1, 1, 9, 32)])
def test_data_gradient_with_relu_mask(cv, dev, case, bits):
    cin, cout, k, pad, N, H, W = case
    x, w = rnd((N, cin, H, W), 11), rnd((cout, cin, k, k), 12, (2.0 / (k * k * cin)) ** 0.5)
    xr, wr = r16(x).requires_grad_(True), r16(w)
    yy = F.conv2d(xr, wr, None, 1, pad)
    g = rnd(tuple(yy.shape), 13)
    yy.backward(r16(g))
    gb = cv.to_bf16(nhwc(g).to(dev))
    wd = cv.pack_weights_bf16(w.to(dev), 1)
    z = rnd((N, cin, H, W), 14)
    zb = cv.to_bf16(nhwc(z).to(dev))
    if bits:
        if cin % 32:
            pytest.skip("sign bits need 32-channel words")
        words = packed_sign(zb.float())
        zb._rn_sign = words
        old = cv.BITMASKS, cv.BITMASKS_BF16
        cv.BITMASKS = cv.BITMASKS_BF16 = True
        try:
            dx = cv.dgrad_bf16(gb, wd, (H, W), cin, k, pad, mask=zb, mask_mode=2)
        finally:
            cv.BITMASKS, cv.BITMASKS_BF16 = old
    else:
        dx = cv.dgrad_bf16(gb, wd, (H, W), cin, k, pad, mask=zb, mask_mode=2)
    want = xr.grad * (r16(z) > 0)
    close_bf16(nchw(dx.float()), want)


def test_sign_bits_of_the_stored_result(cv, dev):
    cin, cout, N, H, W = 64, 320, 2, 13, 21              # two channel tiles, the second a quarter full; 546 rows
    x, w = rnd((N, cin, H, W), 21), rnd((cout, cin, 3, 3), 22, (2.0 / (9 * cin)) ** 0.5)
    xb, wp = cv.to_bf16(nhwc(x).to(dev)), cv.pack_weights_bf16(w.to(dev), 0)
    old = cv.BITMASKS, cv.BITMASKS_BF16
    cv.BITMASKS = cv.BITMASKS_BF16 = True
    try:
        y = cv.fprop_bf16(xb, wp, cout, 3, 1, 1, act=cv.ACT_RELU, sign=True)
    finally:
        cv.BITMASKS, cv.BITMASKS_BF16 = old
    words = getattr(y, "_rn_sign", None)
    assert words is not None
    assert torch.equal(words, packed_sign(y.float()))
    assert 0.2 < float((y.float() > 0).float().mean()) < 0.8


def test_grouped_pyramid_launch(cv, dev):
    """The five levels of a head layer as one grid (D/model.py:110-205): every level against its own single launch of the other kernel."""
    cin = cout = 256
    w = rnd((cout, cin, 3, 3), 31, (2.0 / (9 * cin)) ** 0.5)
    wp = cv.pack_weights_bf16(w.to(dev), 0)
    shift = rnd((cout,), 32, 0.1).to(dev)
    sizes = [(2, 34, 60), (2, 17, 30), (2, 9, 15), (2, 5, 8), (2, 3, 4)]
    xs = [cv.to_bf16(nhwc(rnd((n, cin, h, ww), 33 + i)).to(dev)) for i, (n, h, ww) in enumerate(sizes)]
    ys = [torch.empty((n, h, ww, cout), dtype=torch.bfloat16, device=dev) for (n, h, ww) in sizes]
    problems = [dict(x=x, y=y, geom=(x.shape[1], x.shape[2], cout, 3, 3, 1, 1, -1, 0)) for x, y in zip(xs, ys)]
    cv.conv_igemm_bf16_grouped(problems, wp, shift=shift, act=cv.ACT_RELU)
    cv.set_option(cv.OPT_BF16_P8, 0)
    for x, y in zip(xs, ys):
        ref = cv.fprop_bf16(x, wp, cout, 3, 1, 1, shift=shift, act=cv.ACT_RELU)
        close_bf16(y.float(), ref.float())
        want = F.relu(F.conv2d(nchw(x.float().cpu()), r16(w), None, 1, 1) + shift.cpu().view(1, -1, 1, 1))
        close_bf16(nchw(y.float()), want)
    cv.set_option(cv.OPT_BF16_P8, 2)


def test_dominant_layer_at_full_size(cv, dev):
    """BASELINE configs[1]'s dominant layer (8 x 135 x 240, 256 -> 256, 3x3) at size: linearity in the weights' scale and agreement with the
    128 x 128 kernel -- the size-independent checks; the oracle runs on a crop of the first image."""
    N, H, W, C = 8, 135, 240, 256
    g = torch.Generator(device="cpu").manual_seed(41)
    x = torch.randn((N, H, W, C), generator=g).to(dev)
    w = rnd((C, C, 3, 3), 42, (2.0 / (9 * C)) ** 0.5)
    xb, wp = cv.to_bf16(x), cv.pack_weights_bf16(w.to(dev), 0)
    y = cv.fprop_bf16(xb, wp, C, 3, 1, 1)
    cv.set_option(cv.OPT_BF16_P8, 0)
    y_old = cv.fprop_bf16(xb, wp, C, 3, 1, 1)
    cv.set_option(cv.OPT_BF16_P8, 2)
    close_bf16(y.float(), y_old.float())
    crop = nchw(xb[:1, :24].float().cpu())
    want = F.conv2d(crop, r16(w), None, 1, 1)[:, :, :23]       # the crop's last row lacks its lower neighbour
    close_bf16(nchw(y[:1, :23].float()), want)
    y2 = cv.fprop_bf16(xb, cv.pack_weights_bf16((2.0 * w).to(dev), 0), C, 3, 1, 1)
    assert torch.equal(y2.float(), 2.0 * y.float())            # a power of two scales every product and sum exactly
