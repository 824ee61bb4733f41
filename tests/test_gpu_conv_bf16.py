"""GPU parity of the bf16 MFMA convolution engine (rn_conv_igemm_bf16 / rn_conv_wgrad_bf16, BASELINE configs[2]) against
torch's fp32 CPU convolution of the SAME bf16-rounded operands.

What can be demanded: with both operands already rounded to bf16, every product is exact in fp32 and the kernels
accumulate in fp32, so an fp32 result differs from the fp32 reference only by summation order -- TOL_F32 = 2e-5 of the
output's max magnitude (measured ~1e-6), i.e. the kernels introduce NO error of their own.  A bf16 result is that value
rounded once: at most half a bf16 ulp = 2^-9 = 0.2 % of each element, checked as |got - ref| <= 2^-8 |ref| + 1e-3 max|ref|.
Against the fp32 oracle on UNROUNDED operands (the reference's arithmetic) the bf16 path is then off by the input
rounding itself, ~3e-3 of the max per layer: test_bf16_vs_unrounded_fp32 records that number (TOL_VS_FP32 = 2e-2).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL_F32, TOL_VS_FP32 = 2e-5, 2e-2


@pytest.fixture(scope="module")
def cv(dev):
    from retinanet_mi355x import conv
    return conv


def rnd(shape, seed, std=1.0):
    from retinanet_mi355x import synth
    return torch.from_numpy(synth.normal(shape, seed, std))


def r16(t):
    return t.bfloat16().float()                       # the value a bf16 tensor holds, as fp32 (CPU reference side)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close_f32(got, want, tol=TOL_F32):
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    assert got.shape == want.shape, (got.shape, want.shape)
    err, ref = float((got - want).abs().max()), float(want.abs().max()) + 1e-12
    assert err <= tol * ref, "max err %.3e vs max |ref| %.3e" % (err, ref)


def close_bf16(got, want):
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    assert got.shape == want.shape
    bound = want.abs() * 2.0 ** -8 + 1e-3 * float(want.abs().max())
    bad = (got - want).abs() > bound
    assert not bool(bad.any()), "%d elements off by more than a bf16 rounding, worst %.3e" % (int(bad.sum()), float((got - want).abs().max()))


CASES = [  # cin, cout, k, stride, pad, N, H, W
    (64, 256, 3, 1, 1, 2, 19, 23),
    (128, 128, 3, 2, 1, 2, 17, 21),
    (64, 256, 1, 1, 0, 2, 15, 17),
    (256, 512, 1, 2, 0, 1, 17, 15),
    (256, 108, 3, 1, 1, 1, 9, 15),       # head output: Cout % 4 == 0 only (fp32 result)
    (40, 72, 3, 1, 1, 1, 7, 9),          # Cin % 8 == 0 but not a multiple of the K-step: general staging path
    (512, 128, 3, 1, 1, 3, 5, 7),        # long K, several images inside one tile
    (64, 64, 3, 1, 1, 2, 19, 23),        # at most 64 output channels: the 256 x 64 tile (round 4), a ragged last row tile
    (256, 64, 1, 1, 0, 3, 9, 13),        # the same, 1x1, images inside the tile
    (64, 40, 3, 1, 1, 1, 17, 16),        # the same with a ragged channel tile (Cout = 40)
]


@pytest.mark.parametrize("case", CASES)
def test_fprop_bf16(cv, dev, case):
    cin, cout, k, stride, pad, N, H, W = case
    x, w = rnd((N, cin, H, W), 1), rnd((cout, cin, k, k), 2, (2.0 / (k * k * cin)) ** 0.5)
    scale, shift = rnd((cout,), 3, 0.3) + 1.0, rnd((cout,), 4, 0.2)
    xb = cv.to_bf16(nhwc(x).to(dev))
    wp = cv.pack_weights_bf16(w.to(dev), 0)
    want = F.conv2d(r16(x), r16(w), None, stride, pad)
    y32 = cv.fprop_bf16(xb, wp, cout, k, stride, pad, out_dtype=torch.float32)
    close_f32(nchw(y32), want)
    # fused epilogue: folded batch-norm, residual add, ReLU; bf16 output
    res = rnd(tuple(want.shape), 5)
    resb = cv.to_bf16(nhwc(res).to(dev))
    want2 = F.relu(want * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + r16(res))
    if cout % 8 == 0:                                  # a bf16 result moves 8 channels (16 bytes) per lane
        y2 = cv.fprop_bf16(xb, wp, cout, k, stride, pad, scale=scale.to(dev), shift=shift.to(dev), add=resb, add_mode=1,
                           act=cv.ACT_RELU)
        assert y2.dtype == torch.bfloat16
        close_bf16(nchw(y2.float()), want2)
    else:
        with pytest.raises(RuntimeError):
            cv.fprop_bf16(xb, wp, cout, k, stride, pad)
    y3 = cv.fprop_bf16(xb, wp, cout, k, stride, pad, out_dtype=torch.float32, scale=scale.to(dev), shift=shift.to(dev),
                       add=resb, add_mode=1, act=cv.ACT_RELU)
    close_f32(nchw(y3), want2)


@pytest.mark.parametrize("case", [c for c in CASES if c[3] == 1])
def test_dgrad_and_wgrad_bf16(cv, dev, case):
    cin, cout, k, stride, pad, N, H, W = case
    x, w = rnd((N, cin, H, W), 11), rnd((cout, cin, k, k), 12, (2.0 / (k * k * cin)) ** 0.5)
    xr, wr = r16(x).requires_grad_(True), r16(w).requires_grad_(True)
    y = F.conv2d(xr, wr, None, stride, pad)
    g = rnd(tuple(y.shape), 13)
    y.backward(r16(g))
    gb = cv.to_bf16(nhwc(g).to(dev))
    xb = cv.to_bf16(nhwc(x).to(dev))
    # data gradient: the same kernel under the transposed coordinate map, with a ReLU mask of the destination and an addend
    cout_pad = (cout + 7) // 8 * 8
    if cout_pad == cout:
        wd = cv.pack_weights_bf16(w.to(dev), 1)
        z = rnd((N, cin, H, W), 14)
        zb = cv.to_bf16(nhwc(z).to(dev))
        dx = cv.dgrad_bf16(gb, wd, (H, W), cin, k, pad, mask=zb, mask_mode=2)
        want = xr.grad * (r16(z) > 0)
        close_bf16(nchw(dx.float()), want)
    # weight gradient (+ column sums of dy)
    kp = (k * k * cin + 31) // 32 * 32
    if cout % 8 == 0:
        dw = torch.zeros((cout, kp), device=dev)
        cs = torch.zeros(cout, device=dev)
        cv.wgrad_bf16(gb, xb, dw, cout, k, stride, pad, colsum=cs)
        got = dw[:, :k * k * cin].view(cout, k, k, cin).permute(0, 3, 1, 2)
        close_f32(got, wr.grad, tol=5e-5)
        close_f32(cs, r16(g).sum(dim=(0, 2, 3)), tol=5e-5)
        assert float(dw[:, k * k * cin:].abs().max()) == 0.0 if kp > k * k * cin else True


def test_wgrad_bf16_strided_and_many_slices(cv, dev):
    """Stride 2 (bottleneck conv2 of the first block of a layer) and a K range long enough for many slices and several
    pixel-table batches per slice."""
    cin, cout, k, stride, pad, N, H, W = 64, 128, 3, 2, 1, 2, 61, 75
    x, w = rnd((N, cin, H, W), 21), rnd((cout, cin, k, k), 22, 0.05)
    xr, wr = r16(x).requires_grad_(True), r16(w).requires_grad_(True)
    y = F.conv2d(xr, wr, None, stride, pad)
    g = rnd(tuple(y.shape), 23)
    y.backward(r16(g))
    dw = torch.zeros((cout, k * k * cin), device=dev)
    cv.wgrad_bf16(cv.to_bf16(nhwc(g).to(dev)), cv.to_bf16(nhwc(x).to(dev)), dw, cout, k, stride, pad)
    close_f32(dw.view(cout, k, k, cin).permute(0, 3, 1, 2), wr.grad, tol=5e-5)


def test_head_output_slice_fp32_with_sigmoid(cv, dev):
    """A classification output layer writing fp32 sigmoid values straight into its slice of the concatenated [B, A, C]
    tensor (D/model.py:196-205), from bf16 activations."""
    B, H, W, C = 2, 9, 15, 72
    x, w, b = rnd((B, 256, H, W), 31), rnd((C, 256, 3, 3), 32, 0.02), rnd((C,), 33, 0.5)
    A = H * W * 9 + 100
    out = torch.full((B, A, 8), -7.0, device=dev)
    view = out.view(B, -1)[:, 50 * 8:]
    cv.conv_igemm_bf16(cv.to_bf16(nhwc(x).to(dev)), cv.pack_weights_bf16(w.to(dev), 0), view, (H, W, C, 3, 3, 1, 1, -1, 0),
                       shift=b.to(dev), act=cv.ACT_SIGMOID, y_batch_stride=A * 8)
    want = torch.sigmoid(F.conv2d(r16(x), r16(w), b, 1, 1)).permute(0, 2, 3, 1).reshape(B, -1, 8)
    close_f32(out[:, 50:50 + H * W * 9], want)
    assert float(out[:, :50].max()) == -7.0 and float(out[:, 50 + H * W * 9:].max()) == -7.0


def test_upsample_add_epilogue_bf16(cv, dev):
    """FPN lateral 1x1 with the nearest-upsampled, cropped coarser map added in the epilogue (D/model.py:88-108)."""
    x, w = rnd((1, 512, 9, 13), 41), rnd((256, 512, 1, 1), 42, 0.05)
    coarse = rnd((1, 256, 5, 7), 43)
    y = cv.fprop_bf16(cv.to_bf16(nhwc(x).to(dev)), cv.pack_weights_bf16(w.to(dev), 0), 256, 1, 1, 0, out_dtype=torch.float32,
                      add=cv.to_bf16(nhwc(coarse).to(dev)), add_mode=2, add_hw=(5, 7))
    up = F.interpolate(r16(coarse), scale_factor=2, mode="nearest")[:, :, :9, :13]
    close_f32(nchw(y), F.conv2d(r16(x), r16(w)) + up)


def test_bf16_vs_unrounded_fp32(cv, dev):
    """How far the bf16 path is from the reference's fp32 arithmetic on one layer (operand rounding + output rounding)."""
    x, w = rnd((2, 256, 17, 19), 51), rnd((256, 256, 3, 3), 52, (2.0 / 2304) ** 0.5)
    y = cv.fprop_bf16(cv.to_bf16(nhwc(x).to(dev)), cv.pack_weights_bf16(w.to(dev), 0), 256, 3, 1, 1)
    want = F.conv2d(x, w, None, 1, 1)
    err = float((nchw(y.float()).cpu() - want).abs().max()) / float(want.abs().max())
    assert err <= TOL_VS_FP32, err
    print("bf16 conv vs fp32 reference: max err %.2e of max |y|" % err)


def test_casts_round_to_nearest_even_and_keep_nan(cv, dev):
    v = torch.tensor([1.0, 1.00390625, 1.01171875, -3.14159, 65504.0, 1e-40, float("nan"), float("inf")], device=dev)
    got = cv.to_bf16(v)
    assert torch.equal(got[:6].cpu(), v[:6].cpu().bfloat16())
    assert bool(torch.isnan(got[6])) and bool(torch.isinf(got[7]))
    assert torch.equal(cv.to_f32(got)[:6].cpu(), v[:6].cpu().bfloat16().float())


def test_bf16_dominant_layer_1080p(cv, dev):
    """The dominant layer of the benchmark (3x3 256->256 at 135x240, one image) forward, data and weight gradient."""
    N, C, H, W = 1, 256, 135, 240
    x, w = rnd((N, C, H, W), 61), rnd((C, C, 3, 3), 62, (2.0 / 2304) ** 0.5)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    xr, wr = r16(x).requires_grad_(True), r16(w).requires_grad_(True)
    y = F.conv2d(xr, wr, None, 1, 1)
    g = rnd(tuple(y.shape), 63)
    y.backward(r16(g))
    xb, gb = cv.to_bf16(nhwc(x).to(dev)), cv.to_bf16(nhwc(g).to(dev))
    close_f32(nchw(cv.fprop_bf16(xb, cv.pack_weights_bf16(w.to(dev), 0), C, 3, 1, 1, out_dtype=torch.float32)), y)
    dx = cv.conv_igemm_bf16(gb, cv.pack_weights_bf16(w.to(dev), 1), torch.empty((N, H, W, C), device=dev), (H, W, C, 3, 3, 1, -1, 1, 0))
    close_f32(nchw(dx), xr.grad)
    dw = torch.zeros((C, 9 * C), device=dev)
    cv.wgrad_bf16(gb, xb, dw, C, 3, 1, 1)
    close_f32(dw.view(C, 3, 3, C).permute(0, 3, 1, 2), wr.grad, tol=1e-4)


def test_grouped_weight_gradient_over_pyramid_levels(cv, dev):
    """rn_conv_wgrad_bf16_grouped: the levels of a head layer share one weight tensor, so its gradient is the SUM over the levels
    (D/model.py:110-205) -- one launch against torch's autograd over all levels and against the per-level launches."""
    cin, cout, ld, k, N = 64, 72, 80, 3, 2                      # dy rows wider than Cout (the head outputs' padded gradient slices)
    sizes = [(19, 23), (10, 12), (5, 6), (3, 3), (2, 2)]
    w = rnd((cout, cin, k, k), 31, 0.05)
    wr = r16(w).requires_grad_(True)
    xs, gs = [], []
    for i, (H, W) in enumerate(sizes):
        x, g = rnd((N, cin, H, W), 32 + i), rnd((N, ld, H, W), 42 + i)
        F.conv2d(r16(x), wr, None, 1, 1).backward(r16(g)[:, :cout])
        xs.append(cv.to_bf16(nhwc(x).to(dev)))
        gs.append(cv.to_bf16(nhwc(g).to(dev)))
    kp = (k * k * cin + 31) // 32 * 32
    dw, cs = torch.zeros((cout, kp), device=dev), torch.zeros(cout, device=dev)
    cv.wgrad_bf16_grouped(gs, xs, dw, cout, k, 1, 1, colsum=cs)
    dw1, cs1 = torch.zeros((cout, kp), device=dev), torch.zeros(cout, device=dev)
    for g, x in zip(gs, xs):
        cv.wgrad_bf16(g, x, dw1, cout, k, 1, 1, colsum=cs1)
    close_f32(dw, dw1, tol=2e-5)
    close_f32(cs, cs1, tol=2e-5)
    close_f32(dw[:, :k * k * cin].view(cout, k, k, cin).permute(0, 3, 1, 2), wr.grad, tol=5e-5)
    want_cs = sum(r16(nchw(g.float().cpu()))[:, :cout].sum(dim=(0, 2, 3)) for g in gs)
    close_f32(cs, want_cs, tol=5e-5)
    # a 256 -> 256 layer at larger levels: the K-slice placement by XCD (tiles >= 4, >= 8 slices) inside a grouped grid
    cin = cout = 256
    sizes = [(34, 60), (17, 30), (9, 15)]
    xs = [cv.to_bf16(nhwc(rnd((N, cin, H, W), 52 + i)).to(dev)) for i, (H, W) in enumerate(sizes)]
    gs = [cv.to_bf16(nhwc(rnd((N, cout, H, W), 62 + i)).to(dev)) for i, (H, W) in enumerate(sizes)]
    dw, dw1 = torch.zeros((cout, 9 * cin), device=dev), torch.zeros((cout, 9 * cin), device=dev)
    cv.wgrad_bf16_grouped(gs, xs, dw, cout, 3, 1, 1)
    for g, x in zip(gs, xs):
        cv.wgrad_bf16(g, x, dw1, cout, 3, 1, 1)
    close_f32(dw, dw1, tol=2e-5)


@pytest.mark.parametrize("with_add", [False, True])
def test_stride2_shortcut_gradient_on_the_output_grid(cv, dev, with_add):
    """The bf16 engine's data gradient of a 1x1 stride-2 shortcut (D/model.py:265-270; engine.py: Layer.bwd_data): one launch on the
    OUTPUT grid stored at the even input positions of a zeroed tensor -- or of the addend, in place -- against the generic form that
    tries the tap at every input pixel: the same products in the same order, so the two must agree bit for bit; odd sizes included."""
    cin, cout, N, H, W = 64, 128, 2, 19, 23
    Ho, Wo = cv.out_size(H, 1, 2, 0), cv.out_size(W, 1, 2, 0)
    w = rnd((cout, cin, 1, 1), 71, 0.1)
    g = cv.to_bf16(nhwc(rnd((N, cout, Ho, Wo), 72)).to(dev))
    add = cv.to_bf16(nhwc(rnd((N, cin, H, W), 73)).to(dev)) if with_add else None
    wd = cv.pack_weights_bf16(w.to(dev), 1)
    kw = dict(add=add, add_mode=1) if with_add else {}
    want = cv.dgrad_any_bf16(g, wd, (H, W), cin, 1, 2, 0, **kw)
    dx = add.clone() if with_add else torch.zeros((N, H, W, cin), dtype=torch.bfloat16, device=dev)
    cv.conv_igemm_bf16(g, wd, dx, (Ho, Wo, cin, 1, 1, 1, -1, 0, 0), out_map=(2, 0, 0, H, W), add=dx if with_add else None,
                       add_mode=1 if with_add else 0)
    assert torch.equal(dx, want)
    ref = F.conv_transpose2d(r16(nchw(g.float().cpu())), r16(w), stride=2, output_padding=(H - 1 - 2 * (Ho - 1), W - 1 - 2 * (Wo - 1)))
    if with_add:
        ref = ref + nchw(add.float().cpu())
    close_bf16(nchw(dx.float()), ref)


@pytest.mark.parametrize("case", [(4, 64, 7, 2, 3, 2, 45, 61), (4, 64, 7, 2, 3, 1, 128, 96), (32, 48, 3, 1, 1, 2, 17, 19)])
def test_fp32_tensors_with_bf16_products(cv, dev, case):
    """rn_conv_desc.w_format 2 (the fp32 stem of the bf16 / fp8 engines, D/model.py:208-232): fp32 activations and packed fp32 weights,
    products formed from their FIRST bf16 terms only -- exactly the fp32 convolution of the bf16-rounded operands (products of bf16 values
    are exact in fp32, only the summation order differs: the 2e-5 of the other fp32-result tests), with the folded batch norm and ReLU."""
    cin, cout, k, stride, pad, N, H, W = case
    x, w = rnd((N, cin, H, W), 81), rnd((cout, cin, k, k), 82, (2.0 / (k * k * cin)) ** 0.5)
    scale, shift = rnd((cout,), 83, 0.3) + 1.0, rnd((cout,), 84, 0.2)
    kw_pad = 8 if k == 7 else k
    wp = cv.pack_weights(w.to(dev), 0, kw_pad=kw_pad, c_pad=(cin + 3) // 4 * 4)
    Ho, Wo = cv.out_size(H, k, stride, pad), cv.out_size(W, k, stride, pad)
    y = torch.empty((N, Ho, Wo, cout), device=dev)
    cv.conv_igemm(nhwc(x).to(dev), wp, y, (Ho, Wo, cout, k, kw_pad, stride, 1, -pad, 0), scale=scale.to(dev), shift=shift.to(dev),
                  act=cv.ACT_RELU, bf16_products=True)
    want = F.relu(F.conv2d(r16(x), r16(w), None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    close_f32(nchw(y), want)
    # and it is NOT the fp32 path's result: the operands' second and third terms are missing
    y32 = torch.empty_like(y)
    cv.conv_igemm(nhwc(x).to(dev), wp, y32, (Ho, Wo, cout, k, kw_pad, stride, 1, -pad, 0), scale=scale.to(dev), shift=shift.to(dev), act=cv.ACT_RELU)
    assert float((y - y32).abs().max()) > 1e-4 * float(y32.abs().max())
