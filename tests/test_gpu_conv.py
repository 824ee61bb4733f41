"""GPU parity of the fp32 MFMA convolution engine (C ABI: rn_conv_igemm / rn_conv_wgrad / pack / elementwise)
against torch's fp32 CPU kernels (the same third-party arithmetic the reference runs on).

Tolerance: 1e-4 of the output's max magnitude (north_star: fp32 within 1e-4); achieved ~1e-6 -- the MFMA is an
exact fp32 fma chain, only the summation order differs.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["native", "split", "split-in-kernel", "split3"])
def cv(dev, request):
    """Every test of this module runs in all product modes of the fp32 kernels (include/retinanet_mi355x.h:
    RN_FP32_NATIVE / RN_FP32_SPLIT / RN_FP32_SPLIT3), the split mode with the weights' terms prepared by rn_split_weights (w_format 1)
    and with both operands split inside the kernels, split3 with the fp16 two-term kernels wherever they exist -- against the same
    fp64 / fp32 references with the same tolerances."""
    from retinanet_mi355x import conv
    before = conv.get_fp32_mfma(), conv.PRESPLIT
    conv.set_fp32_mfma(request.param.split("-")[0])
    conv.PRESPLIT = request.param in ("split", "split3")
    yield conv
    conv.set_fp32_mfma(before[0])
    conv.PRESPLIT = before[1]


def rnd(shape, seed, std=1.0):
    from retinanet_mi355x import synth
    return torch.from_numpy(synth.normal(shape, seed, std))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(got, want, tol=1e-4):
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = float((got - want).abs().max())
    ref = float(want.abs().max()) + 1e-12
    assert err <= tol * ref, "max err %.3e vs max |ref| %.3e" % (err, ref)


FWD_CASES = [  # cin, cout, k, stride, pad, N, H, W
    (64, 256, 3, 1, 1, 2, 19, 23),
    (128, 128, 3, 2, 1, 2, 17, 21),
    (64, 256, 1, 1, 0, 2, 15, 17),
    (256, 512, 1, 2, 0, 1, 17, 15),
    (64, 64, 3, 1, 1, 2, 21, 19),        # Cout <= 64: 256x64 tile
    (256, 108, 3, 1, 1, 1, 9, 15),       # head output, Cout not a multiple of 32
    (32, 40, 3, 1, 1, 1, 7, 9),          # K = 288, small
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_fprop_dgrad_wgrad(cv, dev, case):
    cin, cout, k, stride, pad, N, H, W = case
    x = rnd((N, cin, H, W), 1)
    w = rnd((cout, cin, k, k), 2, (2.0 / (k * k * cin)) ** 0.5)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, stride, pad)
    gy = rnd(tuple(y_ref.shape), 3)
    (y_ref * gy).sum().backward()

    xg, wg = nhwc(x).to(dev), w.to(dev)
    wp = cv.pack_weights(wg, 0)
    y = cv.fprop(xg, wp, cout, k, stride, pad)
    close(nchw(y), y_ref)

    # dgrad (conv Cout padded to a multiple of 4 where needed)
    cpad = (cout + 3) // 4 * 4
    gyg = nhwc(gy).to(dev)
    if cpad != cout:
        gyg = F.pad(gyg, (0, cpad - cout))
    wd = cv.pack_weights(wg, 1, c_pad=cpad)
    dx = cv.dgrad(gyg.contiguous(), wd, (H, W), cin, k, stride, pad)
    close(nchw(dx), xr.grad)

    # wgrad, accumulated in two halves of the batch when possible, then unpacked to OIHW
    dw = torch.zeros_like(wp)
    gy_c = gyg.contiguous()
    if N > 1:
        cv.wgrad(gy_c[:1].contiguous(), xg[:1].contiguous(), dw, cout, k, stride, pad)
        cv.wgrad(gy_c[1:].contiguous(), xg[1:].contiguous(), dw, cout, k, stride, pad)
    else:
        cv.wgrad(gy_c, xg, dw, cout, k, stride, pad)
    dweight, _, _ = cv.unpack_wgrad(dw, wp, tuple(w.shape))
    close(dweight, wr.grad)


WGRAD_LONG_CASES = [  # cin, cout, k, stride, pad, N, H, W : tens of K slices, slices crossing image boundaries,
    (64, 64, 3, 1, 1, 3, 67, 91),        # several pixel-table batches per slice; 64 x 256 tile
    (64, 256, 1, 1, 0, 3, 61, 83),       # 256 x 64 tile
    (128, 128, 3, 2, 1, 3, 131, 93),     # 128 x 128 tile, stride 2, odd sizes
    (256, 72, 3, 1, 1, 2, 45, 80),       # ld of dY (72) is not the tile width; columns past Cout
]


@pytest.mark.parametrize("case", WGRAD_LONG_CASES)
def test_wgrad_long_k(cv, dev, case):
    """Weight gradient and dY column sums over a long, split K range (the shapes the training step runs), against
    torch's fp64 conv2d backward."""
    cin, cout, k, stride, pad, N, H, W = case
    x = rnd((N, cin, H, W), 11)
    w = rnd((cout, cin, k, k), 12, (2.0 / (k * k * cin)) ** 0.5)
    wr = w.double().requires_grad_(True)
    y_ref = F.conv2d(x.double(), wr, None, stride, pad)
    gy = rnd(tuple(y_ref.shape), 13)
    (y_ref * gy.double()).sum().backward()
    xg, gyg = nhwc(x).to(dev), nhwc(gy).to(dev)
    wp = cv.pack_weights(w.to(dev), 0)
    dw = torch.zeros_like(wp)
    cs = torch.zeros(cout, device=dev)
    cv.wgrad(gyg, xg, dw, cout, k, stride, pad, colsum=cs)
    dweight, _, _ = cv.unpack_wgrad(dw, wp, tuple(w.shape))
    close(dweight, wr.grad, tol=2e-5)
    close(cs, gy.double().sum(dim=(0, 2, 3)), tol=2e-5)


def test_winograd_group_fprop_and_dgrad(cv, dev):
    """Winograd F(4x4,3x3) path of the head towers on a small pyramid (odd sizes: partial tiles on both edges): forward
    with bias + ReLU, data gradient with ReLU mask + accumulated gradient, against torch's fp64 conv2d.
    Tolerance 1e-4 of the max magnitude (north_star); F(4x4,3x3) in fp32 measures ~1e-5."""
    cin = cout = 64
    w = rnd((cout, cin, 3, 3), 21, (2.0 / (9 * cin)) ** 0.5)
    b = rnd((cout,), 22, 0.1)
    shapes = [(2, 19, 23), (2, 10, 12), (2, 5, 6), (2, 3, 3), (2, 2, 2)]
    xs = [rnd((n, cin, h, ww), 30 + i) for i, (n, h, ww) in enumerate(shapes)]
    U = cv.wino_weights(w.to(dev), 0)
    ys = cv.wino_conv_group([nhwc(x).to(dev) for x in xs], U, shift=b.to(dev), act=cv.ACT_RELU)
    for x, y in zip(xs, ys):
        ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1))
        close(nchw(y), ref, tol=1e-4)
    # data gradient: dx = mask * conv_transpose(dy, w) + add
    Ud = cv.wino_weights(w.to(dev), 1)
    gys = [rnd((n, cout, h, ww), 40 + i) for i, (n, h, ww) in enumerate(shapes)]
    adds = [rnd((n, cin, h, ww), 50 + i) for i, (n, h, ww) in enumerate(shapes)]
    dxs = cv.wino_conv_group([nhwc(g).to(dev) for g in gys], Ud, adds=[nhwc(a).to(dev) for a in adds],
                             masks=[nhwc(x).to(dev) for x in xs], mask_mode=2)
    for x, g, a, dx in zip(xs, gys, adds, dxs):
        ref = (F.conv_transpose2d(g.double(), w.double(), None, 1, 1) + a.double()) * (x.double() > 0)
        close(nchw(dx), ref, tol=1e-4)


@pytest.mark.parametrize("case", [(128, 192, 1, 13, 18), (192, 64, 3, 7, 5), (64, 128, 2, 4, 4)])
def test_winograd_single_epilogues(cv, dev, case):
    """Winograd path on one problem with Cin != Cout: folded batch-norm scale / shift + ReLU forward; data gradient with
    the scale folded into the weights, ReLU mask applied BEFORE the addend (mask_mode 1); weight gradient with the kept
    input transform.  Against torch fp64."""
    cin, cout, n, h, ww = case
    w = rnd((cout, cin, 3, 3), 91, (2.0 / (9 * cin)) ** 0.5)
    sc, sh = rnd((cout,), 92, 0.3).abs() + 0.5, rnd((cout,), 93, 0.1)
    x = rnd((n, cin, h, ww), 94)
    xg = nhwc(x).to(dev)
    (y,), V = cv.wino_conv_group([xg], cv.wino_weights(w.to(dev), 0), scale=sc.to(dev), shift=sh.to(dev), act=cv.ACT_RELU,
                                 keep_v=True)
    conv = F.conv2d(x.double(), w.double(), None, 1, 1)
    close(nchw(y), torch.relu(conv * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)), tol=1e-4)
    g = rnd((n, cout, h, ww), 95)
    a = rnd((n, cin, h, ww), 96)
    (dx,) = cv.wino_conv_group([nhwc(g).to(dev)], cv.wino_weights(w.to(dev), 1, scale=sc.to(dev)), adds=[nhwc(a).to(dev)],
                               masks=[xg], mask_mode=1)
    ref = F.conv_transpose2d(g.double() * sc.double().view(1, -1, 1, 1), w.double(), None, 1, 1) * (x.double() > 0) + a.double()
    close(nchw(dx), ref, tol=1e-4)
    wr = w.double().requires_grad_(True)
    (F.conv2d(x.double(), wr, None, 1, 1) * g.double()).sum().backward()
    wp = cv.pack_weights(w.to(dev), 0)
    dw = torch.zeros_like(wp)
    cs = torch.zeros(cout, device=dev)
    cv.wino_wgrad_group([nhwc(g).to(dev)], [xg], dw, cs, V=V)
    close(cv.unpack_wgrad(dw, wp, tuple(w.shape))[0], wr.grad, tol=1e-4)
    close(cs, g.double().sum(dim=(0, 2, 3)), tol=2e-5)


def test_winograd_group_wgrad(cv, dev):
    """Winograd weight gradient + bias gradient over a small pyramid (partial tiles), against torch's fp64 conv2d backward."""
    cin, cout = 64, 128
    w = rnd((cout, cin, 3, 3), 61, (2.0 / (9 * cin)) ** 0.5)
    shapes = [(2, 19, 23), (2, 10, 12), (2, 5, 6), (2, 3, 3), (2, 2, 2)]
    xs = [rnd((n, cin, h, ww), 70 + i) for i, (n, h, ww) in enumerate(shapes)]
    gys = [rnd((n, cout, h, ww), 80 + i) for i, (n, h, ww) in enumerate(shapes)]
    wr = w.double().requires_grad_(True)
    tot = 0
    for x, g in zip(xs, gys):
        tot = tot + (F.conv2d(x.double(), wr, None, 1, 1) * g.double()).sum()
    tot.backward()
    wp = cv.pack_weights(w.to(dev), 0)
    dw = torch.zeros_like(wp)
    cs = torch.zeros(cout, device=dev)
    cv.wino_wgrad_group([nhwc(g).to(dev) for g in gys], [nhwc(x).to(dev) for x in xs], dw, cs)
    dweight, _, _ = cv.unpack_wgrad(dw, wp, tuple(w.shape))
    close(dweight, wr.grad, tol=1e-4)
    close(cs, sum(g.double().sum(dim=(0, 2, 3)) for g in gys), tol=2e-5)


def test_stem_conv_bn_relu(cv, dev):
    """7x7 s2 p3 on a 3-channel NCHW image: NHWC4 staging, kw padded to 8, folded frozen BN + ReLU epilogue."""
    N, H, W = 2, 37, 45
    img = rnd((N, 3, H, W), 4)
    w = rnd((64, 3, 7, 7), 5, 0.1)
    gamma, beta = rnd((64,), 6, 0.3) + 1.0, rnd((64,), 7, 0.1)
    mean, var = rnd((64,), 8, 0.1), rnd((64,), 9, 0.1).abs() + 0.5
    ir, wr = img.clone(), w.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.relu(F.batch_norm(F.conv2d(ir, wr, None, 2, 3), mean, var, gr, br, False, 0.0, 1e-5))
    gy = rnd(tuple(y_ref.shape), 10)
    (y_ref * gy).sum().backward()

    x4 = cv.nchw_to_nhwc4(img.to(dev))
    assert x4.shape == (N, H, W, 4) and float(x4[..., 3].abs().max()) == 0.0
    wp = cv.pack_weights(w.to(dev), 0, kw_pad=8, c_pad=4)
    scale, shift, rstd = cv.bn_fold(gamma.to(dev), beta.to(dev), mean.to(dev), var.to(dev))
    y = cv.fprop(x4, wp, 64, 7, 2, 3, kw_pad=8, scale=scale, shift=shift, act=cv.ACT_RELU)
    close(nchw(y), y_ref)
    # backward of the fused layer: g = dy*(y>0); dW = scale*wgrad(g,x); dgamma/dbeta from dWraw, W and colsum(g)
    g = nhwc(gy).to(dev).contiguous()
    cv.relu_mask_(g, y)
    dw = torch.zeros_like(wp)
    cs = torch.zeros(64, device=dev)
    cv.wgrad(g, x4, dw, 64, 7, 2, 3, kw_pad=8, colsum=cs)
    close(cs, cv.colsum(g), 1e-5)
    dweight, dgamma, dbeta = cv.unpack_wgrad(dw, wp, (64, 3, 7, 7), kw_pad=8, c_pad=4, scale=scale, mean=mean.to(dev),
                                             rstd=rstd, colsum=cs, want_bn=True)
    close(dweight, wr.grad)
    close(dgamma, gr.grad)
    close(dbeta, br.grad)


def test_residual_block_epilogues(cv, dev):
    """conv + BN + residual + ReLU in one epilogue; dgrad with gradient accumulation and ReLU mask."""
    N, C, H, W = 2, 64, 13, 11
    x = rnd((N, C, H, W), 11)
    res = rnd((N, 128, H, W), 12)
    w = rnd((128, C, 3, 3), 13, 0.06)
    scale_c, shift_c = rnd((128,), 14, 0.2) + 1.0, rnd((128,), 15, 0.1)
    xr = x.clone().requires_grad_(True)
    y_ref = F.relu(F.conv2d(xr, w, None, 1, 1) * scale_c[None, :, None, None] + shift_c[None, :, None, None] + res)
    xg = nhwc(x).to(dev)
    wp = cv.pack_weights(w.to(dev), 0)
    y = cv.fprop(xg, wp, 128, 3, 1, 1, scale=scale_c.to(dev), shift=shift_c.to(dev), add=nhwc(res).to(dev),
                 add_mode=1, act=cv.ACT_RELU)
    close(nchw(y), y_ref)
    # dgrad: dx = dgrad(g, scale*W) + other, masked by (zmask > 0)
    gy = rnd(tuple(y_ref.shape), 16)
    other = rnd((N, C, H, W), 17)
    zmask = rnd((N, C, H, W), 18)
    g_ref = gy * (y_ref > 0)
    g_ref_x, = torch.autograd.grad(F.conv2d(xr, w * scale_c[:, None, None, None], None, 1, 1), xr, g_ref)
    want = (g_ref_x + other) * (zmask > 0)
    g = nhwc(gy).to(dev).contiguous()
    cv.relu_mask_(g, y)
    wd = cv.pack_weights(w.to(dev), 1, scale=scale_c.to(dev))
    dx = cv.dgrad(g, wd, (H, W), C, 3, 1, 1, add=nhwc(other).to(dev), add_mode=1, mask=nhwc(zmask).to(dev))
    close(nchw(dx), want)
    # mask applied before the add (P6: heads' gradient added unmasked to the masked P7 contribution)
    dx1 = cv.dgrad(g, wd, (H, W), C, 3, 1, 1, add=nhwc(other).to(dev), add_mode=1, mask=nhwc(zmask).to(dev), mask_mode=1)
    close(nchw(dx1), g_ref_x * (zmask > 0) + other)
    # ReLU-on-load of the input (P7 = conv(ReLU(P6)))
    y_in = cv.fprop(xg, wp, 128, 3, 1, 1, in_relu=True)
    close(nchw(y_in), F.conv2d(F.relu(x), w, None, 1, 1))
    dwr = torch.zeros_like(wp)
    cv.wgrad(g, xg, dwr, 128, 3, 1, 1, in_relu=True)
    wr2 = w.clone().requires_grad_(True)
    (F.conv2d(F.relu(x), wr2, None, 1, 1) * g_ref).sum().backward()
    close(cv.unpack_wgrad(dwr, wp, tuple(w.shape))[0], wr2.grad)


def test_fpn_upsample_add_crop(cv, dev):
    """Lateral 1x1 conv + nearest-upsampled coarser map, cropped to the lateral's size (D/model.py:88-108)."""
    N = 2
    c_lat = rnd((N, 128, 9, 13), 19)        # finer level
    coarse = rnd((N, 256, 5, 7), 20)        # up2 -> 10 x 14, cropped to 9 x 13
    w = rnd((256, 128, 1, 1), 21, 0.1)
    b = rnd((256,), 22, 0.1)
    want = F.conv2d(c_lat, w, b) + F.interpolate(coarse, scale_factor=2, mode="nearest")[:, :, :9, :13]
    wp = cv.pack_weights(w.to(dev), 0)
    y = cv.fprop(nhwc(c_lat).to(dev), wp, 256, 1, 1, 0, shift=b.to(dev), add=nhwc(coarse).to(dev), add_mode=2,
                 add_hw=(5, 7))
    close(nchw(y), want)
    # backward of the upsample+crop: children summed into the coarse map
    g = rnd((N, 256, 9, 13), 23)
    cr = coarse.clone().requires_grad_(True)
    (F.interpolate(cr, scale_factor=2, mode="nearest")[:, :, :9, :13] * g).sum().backward()
    acc = rnd((N, 256, 5, 7), 24)
    dst = nhwc(acc).to(dev).contiguous()
    cv.upsample_add_bwd(nhwc(g).to(dev).contiguous(), dst)
    close(nchw(dst), acc + cr.grad)


def test_head_output_into_concat_buffer(cv, dev):
    """Head output conv writes its level's slice of the concatenated [B, A, n] tensor, sigmoid fused
    (permute + view + cat of D/model.py:155-157, 196-205, 302-304 for free)."""
    N, C = 2, 8
    feats = [rnd((N, 64, 5, 7), 30), rnd((N, 64, 3, 4), 31)]
    w = rnd((9 * C, 64, 3, 3), 32, 0.05)
    b = rnd((9 * C,), 33, 0.1)
    want = torch.cat([torch.sigmoid(F.conv2d(f, w, b, 1, 1)).permute(0, 2, 3, 1).contiguous().view(N, -1, C)
                      for f in feats], dim=1)
    A = want.shape[1]
    out = torch.zeros((N, A, C), dtype=torch.float32, device=dev)
    wp = cv.pack_weights(w.to(dev), 0)
    off = 0
    for f in feats:
        h, wd = f.shape[2], f.shape[3]
        view = out.view(N, -1)[:, off * C:]
        cv.conv_igemm(nhwc(f).to(dev), wp, view, (h, wd, 9 * C, 3, 3, 1, 1, -1, 0), shift=b.to(dev),
                      act=cv.ACT_SIGMOID, y_batch_stride=A * C)
        off += h * wd * 9
    close(out, want)
    # sigmoid backward + channel padding for the dgrad/wgrad GEMMs
    dy = rnd((N, A, C), 34).to(dev)
    rows = 5 * 7
    pad = cv.sigmoid_bwd_pad(dy.data_ptr(), out.data_ptr(), N, rows, 9 * C, 96, A * C, dev)
    lvl = lambda t: t[:, :rows * 9].reshape(N, rows, 9 * C)
    wantp = F.pad(lvl(dy) * lvl(out) * (1 - lvl(out)), (0, 96 - 9 * C)).reshape(-1, 96)
    close(pad, wantp)
    # second level: pointer offset into the concatenated tensors, identity (regression-style) variant
    rows2 = 3 * 4
    pad2 = cv.sigmoid_bwd_pad(dy.data_ptr() + 4 * rows * 9 * C, None, N, rows2, 9 * C, 80, A * C, dev)
    want2 = F.pad(dy[:, rows * 9:].reshape(N, rows2, 9 * C), (0, 80 - 9 * C)).reshape(-1, 80)
    close(pad2, want2)


def test_maxpool_forward_backward_with_ties(cv, dev):
    N, C, H, W = 2, 8, 13, 15
    x = F.relu(rnd((N, C, H, W), 40))                    # many exact zeros: ties inside windows
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    gy = rnd(tuple(y_ref.shape), 41)
    (y_ref * gy).sum().backward()
    xg = nhwc(x).to(dev)
    y, arg = cv.maxpool_fwd(xg, want_argmax=True)
    assert torch.equal(nchw(y).cpu(), y_ref.detach())
    assert torch.equal(cv.maxpool_fwd(xg), y)
    dx = cv.maxpool_bwd(xg, nhwc(gy).to(dev), arg, relu_mask=False)
    close(nchw(dx), xr.grad, 1e-6)
    dxm = cv.maxpool_bwd(xg, nhwc(gy).to(dev), arg, relu_mask=True)
    close(nchw(dxm), xr.grad * (x > 0), 1e-6)


def test_colsum(cv, dev):
    g = rnd((3, 7, 11, 40), 50)
    close(cv.colsum(g.to(dev)), g.reshape(-1, 40).sum(0), 1e-5)
    close(cv.colsum(g.to(dev), C=36), g.reshape(-1, 40)[:, :36].sum(0), 1e-5)
    acc = cv.colsum(g.to(dev))
    cv.colsum(g.to(dev), out=acc)
    close(acc, 2 * g.reshape(-1, 40).sum(0), 1e-5)


def test_layer_shapes_of_the_benchmark_config(cv, dev):
    """The dominant benchmark shape (3x3 256->256 on a P3-sized map, B=1 slice of cfg2) against torch CPU."""
    x = rnd((1, 256, 135, 240), 60)
    w = rnd((256, 256, 3, 3), 61, (2.0 / 2304) ** 0.5)
    b = rnd((256,), 62, 0.1)
    want = F.relu(F.conv2d(x, w, b, 1, 1))
    y = cv.fprop(nhwc(x).to(dev), cv.pack_weights(w.to(dev), 0), 256, 3, 1, 1, shift=b.to(dev), act=cv.ACT_RELU)
    close(nchw(y), want)


@pytest.mark.parametrize("case", [(128, 96, 3, 1, 2, 17, 21), (64, 128, 3, 1, 1, 18, 20), (128, 128, 3, 1, 1, 1, 1)])
def test_stride2_dgrad_by_parity_classes(cv, dev, case):
    """Stride-2 data gradient as 4 tap-subset convolutions written to strided positions == the generic form,
    with gradient accumulation and ReLU mask applied per class."""
    cin, cout, k, pad, N, H, W = case
    x = rnd((N, cin, H, W), 70)
    w = rnd((cout, cin, k, k), 71, 0.05)
    sc = rnd((cout,), 72, 0.2) + 1.0
    xr = x.clone().requires_grad_(True)
    y = F.conv2d(xr, w * sc[:, None, None, None], None, 2, pad)
    gy = rnd(tuple(y.shape), 73)
    (y * gy).sum().backward()
    other, zmask = rnd((N, cin, H, W), 74), rnd((N, cin, H, W), 75)
    want = (xr.grad + other) * (zmask > 0)
    wcls = [cv.pack_weights(w.to(dev), 1, scale=sc.to(dev), taps=c[2]) for c in cv.s2_classes(k, pad)]
    dx = cv.dgrad_s2_classes(nhwc(gy).to(dev), wcls, (H, W), cin, k, pad, add=nhwc(other).to(dev), add_mode=1,
                             mask=nhwc(zmask).to(dev))
    close(nchw(dx), want)


def test_shortcut_1x1_stride2_gradient_added_at_even_positions(cv, dev):
    """Bottleneck first block: d(input) = dgrad(conv1 1x1 s1) + lateral + [1x1 s2 shortcut gradient at even pixels],
    masked -- the shortcut part stays on its compact grid (add2)."""
    N, C, H, W, Cd = 2, 64, 13, 15, 128
    x = rnd((N, C, H, W), 80)
    w1 = rnd((32, C, 1, 1), 81, 0.1)
    wd = rnd((Cd, C, 1, 1), 82, 0.1)
    xr = x.clone().requires_grad_(True)
    y1 = F.conv2d(xr, w1)
    yd = F.conv2d(xr, wd, None, 2)
    g1, gd = rnd(tuple(y1.shape), 83), rnd(tuple(yd.shape), 84)
    ((y1 * g1).sum() + (yd * gd).sum()).backward()
    lateral, zmask = rnd((N, C, H, W), 85), rnd((N, C, H, W), 86)
    want = (xr.grad + lateral) * (zmask > 0)
    compact = torch.empty((N, yd.shape[2], yd.shape[3], C), device=dev)
    cv.conv_igemm(nhwc(gd).to(dev), cv.pack_weights(wd.to(dev), 1), compact, (yd.shape[2], yd.shape[3], C, 1, 1, 1, -1, 0, 0))
    dx = cv.dgrad(nhwc(g1).to(dev), cv.pack_weights(w1.to(dev), 1), (H, W), C, 1, 1, 0, add=nhwc(lateral).to(dev),
                  add_mode=1, mask=nhwc(zmask).to(dev), add2=compact)
    close(nchw(dx), want)


SPLITK_CASES = [  # cin, cout, k, stride, pad, N, H, W : few output tiles, long K -> rn_conv_igemm_splitk
    (512, 256, 3, 2, 1, 2, 17, 15),      # the P6 shape class: 3x3 stride 2 on many channels
    (256, 256, 3, 1, 1, 3, 4, 4),        # layer4-of-a-crop class: a handful of pixels
    (512, 40, 3, 1, 1, 2, 5, 7),         # narrow Cout (256x64 tile), Cout not a multiple of 32
    (1024, 256, 1, 1, 0, 1, 9, 11),      # 1x1 with long K
]


@pytest.mark.parametrize("case", SPLITK_CASES)
def test_splitk_fprop_and_dgrad(cv, dev, case):
    """The split-K path (chosen by rn_conv_splitk_workspace_bytes) with bias + residual add + ReLU against F.conv2d,
    and the data gradient; results must be reproducible bit for bit (no atomics)."""
    from retinanet_mi355x import _hip
    cin, cout, k, stride, pad, N, H, W = case
    x = rnd((N, cin, H, W), 1)
    w = rnd((cout, cin, k, k), 2, (2.0 / (k * k * cin)) ** 0.5)
    bias = rnd((cout,), 4)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, w, bias, stride, pad)
    res = rnd(tuple(y_ref.shape), 5)
    out_ref = F.relu(y_ref + res)
    xg, wg = nhwc(x).to(dev), w.to(dev)
    wp = cv.pack_weights(wg, 0)
    Ho, Wo = y_ref.shape[2], y_ref.shape[3]
    geom = (Ho, Wo, cout, k, k, stride, 1, -pad, 0)
    d = cv._make_desc(xg, geom, cv.ACT_RELU, 1, (0, 0), 0, False, None, None, None, None)
    import ctypes
    assert _hip.load().rn_conv_splitk_workspace_bytes(ctypes.byref(d)) > 0, "case no longer takes the split-K path"
    resg = nhwc(res).contiguous().to(dev)
    y1 = cv.fprop(xg, wp, cout, k, stride, pad, shift=bias.to(dev), act=cv.ACT_RELU, add=resg, add_mode=1)
    close(nchw(y1), out_ref.detach())
    y2 = cv.fprop(xg, wp, cout, k, stride, pad, shift=bias.to(dev), act=cv.ACT_RELU, add=resg, add_mode=1)
    assert torch.equal(y1, y2)
    if stride == 1:
        gy = rnd(tuple(y_ref.shape), 3)
        (y_ref * gy).sum().backward()
        cpad = (cout + 3) // 4 * 4
        gyg = nhwc(gy).to(dev)
        if cpad != cout:
            gyg = F.pad(gyg, (0, cpad - cout))
        wd = cv.pack_weights(wg, 1, c_pad=cpad)
        dx = cv.dgrad(gyg.contiguous(), wd, (H, W), cin, k, stride, pad)
        close(nchw(dx), xr.grad)


def test_split_weights_terms_add_up_exactly(cv, dev):
    """rn_split_weights: [rows][Kpad/16][h, m, l][16] bf16 with h + m + l == w exactly for 2^20 values spread over 24 decades
    (csrc/mfma_split.h: the kernels' in-register split is the same code), h the nearest bf16, and the residuals within their bounds."""
    w = rnd((1024, 1024), 5) * torch.logspace(-12, 12, 1024 * 1024).view(1024, 1024)
    w[3, 7] = 0.0
    wp = w.to(dev).contiguous()
    cv.split_weights(wp)
    rec = wp._rn_split.view(torch.bfloat16).view(1024, 64, 3, 16).cpu()
    total = rec.double().sum(dim=2).reshape(1024, 1024)
    assert torch.equal(total, w.double())
    h, m, l = (rec[:, :, k].reshape(1024, 1024).double() for k in range(3))
    assert torch.equal(h.float(), w.to(torch.bfloat16).float())                   # h is the nearest bf16
    ax = w.double().abs()
    assert bool((m.abs() <= ax * 2.0 ** -8).all()) and bool((l.abs() <= ax * 2.0 ** -16).all())


def test_presplit_operand_is_refused_in_native_mode(cv, dev):
    import ctypes
    from retinanet_mi355x import _hip
    if cv.get_fp32_mfma() != "native":
        pytest.skip("native mode only")
    x = rnd((1, 8, 8, 16), 1).to(dev)
    wp = cv.pack_weights(rnd((32, 16, 1, 1), 2).to(dev), 0)
    cv.split_weights(wp)
    y = torch.empty((1, 8, 8, 32), device=dev)
    d = _hip.ConvDesc(1, 8, 8, 16, 8, 8, 32, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 8, 8, 0, 0, 0, 0, 8 * 8 * 16, 8 * 8 * 32, 8 * 8 * 32, 0, 1)
    rc = _hip.load().rn_conv_igemm(ctypes.byref(d), x.data_ptr(), wp._rn_split.data_ptr(), y.data_ptr(), None, None, None, None, None, _hip.stream())
    assert rc != 0


def test_product_mode_follows_the_environment(dev):
    """RN_FP32_MFMA selects the library's initial mode (include/retinanet_mi355x.h); without it: RN_FP32_DEFAULT = split3."""
    import os
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); from retinanet_mi355x import conv; print(conv.get_fp32_mfma())" % os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-playground_amd")
    for env, want in (({"RN_FP32_MFMA": "native"}, "native"), ({"RN_FP32_MFMA": "split"}, "split"), ({"RN_FP32_MFMA": "split3"}, "split3"),
                      ({}, "split3")):
        e = {k: v for k, v in os.environ.items() if k != "RN_FP32_MFMA"}
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert out.stdout.strip().splitlines()[-1] == want, (env, out.stdout, out.stderr[-500:])


def test_wide_dynamic_range_elementwise(cv, dev):
    """Positive operands spread over 17 decades (no cancellation): EVERY output, gradient and weight-gradient element within 1e-5
    RELATIVE of fp64 -- in split mode this holds only if the three terms of an operand really add up to it and the dropped
    products are as small as csrc/mfma_split.h says (a two-term split would sit at 4e-6 per product, a bf16 product at 4e-3)."""
    g = torch.Generator().manual_seed(3)
    N, H, W, cin, cout = 2, 24, 40, 512, 256
    x = torch.exp((torch.rand(N, cin, H, W, generator=g) - 0.5) * 40).float()
    w = torch.exp((torch.rand(cout, cin, 1, 1, generator=g) - 0.5) * 40).float()
    gy = torch.exp((torch.rand(N, cout, H, W, generator=g) - 0.5) * 40).float()
    xd, wd = x.double(), w.double().view(cout, cin)
    y_ref = torch.einsum("oc,nchw->nohw", wd, xd)
    dx_ref = torch.einsum("oc,nohw->nchw", wd, gy.double())
    dw_ref = torch.einsum("nohw,nchw->oc", gy.double(), xd)
    finite = lambda t: bool(torch.isfinite(t.float()).all())           # the references must fit fp32 for the comparison to mean anything
    assert finite(y_ref) and finite(dx_ref) and finite(dw_ref)
    wp, wdp = cv.pack_weights(w.to(dev), 0), cv.pack_weights(w.to(dev), 1)
    y = cv.fprop(nhwc(x).to(dev), wp, cout, 1, 1, 0)
    dx = cv.dgrad(nhwc(gy).to(dev), wdp, (H, W), cin, 1, 1, 0)
    dw = torch.zeros_like(wp)
    cv.wgrad(nhwc(gy).to(dev), nhwc(x).to(dev), dw, cout, 1, 1, 0)
    dwu = cv.unpack_wgrad(dw, wp, tuple(w.shape))[0]
    for got, ref in ((nchw(y), y_ref), (nchw(dx), dx_ref), (dwu.view(cout, cin), dw_ref)):
        rel = ((got.cpu().double() - ref).abs() / ref.abs()).max()
        assert float(rel) <= 1e-5, float(rel)


def test_split_mode_corners_that_differ_from_ieee_fp32(dev):
    """RN_FP32_SPLIT (the library default) is not IEEE in three corners, documented in include/retinanet_mi355x.h, DESIGN.md 4.6
    and INTEGRATION.md; this pins them so a change is noticed:
      * an infinite operand gives NaN (inf - bf16(inf) in the residual), where the fp32 MFMA gives inf;
      * a finite operand above the largest bf16 (3.3895e38) rounds its leading term to inf and gives NaN as well;
      * operands below the bf16 NORMAL range (|x| < 1.18e-38) lose their residual terms (flushed by the matrix core) and keep only
        a subnormal bf16 leading term: the result is correct to a few per cent instead of 2^-24 -- on values below 1e-37
        (measured here: 1.0e-2 relative with operands of 1e-39).
    Everything finite and normal is covered by the other tests of this module at 1e-5 relative."""
    from retinanet_mi355x import conv
    before = conv.get_fp32_mfma(), conv.PRESPLIT
    cin, cout, N, H, W = 256, 128, 1, 8, 16               # K = 256 >= rn_fp32_split_min_k: the split kernels take it
    w = torch.full((cout, cin, 1, 1), 0.5)
    try:
        out = {}
        for mode in ("native", "split"):
            conv.set_fp32_mfma(mode)
            conv.PRESPLIT = True
            wp = conv.pack_weights(w.to(dev), 0)
            for name, val in (("inf", float("inf")), ("huge", 3.4e38), ("tiny", 3e-39)):
                x = torch.ones(N, H, W, cin) * (1e-39 if name == "tiny" else 1.0)
                x[0, 2, 3, 5] = val
                out[mode, name] = conv.fprop(x.to(dev), wp, cout, 1, 1, 0).cpu()
        assert torch.isinf(out["native", "inf"][0, 2, 3]).all() and torch.isfinite(out["native", "inf"][0, 2, 4]).all()
        assert torch.isnan(out["split", "inf"][0, 2, 3]).all() and torch.isfinite(out["split", "inf"][0, 2, 4]).all()
        assert torch.isnan(out["split", "huge"][0, 2, 3]).all()                        # native: 0.5 * 3.4e38 + 127.5, finite
        assert torch.isfinite(out["native", "huge"][0, 2, 3]).all()
        t_n, t_s = out["native", "tiny"], out["split", "tiny"]
        assert torch.isfinite(t_s).all() and float((t_n - t_s).abs().max()) <= 2.0 ** -4 * float(t_n.abs().max()) + 1e-44
    finally:
        conv.set_fp32_mfma(before[0])
        conv.PRESPLIT = before[1]
