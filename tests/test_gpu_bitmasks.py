"""Sign bits of ReLU outputs (round 4; csrc/common.h rn_sign_store / rn_mask_load4, rn_conv_desc.sign_out, mask_mode | RN_MASK_BITS).

The backward pass needs of a ReLU output only y > 0 (D/utils.py:60-80: the gradient of `out = self.relu(out)`); the engine used to
re-read the fp32 activation for it in every data-gradient epilogue (13 GB per training step).  Producers now also write one bit per
element, consumers read the bits.  Nothing about the arithmetic changes, so everything here is EXACT:
  * the words a producer writes are the packed (y > 0) of the tensor it wrote -- every producer kernel family (16x16x32 split kernel,
    32x32x16 split / native tiles wide and narrow, grouped launch, split-K finish, Winograd output transform);
  * a consumer gives bit-identical results from the bits and from the fp32 tensor -- plain, strided (stride-2 parity classes),
    mask before / after the addend, grouped, Winograd, max-pool backward;
  * a whole training step gives bit-identical parameter gradients with the bits on and off (fixed-order reductions).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["split", "native", "split3"])
def cv(dev, request):
    from retinanet_mi355x import conv
    before = conv.get_fp32_mfma(), conv.BITMASKS
    conv.set_fp32_mfma(request.param)
    conv.BITMASKS = True
    yield conv
    conv.set_fp32_mfma(before[0])
    conv.BITMASKS = before[1]


def rnd(shape, seed, std=1.0):
    from retinanet_mi355x import synth
    return torch.from_numpy(synth.normal(shape, seed, std))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def packed_sign(y):
    """(y > 0) of a dense fp32 tensor packed as the kernels pack it: bit (e & 31) of word (e >> 5)."""
    b = (y.reshape(-1) > 0).to(torch.int64).reshape(-1, 32)
    w = (b << torch.arange(32, device=y.device)).sum(1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


PRODUCERS = [  # cin, cout, k, stride, pad, N, H, W
    (64, 256, 1, 1, 0, 2, 19, 23),       # wide 1x1: the 16x16x32 kernel in split mode
    (256, 64, 1, 1, 0, 1, 33, 31),       # narrow (256 x 64 tile)
    (64, 64, 3, 1, 1, 2, 19, 23),        # narrow 3x3
    (128, 320, 3, 1, 1, 1, 30, 34),      # Cout = 320: a column tile with 64 valid columns
    (32, 128, 3, 2, 1, 2, 21, 17),       # stride 2
    (4, 64, 7, 2, 3, 1, 40, 56),         # the stem's shape (4-channel input, 7x7 stride 2)
]


@pytest.mark.parametrize("case", PRODUCERS)
def test_producer_writes_the_sign_of_what_it_stored(cv, dev, case):
    cin, cout, k, stride, pad, N, H, W = case
    x, w, b = rnd((N, cin, H, W), 1), rnd((cout, cin, k, k), 2, (2.0 / (k * k * cin)) ** 0.5), rnd((cout,), 3)
    res = rnd((N, cout, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1), 4)
    wp = cv.pack_weights(w.to(dev), 0, kw_pad=8 if cin == 4 else None, c_pad=4 if cin == 4 else None)
    kw = dict(kw_pad=8) if cin == 4 else {}
    y = cv.fprop(nhwc(x).to(dev), wp, cout, k, stride, pad, shift=b.to(dev), act=cv.ACT_RELU, add=nhwc(res).to(dev), add_mode=1, sign=True, **kw)
    bits = getattr(y, "_rn_sign", None)
    assert bits is not None and bits.numel() == y.numel() // 32
    assert 0.2 < float((y > 0).float().mean()) < 0.8
    assert torch.equal(bits, packed_sign(y))
    y2 = cv.fprop(nhwc(x).to(dev), wp, cout, k, stride, pad, shift=b.to(dev), act=cv.ACT_RELU, add=nhwc(res).to(dev), add_mode=1, **kw)
    assert torch.equal(y, y2) and getattr(y2, "_rn_sign", None) is None       # asking for the bits changes nothing else


def test_reusing_an_output_tensor_drops_the_old_bits(cv, dev):
    """A producer that writes into a tensor an EARLIER producer left sign bits on (the public conv_* API with its own destination) and
    is not asked for bits itself must not leave the old ones behind: a consumer would mask with the signs of a tensor that no longer
    exists.  With sign=True the new bits replace the old."""
    cin, cout, N, H, W = 64, 128, 1, 17, 19
    x1, x2 = nhwc(rnd((N, cin, H, W), 1)).to(dev), nhwc(rnd((N, cin, H, W), 2)).to(dev)
    wp = cv.pack_weights(rnd((cout, cin, 1, 1), 3, 0.2).to(dev), 0)
    y = torch.empty((N, H, W, cout), device=dev)
    geom = (H, W, cout, 1, 1, 1, 1, 0, 0)
    cv.conv_igemm(x1, wp, y, geom, act=cv.ACT_RELU, sign=True)
    first = y._rn_sign.clone()
    assert torch.equal(first, packed_sign(y))
    cv.conv_igemm(x2, wp, y, geom, act=cv.ACT_RELU, sign=False)              # same destination, no bits asked for
    assert getattr(y, "_rn_sign", None) is None
    cv.conv_igemm(x2, wp, y, geom, act=cv.ACT_RELU, sign=True)
    assert torch.equal(y._rn_sign, packed_sign(y)) and not torch.equal(y._rn_sign, first)


def _with_and_without_bits(cv, z, fn):
    """fn(mask tensor) with z's sign bits attached and with the plain fp32 z: must be bit-identical."""
    zb = z.clone()
    zb._rn_sign = packed_sign(z)
    a, b = fn(zb), fn(z.clone())
    assert torch.equal(a, b)
    return a


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("shape", [(256, 256, 3), (256, 64, 1), (64, 256, 1), (64, 64, 3)])
def test_data_gradient_reads_bits_like_the_tensor(cv, dev, shape, mode):
    cin, cout, k = shape
    N, H, W = 2, 21, 27
    w, gy = rnd((cout, cin, k, k), 5, 0.03), rnd((N, cout, H, W), 6)
    z, addend = nhwc(rnd((N, cin, H, W), 7)).to(dev), nhwc(rnd((N, cin, H, W), 8)).to(dev)
    wd = cv.pack_weights(w.to(dev), 1)
    g = nhwc(gy).to(dev)
    got = _with_and_without_bits(cv, z, lambda m: cv.dgrad(g, wd, (H, W), cin, k, 1, k // 2, add=addend, add_mode=1, mask=m, mask_mode=mode))
    dx = F.conv_transpose2d(gy.double(), w.double(), None, 1, k // 2).permute(0, 2, 3, 1)
    zc, ac = z.cpu().double(), addend.cpu().double()
    want = torch.where(zc > 0, dx + ac, torch.zeros_like(dx)) if mode == 2 else torch.where(zc > 0, dx, torch.zeros_like(dx)) + ac
    assert float((got.cpu().double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_stride2_parity_classes_read_bits_at_strided_positions(cv, dev):
    """The four output-parity classes of a stride-2 data gradient store at strided positions of dx and read the mask there."""
    cin, cout, N, H, W = 128, 256, 2, 37, 29
    w, gy = rnd((cout, cin, 3, 3), 9, 0.03), rnd((N, cout, (H + 1) // 2, (W + 1) // 2), 10)
    z = nhwc(rnd((N, cin, H, W), 11)).to(dev)
    wcls = [cv.pack_weights(w.to(dev), 1, taps=c[2]) for c in cv.s2_classes(3, 1)]
    g = nhwc(gy).to(dev)
    got = _with_and_without_bits(cv, z, lambda m: cv.dgrad_s2_classes(g, wcls, (H, W), cin, 3, 1, mask=m, mask_mode=2))
    dx = F.conv_transpose2d(gy.double(), w.double(), None, 2, 1, output_padding=(H - ((gy.shape[2] - 1) * 2 + 1), W - ((gy.shape[3] - 1) * 2 + 1)))
    want = torch.where(z.cpu().double() > 0, dx.permute(0, 2, 3, 1), torch.zeros(1, dtype=torch.float64))
    assert float((got.cpu().double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_winograd_group_writes_and_reads_bits(cv, dev):
    C, B = 256, 2
    w, b = rnd((C, C, 3, 3), 11, 0.03), rnd((C,), 12)
    xs = [nhwc(rnd((B, C, 35, 41), 13)).to(dev), nhwc(rnd((B, C, 18, 21), 14)).to(dev)]
    U = cv.wino_weights(w.to(dev), 0)
    ys = cv.wino_conv_group(xs, U, shift=b.to(dev), act=cv.ACT_RELU, sign=True)
    for y in ys:
        assert torch.equal(y._rn_sign, packed_sign(y))
    Ud = cv.wino_weights(w.to(dev), 1)
    gs = [torch.randn_like(y) for y in ys]
    with_bits = cv.wino_conv_group(gs, Ud, masks=ys, mask_mode=2)
    plain = cv.wino_conv_group(gs, Ud, masks=[y.clone() for y in ys], mask_mode=2)
    for a, b_ in zip(with_bits, plain):
        assert torch.equal(a, b_)


def test_grouped_launch_and_splitk_finish_write_bits(cv, dev):
    cin, cout, B = 64, 256, 2
    w, b = rnd((cout, cin, 3, 3), 9, 0.05), rnd((cout,), 10)
    wp = cv.pack_weights(w.to(dev), 0)
    xs = [nhwc(rnd((B, cin, h, w_), 20 + i)).to(dev) for i, (h, w_) in enumerate([(17, 23), (9, 12), (5, 6)])]
    probs = [{"x": x, "y": torch.empty((B, x.shape[1], x.shape[2], cout), device=dev), "geom": (x.shape[1], x.shape[2], cout, 3, 3, 1, 1, -1, 0),
              "sign": True} for x in xs]
    cv.conv_igemm_grouped(probs, wp, shift=b.to(dev), act=cv.ACT_RELU)
    for pr in probs:
        assert torch.equal(pr["y"]._rn_sign, packed_sign(pr["y"]))
    # few output tiles and a long reduction: the split-K form and its finish kernel (fpn.P6's shape class)
    cin, cout = 2048, 256
    x, w = nhwc(rnd((1, cin, 17, 15), 30)).to(dev), rnd((cout, cin, 3, 3), 31, 0.01)
    y = cv.fprop(x, cv.pack_weights(w.to(dev), 0), cout, 3, 2, 1, shift=rnd((cout,), 32).to(dev), sign=True)
    assert torch.equal(y._rn_sign, packed_sign(y))


def test_maxpool_backward_reads_the_stem_bits(cv, dev):
    x = torch.randn(2, 37, 45, 64, device=dev)
    y, arg = cv.maxpool_fwd(x, want_argmax=True)
    dy = torch.randn_like(y)
    xb = x.clone()
    xb._rn_sign = packed_sign(x)
    assert torch.equal(cv.maxpool_bwd(xb, dy, arg, relu_mask=True), cv.maxpool_bwd(x, dy, arg, relu_mask=True))


@pytest.mark.parametrize("wino", [True, False])
def test_training_step_gradients_are_bit_identical_with_and_without_bits(dev, wino):
    """ResNet-50 at the goldens' size, fixed-order weight-gradient reductions: every parameter gradient the same bits either way,
    and the engine really took the bits (every ReLU output of the saved forward carries them)."""
    import golden_cases as gc
    from retinanet_mi355x import conv, modules
    before = conv.BITMASKS, conv.get_option(conv.OPT_DETERMINISTIC)
    conv.set_deterministic(True)
    try:
        grads = {}
        for on in (True, False):
            conv.BITMASKS = on
            fn, sd, img, ann = gc.model_case("resnet50", True)
            net = modules.resnet50(num_classes=4)
            net.load_state_dict(sd)
            net = net.to(dev).train()
            net.freeze_bn()
            net._engine.use_wino = wino
            if on:
                with torch.no_grad():
                    S = net._engine.forward(net._tensor_dict(), img.to(dev), save=True)[2]
                acts = net._engine.relu_outputs(S)
                assert len(acts) > 60 and all(getattr(t, "_rn_sign", None) is not None for t in acts.values()), \
                    [n for n, t in acts.items() if getattr(t, "_rn_sign", None) is None]
                for name, t in acts.items():
                    assert torch.equal(t._rn_sign, packed_sign(t)), name
            sum(l.mean() for l in net([img.to(dev), ann.to(dev)])).backward()
            grads[on] = {n: p.grad.clone() for n, p in net.named_parameters()}
        for n in grads[True]:
            assert torch.equal(grads[True][n], grads[False][n]), n
    finally:
        conv.BITMASKS = before[0]
        conv.set_deterministic(before[1])


# ------------------------------------------------------------------------------------------------ the bf16 engine's tensors
def test_bf16_producer_and_consumer(dev):
    """conv_bf16.hip: a lane finishes 8 consecutive bf16 channels = one byte of sign bits, four lanes a word."""
    from retinanet_mi355x import conv as cv
    before = cv.BITMASKS
    cv.BITMASKS = True
    try:
        cin, cout, N, H, W = 64, 256, 2, 19, 23
        x, w, b = rnd((N, cin, H, W), 1), rnd((cout, cin, 3, 3), 2, 0.06), rnd((cout,), 3)
        xb = cv.to_bf16(nhwc(x).to(dev))
        wp = cv.pack_weights_bf16(w.to(dev), 0)
        y = torch.empty((N, H, W, cout), dtype=torch.bfloat16, device=dev)
        cv.conv_igemm_bf16(xb, wp, y, (H, W, cout, 3, 3, 1, 1, -1, 0), shift=b.to(dev), act=cv.ACT_RELU, sign=True)
        assert 0.2 < float((y > 0).float().mean()) < 0.8
        assert torch.equal(y._rn_sign, packed_sign(y.float()))
        # consumer: the data gradient of a 3x3 layer masked by y, from the bits and from the tensor
        gy = cv.to_bf16(nhwc(rnd((N, 128, H, W), 6)).to(dev))
        wd = cv.pack_weights_bf16(rnd((128, cout, 3, 3), 5, 0.03).to(dev), 1)
        plain = y.clone()
        for mode in (1, 2):
            a = cv.dgrad_bf16(gy, wd, (H, W), cout, 3, 1, mask=y, mask_mode=mode)
            b_ = cv.dgrad_bf16(gy, wd, (H, W), cout, 3, 1, mask=plain, mask_mode=mode)
            assert torch.equal(a, b_)
    finally:
        cv.BITMASKS = before


def test_bf16_training_step_with_and_without_bits(dev):
    """The bf16 engine: same losses (the forward does not change), every ReLU output carries its bits, parameter gradients equal up to
    the order of the bf16 weight gradient's fp32 atomics (1e-6 relative)."""
    import golden_cases as gc
    from retinanet_mi355x import conv, modules
    before = conv.BITMASKS, conv.BITMASKS_BF16
    conv.BITMASKS_BF16 = True                              # (opt-in for this engine: measured neutral on its step)
    try:
        out = {}
        for on in (True, False):
            conv.BITMASKS = on
            fn, sd, img, ann = gc.model_case("resnet50", True)
            net = modules.resnet50(num_classes=4)
            net.load_state_dict(sd)
            net = net.to(dev)
            net.set_compute_dtype("bf16")
            net.train()
            net.freeze_bn()
            if on:
                with torch.no_grad():
                    S = net._engine.forward(net._tensor_dict(), img.to(dev), save=True)[2]
                acts = net._engine.relu_outputs(S)
                missing = [n for n, t in acts.items() if getattr(t, "_rn_sign", None) is None]
                assert not missing, missing
            losses = net([img.to(dev), ann.to(dev)])
            sum(l.mean() for l in losses).backward()
            out[on] = ([float(l.detach()) for l in losses], {n: p.grad.clone() for n, p in net.named_parameters()})
        assert out[True][0] == out[False][0]
        for n, g in out[True][1].items():
            h = out[False][1][n]
            assert float((g - h).double().norm()) <= 1e-6 * float(h.double().norm()) + 1e-30, n
    finally:
        conv.BITMASKS, conv.BITMASKS_BF16 = before
