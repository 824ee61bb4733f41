"""BASELINE configs[1] at the benchmark's batch: ResNet-50 directional, 1920x1080, BATCH 8 (what bench.py times), and the
fixed-order weight-gradient mode.

The oracle needs ~16 s of CPU per 1080p image, so batch 8 is pinned through the batch-1 runs, which ARE oracle-checked
(tests/test_gpu_model.py::test_cfg2_full_size_against_oracle; image 0 here is that test's image):

* with the split-K form off (RN_OPT_SPLITK = 0) every output element of every convolution is ONE K loop in a fixed order,
  whatever the number of images in the launch -- so the forward of image i inside the batch-8 launch must be BIT-IDENTICAL to
  the forward of image i alone.  That witnesses the batch strides, the Winograd tile count T (padded to 256; depends on B),
  the grouped launches' per-problem tables and the pooled argmax at B = 8 element by element (D/model.py:284-306);
* the three losses of the batch are the means of the eight per-image losses (D/losses.py:47-357: per-image normalisation,
  then the mean over the batch), and every parameter gradient is the mean of the eight per-image gradients.  The masks are
  identical (bit-identical forward), so what remains is the order of the fp32 sums: measured 1e-7 .. 3e-6 per tensor
  (L2-relative), bound 1e-5;
* image 0 of the eight against oracle.model.train_forward (losses 1e-4).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, B = 1080, 1920, 8
LOSS_TOL = 1e-6
GRAD_L2_TOL = 1e-5


@pytest.fixture(params=["native", "split", "split3"])
def mfma(request, dev):
    from retinanet_mi355x import conv
    before = conv.get_fp32_mfma()
    conv.set_fp32_mfma(request.param)
    yield request.param
    conv.set_fp32_mfma(before)


@pytest.fixture
def no_splitk(dev):
    from retinanet_mi355x import conv
    before = conv.get_option(conv.OPT_SPLITK)
    conv.set_option(conv.OPT_SPLITK, False)
    yield
    conv.set_option(conv.OPT_SPLITK, before)


def _net(dev, wino):
    from retinanet_mi355x import modules, synth
    net = modules.resnet50(num_classes=8)
    net.load_state_dict(synth.state_dict("resnet50", 8, 12, seed=2))
    net = net.to(dev)
    net._engine.use_wino = wino
    net.train()
    net.freeze_bn()
    return net


def _step(net, img, ann):
    for p in net.parameters():
        p.grad = None
    losses = net([img, ann])
    sum(l.mean() for l in losses).backward()
    return [float(l.detach()) for l in losses], {n: p.grad.detach().clone() for n, p in net.named_parameters()}


@pytest.mark.parametrize("mode", ["wino", "direct"])
def test_cfg2_batch8_equals_the_eight_single_image_runs(dev, mode, mfma, no_splitk):
    from retinanet_mi355x import synth
    net = _net(dev, mode == "wino")
    eng = net._engine
    img = synth.frames(B, H, W, seed=0).to(dev)                    # frames(8)[0] == frames(1)[0]: the oracle-checked image
    ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)
    # ---- forward, element by element
    with torch.no_grad():
        reg8, cls8, _ = eng.forward(net._tensor_dict(), img, save=True)
        for i in (0, 3, 7):
            reg1, cls1, _ = eng.forward(net._tensor_dict(), img[i:i + 1], save=True)
            assert torch.equal(reg8[i:i + 1], reg1), "regression head of image %d differs inside the batch" % i
            assert torch.equal(cls8[i:i + 1], cls1), "classification head of image %d differs inside the batch" % i
        del reg8, cls8, reg1, cls1
    # ---- losses and gradients
    loss8, grad8 = _step(net, img, ann)
    mean_loss = np.zeros(3)
    mean_grad = {n: torch.zeros_like(g, dtype=torch.float64) for n, g in grad8.items()}
    first = None
    for i in range(B):
        l1, g1 = _step(net, img[i:i + 1], ann[i:i + 1])
        first = l1 if first is None else first
        mean_loss += np.array(l1) / B
        for n, g in g1.items():
            mean_grad[n] += g.double() / B
    assert np.allclose(loss8, mean_loss, rtol=LOSS_TOL, atol=0), (loss8, mean_loss.tolist())
    worst = ("", 0.0)
    for n, g in grad8.items():
        want = mean_grad[n]
        err = float((g.double() - want).norm() / (want.norm() + 1e-30))
        worst = max(worst, (n, err), key=lambda t: t[1])
        assert err <= GRAD_L2_TOL, "%s: batch-8 gradient vs mean of the single-image gradients: %.3e" % (n, err)
    print("batch 8 vs 8 x batch 1 [%s, %s]: losses %s, worst gradient %s %.2e" % (mode, mfma, loss8, worst[0], worst[1]))
    # ---- image 0 against the oracle (the CPU run test_gpu_model.py shares: ~16 s once per session)
    import test_gpu_model as tgm
    o = tgm._cfg2_oracle()
    assert np.allclose(first, o["losses"], rtol=1e-4), (first, o["losses"])


@pytest.mark.parametrize("mode", ["wino", "direct"])
def test_deterministic_weight_gradients_are_bit_reproducible(dev, mode, mfma):
    """RN_OPT_DETERMINISTIC: K slices into slabs + one ordered combine instead of fp32 atomics -- two runs of the same step give
    bit-identical gradients for every parameter (the reference's CPU autograd is reproducible; the default mode's gradients
    differ in their last bits from run to run)."""
    from retinanet_mi355x import conv, synth
    net = _net(dev, mode == "wino")
    h, w = 360, 640                                                 # every layer has several K slices at this size
    img = synth.frames(2, h, w, seed=5).to(dev)
    ann = synth.labels_dir(2, 6, h, w, 8, seed=6, size_px=(40, 120)).to(dev)
    before = conv.get_option(conv.OPT_DETERMINISTIC)
    conv.set_deterministic(True)
    try:
        l_a, g_a = _step(net, img, ann)
        l_b, g_b = _step(net, img, ann)
    finally:
        conv.set_option(conv.OPT_DETERMINISTIC, before)
    assert l_a == l_b
    for n in g_a:
        assert torch.equal(g_a[n], g_b[n]), "%s differs between two runs in deterministic mode" % n
    # and it is the same gradient as the atomic form up to summation order
    l_c, g_c = _step(net, img, ann)
    for n in g_a:
        err = float((g_a[n].double() - g_c[n].double()).norm() / (g_c[n].double().norm() + 1e-30))
        assert err <= 1e-5, (n, err)


@pytest.mark.parametrize("mode", ["wino", "direct"])
def test_tower_streams_change_no_bit(dev, mode, mfma, monkeypatch):
    """The two head towers run on a side stream each (engine.py: tower_streams; forward and backward).  In deterministic mode the
    step's result does not depend on timing, so the run on two streams must equal the run on the caller's stream BIT FOR BIT -- any
    missing event (a tower reading the pyramid gradient, a Winograd scratch or a saved transform too early) shows up here."""
    from retinanet_mi355x import conv, synth
    net = _net(dev, mode == "wino")
    h, w = 360, 640
    img = synth.frames(2, h, w, seed=7).to(dev)
    ann = synth.labels_dir(2, 6, h, w, 8, seed=8, size_px=(40, 120)).to(dev)
    before = conv.get_option(conv.OPT_DETERMINISTIC)
    conv.set_deterministic(True)
    try:
        monkeypatch.setenv("RN_TOWER_STREAMS", "0")
        assert net._engine.tower_streams(dev) is None
        l_one, g_one = _step(net, img, ann)
        monkeypatch.setenv("RN_TOWER_STREAMS", "1")
        assert net._engine.tower_streams(dev) is not None
        for _ in range(3):                                          # several runs: a race need not show the first time
            l_two, g_two = _step(net, img, ann)
            assert l_one == l_two
            for n in g_one:
                assert torch.equal(g_one[n], g_two[n]), "%s differs between one stream and two" % n
    finally:
        conv.set_option(conv.OPT_DETERMINISTIC, before)


@pytest.mark.parametrize("mode", ["wino", "direct"])
def test_weight_gradient_stream_changes_no_bit(dev, mode, mfma, monkeypatch):
    """Round 5: the backbone's, the pyramid's and the stem's weight gradients run on a side stream (engine.py: wgrad_streams; for Winograd
    layers the 36 reductions only), forked where a layer's output gradient is ready and joined where a bucket's unpack launch reads the
    accumulators; their operands are held by reference until that join.  In deterministic mode the step's result does not depend on
    timing: with the stream off, on, and with two of them the gradients must be equal BIT FOR BIT -- a missing event (a reduction reading
    A dy A^T before its transform, an accumulator unpacked before its last weight gradient, an operand's memory handed out again too
    early, an amax table made on the wrong stream) shows up here."""
    from retinanet_mi355x import conv, synth
    net = _net(dev, mode == "wino")
    h, w = 360, 640
    img = synth.frames(2, h, w, seed=9).to(dev)
    ann = synth.labels_dir(2, 6, h, w, 8, seed=10, size_px=(40, 120)).to(dev)
    before = conv.get_option(conv.OPT_DETERMINISTIC)
    conv.set_deterministic(True)
    try:
        monkeypatch.setenv("RN_WGRAD_STREAMS", "0")
        assert net._engine.wgrad_streams(dev) is None
        l_one, g_one = _step(net, img, ann)
        for n_streams in ("1", "2"):
            monkeypatch.setenv("RN_WGRAD_STREAMS", n_streams)
            assert len(net._engine.wgrad_streams(dev)) == int(n_streams)
            for _ in range(3):                                      # several runs: a race need not show the first time
                l_two, g_two = _step(net, img, ann)
                assert l_one == l_two
                for n in g_one:
                    assert torch.equal(g_one[n], g_two[n]), "%s differs with %s weight-gradient stream(s)" % (n, n_streams)
    finally:
        conv.set_option(conv.OPT_DETERMINISTIC, before)
