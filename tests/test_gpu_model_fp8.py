"""BASELINE configs[4]: the detector's forward pass with e4m3 activations and weights on the fp8 MFMA (net.calibrate_fp8:
csrc/conv_fp8*.hip, Engine(dtype="fp8")), against the REFERENCE's fp32 outputs (tests/golden/model*.npz), and (round 5) the training step
built on it (fp8 forward, bf16 gradients).

The reference has no fp8 mode: fp8 parity is unpinned by it; what these tests pin is that the fp8 schedule computes the same function
within a stated error budget.  The budget (tools/fp8_error_budget.py, profiles/r05_fp8_error_budget.txt: ResNet-101 at 1080p, one
layer group at a time in e4m3): with EVERY layer in e4m3 the scores are 14.7 % (max) / 5.7 % (rms) off the fp32 forward, and most of
that comes from the residual stream being re-quantised at every block and from the pyramid; with the last convolution of every
bottleneck, the shortcuts and the FPN in bf16 -- the engine's default since round 5 -- 8.0 % / 2.0 %, boxes 3.1 % / 1.0 %.  The bounds
below are that design target (scores <= 8 % of the largest score, boxes <= 5 % of the largest coordinate), not the last measurement
(measured on the goldens -- ResNet-50 / ResNet-101 at 72x104, ResNet-34 at 112x112, scales from the test image and one other frame or
from two disjoint frames: scores 3.2-4.6 %, boxes 2.0-2.9 %).
"""
import numpy as np
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu

SCORE_TOL, BOX_TOL = 0.08, 0.05


def _rel(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / (np.abs(want).max() + 1e-12)), float(np.sqrt(((got - want) ** 2).mean()) / (np.sqrt((want ** 2).mean()) + 1e-12))


@pytest.mark.parametrize("calib", ["with_eval_frame", "disjoint"])
@pytest.mark.parametrize("arch", ["resnet50", "resnet101", "resnet34"])
def test_fp8_forward_against_the_fp32_goldens(dev, golden, arch, calib):
    """calib "disjoint" (round 4): the activation scales come from frames the evaluation never sees (two other seeds of the same
    generator, margin 1.25 for the magnitudes they did not reach) -- what a deployment does; "with_eval_frame" keeps round 3's
    set-up (the evaluated frame is one of the two calibration frames) as the best case beside it."""
    from retinanet_mi355x import modules, synth
    z = golden(gc.MODEL_CASES[arch][0])
    fn, sd, img, ann = gc.model_case(arch, True)
    net = getattr(modules, arch)(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    img = img.to(dev)
    H, W = img.shape[2], img.shape[3]
    other = synth.frames(img.shape[0], H, W, seed=123).to(dev)
    if calib == "disjoint":
        scales = net.calibrate_fp8(torch.cat([other, synth.frames(img.shape[0], H, W, seed=124).to(dev)]), margin=1.25)
    else:
        scales = net.calibrate_fp8(torch.cat([img, other]))
    assert net._engine.fp8 and len(scales) > 30 and set(net._engine.fp8_scale_names()) <= set(scales)
    boxes, cls = net(img, LOCALIZE=True)
    assert cls.dtype == torch.float32 and torch.isfinite(cls).all() and torch.isfinite(boxes).all()
    s_max, s_rms = _rel(cls.cpu().numpy(), z["%s_dir_cls" % arch])
    b_max, b_rms = _rel(boxes.cpu().numpy(), z["%s_dir_boxes" % arch])
    print("fp8 %s (%s): scores max %.3e rms %.3e | boxes max %.3e rms %.3e" % (arch, calib, s_max, s_rms, b_max, b_rms))
    assert s_max <= SCORE_TOL and b_max <= BOX_TOL, (s_max, b_max)
    # the other eval modes run on the same tensors
    s, c, b = net(img[:1])
    assert s.shape[0] == c.shape[0] == b.shape[0]
    if arch == "resnet34":                                    # the training step (fp8 forward, bf16 gradients) exists for the bottleneck networks
        with pytest.raises(RuntimeError, match="bottleneck"):
            net.train()
            net([img, ann.to(dev)])
    net.eval()
    net.set_compute_dtype("fp32")                             # and back: the fp32 engine again meets 1e-4
    _, cls32 = net(img, LOCALIZE=True)
    assert _rel(cls32.cpu().numpy(), z["%s_dir_cls" % arch])[0] <= 1e-4


def test_calibration_runs_on_the_fp32_direct_kernels_whatever_the_current_engine(dev, golden):
    """Round-3 advisor finding: the magnitudes are recorded after the fp32 direct kernels only, so a calibration started from a
    bf16 engine (or with Winograd on in eval) recorded a handful of layers and the fp8 forward silently used scale 0 for the rest.
    Now calibrate_fp8 switches to the fp32 direct path itself -- the scales from a bf16 model equal those from an fp32 one -- and
    an fp8 forward with a scale missing raises instead of saturating."""
    from retinanet_mi355x import modules
    fn, sd, img, ann = gc.model_case("resnet50", True)
    img = img.to(dev)
    nets = []
    for start in ("fp32", "bf16", "wino_eval"):
        net = modules.resnet50(num_classes=4)
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        if start == "bf16":
            net.set_compute_dtype("bf16")
        if start == "wino_eval":
            net._engine.wino_eval = True
        nets.append((net, net.calibrate_fp8(img)))
        assert net._engine.fp8
    ref = nets[0][1]
    for net, scales in nets[1:]:
        assert scales.keys() == ref.keys() and all(scales[k] == ref[k] for k in ref)
    z = golden(gc.MODEL_CASES["resnet50"][0])
    _, cls = nets[1][0](img, LOCALIZE=True)
    assert _rel(cls.cpu().numpy(), z["resnet50_dir_cls"])[0] <= SCORE_TOL
    net = nets[2][0]
    del net._engine.fp8_scales["layer3.2.conv2"]
    with pytest.raises(RuntimeError, match="scales are missing"):
        net(img, LOCALIZE=True)


@pytest.mark.parametrize("arch", ["resnet50", "resnet101"])
def test_fp8_training_step_against_the_fp32_goldens(dev, golden, arch):
    """Round 5, BASELINE configs[4] as a TRAINING configuration: forward on the fp8 kernels (e4m3 activations saved), loss in fp32,
    data and weight gradients on the bf16 kernels reading the e4m3 activations through rn_fp8_to_bf16.  Against the reference's fp32
    goldens: the three losses within 5 % (measured 0.3-2 %: per-tensor e4m3 activations through 50 / 101 layers), every parameter
    gradient finite, and in DIRECTION: cosine >= 0.9 for at least 85 % of the tensors (measured: printed below).  The reference has no
    fp8 mode: this pins that the mixed schedule computes the same function."""
    from retinanet_mi355x import modules, synth
    z = golden(gc.MODEL_CASES[arch][0])
    fn, sd, img, ann = gc.model_case(arch, True)
    net = getattr(modules, arch)(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    img, ann = img.to(dev), ann.to(dev)
    other = synth.frames(img.shape[0], img.shape[2], img.shape[3], seed=123).to(dev)
    net.calibrate_fp8(torch.cat([img, other]), margin=1.25)
    net.train()
    net.freeze_bn()
    losses = net([img, ann])
    got = np.array([float(l.detach()) for l in losses])
    want = z["%s_dir_losses" % arch]
    sum(l.sum() for l in losses).backward()
    cos = []
    for name, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        full = "%s_dir_g_%s" % (arch, name)
        if full in z.files and np.abs(z[full]).max() > 0:
            g, w = p.grad.detach().cpu().numpy().ravel().astype(np.float64), z[full].ravel().astype(np.float64)
            cos.append(float(g @ w / (np.linalg.norm(g) * np.linalg.norm(w) + 1e-300)))
    cos = np.array(cos)
    print("fp8-forward training %s: losses %s vs %s (rel %s); gradient cosine min %.4f median %.4f, >= 0.9: %.0f %%"
          % (arch, got, want, np.abs(got - want) / np.abs(want), cos.min(), np.median(cos), 100 * (cos >= 0.9).mean()))
    # (ResNet-101's golden case has a classification loss of 34: random weights drive many scores into the sigmoid's tails, where
    # -log(1 - p) amplifies a 4 % score error; its bound is 15 %, every other loss 5 %)
    tol = np.array([0.15 if arch == "resnet101" else 0.05, 0.05, 0.05])
    assert np.all(np.abs(got - want) <= tol * np.abs(want)), (got, want)
    assert (cos >= 0.9).mean() >= 0.85 and np.median(cos) >= 0.95, (cos.min(), np.median(cos))
