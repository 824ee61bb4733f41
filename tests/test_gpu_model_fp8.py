"""BASELINE configs[4], first cut: the detector's forward pass with e4m3 activations and weights on the fp8 MFMA
(net.calibrate_fp8: csrc/conv_fp8.hip, Engine(dtype="fp8")), against the REFERENCE's fp32 outputs (tests/golden/model*.npz).

The reference has no fp8 mode: fp8 parity is unpinned by it; what these tests pin is that the fp8 schedule computes the same
function within what 3-bit significands allow.  Measured on the goldens (ResNet-50 / ResNet-101 at 72x104, ResNet-34 at
112x112; activation scales calibrated on the test image and one other frame): classification scores (post-sigmoid, ~0.01)
within 3.4-6.9 % of the largest score (2.1-2.6 % rms), decoded boxes within 2.4-3.3 % of the largest coordinate (1.2-2.3 % rms);
bounds ~2x the measured maxima.  Inference only.
"""
import numpy as np
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu

SCORE_TOL, BOX_TOL = 0.12, 0.08


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
    with pytest.raises(RuntimeError, match="inference-only"):
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
