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


@pytest.mark.parametrize("arch", ["resnet50", "resnet101", "resnet34"])
def test_fp8_forward_against_the_fp32_goldens(dev, golden, arch):
    from retinanet_mi355x import modules, synth
    z = golden(gc.MODEL_CASES[arch][0])
    fn, sd, img, ann = gc.model_case(arch, True)
    net = getattr(modules, arch)(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    img = img.to(dev)
    H, W = img.shape[2], img.shape[3]
    other = synth.frames(img.shape[0], H, W, seed=123).to(dev)
    scales = net.calibrate_fp8(torch.cat([img, other]))
    assert net._engine.fp8 and len(scales) > 30
    boxes, cls = net(img, LOCALIZE=True)
    assert cls.dtype == torch.float32 and torch.isfinite(cls).all() and torch.isfinite(boxes).all()
    s_max, s_rms = _rel(cls.cpu().numpy(), z["%s_dir_cls" % arch])
    b_max, b_rms = _rel(boxes.cpu().numpy(), z["%s_dir_boxes" % arch])
    print("fp8 %s: scores max %.3e rms %.3e | boxes max %.3e rms %.3e" % (arch, s_max, s_rms, b_max, b_rms))
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
