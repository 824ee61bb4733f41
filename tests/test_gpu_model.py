"""GPU parity of the whole detector (drop-in ResNet on the HIP engine) against the golden vectors the
REFERENCE produced on CPU (tests/golden/model.npz) and against the oracle.

Tolerances: fp32 losses / boxes / scores within 1e-4 relative (north_star).  Parameter gradients: L2 norm within
1e-3, and element-wise L2-relative error <= 2e-3 with no element further than 1e-2 of the gradient's max
magnitude.  (Measured: ~1e-6 everywhere, except where ONE pre-activation lies within fp32 rounding of zero and
the different summation order of the MFMA tiles flips its ReLU mask -- that moves a single bias-gradient element
by ~3e-3 of the max; seen once on ResNet-18 layer1.1.bn1.bias, see tools/dbg_grad.py.)
"""
import numpy as np
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


def _build(arch, directional, dev):
    from retinanet_mi355x import modules
    sd, img, ann = gc.model_inputs(arch, directional)
    net = getattr(modules, arch)(num_classes=4, directional=directional)
    missing = net.load_state_dict(sd)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.to(dev), img.to(dev), ann.to(dev), sd


def rel_close(got, want, tol):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want).max()
    assert err <= tol * (np.abs(want).max() + 1e-12), "max err %.3e vs max|ref| %.3e" % (err, np.abs(want).max())


def grad_close(got, want, name):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    l2 = np.sqrt(((got - want) ** 2).sum()) / (np.sqrt((want ** 2).sum()) + 1e-30)
    mx = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
    assert l2 <= 2e-3 and mx <= 1e-2, "%s: L2-rel %.3e, max-rel %.3e" % (name, l2, mx)


def test_state_dict_keys_match_reference_layout(dev):
    from retinanet_mi355x import arch, modules
    for name, n_entries in (("resnet18", 156), ("resnet50", 354), ("resnet101", 660)):
        net = getattr(modules, name)(num_classes=8)
        sd = net.state_dict()
        assert len(sd) == n_entries
        want = arch.state_dict_shapes(name, 8, 12)
        assert list(sd.keys()) == list(want.keys())
        assert all(tuple(sd[k].shape) == tuple(v) for k, v in want.items())
    children = [n for n, _ in modules.resnet18(num_classes=8).named_children()]
    assert children == ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4", "fpn",
                        "regressionModel", "classificationModel", "anchors", "regressBoxes", "clipBoxes", "focalLoss"]


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_directional_eval_localize(dev, golden, arch):
    z = golden("model")
    net, img, ann, _ = _build(arch, True, dev)
    net.eval()
    boxes, cls = net(img, LOCALIZE=True)
    rel_close(cls.cpu().numpy(), z["%s_dir_cls" % arch], 1e-4)
    rel_close(boxes.cpu().numpy(), z["%s_dir_boxes" % arch], 1e-4)


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_directional_train_losses_and_gradients(dev, golden, arch):
    z = golden("model")
    net, img, ann, _ = _build(arch, True, dev)
    net.train()
    net.freeze_bn()
    cls_l, reg_l, vp_l = net([img, ann])
    got = [float(cls_l), float(reg_l), float(vp_l)]
    assert np.allclose(got, z["%s_dir_losses" % arch], rtol=1e-4), (got, z["%s_dir_losses" % arch])
    (cls_l + reg_l + vp_l).sum().backward()
    checked = 0
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        key = "%s_dir_gsum_%s" % (arch, name)
        g = p.grad.detach().cpu().numpy().astype(np.float64)
        ref_norm = z[key][2]
        assert abs(np.sqrt((g ** 2).sum()) - ref_norm) <= 1e-3 * ref_norm + 1e-9, (name, np.sqrt((g ** 2).sum()), ref_norm)
        full = "%s_dir_g_%s" % (arch, name)
        if full in z.files:
            grad_close(g, z[full], name)
            checked += 1
    assert checked > 30


def test_flat2d_train_and_eval(dev, golden):
    z = golden("model")
    net, img, ann, _ = _build("resnet18", False, dev)
    net.train()
    net.freeze_bn()
    out = net([img, ann])
    assert len(out) == 2
    assert np.allclose([float(x) for x in out], z["resnet18_2d_losses"], rtol=1e-4)
    (out[0] + out[1]).sum().backward()
    for name, p in net.named_parameters():
        ref_norm = z["resnet18_2d_gsum_" + name][2]
        gn = float(p.grad.double().norm())
        assert abs(gn - ref_norm) <= 1e-3 * ref_norm + 1e-9, (name, gn, ref_norm)
    net.eval()
    boxes, cls = net(img, LOCALIZE=True)
    rel_close(cls.cpu().numpy(), z["resnet18_2d_cls"], 1e-4)
    rel_close(boxes.cpu().numpy(), z["resnet18_2d_boxes"], 1e-4)


def test_eval_modes_against_oracle(dev):
    """Single-frame, MULTI_FRAME and LOCALIZE outputs of the engine vs the oracle model run on CPU with the same
    weights; post-process compared on the oracle's own decoded inputs where ordering could depend on 1e-7 ties."""
    from oracle import model as omodel
    net, img, ann, sd = _build("resnet18", True, dev)
    net.eval()
    with torch.no_grad():
        o_boxes, o_cls = omodel.eval_forward(img.cpu(), sd, "resnet18", LOCALIZE=True)
    boxes, cls = net(img, LOCALIZE=True)
    rel_close(cls.cpu().numpy(), o_cls.numpy(), 1e-4)
    rel_close(boxes.cpu().numpy(), o_boxes.numpy(), 1e-4)
    s, c, b = net(img[:1])
    assert s.shape[0] == c.shape[0] == b.shape[0] and b.shape[1] == 20 and c.dtype == torch.int64
    s4, c4, b4, im4 = net(img, MULTI_FRAME=True)
    assert s4.shape[0] == im4.shape[0] and set(im4.cpu().tolist()) <= {0, 1}


def test_head_reinit_invalidates_packed_weights(dev):
    """Callers replace head Parameters after construction (train_detector_3D_angle.py:290-291); the packed copies
    must follow."""
    net, img, ann, _ = _build("resnet18", True, dev)
    net.eval()
    _, cls0 = net(img, LOCALIZE=True)
    net.classificationModel.output.weight = torch.nn.Parameter(
        torch.rand([9 * 4, 256, 3, 3], device=dev) * 1e-2)
    _, cls1 = net(img, LOCALIZE=True)
    assert float((cls1 - cls0).abs().max()) > 1e-4
    with torch.no_grad():
        net.classificationModel.output.bias.add_(1.0)                       # in-place update bumps the version
    _, cls2 = net(img, LOCALIZE=True)
    assert float((cls2 - cls1).abs().max()) > 1e-3


def test_training_step_changes_loss(dev):
    """Adam + clip_grad_norm_ as the reference trainer does (train_detector_3D_angle.py:337, 383-387)."""
    net, img, ann, _ = _build("resnet18", True, dev)
    net.train()
    net.freeze_bn()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    first = None
    for _ in range(3):
        opt.zero_grad()
        losses = net([img, ann])
        loss = sum(l.mean() for l in losses)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 0.1)
        opt.step()
        first = float(loss) if first is None else first
    assert float(loss) < first


def test_fused_clip_adam_matches_torch(dev):
    """retinanet_mi355x.optim.ClipAdam == clip_grad_norm_(0.1) + torch.optim.Adam(lr=1e-4) over 3 steps, and the
    engine's packed-weight cache notices the in-place update."""
    from retinanet_mi355x import optim
    torch.manual_seed(0)
    shapes = [(64, 3, 7, 7), (64,), (5000,), (256, 256, 3, 3), (1,)]
    pa = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.1) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ref = torch.optim.Adam(pb, lr=1e-4)
    mine = optim.ClipAdam(pa, lr=1e-4, max_norm=0.1)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(mine, mode="min", patience=0)          # the trainer's scheduler
    sched.step(1.0); sched.step(2.0)
    assert mine.param_groups[0]["lr"] < 1e-4
    mine.param_groups[0]["lr"] = 1e-4
    for it in range(3):
        grads = [torch.randn(s, device=dev) * (10.0 if it == 1 else 0.001) for s in shapes]   # clipped and unclipped steps
        for p, q, g in zip(pa, pb, grads):
            p.grad, q.grad = g.clone(), g.clone()
        v0 = pa[0]._version
        want_norm = torch.nn.utils.clip_grad_norm_(pb, 0.1)
        ref.step()
        got_norm = mine.step()
        assert pa[0]._version > v0
        assert abs(float(got_norm) - float(want_norm)) <= 1e-5 * float(want_norm)
        for p, q in zip(pa, pb):
            assert float((p - q).abs().max()) <= 1e-6 * float(q.abs().max()) + 1e-9
            assert float((p.grad - q.grad).abs().max()) <= 1e-6 * float(q.grad.abs().max()) + 1e-12   # clipped grads written back


def test_cfg1_resnet18_2d_512(dev):
    """BASELINE configs[0]: ResNet-18 2D RetinaNet, 512x512 random tensors, 10 random GT boxes, batch 2,
    forward + FocalLoss -- against the oracle on CPU with the same weights."""
    from oracle import model as omodel
    from retinanet_mi355x import modules, synth
    sd = synth.state_dict("resnet18", num_classes=8, n_reg=4, seed=2, head_scale=3e-4)
    img = synth.frames(2, 512, 512, seed=0)
    ann = synth.labels_2d(2, 10, 512, 512, 8, seed=1, size_px=(40, 160))
    with torch.no_grad():
        want = omodel.train_forward(img, ann, sd, "resnet18")
    net = modules.resnet18(num_classes=8, directional=False)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    got = net([img.to(dev), ann.to(dev)])
    assert len(got) == 2
    assert np.allclose([float(x) for x in got], [float(x) for x in want], rtol=1e-4), (got, want)
