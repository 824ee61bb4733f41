"""GPU parity of the whole detector (drop-in ResNet on the HIP engine) against the golden vectors the
REFERENCE produced on CPU (tests/golden/model.npz, model_deep.npz) and against the oracle.

Tolerances: fp32 losses / boxes / scores within 1e-4 relative (north_star).  Parameter gradients are compared tensor by
tensor (L2-relative error, and the largest element error relative to the gradient's max magnitude).  Measured on the
MI355X (profiles/r02_gradient_errors.txt): at the goldens' small sizes 3e-7..2e-6 with the direct kernels and 1e-6..1e-5
with the Winograd F(4x4,3x3) layers on; the bounds are ~10x that (GRAD_L2).  ONE effect is outside them and is not an
error of the kernels: an activation that lies within rounding of zero (|x| ~ 1e-6 of the layer's max) comes out on the
other side of the ReLU than in the CPU run, its mask flips, and everything upstream of that element moves -- measured:
ResNet-18, batch 2, ONE flipped mask in 17 920 (regressionModel.conv3 at the P4 level, activation 1.4e-6 against a max of
27) puts fpn.P4_2.bias at 3.8e-4 (tools/dbg/wino_bias.py).  Which run has such an element is a matter of its last bit
-- kernel family, product mode (tools/dbg/mode_flips.py: one element at 6e-8 of its tensor's maximum in layer1.1 of
ResNet-18 moves the 8 tensors below it by 2e-4 .. 1.3e-3; one at 8e-8 in layer4.0 of ResNet-101, 3x4 pixels there at the
golden's size, moves the 189 tensors below it by ~8e-4 and layer4.0.bn1.bias by 4.8e-3).  So: at least FLIP_FREE of
the compared tensors must meet the tight bound and every tensor the loose one (GRAD_L2_FLIP / GRAD_MAX) -- a systematic
error 10x the measured one fails the first -- OR the test LOCATES the flipped element(s): every ReLU output of this GPU run
is compared, element by element, with the oracle's CPU run (located_flips: Engine.relu_outputs vs oracle.model's taps); at
most MAX_FLIPS elements may differ, only the parameters inside their backward cone (Engine.backward_cone: the producing
layer and what the backward pass runs after it along the data-gradient path) may use the bounds a flip can reach at these
sizes (FLIPPED_L2 / FLIPPED_MAX), and EVERY tensor outside the cone must meet the tight bound.  The kernels themselves are held to 1e-4 .. 2e-5 against fp64 in
every mode by tests/test_gpu_conv.py, where no ReLU intervenes.
At the full benchmark size (test_cfg2_full_size_against_oracle: 1e8 activations, hundreds of such flips, accumulating
towards the stem) heads and FPN stay at <= 1.1e-5 and the backbone reaches 1e-4 .. 6e-4 with EITHER kernel family, so
that test has its own bounds per group, again ~10x the measured values.
"""
import json
import os

import numpy as np
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu

GRAD_L2 = {"direct": 3e-5, "wino": 5e-5}             # tight: ~10x measured, >= FLIP_FREE of the tensors
GRAD_L2_FLIP, GRAD_MAX = 2e-3, 5e-3                   # loose: a flipped ReLU mask upstream (see above), every tensor
NORM_TOL = {"direct": 2e-5, "wino": 2e-4}
CFG2_L2 = {"head": 1e-4, "backbone": 2e-3}            # measured 1.1e-5 / 6.1e-4 .. 7.7e-4 (round 4: backbone 5e-3 -> 2e-3, and only
                                                      # inside the backward cone of a LOCATED sign flip: test_cfg2_full_size_against_oracle)
STATS = {}


def _note(test, name, l2, mx):
    w = STATS.setdefault(test, {"worst_l2": 0.0, "worst_max": 0.0, "n": 0})
    if l2 >= w["worst_l2"]:
        w["worst_l2"], w["worst_l2_name"] = l2, name
    if mx >= w["worst_max"]:
        w["worst_max"], w["worst_max_name"] = mx, name
    w["n"] += 1
    w.setdefault("all", {})[name] = [l2, mx]
    _dump()


def _dump():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "gpu_stats.json"), "w") as f:
            json.dump(STATS, f, indent=1, sort_keys=True)


def _build(arch, directional, dev, wino=None):
    from retinanet_mi355x import modules
    fn, sd, img, ann = gc.model_case(arch, directional)
    net = getattr(modules, arch)(num_classes=4, directional=directional)
    missing = net.load_state_dict(sd)
    assert not missing.missing_keys and not missing.unexpected_keys
    if wino is not None:
        net._engine.use_wino = wino
    return net.to(dev), img.to(dev), ann.to(dev), sd


def rel_close(got, want, tol):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    err = np.abs(got - want).max()
    assert err <= tol * (np.abs(want).max() + 1e-12), "max err %.3e vs max|ref| %.3e" % (err, np.abs(want).max())


MEASURE_ONLY = os.environ.get("RN_TEST_MEASURE") == "1"       # collect the statistics, assert nothing about gradients


def grad_close(got, want, name, test, l2_tol=GRAD_L2_FLIP, max_tol=GRAD_MAX):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    l2 = np.sqrt(((got - want) ** 2).sum()) / (np.sqrt((want ** 2).sum()) + 1e-30)
    mx = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
    _note(test, name, float(l2), float(mx))
    if not MEASURE_ONLY:
        assert l2 <= l2_tol and mx <= max_tol, "%s [%s]: L2-rel %.3e, max-rel %.3e" % (name, test, l2, mx)


FLIP_FREE = 0.9                                                # share of the tensors that must meet the tight bounds


def most_within(test, tol, frac=FLIP_FREE):
    vals = [v[0] for v in STATS[test]["all"].values()]
    ok = sum(v <= tol for v in vals)
    if not MEASURE_ONLY:
        assert ok >= frac * len(vals), "%s: only %d of %d gradients within %.1e (worst %.2e)" % (test, ok, len(vals), tol, max(vals))


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


ARCHS = ["resnet18", "resnet50", "resnet34", "resnet101"]


@pytest.mark.parametrize("arch", ARCHS)
def test_directional_eval_localize(dev, golden, arch):
    z = golden(gc.MODEL_CASES[arch][0])
    net, img, ann, _ = _build(arch, True, dev)
    net.eval()
    boxes, cls = net(img, LOCALIZE=True)
    rel_close(cls.cpu().numpy(), z["%s_dir_cls" % arch], 1e-4)
    rel_close(boxes.cpu().numpy(), z["%s_dir_boxes" % arch], 1e-4)


@pytest.fixture(params=["native", "split", "split3"])
def mfma(request, dev):
    """Both product modes of the fp32 convolution kernels (include/retinanet_mi355x.h: RN_FP32_NATIVE / RN_FP32_SPLIT)."""
    from retinanet_mi355x import conv
    before = conv.get_fp32_mfma()
    conv.set_fp32_mfma(request.param)
    yield request.param
    conv.set_fp32_mfma(before)


FLIPPED_L2, FLIPPED_MAX = 1e-2, 5e-2                           # reach of one flipped mask at the goldens' sizes (see the module docstring)
MAX_FLIPS = 4                                                  # more activations than this on the wrong side of zero is not rounding


def located_flips(net, img, sd, arch):
    """{activation name: number of elements} whose sign differs between THIS GPU run (current product mode and kernel family)
    and the oracle's CPU run of the same network: every ReLU output of the forward pass, compared element by element
    (Engine.relu_outputs vs oracle.model's taps)."""
    from oracle import model as omodel
    taps = {}
    with torch.no_grad():
        omodel.forward_heads(img.cpu(), sd, arch, taps=taps)
    eng = net._engine
    with torch.no_grad():
        S = eng.forward(net._tensor_dict(), img, save=True)[2]
    got = eng.relu_outputs(S)
    assert sorted(got) == sorted(taps), sorted(set(got) ^ set(taps))
    flips = {}
    for name, t in got.items():
        want = taps[name] > 0                                  # NCHW on the CPU
        have = (t > 0).permute(0, 3, 1, 2).cpu()
        assert have.shape == want.shape, (name, have.shape, want.shape)
        n = int((have != want).sum())
        if n:
            flips[name] = n
    return flips


def check_gradients(net, img, sd, arch, mode, test, errs, norm_err):
    """The two-level rule of the module docstring.  errs / norm_err: {parameter: (L2, max)} / {parameter: relative norm error}."""
    worst = max(errs, key=lambda n: errs[n][0])
    tight = sum(e[0] <= GRAD_L2[mode] for e in errs.values()) >= FLIP_FREE * len(errs) and \
        sum(e <= NORM_TOL[mode] for e in norm_err.values()) >= FLIP_FREE * len(norm_err)
    loose = all(e[0] <= GRAD_L2_FLIP and e[1] <= GRAD_MAX for e in errs.values()) and all(e <= GRAD_L2_FLIP for e in norm_err.values())
    if tight and loose:
        return
    # Outside the bounds: the ONLY accepted cause is an activation on the other side of zero than in the CPU run.  Find it
    # (layer, pyramid level), take the parameters its mask can reach (Engine.backward_cone) -- those may move by what one flipped
    # mask moves at these sizes -- and hold EVERY other tensor to the tight bound.
    flips = located_flips(net, img, sd, arch)
    STATS[test]["flips"] = flips
    assert flips, "%s: worst %s %s and every ReLU output has the sign pattern of the CPU run" % (test, worst, errs[worst])
    assert sum(flips.values()) <= MAX_FLIPS, "%s: %s" % (test, flips)
    cone = set()
    for act in flips:
        cone.update(net._engine.backward_cone(act))
    outside = {n: e for n, e in errs.items() if n not in cone}
    for n, e in outside.items():
        assert e[0] <= GRAD_L2[mode], "%s: %s %s is outside the cone of %s and outside the tight bound" % (test, n, e, flips)
    for n, e in norm_err.items():
        assert e <= (FLIPPED_L2 if n in cone else 10 * NORM_TOL[mode]), "%s: norm of %s off by %.2e (flips %s)" % (test, n, e, flips)
    for n, e in errs.items():
        assert e[0] <= FLIPPED_L2 and e[1] <= FLIPPED_MAX, "%s: %s %s inside the cone of %s" % (test, n, e, flips)


@pytest.mark.parametrize("mode", ["wino", "direct"])
@pytest.mark.parametrize("arch", ARCHS)
def test_directional_train_losses_and_gradients(dev, golden, arch, mode, mfma):
    """Losses within 1e-4; parameter gradients by the two-level rule of the module docstring."""
    z = golden(gc.MODEL_CASES[arch][0])
    net, img, ann, sd = _build(arch, True, dev, wino=(mode == "wino"))
    net.train()
    net.freeze_bn()
    cls_l, reg_l, vp_l = net([img, ann])
    got = [float(cls_l.detach()), float(reg_l.detach()), float(vp_l.detach())]
    assert np.allclose(got, z["%s_dir_losses" % arch], rtol=1e-4), (got, z["%s_dir_losses" % arch])
    (cls_l + reg_l + vp_l).sum().backward()
    test = "train_%s_%s_%s" % (arch, mode, mfma)
    errs, norm_err = {}, {}
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        g = p.grad.detach().cpu().numpy().astype(np.float64)
        ref_norm = z["%s_dir_gsum_%s" % (arch, name)][2]
        norm_err[name] = abs(np.sqrt((g ** 2).sum()) - ref_norm) / (ref_norm + 1e-9)
        full = "%s_dir_g_%s" % (arch, name)
        if full in z.files:
            want = z[full].astype(np.float64)
            l2 = np.sqrt(((g - want) ** 2).sum()) / (np.sqrt((want ** 2).sum()) + 1e-30)
            mx = np.abs(g - want).max() / (np.abs(want).max() + 1e-30)
            errs[name] = (float(l2), float(mx))
            _note(test, name, float(l2), float(mx))
    assert len(errs) > 30
    if MEASURE_ONLY:
        return
    check_gradients(net, img, sd, arch, mode, test, errs, norm_err)


def test_flat2d_train_and_eval(dev, golden):
    z = golden("model")
    net, img, ann, _ = _build("resnet18", False, dev)
    net.train()
    net.freeze_bn()
    out = net([img, ann])
    assert len(out) == 2
    assert np.allclose([float(x.detach()) for x in out], z["resnet18_2d_losses"], rtol=1e-4)
    (out[0] + out[1]).sum().backward()
    for name, p in net.named_parameters():
        ref_norm = z["resnet18_2d_gsum_" + name][2]
        gn = float(p.grad.double().norm())
        assert abs(gn - ref_norm) <= NORM_TOL["wino"] * ref_norm + 1e-9, (name, gn, ref_norm)
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
        first = float(loss.detach()) if first is None else first
    assert float(loss.detach()) < first


def test_training_trajectories_agree_between_product_modes(dev):
    """Twelve optimizer steps (clip 0.1 + Adam 1e-4, the reference trainer's recipe) from the same weights, twice with the fp32
    products on the fp32 MFMA and once as split operands on the bf16 MFMA -- in FIXED-ORDER mode (conv.set_deterministic: the
    weight gradients' reduction order no longer depends on arrival), so that two runs of one mode are bit-identical and whatever
    separates the modes is the products' rounding alone, amplified by twelve Adam steps (round 3 ran this with the atomics on and
    had to allow 2e-2 for their run-to-run drift, which hid a systematic difference of that size).  A reduced-precision product --
    bf16 -- differs at 1e-3 in the FIRST loss."""
    from retinanet_mi355x import conv, optim
    before = conv.get_fp32_mfma(), conv.get_option(conv.OPT_DETERMINISTIC)

    def run(mode):
        conv.set_fp32_mfma(mode)
        net, img, ann, _ = _build("resnet50", True, dev)
        net.train()
        net.freeze_bn()
        opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
        seq = []
        for _ in range(12):
            opt.zero_grad(set_to_none=True)
            loss = sum(l.mean() for l in net([img, ann]))
            loss.backward()
            opt.step()
            seq.append(float(loss.detach()))
        return np.array(seq)
    conv.set_deterministic(True)
    try:
        n1, n2, sp = run("native"), run("native"), run("split")
    finally:
        conv.set_fp32_mfma(before[0])
        conv.set_deterministic(before[1])
    assert n1[-1] < n1[0]
    rel = lambda a, b: np.abs(a - b) / np.abs(a)
    same, cross = rel(n1, n2), rel(n1, sp)
    print("trajectory, fixed order: same-mode drift", same.max(), "| split vs native", cross)
    STATS["trajectory_fixed_order"] = {"same_mode": same.tolist(), "split_vs_native": cross.tolist()}
    _dump()
    assert same.max() == 0.0, same                                          # fixed order: the same bits, step after step
    assert cross[:3].max() <= 1e-5, cross                                   # fp32 rounding while the dynamics have not amplified it
    assert cross.max() <= TRAJ_CROSS, cross


TRAJ_CROSS = 1e-2     # split vs native after 12 steps at fixed order: measured 2e-7 .. 1e-6 for steps 1-4, 3.4e-3 at step 12 (round 3's bound with the atomics on: 2e-2 + 5 x the same-mode drift)


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


def test_clip_adam_hyperparameters_live_on_the_device(dev):
    """ClipAdam keeps lr / max_norm / betas / eps / grad_scale in device memory (rn_opt_clip_adam_hp): a scheduler's new lr
    reaches a REPLAYED hipGraph (sync_hyperparameters outside the graph), a change inside a capture raises instead of being
    frozen in silently, grad_scale = 1/world equals pre-scaled gradients, and state_dict / load_state_dict carry the step
    number and both moments (a resumed run keeps Adam's bias correction)."""
    from retinanet_mi355x import optim
    torch.manual_seed(1)
    shapes = [(300, 7), (64,), (5000,)]
    grads = [torch.randn(s, device=dev) * 0.01 for s in shapes]

    def params():
        torch.manual_seed(2)
        return [torch.nn.Parameter(torch.randn(s, device=dev) * 0.1) for s in shapes]
    # (a) lr change between two replays of a captured step
    pa, pb = params(), params()
    ref = torch.optim.Adam(pb, lr=1e-3)
    mine = optim.ClipAdam(pa, lr=1e-3, max_norm=0.0)
    for p, q, g in zip(pa, pb, grads):
        p.grad, q.grad = g.clone(), g.clone()
    mine.step(); ref.step()                                    # eager warm-up: pointer table
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        mine.step()
    graph.replay(); ref.step()
    graph.replay(); ref.step()                                 # (capture itself executes nothing)
    mine.param_groups[0]["lr"] = ref.param_groups[0]["lr"] = 1e-4
    assert mine.sync_hyperparameters()
    graph.replay(); ref.step()
    for p, q in zip(pa, pb):
        assert float((p - q).abs().max()) <= 2e-6 * float(q.abs().max()), "the new lr did not reach the replayed graph"
    mine.param_groups[0]["lr"] = 5e-5
    g2, side, msg = torch.cuda.CUDAGraph(), torch.cuda.Stream(), ""
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g2.capture_begin()
        try:
            mine.step()                                        # raises before anything is launched into the capture
        except RuntimeError as e:
            msg = str(e)
        finally:
            g2.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    assert "inside a graph capture" in msg
    # (b) grad_scale: a gradient SUM over 4 ranks with grad_scale 1/4 == averaged gradients
    pa, pb = params(), params()
    one = optim.ClipAdam(pa, lr=1e-3, max_norm=0.1)
    two = optim.ClipAdam(pb, lr=1e-3, max_norm=0.1, grad_scale=0.25)
    for p, q, g in zip(pa, pb, grads):
        p.grad, q.grad = g.clone(), 4.0 * g
    n1, n2 = float(one.step()), float(two.step())
    assert abs(n1 - n2) <= 1e-6 * n1
    for p, q in zip(pa, pb):
        assert float((p - q).abs().max()) <= 1e-6 * float(q.abs().max())
    # (c) state_dict round trip: resumed optimizer continues bit for bit
    sd = one.state_dict()
    assert sd["step"] == 1 and len(sd["m"]) == len(shapes)
    pc = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    three = optim.ClipAdam(pc, lr=7e-3, max_norm=0.5)
    three.load_state_dict(sd)
    for p, q, g in zip(pa, pc, grads):
        p.grad, q.grad = g.clone(), g.clone()
    one.step(); three.step()
    assert int(three.step_dev) == 2
    for p, q in zip(pa, pc):
        assert torch.equal(p.detach(), q.detach())


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
    assert np.allclose([float(x.detach()) for x in got], [float(x) for x in want], rtol=1e-4), (got, want)


# ------------------------------------------------------------------------------------------------ per-call state
def _grads(net, img, ann):
    for p in net.parameters():
        p.grad = None
    losses = net([img, ann])
    sum(l.mean() for l in losses).backward()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters()}


def test_second_forward_before_backward_does_not_leak_state(dev):
    """forward(b1); forward(b2); backward(b1) must give the gradients of a lone step on b1: the Winograd input transforms
    kept for the weight gradients belong to the call, not to the (shared) layer objects."""
    from retinanet_mi355x import synth
    net, img, ann, _ = _build("resnet50", True, dev, wino=True)
    net.train()
    net.freeze_bn()
    H, W = gc.MODEL_HW
    img2 = synth.frames(1, H, W, seed=77).to(dev)
    ann2 = synth.labels_dir(1, 5, H, W, num_classes=4, seed=78, size_px=(24, 60)).to(dev)
    want = _grads(net, img, ann)
    for p in net.parameters():
        p.grad = None
    l1 = net([img, ann])
    l2 = net([img2, ann2])                                   # same shapes, other data, before l1's backward
    with torch.no_grad():
        net([img2, ann2])                                    # and a train-mode pass that will never run backward
    sum(l.mean() for l in l1).backward()
    worst = 0.0
    for n, p in net.named_parameters():
        err = float((p.grad - want[n]).norm() / (want[n].norm() + 1e-30))
        worst = max(worst, err)
    assert worst <= 1e-4, worst                              # fp32 atomics in wgrad: order-dependent last bits only
    del l2


def test_flat_gradient_buffer(dev):
    """Engine.set_flat_grads: same gradients, p.grad pointers stable from step to step, and gradient accumulation over two
    backward calls (p.grad still living in the buffer) is not corrupted by the second backward."""
    net, img, ann, _ = _build("resnet18", True, dev)
    net.train()
    net.freeze_bn()
    want = _grads(net, img, ann)
    net.use_flat_gradients()
    got = _grads(net, img, ann)
    ptr1 = {n: p.grad.data_ptr() for n, p in net.named_parameters()}
    for n in want:
        assert float((got[n] - want[n]).norm()) <= 1e-4 * float(want[n].norm()) + 1e-12, n
    got2 = _grads(net, img, ann)                              # second step: zero_grad(set_to_none) happened inside
    ptr2 = {n: p.grad.data_ptr() for n, p in net.named_parameters()}
    assert ptr1 == ptr2, "gradient pointers moved between steps"
    arena = net._engine._flat["arena"]
    lo, hi = arena.data_ptr(), arena.data_ptr() + 4 * arena.numel()
    assert all(lo <= q < hi for q in ptr2.values())
    # accumulation: backward again WITHOUT clearing p.grad -> p.grad must become 2x
    losses = net([img, ann])
    sum(l.mean() for l in losses).backward()
    for n, p in net.named_parameters():
        assert float((p.grad - 2 * got2[n]).norm()) <= 2e-4 * float(got2[n].norm()) + 1e-12, n
    net.use_flat_gradients(False)


def test_data_parallel_is_refused_clearly(dev):
    net, img, ann, _ = _build("resnet18", True, dev)
    net.eval()
    dp = torch.nn.DataParallel(net, device_ids=[0, 0])
    with pytest.raises(RuntimeError, match="one process per GPU"):
        dp(torch.cat([img, img]))


def test_label_check_is_eager_and_the_reference_loop_skips_the_iteration(dev):
    """A training batch without any label: the reference raises INSIDE FocalLoss (D/losses.py:362), so its trainer's
    try / except (train_detector_3D_angle.py:367-408) never reaches backward / clip / step for that iteration.  Same here by
    default: one reference-style iteration on an all-empty batch leaves every parameter untouched and the loss history clean."""
    from retinanet_mi355x import ops
    net, img, ann, _ = _build("resnet18", True, dev)
    net.train()
    net.freeze_bn()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    empty = torch.full_like(ann, -1.0)
    loss_hist = []
    try:                                                      # the reference's iteration, line for line in spirit (:367-408)
        opt.zero_grad()
        cls_l, reg_l, vp_l = net([img, empty])
        loss = cls_l.mean() + reg_l.mean() + vp_l.mean()
        if not bool(loss == 0):
            loss.backward()
            torch.nn.utils.clip_grad_norm_(net.parameters(), 0.1)
            opt.step()
            loss_hist.append(float(loss))
    except Exception as e:
        assert "non-empty TensorList" in str(e)
    assert not loss_hist
    for n, p in net.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
    net([img, ann])                                           # the next (good) batch is not discarded
    ops.flush_label_checks()


def test_deferred_label_check_is_opt_in(dev):
    """RN_DEFERRED_LABEL_CHECK=1 (bench.py, captured-graph loops): no host sync per step; the error arrives one call late."""
    from retinanet_mi355x import ops
    net, img, ann, _ = _build("resnet18", True, dev)
    net.train()
    empty = torch.full_like(ann, -1.0)
    os.environ["RN_DEFERRED_LABEL_CHECK"] = "1"
    try:
        out = net([img, empty])
        assert torch.isnan(out[2]).all()                      # vp loss 0/0 over zero labelled images
        with pytest.raises(RuntimeError, match="non-empty TensorList"):
            ops.flush_label_checks()
    finally:
        del os.environ["RN_DEFERRED_LABEL_CHECK"]
    with pytest.raises(RuntimeError, match="non-empty TensorList"):
        net([img, empty])                                     # default: raised by this very forward
    net([img, ann])
    ops.flush_label_checks()


# ------------------------------------------------------------------------------------------------ cfg2 at full size
CFG2 = {}


def _cfg2_oracle():
    """ResNet-50, 1080x1920, batch 1, the benchmark's synthetic inputs: oracle forward + three losses + backward on the
    CPU (about 16 s on the GPU box's cores), once per session."""
    if not CFG2:
        from oracle import model as omodel
        from retinanet_mi355x import synth
        H, W = 1080, 1920
        sd = synth.state_dict("resnet50", 8, 12, seed=2)
        img = synth.frames(1, H, W, seed=0)
        ann = synth.labels_dir(1, 10, H, W, 8, seed=1)
        params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
                  for k, v in sd.items()}
        taps = {}
        losses = omodel.train_forward(img, ann, params, "resnet50", taps=taps)
        signs = {k: (v > 0) for k, v in taps.items()}              # 1 byte per activation: the ReLU outputs' sign pattern
        del taps
        sum(l.mean() for l in losses).backward()
        with torch.no_grad():
            boxes, cls = omodel.eval_forward(img, sd, "resnet50", LOCALIZE=True)
        CFG2.update(sd=sd, img=img, ann=ann, losses=[float(l.detach()) for l in losses], boxes=boxes, cls=cls, signs=signs,
                    grads={k: p.grad for k, p in params.items() if getattr(p, "grad", None) is not None})
    return CFG2


CFG2_PARAMS = (["conv1.weight", "bn1.weight", "bn1.bias"]
               + ["layer%d.%d.conv2.weight" % (l, b) for l, b in ((1, 0), (2, 0), (2, 3), (3, 0), (3, 5), (4, 0), (4, 2))]
               + ["layer2.0.downsample.0.weight", "layer3.0.conv1.weight", "layer4.2.conv3.weight", "layer4.2.bn3.weight",
                  "layer3.2.bn2.bias", "layer1.2.bn3.weight"]
               + ["fpn.%s.%s" % (n, t) for n in ("P3_1", "P3_2", "P4_1", "P4_2", "P5_1", "P5_2", "P6", "P7_2") for t in ("weight", "bias")]
               + ["%s.%s.%s" % (m, c, t) for m in ("regressionModel", "classificationModel")
                  for c in ("conv1", "conv2", "conv3", "conv4", "output") for t in ("weight", "bias")])


@pytest.mark.parametrize("mode", ["wino", "direct"])
def test_cfg2_full_size_against_oracle(dev, mode, mfma):
    """BASELINE configs[1] at its real size (ResNet-50, 1920x1080, the benchmark's inputs, batch 1): HIP training forward +
    three losses + backward against the oracle on CPU -- the five real pyramid sizes 135x240 ... 9x15, both FPN crop
    branches, the grouped Winograd path with its padded tile count (mode "wino", what bench.py runs) and the direct
    kernels (mode "direct" = RN_WINOGRAD=0).  D/model.py:284-309, D/losses.py:27-362."""
    from retinanet_mi355x import modules
    o = _cfg2_oracle()
    net = modules.resnet50(num_classes=8)
    net.load_state_dict(o["sd"])
    net = net.to(dev)
    net._engine.use_wino = mode == "wino"
    net.train()
    net.freeze_bn()
    img, ann = o["img"].to(dev), o["ann"].to(dev)
    losses = net([img, ann])
    got = [float(l.detach()) for l in losses]
    assert np.allclose(got, o["losses"], rtol=1e-4), (got, o["losses"])
    sum(l.mean() for l in losses).backward()
    assert len(CFG2_PARAMS) >= 30
    test = "cfg2_full_%s_%s" % (mode, mfma)
    named = dict(net.named_parameters())
    errs = {}
    for name in CFG2_PARAMS:                                   # measured first, judged below (once the flips are located)
        grad_close(named[name].grad.cpu().numpy(), o["grads"][name].numpy(), name, test, l2_tol=np.inf, max_tol=np.inf)
        errs[name] = STATS[test]["all"][name]
    for name, p in named.items():                              # and every other gradient by its norm
        want = float(o["grads"][name].double().norm())
        assert abs(float(p.grad.double().norm()) - want) <= 2e-3 * want + 1e-12, name
    for p in net.parameters():
        p.grad = None
    # The backbone's 1e-4 .. 8e-4 is attributed to ReLU outputs within rounding of zero that land on the other side than in the CPU
    # run.  LOCATE them at this size -- every ReLU output of this run against the oracle's, element by element -- and apply the rule
    # of the small tests (round 4): only the parameters inside the backward cone of a located flip (Engine.backward_cone) may use the
    # bound a flip can reach (CFG2_L2["backbone"], 2.6x the measured worst); EVERY tensor outside the cones must meet the tight one.
    with torch.no_grad():
        S = net._engine.forward(net._tensor_dict(), img, save=True)[2]
    total, flips, by_group, cone = 0, 0, {}, set()
    for aname, t in net._engine.relu_outputs(S).items():
        n = int(((t > 0).permute(0, 3, 1, 2).cpu() != o["signs"][aname]).sum())
        total += t.numel()
        flips += n
        if n:
            cone.update(net._engine.backward_cone(aname))
        grp = aname.split(".")[0] if aname.startswith("layer") else ("heads" if "Model" in aname else aname.split("@")[0])
        by_group[grp] = by_group.get(grp, 0) + n
    del S
    outside = [n for n in errs if n not in cone]
    STATS[test].update(relu_outputs=total, sign_flips=flips, sign_flips_by_group=by_group, params_outside_every_cone=outside)
    _dump()
    assert 0 < total and flips <= 2e-5 * total, (flips, total, by_group)      # measured: a few hundred among 1.4e8
    if not MEASURE_ONLY:
        for name, (l2, mx) in errs.items():
            tol = CFG2_L2["backbone"] if name in cone else CFG2_L2["head"]
            assert l2 <= tol and mx <= 2 * GRAD_MAX, "%s [%s]: L2-rel %.3e, max-rel %.3e (%s the cone of a located flip: %s)" % (
                name, test, l2, mx, "inside" if name in cone else "OUTSIDE", by_group)
        # the heads and the pyramid see few flips and short cones: whatever the cone rule allows them, they stay at their measured level
        for name, (l2, mx) in errs.items():
            if name.startswith(("fpn.", "regressionModel.", "classificationModel.")):
                assert l2 <= CFG2_L2["head"], (name, l2)
    net.eval()
    boxes, cls = net(img, LOCALIZE=True)
    rel_close(cls.cpu().numpy(), o["cls"].numpy(), 1e-4)
    rel_close(boxes.cpu().numpy(), o["boxes"].numpy(), 1e-4)


# ------------------------------------------------------------------------------------------------ hipGraph capture
def test_whole_step_replays_from_a_captured_graph(dev):
    """The training step (forward, loss, backward, fused clip + Adam) is a fixed launch sequence with no host decision in
    it (engine.py): captured once into a hipGraph and replayed, it trains like the eager loop (loss trajectory equal up to
    the order of the weight-gradient atomics; the optimizer's step number lives on the device, rn_opt_clip_adam_dev)."""
    from retinanet_mi355x import optim

    def make():
        net, img, ann, _ = _build("resnet50", True, dev)
        net.train()
        net.freeze_bn()
        net.use_flat_gradients()
        opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = sum(l.mean() for l in net([img, ann]))
            loss.backward()
            opt.step()
            return loss
        return step, opt
    step, _ = make()
    eager = [float(step().detach()) for _ in range(6)]
    step, opt = make()
    got = [float(step().detach()) for _ in range(2)]            # eager warm-up: job tables, gradient buffer, pointer table
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()
    got.append(float(loss.detach()))                             # (capture does not execute: the value comes from the first replay)
    got = got[:2]
    for _ in range(4):
        g.replay()
        got.append(float(loss.detach()))
    assert int(opt.step_dev) == 6
    assert len(set(eager)) == 6                                  # the parameters do move from step to step
    assert np.allclose(got, eager, rtol=2e-3), (got, eager)
