"""BASELINE configs[2] and configs[4] at THEIR size on one GPU (round 4; the small-size twins are test_gpu_model_bf16.py and
test_gpu_model_fp8.py):

* cfg3's per-GPU leg -- ResNet-50, 1920x1080, bf16 activations / MFMA: batch 1 against the oracle's CPU run of the benchmark's
  inputs (the run tests/test_gpu_model.py::test_cfg2_full_size_against_oracle shares: losses within 1e-2, every head / pyramid
  gradient tensor within cosine 0.97 of the fp32 one, the sampled backbone tensors too), and batch 8 -- the per-GPU batch of
  configs[2] -- equal to the mean of its eight single-image runs (the pattern of test_gpu_batch8.py: the forward of image i
  inside the batch is bit-identical to image i alone; the batch's losses and parameter gradients are the means).
* cfg5's per-GPU leg -- ResNet-101, 1920x1080, fp8 (e4m3) forward, batch 2, against the oracle's fp32 forward on the CPU with
  activation scales calibrated on frames the evaluation does not see.  Bounds = the design target of the round-5 error budget
  (tools/fp8_error_budget.py, profiles/r05_fp8_error_budget.txt): scores within 8.5 % of the largest score and 3 % rms, boxes within
  5 % -- met by keeping the residual stream and the FPN in bf16 (every layer in e4m3, round 4's configuration: 14.7 % / 5.7 %; the
  budget run compares with the fp32 ENGINE and reads 7.97 %; the half per cent is the oracle's own CPU summation order on top).
  Round 5 also: the same network at ITS batch (16, every image bit-identical to its single-image run) and as a TRAINING step (fp8
  forward, bf16 gradients) against the bf16 engine's step.
The reference has neither mode: parity of the low-precision arithmetic is unpinned by it; what is pinned is that these schedules
compute the reference's FUNCTION at the benchmark's sizes (five real pyramid sizes, both FPN crop branches, grouped head launches).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H, W = 1080, 1920
LOSS_TOL, COS_MIN = 1e-2, 0.97


def _bf16_net(dev):
    from retinanet_mi355x import modules, synth
    net = modules.resnet50(num_classes=8)
    net.load_state_dict(synth.state_dict("resnet50", 8, 12, seed=2))
    net = net.to(dev)
    net.set_compute_dtype("bf16")
    net.train()
    net.freeze_bn()
    return net


def _step(net, img, ann):
    for p in net.parameters():
        p.grad = None
    losses = net([img, ann])
    sum(l.mean() for l in losses).backward()
    return [float(l.detach()) for l in losses], {n: p.grad.detach().clone() for n, p in net.named_parameters()}


def test_bf16_cfg2_size_against_oracle(dev):
    import test_gpu_model as tgm
    o = tgm._cfg2_oracle()
    net = _bf16_net(dev)
    losses, grads = _step(net, o["img"].to(dev), o["ann"].to(dev))
    want = np.array(o["losses"])
    rel = np.abs(np.array(losses) - want) / np.abs(want)
    cos = {}
    for name in tgm.CFG2_PARAMS:
        g, w = grads[name].double().cpu().reshape(-1), o["grads"][name].double().reshape(-1)
        assert torch.isfinite(g).all(), name
        cos[name] = float(g @ w / (g.norm() * w.norm() + 1e-300))
    head = {n: c for n, c in cos.items() if n.startswith(("fpn.", "regressionModel.", "classificationModel."))}
    ratio = np.array([float(grads[n].double().norm() / (o["grads"][n].double().norm() + 1e-300)) for n in grads])
    print("bf16 1080p batch 1: losses %s vs %s (rel %s); cosine heads+FPN min %.4f (%s), backbone sample min %.4f; norm ratio median %.4f"
          % (losses, o["losses"], rel, min(head.values()), min(head, key=head.get),
             min(c for n, c in cos.items() if n not in head), float(np.median(ratio))))
    tgm.STATS["bf16_cfg2_full"] = {"loss_rel": rel.tolist(), "cosine": cos, "norm_ratio_median": float(np.median(ratio))}
    tgm._dump()
    assert np.all(rel <= LOSS_TOL), (losses, o["losses"])
    assert all(c >= COS_MIN for c in head.values()), {n: c for n, c in head.items() if c < COS_MIN}
    assert np.mean([c >= COS_MIN for c in cos.values()]) >= 0.9 and min(cos.values()) > 0.8, cos
    assert abs(float(np.median(ratio)) - 1.0) <= 0.05
    # the eval branch on the same frame: scores and decoded boxes against the oracle's fp32 ones (3e-2, as at the small size)
    net.eval()
    boxes, cls = net(o["img"].to(dev), LOCALIZE=True)
    assert float((cls.cpu() - o["cls"]).abs().max()) <= 3e-2 * float(o["cls"].abs().max())
    assert float((boxes.cpu() - o["boxes"]).abs().max()) <= 3e-2 * float(o["boxes"].abs().max())


def test_bf16_batch8_equals_the_eight_single_image_runs(dev):
    """configs[2]'s per-GPU batch (64 images over 8 GPUs) in bf16.  Every output element of the bf16 convolutions is one K loop in
    a fixed order whatever the number of images in the launch, so image i inside the batch-8 forward is bit-identical to image i
    alone; the loss is the mean over images, so the gradient flowing into image i is scaled by 1/8 -- a power of two, which bf16
    rounding commutes with -- and every parameter gradient is the mean of the single-image gradients up to the order of the fp32
    sums (atomics)."""
    from retinanet_mi355x import synth
    B = 8
    net = _bf16_net(dev)
    eng = net._engine
    img = synth.frames(B, H, W, seed=0).to(dev)
    ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)
    with torch.no_grad():
        reg8, cls8, _ = eng.forward(net._tensor_dict(), img, save=True)
        for i in (0, 5):
            reg1, cls1, _ = eng.forward(net._tensor_dict(), img[i:i + 1], save=True)
            assert torch.equal(reg8[i:i + 1], reg1) and torch.equal(cls8[i:i + 1], cls1), "image %d differs inside the batch" % i
        del reg8, cls8, reg1, cls1
    loss8, grad8 = _step(net, img, ann)
    mean_loss = np.zeros(3)
    mean_grad = {n: torch.zeros_like(g, dtype=torch.float64) for n, g in grad8.items()}
    for i in range(B):
        l1, g1 = _step(net, img[i:i + 1], ann[i:i + 1])
        mean_loss += np.array(l1) / B
        for n, g in g1.items():
            mean_grad[n] += g.double() / B
    assert np.allclose(loss8, mean_loss, rtol=1e-6, atol=0), (loss8, mean_loss.tolist())
    worst = ("", 0.0)
    for n, g in grad8.items():
        err = float((g.double() - mean_grad[n]).norm() / (mean_grad[n].norm() + 1e-30))
        worst = max(worst, (n, err), key=lambda t: t[1])
    print("bf16 batch 8 vs 8 x batch 1: losses %s, worst gradient %s %.2e" % (loss8, worst[0], worst[1]))
    assert worst[1] <= 1e-5, worst


def test_fp8_resnet101_cfg5_size_against_oracle(dev):
    """configs[4]: ResNet-101, 1920x1080, e4m3 activations and weights -- forward (the fp8 engine is inference-only), batch 2,
    LOCALIZE outputs (every anchor's scores and decoded box) against the oracle's fp32 CPU forward."""
    from oracle import model as omodel
    from retinanet_mi355x import modules, synth
    B = 2
    sd = synth.state_dict("resnet101", 8, 12, seed=2)
    img = synth.frames(B, H, W, seed=0)
    with torch.no_grad():
        o_boxes, o_cls = omodel.eval_forward(img, sd, "resnet101", LOCALIZE=True)
    net = modules.resnet101(num_classes=8)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    calib = torch.cat([synth.frames(1, H, W, seed=123), synth.frames(1, H, W, seed=124)]).to(dev)    # not the evaluated frames
    res = {}
    for margin in (1.0, 1.25):                      # 1.0: the calibration frames' own maxima; 1.25: head-room for what they did not reach
        scales = net.calibrate_fp8(calib, margin=margin)
        assert net._engine.fp8 and set(net._engine.fp8_scale_names()) <= set(scales)
        boxes, cls = net(img.to(dev), LOCALIZE=True)
        s_err = (cls.cpu() - o_cls).abs()
        res[margin] = (float(s_err.max() / o_cls.abs().max()), float((s_err.double() ** 2).mean().sqrt() / (o_cls.double() ** 2).mean().sqrt()))
    print("fp8 resnet101 1080p: scores (max, rms) by calibration margin", res)
    assert cls.shape == o_cls.shape == (B, 389205, 8) and boxes.shape == o_boxes.shape
    assert torch.isfinite(cls).all() and torch.isfinite(boxes).all()
    s_err = (cls.cpu() - o_cls).abs()
    b_err = (boxes.cpu() - o_boxes).abs()
    s_max, s_rms = float(s_err.max() / o_cls.abs().max()), float((s_err.double() ** 2).mean().sqrt() / (o_cls.double() ** 2).mean().sqrt())
    b_max, b_rms = float(b_err.max() / o_boxes.abs().max()), float((b_err.double() ** 2).mean().sqrt() / (o_boxes.double() ** 2).mean().sqrt())
    print("fp8 resnet101 1080p batch 2: scores max %.3e rms %.3e | boxes max %.3e rms %.3e" % (s_max, s_rms, b_max, b_rms))
    import test_gpu_model as tgm
    tgm.STATS["fp8_cfg5_full"] = {"scores_max": s_max, "scores_rms": s_rms, "boxes_max": b_max, "boxes_rms": b_rms,
                                  "scores_max_rms_by_margin": {str(k): v for k, v in res.items()}}
    tgm._dump()
    assert s_max <= 0.085 and s_rms <= 0.03 and b_max <= 0.05, (s_max, s_rms, b_max)
    # the post-processing branches run on these tensors at this size
    s, c, b, im = net(img.to(dev), MULTI_FRAME=True)
    assert s.shape[0] == c.shape[0] == b.shape[0] == im.shape[0] and (im.numel() == 0 or int(im.max()) <= B - 1)


def test_fp8_resnet101_batch16(dev):
    """configs[4] at ITS batch: ResNet-101, 1920x1080, 16 images per GPU through the e4m3 forward (what `bench.py --arch resnet101 --dtype
    fp8 --batch 16` times).  The scales are calibrated constants and every output element is one K loop in a fixed order whatever the
    number of images in the launch (persistent workgroups walk tiles, they do not split reductions), so image i inside the batch-16
    launch must be BIT-IDENTICAL to image i alone -- batch strides, the grouped pyramid launches' tile tables and the persistent tile
    walk at B = 16 are witnessed element by element; image 0 alone is what test_fp8_resnet101_cfg5_size_against_oracle checks against
    the oracle."""
    from retinanet_mi355x import modules, synth
    B = 16
    net = modules.resnet101(num_classes=8)
    net.load_state_dict(synth.state_dict("resnet101", 8, 12, seed=2))
    net = net.to(dev).eval()
    calib = torch.cat([synth.frames(1, H, W, seed=123), synth.frames(1, H, W, seed=124)]).to(dev)
    net.calibrate_fp8(calib, margin=1.25)
    assert net._engine.fp8
    img = synth.frames(B, H, W, seed=0).to(dev)
    with torch.no_grad():
        boxes, cls = net(img, LOCALIZE=True)
        assert cls.shape == (B, 389205, 8) and boxes.shape[0] == B and torch.isfinite(cls).all() and torch.isfinite(boxes).all()
        for i in (0, 7, 15):
            b1, c1 = net(img[i:i + 1], LOCALIZE=True)
            assert torch.equal(cls[i:i + 1], c1), "scores of image %d differ inside the batch" % i
            assert torch.equal(boxes[i:i + 1], b1), "boxes of image %d differ inside the batch" % i
        # different images give different outputs (the comparison above is not vacuous)
        assert not torch.equal(cls[0], cls[7])


def test_fp8_training_step_cfg5_size_against_the_bf16_step(dev):
    """configs[4] as a TRAINING configuration at its size: ResNet-101, 1920x1080, batch 2 -- forward on the fp8 kernels (e4m3 activations,
    bf16 residual stream), activations saved in those formats, the three losses in fp32, data and weight gradients on the bf16 kernels --
    against the SAME step on the bf16 engine (whose 1080p step is pinned to the oracle by test_bf16_cfg2_size_against_oracle): every
    loss within 2 %, every head / pyramid gradient tensor within cosine 0.95 (VERDICT r4 item 3c's criterion), all gradients finite."""
    from retinanet_mi355x import modules, synth
    B = 2
    sd = synth.state_dict("resnet101", 8, 12, seed=2)
    img = synth.frames(B, H, W, seed=0).to(dev)
    ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)

    def build():
        net = modules.resnet101(num_classes=8)
        net.load_state_dict(sd)
        return net.to(dev)
    ref = build()
    ref.set_compute_dtype("bf16")
    ref.train()
    ref.freeze_bn()
    l16, g16 = _step(ref, img, ann)
    del ref
    torch.cuda.empty_cache()
    net = build().eval()
    net.calibrate_fp8(torch.cat([synth.frames(1, H, W, seed=123), synth.frames(1, H, W, seed=124)]).to(dev), margin=1.25)
    net.train()
    net.freeze_bn()
    l8, g8 = _step(net, img, ann)
    rel = np.abs(np.array(l8) - np.array(l16)) / np.abs(np.array(l16))
    cos = {}
    for n, g in g8.items():
        assert torch.isfinite(g).all(), n
        a, b = g.double().reshape(-1), g16[n].double().reshape(-1)
        if float(b.norm()) > 0:
            cos[n] = float(a @ b / (a.norm() * b.norm() + 1e-300))
    head = {n: c for n, c in cos.items() if n.startswith(("fpn.", "regressionModel.", "classificationModel."))}
    back = np.array([c for n, c in cos.items() if n not in head])
    print("fp8-forward training, resnet101 1080p batch 2: losses %s vs bf16 %s (rel %s); cosine heads + FPN min %.4f (%s); backbone min %.4f median %.4f"
          % (l8, l16, rel, min(head.values()), min(head, key=head.get), back.min(), np.median(back)))
    assert np.all(rel <= 2e-2), (l8, l16)
    assert min(head.values()) >= 0.95, (min(head, key=head.get), min(head.values()))
    assert np.median(back) >= 0.9
