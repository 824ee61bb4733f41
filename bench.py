#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the ResNet-50 directional 3D-RetinaNet at 1920x1080
(BASELINE.json metric; workload = configs[1]: batch 8 per GPU, fp32, synthetic frames + 10 random GT boxes).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = what the reference's training loop does per iteration (train_detector_3D_angle.py:368-387):
zero_grad, forward + loss, sum of the three losses, backward, clip_grad_norm_(0.1), Adam(lr 1e-4) step --
all on inputs already resident in HBM.  With N > 1 every rank trains on its own 8 images (weak scaling) and
gradients are averaged over RCCL inside backward (retinanet_mi355x.ddp).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      the dominant kernel family (implicit-GEMM conv, 128x128 tile): algorithmic (fp32) FLOPs and algorithmic bytes (every
                operand once) of its launches inside the timed steps / their HIP-event durations.  Which roof prices it is the roofline
                model's own rule: arithmetic intensity against the ridge peak / 8 TB/s, where peak is that of the instruction the kernels
                run on -- 2500 / 3 = 833 TF in the default mode (split3: three fp16 MFMAs per fp32 product), 2500 / 6 with three bf16
                terms, 157.3 TF on the fp32 MFMA.  Below the ridge: bound "hbm", achieved / peak in GB/s, the matrix-core fraction beside
                it (mfma_frac); above: bound "mfma".  traffic: HBM bytes per launch from the committed PMC passes of this command.
  step_hbm      the whole step's HBM bytes (same PMC passes) over its wall time, against 8 TB/s
  fp32_native_mfma / fp32_split_bf16x3   the same step in the other two product modes, measured in the same run
  kernels       the same for the other timed kernels, plus the fused IoU+focal loss against the HBM roof
  bf16 / fp8    (--sections all) BASELINE configs[2] / configs[4]: the bf16 engine's step; the fp8 engine's forward.  --dtype bf16 / fp8 make
                them the headline: --dtype fp8 is a TRAINING step (e4m3 forward, bf16 gradients, fp32 master weights)
  cpu_baseline  the CPU restatement (oracle/, torch CPU kernels = what the reference runs on a GPU-less host)
                timed on this box's host cores on a bounded sample (1 image, forward+loss+backward), rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TF = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TF = 2500.0        # dense bf16 MFMA (same guide): the kernels of --dtype bf16, and --
PEAK_SPLIT_TF = PEAK_BF16_MFMA_TF / 6   # -- the fp32 kernels in split-operand mode: six bf16 MFMAs per fp32 product (csrc/mfma_split.h)
PEAK_SPLIT3_TF = PEAK_BF16_MFMA_TF / 3  # -- and in split3 mode: three fp16 MFMAs (same rate as bf16) per fp32 product
PEAK_HBM_GBS = 8000.0             # HBM3E spec; 6.29 TB/s measured copy


_FRAMES = {}


def frames(B, H, W, seed, dev):
    """synth.frames (the tensor the parity tests and cpu_baseline use), generated once per run."""
    from retinanet_mi355x import synth
    key = (B, H, W, seed)
    if key not in _FRAMES:
        _FRAMES[key] = synth.frames(B, H, W, seed=seed)
    return _FRAMES[key].to(dev)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE cfg2: 8)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp8"],
                    help="fp32 = the reference's arithmetic = the headline (BASELINE configs[1]); bf16 = configs[2]'s storage "
                         "format (bf16 activations / MFMA, fp32 accumulation and master weights), reported with dtype bf16; "
                         "fp8 = configs[4]'s arithmetic (e4m3 activations / weights), INFERENCE forward only: a different metric "
                         "(forward images/sec), e.g. --arch resnet101 --dtype fp8 --batch 16")
    ap.add_argument("--sections", default="all", choices=["all", "headline"],
                    help="headline: ONLY the benchmark's own steps (warm-up, timed steps, the same steps with per-kernel events) -- no "
                         "native-MFMA / bf16 / fp8 sections, no forward-only passes, no loss micro-benchmark -- so that a rocprofv3 "
                         "--kernel-trace --stats of this command divides by (warmup + 2 * steps) into the line's kernels.*.ms_per_step")
    ap.add_argument("--fp32-mfma", default="", choices=["", "split", "split3", "native"],
                    help="how the fp32 convolution kernels form their products (include/retinanet_mi355x.h: RN_FP32_SPLIT / "
                         "RN_FP32_NATIVE); default: the library's (RN_FP32_DEFAULT, or the environment's RN_FP32_MFMA)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (forward, loss, backward, clip + Adam) into one hipGraph after the warm-up and "
                         "replay it in the timed region (single GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--ddp-timeline", action="store_true",
                    help="N > 1: rank 0 also prints the per-bucket all-reduce timeline of the last step (stderr)")
    return ap.parse_args()


def loss_kernel_roofline(dev, B, H, W, C=8, N=10, iters=20):
    """Fused IoU + assignment + focal / smooth-L1 / VP loss at the benchmark shape, against the HBM roof: forward (one
    launch incl. the batch reduction) and backward.  Algorithmic bytes: forward reads cls B*A*C*4 +
    anchors A*16 + labels B*N*27*4; backward re-reads those and writes dcls B*A*C*4 + dreg B*A*12*4."""
    from retinanet_mi355x import _hip, ops, synth
    lib = _hip.load()
    A = ops.anchor_count(H, W)
    cls, reg = synth.head_outputs(1, A, C, 12, seed=3)
    cls = cls.to(dev).expand(B, A, C).contiguous()
    reg = reg.to(dev).expand(B, A, 12).contiguous()
    ann = synth.labels_dir(B, N, H, W, C, seed=1).to(dev)
    anc = ops.anchors(H, W, dev)
    ws = torch.zeros(lib.rn_focal_workspace_bytes(B, A), dtype=torch.uint8, device=dev)   # counter zero on entry
    out = torch.empty(3, device=dev)
    g = torch.ones(3, device=dev)
    dcls, dreg = torch.empty_like(cls), torch.empty_like(reg)

    def fwd():
        _hip.check(lib.rn_focal_loss_fwd(cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), ann.data_ptr(), B, A, C, N, 1,
                                         ws.data_ptr(), out.data_ptr(), _hip.stream()), "rn_focal_loss_fwd")

    def bwd():
        _hip.check(lib.rn_focal_loss_bwd(cls.data_ptr(), reg.data_ptr(), anc.data_ptr(), ann.data_ptr(), B, A, C, N, 1,
                                         ws.data_ptr(), g.data_ptr(), dcls.data_ptr(), dreg.data_ptr(), _hip.stream()),
                   "rn_focal_loss_bwd")

    def timeit(run):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    fbytes = B * A * C * 4 + A * 16 + B * N * 27 * 4
    bbytes = fbytes + B * A * C * 4 + B * A * 12 * 4
    res = {}
    for name, run, nbytes in (("focal_loss_fwd", fwd, fbytes), ("focal_loss_bwd", bwd, bbytes)):
        ms = timeit(run)
        gbs = nbytes / (ms * 1e-3) / 1e9
        res[name] = {"kernel": "focal_kernel<dir,%s>" % name[-3:], "bound": "hbm", "achieved": round(gbs, 1),
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "ms": round(ms, 4),
                     "algorithmic_bytes": nbytes}
    # the pair as the training step runs it: one forward + one backward launch per step
    ms = res["focal_loss_fwd"]["ms"] + res["focal_loss_bwd"]["ms"]
    gbs = (fbytes + bbytes) / (ms * 1e-3) / 1e9
    res["focal_loss_fwd_bwd"] = {"kernel": "focal_kernel<dir,fwd> + focal_kernel<dir,bwd>", "bound": "hbm",
                                 "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(gbs / PEAK_HBM_GBS, 4), "ms": round(ms, 4),
                                 "algorithmic_bytes": fbytes + bbytes}
    return res


def bf16_section(dev, args, B, H, W):
    """Supplementary, NOT the headline: BASELINE configs[2]'s storage format on this GPU -- (a) the bf16 MFMA kernels on
    the benchmark's dominant layer (3x3, 256->256 at 135x240, batch B) against the 2.5 PF dense bf16 MFMA peak AND the HBM
    roof (algorithmic bytes: every operand once), (b) the same training step with bf16 activations (fp32 accumulation,
    master weights, gradients, loss)."""
    from retinanet_mi355x import conv as cv, modules, optim, synth
    C, Hh, Ww = 256, 135, 240
    x = cv.to_bf16(torch.randn(B, Hh, Ww, C, device=dev))
    g = cv.to_bf16(torch.randn(B, Hh, Ww, C, device=dev))
    w = torch.randn(C, C, 3, 3, device=dev) * 0.02
    wf = cv.pack_weights_bf16(w, 0)
    y = torch.empty(B, Hh, Ww, C, dtype=torch.bfloat16, device=dev)
    dw = torch.zeros(C, 9 * C, device=dev)
    flops = 2.0 * B * Hh * Ww * C * C * 9

    def timeit(fn, iters=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    out = {}
    for name, fn, nbytes in (("conv_igemm_bf16", lambda: cv.conv_igemm_bf16(x, wf, y, (Hh, Ww, C, 3, 3, 1, 1, -1, 0)),
                              2 * (x.numel() + y.numel() + wf.numel())),
                             ("conv_wgrad_bf16", lambda: cv.wgrad_bf16(g, x, dw, C, 3, 1, 1), 2 * (g.numel() + x.numel()) + 4 * dw.numel())):
        ms = timeit(fn)
        tf, gbs = flops / ms / 1e9, nbytes / ms / 1e6
        out[name] = {"layer": "3x3 256->256 @135x240, batch %d" % B, "bound": "mfma", "achieved": round(tf, 1),
                     "peak": PEAK_BF16_MFMA_TF, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_MFMA_TF, 4), "ms": round(ms, 4),
                     "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes": nbytes}
    del x, g, y, dw
    net = getattr(modules, args.arch)(num_classes=8)
    net.load_state_dict(synth.state_dict(args.arch, 8, 12, seed=2))
    net = net.to(dev)
    net.set_compute_dtype("bf16")
    net.train()
    net.freeze_bn()
    net.use_flat_gradients()
    opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
    img = frames(B, H, W, 0, dev)
    ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = sum(l.mean() for l in net([img, ann]))
        loss.backward()
        opt.step()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = time.time() - t0
    out["training_step"] = {"value": round(B * n / dt, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dt / n, 2),
                            "dtype": "bf16", "steps": n, "final_loss": round(float(loss.detach()), 5),
                            "note": "same step as the headline with bf16 activations / MFMA (fp32 accumulation, master weights, "
                                    "gradients, loss); not the reference's arithmetic, not the headline"}
    return out


PEAK_FP8_MFMA_TF = 5000.0         # dense fp8 MFMA (block-scaled f8f6f4 form), same guide


def fp8_section(dev, args, B, H, W):
    """Supplementary, NOT the headline: BASELINE configs[4]'s arithmetic, first cut (inference / forward only) -- (a) the e4m3
    implicit-GEMM kernel (csrc/conv_fp8.hip, v_mfma_scale_f32_32x32x64_f8f6f4) on two layer shapes of the benchmark against the
    ~5 PF dense fp8 MFMA peak and the HBM roof (algorithmic bytes: every operand once, 1 byte per element), (b) the model's forward
    pass (eval, no post-process) with e4m3 activations / weights against the same pass in fp32, activation scales calibrated on
    the benchmark's frames."""
    from retinanet_mi355x import conv as cv, modules, synth
    out = {}

    def timeit(fn, iters=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    for name, cin, cout, k, hh, ww in (("conv_igemm_fp8_3x3_256", 256, 256, 3, 135, 240), ("conv_igemm_fp8_1x1_1024_256", 1024, 256, 1, 68, 120)):
        x = torch.relu(torch.randn(B, hh, ww, cin, device=dev))
        w = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (k * k * cin)) ** 0.5
        xq = cv.fp8_quantize(x, float(x.max()) / cv.FP8_MAX)
        wq, sw = cv.fp8_quantize_weights(cv.pack_weights(w, 0, presplit=False))
        y = torch.empty(B, hh, ww, cout, dtype=torch.uint8, device=dev)
        scale = (sw * xq._rn_scale).contiguous()
        pad = k // 2
        ms = timeit(lambda: cv.conv_igemm_fp8(xq, wq, y, (hh, ww, cout, k, k, 1, 1, -pad, 0), scale, act=cv.ACT_RELU, out_scale=0.05))
        flops = 2.0 * B * hh * ww * cout * cin * k * k
        nbytes = xq.numel() + wq.numel() + y.numel()
        tf, gbs = flops / ms / 1e9, nbytes / ms / 1e6
        out[name] = {"layer": "%dx%d %d->%d @%dx%d, batch %d" % (k, k, cin, cout, hh, ww, B), "bound": "mfma", "achieved": round(tf, 1),
                     "peak": PEAK_FP8_MFMA_TF, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP8_MFMA_TF, 4), "ms": round(ms, 4),
                     "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes": nbytes}
        del x, xq, wq, y
    out["forward_pass"] = fp8_forward(dev, args.arch, B, H, W, with_fp32=True)
    try:                                       # BASELINE configs[4] at its own size: ResNet-101, 1920x1080, batch 16 per GPU
        torch.cuda.empty_cache()
        out["forward_pass_resnet101_b16"] = fp8_forward(dev, "resnet101", 16, H, W, with_fp32=False)
    except Exception as e:
        out["forward_pass_resnet101_b16"] = {"error": str(e)[:300]}
    return out


def fp8_forward(dev, arch, B, H, W, with_fp32, iters=3):
    """The detector's forward pass (eval: backbone + FPN + heads, no post-process) with e4m3 activations / weights, activation
    scales calibrated on two of the benchmark's frames; with_fp32: the same pass in fp32 first, for the ratio."""
    from retinanet_mi355x import modules, synth
    net = getattr(modules, arch)(num_classes=8)
    net.load_state_dict(synth.state_dict(arch, 8, 12, seed=2))
    net = net.to(dev).eval()
    img = frames(B, H, W, 0, dev)
    P = net._tensor_dict()

    def fwd():
        with torch.no_grad():
            net._engine.forward(P, img, save=False)

    def timeit(fn, n):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    ms32 = timeit(fwd, iters) if with_fp32 else None
    net.calibrate_fp8(img[:2])
    P = net._tensor_dict()
    ms8 = timeit(fwd, iters)                                 # the engine's default policy: bf16 residual stream + FPN, the rest e4m3
    net._engine.fp8_policy = "all"
    ms8_all = timeit(fwd, iters)                             # every layer in e4m3 (round 4's configuration)
    net._engine.fp8_policy = "stream_bf16"
    res = {"value": round(B / (ms8 * 1e-3), 1), "unit": "images/sec", "ms_per_pass": round(ms8, 2), "dtype": "fp8 (e4m3fn)",
           "arch": arch, "batch": B,
           "all_layers_e4m3": {"value": round(B / (ms8_all * 1e-3), 1), "ms_per_pass": round(ms8_all, 2),
                               "note": "RN_FP8_POLICY=all: scores 14.7 % max / 5.7 % rms off the fp32 forward at 1080p (ResNet-101) "
                                       "against 8.0 % / 2.0 % with the default policy (profiles/r05_fp8_error_budget.txt)"},
           "note": "forward only (backbone + FPN + heads, no post-process), fp32 stem and head outputs; the last convolution of every "
                   "bottleneck, the shortcuts and the FPN in bf16 (the residual stream is not re-quantised block after block), everything "
                   "else e4m3; the five pyramid levels of a head layer in one grouped fp8 launch; not the reference's arithmetic, not "
                   "the headline"}
    if ms32 is not None:
        res.update(fp32_ms_per_pass=round(ms32, 2), speedup_over_fp32_forward=round(ms32 / ms8, 2))
    return res


def native_section(dev, args, B, H, W, mode="native"):
    """Supplementary: the SAME training step with the fp32 convolutions in another product mode -- "native": v_mfma_f32_32x32x2_f32
    (RN_FP32_NATIVE), "split": three bf16 terms / six MFMAs (RN_FP32_SPLIT, the default of rounds 2-4) -- on this GPU in this run:
    the reader's yardsticks for the headline."""
    from retinanet_mi355x import conv as cv, modules, optim, prof, synth
    before = cv.get_fp32_mfma()
    cv.set_fp32_mfma(mode)
    peak = PEAK_F32_MFMA_TF if mode == "native" else PEAK_SPLIT_TF
    try:
        net = getattr(modules, args.arch)(num_classes=8)
        net.load_state_dict(synth.state_dict(args.arch, 8, 12, seed=2))
        net = net.to(dev)
        net.train()
        net.freeze_bn()
        net.use_flat_gradients()
        opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
        img = frames(B, H, W, 0, dev)
        ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = sum(l.mean() for l in net([img, ann]))
            loss.backward()
            opt.step()
            return loss
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        n = max(args.steps, 3)
        t0 = time.time()
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        dt = time.time() - t0
        out = {"value": round(B * n / dt, 2), "unit": "images/sec", "ms_per_step": round(1e3 * dt / n, 2), "steps": n,
               "final_loss": round(float(loss.detach()), 5),
               "products": "v_mfma_f32_32x32x2_f32" if mode == "native" else "three bf16 terms per operand, six v_mfma_f32_*_bf16 per product"}
        t = prof.ACTIVE = prof.KernelTimer()
        for _ in range(2):
            step()
        summ = t.summary()
        prof.ACTIVE = None
        for kind in ("conv_igemm_2x2", "conv_igemm_4x1", "conv_wgrad"):
            a = summ.get(kind)
            if a and a["ms_total"] > 0:
                tf = a["work_total"] / (a["ms_total"] * 1e-3) / 1e12
                out[kind] = {"achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                             "ms_per_step": round(a["ms_total"] / 2, 2)}
        # north_star's forward target on the fp32 MFMA: conv kernels of forward-only passes, DIRECT kernels everywhere
        eng = net._engine
        wino_was, eng.use_wino = eng.use_wino, False
        t = prof.ACTIVE = prof.KernelTimer()
        with torch.no_grad():
            for _ in range(3):
                net([img, ann])
        fs = t.summary().values()
        prof.ACTIVE = None
        eng.use_wino = wino_was
        fwork, fms = sum(a["work_total"] for a in fs), sum(a["ms_total"] for a in fs)
        if fms > 0:
            ftf = fwork / (fms * 1e-3) / 1e12
            out["forward_convs"] = {"bound": "mfma", "achieved": round(ftf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                    "frac": round(ftf / peak, 4), "ms_per_pass": round(fms / 3, 2),
                                    "gflop_per_image": round(fwork / 3 / B / 1e9, 1)}
        return out
    finally:
        prof.ACTIVE = None
        cv.set_fp32_mfma(before)


def pmc_traffic(kind):
    """Average HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes of THIS command
    (tools/collect_traffic.sh -> profiles/pmc_traffic.json; FETCH_SIZE doubled per the gfx950 correction).
    PMC counters cannot be read from inside the process, so the number is the last collected one, or None."""
    prefix = {"conv_igemm_2x2": ("conv_igemm_kernel<2, 2", "conv_igemm_grouped_kernel<2, 2", "conv_igemm_split_kernel<2, 2",
                                 "conv_igemm_split_grouped_kernel<2, 2", "conv_igemm_mf16_kernel", "conv_igemm_mf16_grouped_kernel"),
              "conv_igemm_4x1": ("conv_igemm_kernel<4, 1", "conv_igemm_grouped_kernel<4, 1", "conv_igemm_split_kernel<4, 1",
                                 "conv_igemm_split_grouped_kernel<4, 1"),
              "conv_wgrad": ("conv_wgrad_kernel", "conv_wgrad_once_kernel"),
              "wino_input": ("wino_in_kernel", "wino_in_both_kernel", "wino_dy_kernel"),
              "wino_output": ("wino_out_kernel",)}.get(kind)
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if prefix is None or not os.path.exists(path) or not traffic_source()["matches_current_sources"]:
        return None
    rows = [v for k, v in json.load(open(path)).items() if k.startswith(prefix)]
    n = sum(v["launches"] for v in rows)
    return int(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in rows) / n) if n else None


def step_traffic():
    """HBM bytes of one whole training step (all kernels) from the same PMC collection, or None when it is not this run's sources."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(path) or not traffic_source()["matches_current_sources"]:
        return None
    return json.load(open(path)).get("_meta", {}).get("step_total_hbm_bytes")


_TRAFFIC_SOURCE = {}


def traffic_source():
    """Where `traffic` comes from: PMC counters cannot be read inside the process, so the bytes are those of the last committed
    collection (tools/collect_traffic.sh) -- quoted ONLY when that collection was made on the same kernel sources and launch
    schedule as this run (prof.sources_digest; no git on the GPU box), otherwise traffic is null."""
    if not _TRAFFIC_SOURCE:
        from retinanet_mi355x import prof
        path = os.path.join(REPO, "profiles", "pmc_traffic.json")
        meta = json.load(open(path)).get("_meta", {}) if os.path.exists(path) else {}
        now = prof.sources_digest()
        _TRAFFIC_SOURCE.update({"file": "profiles/pmc_traffic.json", "collected": meta.get("collected"),
                                "sources_sha256": meta.get("sources_sha256"), "this_run_sources_sha256": now,
                                "matches_current_sources": bool(meta) and meta.get("sources_sha256") == now,
                                "note": "HBM bytes per launch from two separate rocprofv3 --pmc passes of this command (FETCH_SIZE "
                                        "doubled per the gfx950 correction), collected earlier on the sources named; null when they "
                                        "differ from this run's"})
    return _TRAFFIC_SOURCE


def cpu_baseline(arch, H, W):
    """oracle/ (torch CPU kernels, reference algorithm) on this box's host cores, a bounded sample of the workload:
    the whole model (forward + loss + backward on ONE image, all cores) = `value`, plus the path's components one by one at
    the benchmark's own sizes, with all cores and with one thread: anchors, the fused loss forward and
    forward + backward, box decode and the MULTI_FRAME post-process at the benchmark's batch (B = 8), the
    homography round trip."""
    import numpy as np
    from oracle import anchors as oanchors, boxes as oboxes, homography as ohg, losses as olosses, model as omodel
    from retinanet_mi355x import synth
    threads = torch.get_num_threads()
    sd = synth.state_dict(arch, 8, 12, seed=2)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
              for k, v in sd.items()}
    img = synth.frames(1, H, W, seed=0)
    ann = synth.labels_dir(1, 10, H, W, 8, seed=1)
    t0 = time.time()
    losses = omodel.train_forward(img, ann, params, arch)
    sum(l.mean() for l in losses).backward()
    dt = time.time() - t0
    out = {"value": round(1.0 / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
           "sample": "1 image %dx%d, %s forward+loss+backward, torch CPU fp32 (%.1f s)" % (W, H, arch, dt)}
    del params, losses
    # ---- components (seconds per call; [all cores, 1 thread])
    Bc = 8
    anc = torch.from_numpy(oanchors.anchors_for_image(H, W))
    A = anc.shape[1]
    cls, reg = synth.head_outputs(1, A, 8, 12, seed=3)
    cls, reg = cls.expand(Bc, A, 8).contiguous(), reg.expand(Bc, A, 12).contiguous()
    ann2 = synth.labels_dir(Bc, 10, H, W, 8, seed=1)
    Ps, Hs = synth.camera_matrices(18, seed=5)
    state = synth.vehicle_states(3600, seed=6).numpy()
    cam = np.arange(3600) % 18

    def loss_fb():
        c, r = cls.clone().requires_grad_(True), reg.clone().requires_grad_(True)
        sum(l.mean() for l in olosses.focal_loss_dir(c, r, anc, ann2)).backward()

    def hg_round_trip():
        im = ohg.space_to_im(ohg.state_to_space(state).astype(np.float64), Ps[cam])
        ohg.im_to_state(im, Hs[cam], np.full(3600, 5.0, dtype=np.float32))
    boxes = oboxes.decode_dir(anc, reg)
    comps = (("anchors_1080p", lambda: oanchors.anchors_for_image(H, W)),
             ("loss_fwd_B%d" % Bc, lambda: olosses.focal_loss_dir(cls, reg, anc, ann2)),
             ("loss_fwd_bwd_B%d" % Bc, loss_fb),
             ("decode_B%d" % Bc, lambda: oboxes.decode_dir(anc, reg)),
             ("postprocess_multi_B%d" % Bc, lambda: oboxes.postprocess_multi(cls, boxes)),
             ("homography_3600_boxes_round_trip", hg_round_trip))
    res = {}
    for name, fn in comps:
        row = []
        for nt in (threads, 1):
            torch.set_num_threads(nt)
            with torch.no_grad() if "bwd" not in name else torch.enable_grad():
                fn()                                             # untimed: first-call effects (thread pool, lazy init)
                t0 = time.time()
                fn()
                row.append(round(time.time() - t0, 4))
        res[name] = row
    torch.set_num_threads(threads)
    out["components_seconds_allcores_1thread"] = res
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N rank processes ourselves -- as a
    torch.distributed.run CHILD, before this process has touched the GPU (never re-exec a process that has) -- relay
    their output (rank 0 prints the JSON line) and exit with their code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()                       # does not initialise the GPU on this image
    if have < args.gpus and not os.environ.get("RN_REHEARSE_ONE_GPU"):
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s); not reporting a smaller run as n_gpus=%d"
                         % (args.gpus, have, args.gpus))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    # this loop never looks at a loss on the host: the all-empty-batch check travels asynchronously (ops.check_labels; the
    # library's default reads it on the spot, as the reference's FocalLoss raises inside the forward)
    os.environ.setdefault("RN_DEFERRED_LABEL_CHECK", "1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    from retinanet_mi355x import ddp, modules, optim, prof, synth
    rank, local, world = ddp.init_from_env()
    if world > 1:
        world = dist.get_world_size()                      # what RCCL / gloo actually formed
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s)" % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B, H, W = args.batch, args.height, args.width
    from retinanet_mi355x import conv as cv
    if args.fp32_mfma:
        cv.set_fp32_mfma(args.fp32_mfma)
    fp32_mode = cv.get_fp32_mfma()
    split = fp32_mode in ("split", "split3") and args.dtype == "fp32"
    conv_peak = (PEAK_SPLIT3_TF if fp32_mode == "split3" else PEAK_SPLIT_TF) if split else PEAK_F32_MFMA_TF      # what the fp32 conv kernels are priced against

    net = getattr(modules, args.arch)(num_classes=8)
    net.load_state_dict(synth.state_dict(args.arch, 8, 12, seed=2))          # same weights on every rank
    net = net.to(dev)
    if args.dtype == "bf16":
        net.set_compute_dtype("bf16")
    if args.dtype == "fp8":
        # BASELINE configs[4] as a TRAINING configuration (round 5): forward on the fp8 MFMA with e4m3 activations / weights (the residual
        # stream and the FPN in bf16), activations saved in those formats, loss in fp32, data and weight gradients on the bf16 kernels,
        # fp32 master weights, clip + Adam.  Activation scales: calibrated once on two of the benchmark's frames (static scales).
        net.eval()
        net.calibrate_fp8(frames(2, H, W, 100 + rank, dev))
    net.train()
    net.freeze_bn()
    if world > 1:
        # bucketed all-reduce inside backward, in place on the flat buffer; the SUM stays there and the fused optimizer
        # multiplies by 1/world while it reads the gradients (no 147 MB scaling pass after the last bucket)
        reducer = ddp.GradReducer(timeline=args.ddp_timeline, defer_scale=True)
        net.set_gradient_reducer(reducer)
    else:
        net.use_flat_gradients()                             # same persistent gradient buffer, no exchange
    params = [p for p in net.parameters() if p.requires_grad]
    # clip_grad_norm_(0.1) + Adam(lr 1e-4) (train_detector_3D_angle.py:337, 385-387) as one fused native step
    opt = optim.ClipAdam(params, lr=1e-4, max_norm=0.1, grad_scale=reducer.grad_scale if world > 1 else 1.0)
    img = frames(B, H, W, 0 + rank, dev)                          # the frames the parity tests and cpu_baseline use, resident in HBM
    ann = synth.labels_dir(B, 10, H, W, 8, seed=1 + rank).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        cls_l, reg_l, vp_l = net([img, ann])
        loss = cls_l.mean() + reg_l.mean() + vp_l.mean()                       # train_detector_3D_angle.py:374-378
        loss.backward()
        if world > 1:
            # the replicas' mean losses, what the reference prints and feeds ReduceLROnPlateau (:374-381, 412): one 3-float
            # all-reduce per step, result left on the device
            ddp.mean_losses(cls_l, reg_l, vp_l)
        opt.step()                                                             # clip (:385) + Adam step (:387)
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    run_step, graph_note = step, None
    if args.graph and world == 1:
        # The schedule is a fixed launch sequence with no host decision inside (engine.py): capture one step -- ~700 launches
        # of the fp32 schedule -- and replay it.  Needs >= 2 eager warm-up steps (packed-weight job tables, persistent
        # gradient buffer and the optimizer's pointer table must exist before the capture).
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                graph_loss = step()

            def run_step():
                graph.replay()
                return graph_loss
            run_step()
            torch.cuda.synchronize()
            graph_note = "whole step replayed from one captured hipGraph"
        except Exception as e:                                  # report, and measure the eager path
            run_step, graph_note = step, "capture failed, eager launches measured: %s" % str(e)[:200]
            torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        loss = run_step()
    barrier()
    dt = time.time() - t0
    # Kernel roofline: the SAME K steps again, every launch of the conv / transform kernels bracketed by HIP events on the
    # launch stream (prof.KernelTimer).  Kept out of the throughput region above because ~1 400 event records per step
    # cost 2.4 % of it (74.8 vs 76.5 images/s measured); the kernels, shapes and order are identical.
    timer = None
    if not args.no_kernel_timing:                              # every rank (the steps contain the all-reduce); rank 0 reports
        timer = prof.ACTIVE = prof.KernelTimer()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    prof.ACTIVE = None
    if world > 1:
        dist.barrier()
    final_loss = float(loss.detach())
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    # The library's DEFAULT mode checks the labels eagerly (one host read per step, as the reference's FocalLoss raises inside the
    # forward); the loop above defers that check (RN_DEFERRED_LABEL_CHECK, set in main).  The same K steps in the default mode:
    eager = None
    if world == 1 and args.sections == "all" and not (args.graph and graph_note and "replayed" in graph_note):
        os.environ["RN_EAGER_LABEL_CHECK"] = "1"
        try:
            step()
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            eager = B * args.steps / (time.time() - t1)
        finally:
            del os.environ["RN_EAGER_LABEL_CHECK"]

    if rank == 0:
        arch_label = {"resnet50": "ResNet-50", "resnet101": "ResNet-101", "resnet152": "ResNet-152", "resnet18": "ResNet-18",
                      "resnet34": "ResNet-34"}.get(args.arch, args.arch)
        line = {"metric": "training images/sec at %dx%d, %s 3D-RetinaNet" % (W, H, arch_label), "value": round(value, 3),
                "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": {"fp32": "f32", "bf16": "bf16", "fp8": "fp8 forward (e4m3fn), bf16 gradients"}[args.dtype], "data": "synthetic",
                "config": {"workload": "%s directional 3D-RetinaNet, %dx%d synthetic frames, batch %d per GPU, 10 GT "
                                       "boxes/image, %s, fwd+loss+bwd+clip+Adam (BASELINE configs[%d])"
                                       % (args.arch, W, H, B,
                                          {"fp32": "fp32", "bf16": "bf16 activations / MFMA with fp32 accumulation and master weights",
                                           "fp8": "fp8 (e4m3) forward on the fp8 MFMA with a bf16 residual stream, bf16 data / weight "
                                                  "gradients, fp32 accumulation and master weights"}[args.dtype],
                                          {"fp32": 1, "bf16": 2, "fp8": 4}[args.dtype]),
                           "global_batch": world * B, "parallelism": "dp%d" % world, "final_loss": round(final_loss, 5)},
                # what the process group actually is: an 8-GPU line with backend "nccl" (= RCCL on ROCm) and ranks 8 is an RCCL run
                "backend": dist.get_backend() if world > 1 else None, "ranks": dist.get_world_size() if world > 1 else 1}
        if eager is not None:
            line["value_eager_label_check"] = round(eager, 3)
            line["label_check"] = ("value: all-empty-batch check deferred by one call (RN_DEFERRED_LABEL_CHECK=1, no host read per step); "
                                   "value_eager_label_check: the library's default, one host read per step as the reference's loss raises "
                                   "inside the forward")
        if args.dtype == "fp32":
            line["config"]["fp32_products"] = {
                "split3": "split operands, two fp16 terms: every fp32 operand, scaled by a power of two so that its image's (weights: its output "
                          "channel's) largest magnitude lands in [2^14, 2^15), as hi + lo fp16 terms (|error| <= 2^-23 of the value down to 2^-18 "
                          "of that maximum), three v_mfma_f32_16x16x32_f16 / 32x32x16_f16 products per block into the fp32 accumulator; operands, "
                          "accumulation, epilogues, gradients and optimizer fp32; measured against fp64: rms error of every kernel 0.55-1.17 x the "
                          "fp32 MFMA kernels' on N(0,1), ReLU and 40-binade data (profiles/r05_fp32_split3_errors.txt), every parity test at the "
                          "tolerances of the other modes; layers without an fp16 kernel (<= 64 output channels, the 4-channel stem, the padded "
                          "head-output gradients) run the three-term bf16 form; fp32_split_bf16x3 / fp32_native_mfma below = the same step in "
                          "the other two modes",
                "split": "split operands: every fp32 operand as three bf16 terms (h + m + l == x exactly), six v_mfma_f32_32x32x16_bf16 "
                         "products per block into the fp32 accumulator, dropped terms <= 2^-23 of a product (2^-25 rms); operands, accumulation, "
                         "epilogues, gradients and optimizer fp32 (DESIGN.md 4.6); fp32_native_mfma below = the same step on v_mfma_f32_32x32x2_f32",
                "native": "v_mfma_f32_32x32x2_f32"}[fp32_mode]
        if graph_note:
            line["config"]["launch"] = graph_note
        if timer is not None:
            summ = timer.summary()
            kernels = {}
            for kind, a in summ.items():
                if kind in ("wino_input", "wino_output"):                        # streaming transforms: work = bytes moved
                    gbs = a["work_total"] / (a["ms_total"] * 1e-3) / 1e9 if a["ms_total"] > 0 else 0.0
                    kernels[kind] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                     "frac": round(gbs / PEAK_HBM_GBS, 4), "launches_per_step": a["launches"] // args.steps,
                                     "ms_per_step": round(a["ms_total"] / args.steps, 2), "avg_launch_ms": round(a["ms_avg"], 4),
                                     "algorithmic_bytes": int(a["work_total"] / max(a["launches"], 1)), "traffic": pmc_traffic(kind)}
                    if kernels[kind]["traffic"]:
                        kernels[kind]["traffic_over_algorithmic"] = round(kernels[kind]["traffic"] / kernels[kind]["algorithmic_bytes"], 3)
                    continue
                tf = a["work_total"] / (a["ms_total"] * 1e-3) / 1e12 if a["ms_total"] > 0 else 0.0
                is_bf16 = "_bf16" in kind                                        # conv_igemm_bf16, conv_igemm_bf16_p8, conv_wgrad_bf16
                peak = PEAK_FP8_MFMA_TF if "_fp8" in kind else (PEAK_BF16_MFMA_TF if is_bf16 else conv_peak)
                kernels[kind] = {"bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                 "frac": round(tf / peak, 4), "launches_per_step": a["launches"] // args.steps,
                                 "ms_per_step": round(a["ms_total"] / args.steps, 2), "avg_launch_ms": round(a["ms_avg"], 4)}
                if split and not is_bf16 and "_fp8" not in kind:
                    # achieved = fp32 FLOPs of the convolution; the kernel executes 3 (split3) or 6 (split) 16-bit MFMA FLOPs for each of them
                    kernels[kind]["peak_note"] = ("2500 TF dense fp16 MFMA / 3 MFMAs per fp32 product (launches that run the three-term "
                                                  "bf16 kernels in this mode execute 6)" if fp32_mode == "split3"
                                                  else "2500 TF dense bf16 MFMA / 6 MFMAs per fp32 product")
                    kernels[kind]["frac_of_fp32_mfma_peak"] = round(tf / PEAK_F32_MFMA_TF, 4)
                if a.get("bytes_total") and a["ms_total"] > 0:
                    # Which roof bounds the family (the roofline model's own rule): its arithmetic intensity -- algorithmic FLOPs over
                    # algorithmic bytes, every operand once -- against the ridge point peak / 8 TB/s.  Below the ridge the attainable rate
                    # is intensity x bandwidth and the family is priced against HBM; the matrix-core fraction stays beside it.
                    k = kernels[kind]
                    gbs = a["bytes_total"] / (a["ms_total"] * 1e-3) / 1e9
                    intensity = a["work_total"] / a["bytes_total"]
                    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
                    k["intensity_flop_per_byte"], k["ridge_flop_per_byte"] = round(intensity, 1), round(ridge, 1)
                    k["attainable_tflops"] = round(min(peak, intensity * PEAK_HBM_GBS * 1e-3), 1)
                    k["mfma_frac"], k["hbm_gbs"], k["hbm_frac"] = k["frac"], round(gbs, 1), round(gbs / PEAK_HBM_GBS, 4)
                    if intensity < ridge:
                        k.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": round(gbs / PEAK_HBM_GBS, 4), "achieved_tflops": round(tf, 2), "mfma_peak_tflops": round(peak, 1)})
            dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
            r = dict(kernels[dom])
            r["kernel"] = dom
            r["traffic"] = pmc_traffic(dom)
            r["traffic_source"] = traffic_source()
            # ALGORITHMIC bytes per launch of the same family (every operand once), so that the line itself shows the ratio:
            # traffic well above it = re-reads
            a = summ[dom]
            if a.get("bytes_total"):
                r["algorithmic_bytes"] = int(a["bytes_total"] / max(a["launches"], 1))
                if r["traffic"]:
                    r["traffic_over_algorithmic"] = round(r["traffic"] / r["algorithmic_bytes"], 3)
            line["roofline"] = r
            line["kernels"] = kernels
            step_bytes = step_traffic()
            if step_bytes and world == 1:
                # the WHOLE step against the memory roof: every kernel's HBM bytes (same PMC passes) over the step's wall time
                gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
                line["step_hbm"] = {"bytes_per_step": step_bytes, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(gbs / PEAK_HBM_GBS, 4),
                                    "note": "all kernels of one training step, HBM bytes from the PMC passes named in roofline.traffic_source"}
        if timer is not None and args.sections == "all":
            kernels.update(loss_kernel_roofline(dev, B, H, W))
            # north_star target "MFMA roofline on the ResNet-50-FPN forward": conv kernels of forward-only passes, with the
            # DIRECT kernels everywhere (executed FLOPs = algorithmic FLOPs, so the fraction is a true MFMA utilisation)
            eng = net._engine
            wino_was = eng.use_wino

            def forward_passes():
                t = prof.ACTIVE = prof.KernelTimer()
                with torch.no_grad():
                    for _ in range(3):
                        net([img, ann])
                torch.cuda.synchronize()
                prof.ACTIVE = None
                return t.summary()
            eng.use_wino = False
            fs = forward_passes().values()
            eng.use_wino = wino_was
            fwork, fms = sum(a["work_total"] for a in fs), sum(a["ms_total"] for a in fs)
            ftf = fwork / (fms * 1e-3) / 1e12 if fms > 0 else 0.0
            fpeak = {"bf16": PEAK_BF16_MFMA_TF, "fp8": PEAK_FP8_MFMA_TF}.get(args.dtype, conv_peak)   # the engine's own matrix-core peak
            kernels["forward_convs"] = {"bound": "mfma", "achieved": round(ftf, 2), "peak": round(fpeak, 1),
                                        "unit": "TFLOP/s", "frac": round(ftf / fpeak, 4),
                                        "ms_per_pass": round(fms / 3, 2), "gflop_per_image": round(fwork / 3 / B / 1e9, 1)}
            if split:
                kernels["forward_convs"]["frac_of_fp32_mfma_peak"] = round(ftf / PEAK_F32_MFMA_TF, 4)
            if wino_was and args.dtype == "fp32":                  # (the bf16 / fp8 engines never take the Winograd path)
                # the forward as the training step runs it (Winograd F(4x4,3x3) in the head towers): same algorithmic FLOPs
                # over the time of every conv-path kernel, transforms included -- a throughput, not an MFMA utilisation
                wms = sum(a["ms_total"] for a in forward_passes().values())
                kernels["forward_convs_training"] = {
                    "ms_per_pass": round(wms / 3, 2), "algorithmic_tflops": round(fwork / (wms * 1e-3) / 1e12, 2),
                    "note": "3x3 stride-1 layers with >= 128 channels by Winograd F(4x4,3x3): a quarter of the multiplications there"}
            line["kernels"] = kernels
        if world == 1 and args.dtype == "fp32" and timer is not None and args.sections == "all":
            del net, opt, params
            torch.cuda.empty_cache()
            if split:
                try:                                                # supplementary sections never cost the headline its line
                    line["fp32_native_mfma"] = native_section(dev, args, B, H, W)
                    if fp32_mode == "split3":
                        torch.cuda.empty_cache()
                        line["fp32_split_bf16x3"] = native_section(dev, args, B, H, W, mode="split")
                    line["note"] = ("value: fp32 step with split-operand products on the 16-bit matrix cores (config.fp32_products); the same "
                                    "step with every product on v_mfma_f32_32x32x2_f32, same run: %s images/sec (fp32_native_mfma)%s"
                                    % (line["fp32_native_mfma"].get("value"),
                                       "; with three bf16 terms / six MFMAs per product (the default of rounds 2-4): %s (fp32_split_bf16x3)"
                                       % line["fp32_split_bf16x3"].get("value") if fp32_mode == "split3" else ""))
                except Exception as e:
                    line["fp32_native_mfma"] = {"error": str(e)[:300]}
            try:
                line["bf16"] = bf16_section(dev, args, B, H, W)
            except Exception as e:
                line["bf16"] = {"error": str(e)[:300]}
            try:
                torch.cuda.empty_cache()
                line["fp8"] = fp8_section(dev, args, B, H, W)
            except Exception as e:
                line["fp8"] = {"error": str(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.arch, H, W)
            except Exception as e:
                line["cpu_baseline"] = {"error": str(e)[:300]}
        print(json.dumps(line), flush=True)
    if world > 1:
        if args.ddp_timeline and rank == 0:
            tl = net.__dict__["_reducer"].timeline[-1]
            print("ddp timeline (last step, ms since backward began): backward's last kernel retired at %.2f"
                  % tl["backward_gpu_ms"], file=sys.stderr)
            for b in tl["buckets"]:
                print("  bucket %(bucket)d  %(mbytes)6.1f MB  gradients final %(ready_gpu_ms)8.2f (GPU)  all-reduce issued "
                      "%(launch_host_ms)8.2f  done %(done_host_ms)8.2f (host)" % b, file=sys.stderr)
        dist.barrier()                                            # rank 0 arrives last (it measured the extra kernels alone)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
