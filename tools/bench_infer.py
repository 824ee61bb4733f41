#!/usr/bin/env python3
"""Inference path, BASELINE configs[3]: 18 cameras (p1c1 .. p3c6) of raw 1080p uint8 frames -> batched detect
(MULTI_FRAME, 2-3 cameras per call as one GPU's share would be; here all 18 go through ONE GPU) -> parse_detections
(confidence cut, image NMS, image->state per camera, road-plane NMS) -> state_to_im, and the single-frame path of
perform_3D_detection_on_video_sequences.py.  Prints frames/s.

Two HIP streams: the upload of camera batch k+1 (pinned host memory -> device, 6.2 MB per camera instead of the 24.9 MB
fp32 tensor the reference's loader ships, util_track/mp_loader.py:239-247) runs on a copy stream under the detector of
batch k; ingest (to_tensor + normalize, fused into the stem's layout), network, post-process and the tracker's parsing
all run on the compute stream and never leave the device until the parsed states do.

  python tools/bench_infer.py [--cams 18] [--batch 3] [--iters 5]
  python tools/bench_infer.py --gpus N ...      the cameras dealt over N GPUs, one process each (retinanet_mi355x/multicam.py):
                                                rank r detects cameras r, r+N, ..., the ranks' survivors are gathered, rank 0
                                                merges them and runs parse_detections + state_to_im once per time step; one JSON
                                                line with frames/s per GPU and aggregate.  Starts its own ranks (a
                                                torch.distributed.run child, before this process touches the GPU) or reads
                                                torchrun's environment.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
import homography as hgm  # noqa: E402
import mc3d_post  # noqa: E402
import torch.distributed as dist  # noqa: E402
from retinanet_mi355x import conv as cv, ddp, modules, multicam, synth  # noqa: E402

H, W = 1080, 1920


def detector(dev):
    sd = synth.state_dict("resnet50", 8, 12, seed=2, head_scale=2e-3)
    w = sd["classificationModel.output.weight"]
    sd["classificationModel.output.weight"] = torch.from_numpy((synth.uniform(tuple(w.shape), 901) - 0.5).astype(np.float32) * 0.01)
    sd["classificationModel.output.bias"] = torch.full_like(sd["classificationModel.output.bias"], -5.5)
    sd["regressionModel.output.bias"] = torch.tensor([0.0, 0.0, 0.30, 0.05, 0.05, 0.15, 0.0, 0.20, -0.5, -0.5, 0.5, 0.5]).repeat(9)
    net = modules.resnet50(num_classes=8)
    net.load_state_dict(sd)
    return net.to(dev).eval()


def set_dtype(net, dtype, dev):
    """--dtype bf16: DESIGN.md 4.5; --dtype fp8: DESIGN.md 4.7 (e4m3 activations / weights; the per-tensor scales are calibrated here on
    two noise frames drawn like the benchmark's, through the fp32 engine).  Both opt-in, neither the reference's arithmetic."""
    if dtype == "bf16":
        net.set_compute_dtype("bf16")
    elif dtype == "fp8":
        from retinanet_mi355x import ops
        g = torch.Generator().manual_seed(11)
        f = torch.randint(0, 256, (2, H, W, 3), generator=g, dtype=torch.uint8).to(dev).float() / 255.0
        mean = torch.tensor(ops.IMAGENET_MEAN, device=dev).view(1, 1, 1, 3)
        std = torch.tensor(ops.IMAGENET_STD, device=dev).view(1, 1, 1, 3)
        net.calibrate_fp8(((f - mean) / std).permute(0, 3, 1, 2).contiguous(), margin=1.25)


def tracker(dev, names):
    P, Hm = synth.camera_matrices(len(names), seed=5)
    P2, H2 = synth.camera_matrices(len(names), seed=55)

    def make_hg(Pm, Hh):
        hg = hgm.Homography(device=str(dev))
        hg.correspondence = {n: {"P": Pm[i], "H": Hh[i], "H_inv": np.linalg.inv(Hh[i])} for i, n in enumerate(names)}
        hg.default_correspondence = names[0]
        return hg
    me = mc3d_post.DetectionParser()
    me.sigma_d, me.phi_nms_im, me.phi_nms_space = 0.3, 0.3, 0.1
    me.cameras, me.est_ts = list(names), False
    me.hg = hgm.Homography_Wrapper(hg1=make_hg(P, Hm), hg2=make_hg(P2, H2))
    return me


def launch_ranks(args):
    """--gpus N without a torchrun environment: N rank processes as a torch.distributed.run CHILD, started before this process
    has touched the GPU (never re-exec a process that has); exit with their code."""
    have = torch.cuda.device_count()                       # does not initialise the GPU on this image
    if have < args.gpus and not os.environ.get("RN_REHEARSE_ONE_GPU"):
        raise SystemExit("bench_infer.py: --gpus %d but this node shows %d GPU(s)" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def sharded(args, rank, local, world):
    """The N-GPU protocol of retinanet_mi355x/multicam.py (also runs with world = 1: the same chain with one rank, the yardstick
    for the aggregate).  Every rank: its cameras' uint8 frames in pinned host memory -> copy stream -> detector(MULTI_FRAME) in
    calls of --batch cameras -> survivors with global camera ids; then gather, and on rank 0 merge + parse_detections +
    state_to_im once per time step."""
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    net = detector(dev)
    set_dtype(net, args.dtype, dev)
    names = multicam.CAMERAS[:args.cams] if args.cams <= 18 else ["cam%d" % i for i in range(args.cams)]
    mine = multicam.shard(args.cams, world, rank)
    me = tracker(dev, names)
    g = torch.Generator().manual_seed(7)
    frames_all = torch.randint(0, 256, (args.cams, H, W, 3), generator=g, dtype=torch.uint8)     # every rank draws the same 18 frames ...
    host = frames_all[mine].contiguous().pin_memory()                                             # ... and keeps its cameras'
    del frames_all
    n = len(mine)
    spans = [(a, min(n, a + args.batch)) for a in range(0, n, args.batch)]
    compute, copy = torch.cuda.current_stream(), torch.cuda.Stream()
    bufs = [torch.empty((args.batch, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]

    def upload(k):
        a, b = spans[k]
        with torch.cuda.stream(copy):
            copy.wait_event(free[k % 2])
            bufs[k % 2][:b - a].copy_(host[a:b], non_blocking=True)
            copied[k % 2].record(copy)

    def time_step():
        for e in free:
            e.record(compute)
        parts = []
        if spans:
            upload(0)
        for k, (a, b) in enumerate(spans):
            if k + 1 < len(spans):
                upload(k + 1)                                # under the detector of call k
            compute.wait_event(copied[k % 2])
            s, c, bx, im = net(bufs[k % 2][:b - a], MULTI_FRAME=True)
            free[k % 2].record(compute)
            parts.append((s, c, bx, multicam.to_global(im, mine[a:b])))
        if parts:
            s, c, bx, cam = (torch.cat([p[i] for p in parts]) for i in range(4))
        else:                                                # more ranks than cameras: this rank only takes part in the gather
            s, c = torch.empty(0, device=dev), torch.empty(0, dtype=torch.int64, device=dev)
            bx, cam = torch.empty((0, 20), device=dev), torch.empty(0, dtype=torch.int64, device=dev)
        gathered = multicam.gather_detections(s, c, bx, cam)
        kept = sum(int(p[0].numel()) for p in gathered)
        parsed, checksum = 0, 0.0
        if rank == 0 and kept:
            ms, mc, mb, mcam = multicam.merge(gathered)
            st, lb, sc, cm = me.parse_detections(ms, mc, mb, mcam)
            if isinstance(st, torch.Tensor) and st.shape[0]:
                me.hg.state_to_im(st, name=[names[i] for i in cm.cpu().tolist()])
                parsed = int(st.shape[0])
                checksum = float(torch.nan_to_num(st.double()).sum()) if args.checksum else 0.0
        return kept, parsed, checksum

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    time_step()
    barrier()
    t0 = time.time()
    for _ in range(args.iters):
        kept, parsed, checksum = time_step()
    barrier()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    per_step = dt / args.iters
    if rank == 0:
        label = args.dtype if args.dtype != "fp32" else "fp32, %s products" % cv.get_fp32_mfma()
        print(json.dumps({"metric": "inference frames/sec, %d cameras x %dx%d uint8, ResNet-50 3D-RetinaNet (%s), detect + parse_detections "
                                    "+ state_to_im" % (args.cams, W, H, label),
                          "value": round(args.cams / per_step, 2), "unit": "frames/sec", "n_gpus": world, "ms_per_time_step": round(1e3 * per_step, 2),
                          "frames_per_sec_per_gpu": round(args.cams / per_step / world, 2), "cameras_per_rank": [len(s) for s in multicam.shards(args.cams, world)],
                          "cameras_per_call": args.batch, "time_steps": args.iters, "detections_kept": kept, "objects_parsed": parsed,
                          "parsed_states_checksum": checksum if args.checksum else None,
                          "scaling": "strong (18 cameras whatever N)", "data": "synthetic (uniform-noise frames)",
                          "backend": dist.get_backend() if world > 1 else None, "ranks": dist.get_world_size() if world > 1 else 1,
                          "protocol": "rank r: cameras r, r+N, ...; all_gather of the survivors; rank 0 merges and parses (retinanet_mi355x/multicam.py)"}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="> 1: the cameras dealt over N GPUs, one process each (JSON line)")
    ap.add_argument("--checksum", action="store_true", help="sharded protocol: also print the sum of the parsed states (one host read per step)")
    ap.add_argument("--sharded", action="store_true", help="run the N-GPU protocol with one rank as well (JSON line) and stop")
    ap.add_argument("--cams", type=int, default=18)
    ap.add_argument("--batch", type=int, default=3, help="cameras per detector call (18 cameras over 8 GPUs: 2-3 each)")
    ap.add_argument("--iters", type=int, default=5, help="time steps (one frame from every camera each)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp8"],
                    help="bf16: DESIGN.md 4.5, fp8: DESIGN.md 4.7 (opt-in, not the reference's arithmetic)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    rank, local, world = ddp.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench_infer.py: --gpus %d but the process group has %d rank(s)" % (args.gpus, world))
    if world > 1 or args.sharded:
        return sharded(args, rank, local, world)
    dev = torch.device("cuda:0")
    net = detector(dev)
    set_dtype(net, args.dtype, dev)
    names = ["p%dc%d" % (p, c) for p in (1, 2, 3) for c in range(1, 7)][:args.cams]
    me = tracker(dev, names)
    g = torch.Generator().manual_seed(7)
    host = torch.randint(0, 256, (args.cams, H, W, 3), generator=g, dtype=torch.uint8).pin_memory()   # the decoded, resized frames
    nb = (args.cams + args.batch - 1) // args.batch
    spans = [(k * args.batch, min(args.cams, (k + 1) * args.batch)) for k in range(nb)]
    compute = torch.cuda.current_stream()
    copy = torch.cuda.Stream()
    bufs = [torch.empty((args.batch, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]

    def upload(k, stream):
        a, b = spans[k]
        with torch.cuda.stream(stream):
            stream.wait_event(free[k % 2])                   # the detector call that last read this buffer is done
            bufs[k % 2][:b - a].copy_(host[a:b], non_blocking=True)
            copied[k % 2].record(stream)

    def detect(k):
        a, b = spans[k]
        compute.wait_event(copied[k % 2])
        s, c, bx, im = net(bufs[k % 2][:b - a], MULTI_FRAME=True)
        free[k % 2].record(compute)
        if s.numel() == 0:
            return 0, 0
        st, lb, sc, cm = me.parse_detections(s, c, bx, im + a)                # camera index = global camera
        if not isinstance(st, torch.Tensor):
            return int(s.numel()), 0
        me.hg.state_to_im(st, name=[names[i] for i in cm.cpu().tolist()])     # what the tracker plots / crops from
        return int(s.numel()), int(st.shape[0])

    def time_step(pipelined):
        for e in free:
            e.record(compute)
        kept = parsed = 0
        if pipelined:
            upload(0, copy)
        for k in range(nb):
            if pipelined:
                if k + 1 < nb:
                    upload(k + 1, copy)                      # runs under detect(k)
            else:
                upload(k, compute)
            a, b = detect(k)
            kept, parsed = kept + a, parsed + b
        return kept, parsed

    label = args.dtype if args.dtype != "fp32" else "fp32, %s products" % cv.get_fp32_mfma()
    print("%d cameras, %d per call, ResNet-50 (%s), %dx%d uint8 frames from pinned host memory" % (args.cams, args.batch, label, W, H))
    for label, pipelined in (("one stream (upload, then detect)", False), ("two streams (upload k+1 under detect k)", True)):
        time_step(pipelined)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.iters):
            kept, parsed = time_step(pipelined)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.iters
        print("%-42s %7.2f ms per time step  %6.1f frames/s  (%d detections -> %d parsed objects)"
              % (label, dt * 1e3, args.cams / dt, kept, parsed), flush=True)

    frames = torch.randn(1, 3, H, W, device=dev)              # perform_3D_detection_on_video_sequences.py: one normalised frame
    net(frames)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        s, c, b = net(frames)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    print("%-42s %7.2f ms per frame      %6.1f frames/s  (%d detections kept)" % ("single frame, per-class NMS (noise frame: stress)", dt * 1e3, 1 / dt, s.numel()))
    # The same path on a frame a detector would see: a flat road with a few dozen vehicle-sized bright rectangles.  Random weights
    # then score the anchors around the rectangles apart from the background, the 10 000 candidates per class crowd there and
    # NMS leaves what a real frame leaves -- hundreds, not the ~25 000 survivors of pure noise (which time the NMS, not the path).
    gen = torch.Generator().manual_seed(11)
    scene = torch.full((1, 3, H, W), -1.0)
    for _ in range(40):
        h_, w_ = int(torch.randint(40, 140, (1,), generator=gen)), int(torch.randint(60, 220, (1,), generator=gen))
        y0, x0 = int(torch.randint(300, H - h_, (1,), generator=gen)), int(torch.randint(0, W - w_, (1,), generator=gen))
        scene[0, :, y0:y0 + h_, x0:x0 + w_] = torch.rand(3, 1, 1, generator=gen) * 2.0 + 0.5
    scene = scene.to(dev)
    net(scene)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        s, c, b = net(scene)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    print("%-42s %7.2f ms per frame      %6.1f frames/s  (%d detections kept)" % ("single frame, per-class NMS (40-vehicle scene)", dt * 1e3, 1 / dt, s.numel()))


if __name__ == "__main__":
    main()
