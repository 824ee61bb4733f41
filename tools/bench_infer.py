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
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
import homography as hgm  # noqa: E402
import mc3d_post  # noqa: E402
from retinanet_mi355x import conv as cv, modules, synth  # noqa: E402

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", type=int, default=18)
    ap.add_argument("--batch", type=int, default=3, help="cameras per detector call (18 cameras over 8 GPUs: 2-3 each)")
    ap.add_argument("--iters", type=int, default=5, help="time steps (one frame from every camera each)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"], help="bf16: DESIGN.md 4.5 (opt-in, not the reference's arithmetic)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    net = detector(dev)
    if args.dtype == "bf16":
        net.set_compute_dtype("bf16")
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
