#!/usr/bin/env python3
"""Inference path (BASELINE configs[3] shape, one GPU's share): batched detect (MULTI_FRAME) on 1080p frames +
per-class NMS + homography state<->image for the survivors, and the single-frame path of
perform_3D_detection_on_video_sequences.py.  Prints frames/s.
  python tools/bench_infer.py [--cams 3]"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import modules, ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", type=int, default=3, help="cameras per GPU (18 cameras round-robin over 8 GPUs: 2-3)")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    net = modules.resnet50(num_classes=8)
    sd = synth.state_dict("resnet50", 8, 12, seed=2, head_scale=3e-3)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    with torch.no_grad():                                   # spread the scores so the threshold loop leaves survivors
        w = net.classificationModel.output.weight
        w.copy_((torch.rand_like(w) - 0.5) * 0.06)
        net.classificationModel.output.bias.fill_(-7.0)
    frames = torch.randn(args.cams, 3, 1080, 1920, device=dev)
    Pn, Hn = synth.camera_matrices(18, seed=5)
    P, H = torch.from_numpy(Pn).to(dev), torch.from_numpy(Hn).to(dev)

    def multi():
        s, c, b, im = net(frames, MULTI_FRAME=True)
        # the tracker's next step (MC3D_crop_tracker.py:349-364): image boxes -> state with per-object camera
        if s.numel():
            boxes = b[:, :16].reshape(-1, 8, 2).double()
            idx = im.to(torch.int32)
            st = ops.hg_from_im(boxes, torch.full((boxes.shape[0],), 5.0, device=dev), H, None, idx)
            ops.hg_to_im(st, P, None, idx)
        return s.numel()

    def single():
        s, c, b = net(frames[:1])
        return s.numel()

    for name, fn, nframes in (("MULTI_FRAME x%d" % args.cams, multi, args.cams), ("single frame", single, 1)):
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.iters):
            k = fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.iters
        print("%-18s %7.2f ms / call  %6.1f frames/s  (%d detections kept)" % (name, dt * 1e3, nframes / dt, k), flush=True)


if __name__ == "__main__":
    main()
