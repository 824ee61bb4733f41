#!/usr/bin/env python3
"""Crop-refinement path of the tracker on device, stage by stage, for n tracked objects seen by
18 cameras at 1080p; and what the reference does with the detector's outputs -- four device->host copies
(MC3D_crop_tracker.py:1198-1202) and the post-processing in torch CPU ops (oracle/crop_refine.py restates it).
Latency-bound stages: microseconds per call.
    python tools/bench_crop.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
import golden_cases as gc                  # noqa: E402
import homography as hgm                   # noqa: E402
import mc3d_post                           # noqa: E402
from oracle import crop_refine as ocr      # noqa: E402
from retinanet_mi355x import modules, ops, synth   # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    det = modules.resnet34(num_classes=8)
    det.load_state_dict(synth.state_dict("resnet34", 8, 12, seed=2))
    det = det.to(dev).eval()
    frames = torch.randn(18, 3, 1080, 1920, device=dev)
    for n in (40, 160):
        pre_loc, cam, im_objs, names, (P, H), (P2, H2) = gc.crop_refine_inputs(n_obj=n, seed=91)

        def make_hg(Pm, Hm):
            hg = hgm.Homography(device="cuda:0")
            hg.correspondence = {k: {"P": Pm[i], "H": Hm[i], "H_inv": np.linalg.inv(Hm[i])} for i, k in enumerate(names)}
            hg.default_correspondence = names[0]
            return hg

        class Tracker(mc3d_post.DetectionParser):
            pass
        me = Tracker()
        me.b, me.cs, me.cd_max, me.W, me.cameras, me.device = 1.25, 112, 50, 0.5, list(names), dev
        me.hg = hgm.Homography_Wrapper(hg1=make_hg(P, H), hg2=make_hg(P2, H2))
        me.crop_detector = det
        H1d, H2d, P1d, P2d = mc3d_post._camera_matrices(me, dev)
        pre, camd = pre_loc.to(dev), cam.to(dev)
        im = ops.hg_to_im(pre, P1d, P2d, camd.int(), from_state=True)
        boxes, rois = ops.crop_boxes(im, camd, b=1.25)
        # keep the synthetic crops inside the frame so roi_align does real sampling work
        rois[:, 1:] = torch.tensor([400.0, 300.0, 700.0, 600.0], device=dev) + 40 * torch.rand(n, 4, device=dev)
        crops = ops.roi_align(frames, rois, (112, 112))
        with torch.no_grad():
            reg, cls = det(crops, LOCALIZE=True)
        t_geo = timeit(lambda: ops.crop_boxes(ops.hg_to_im(pre, P1d, P2d, camd.int(), from_state=True), camd, b=1.25))
        t_roi = timeit(lambda: ops.roi_align(frames, rois, (112, 112)))
        t_det = timeit(lambda: det(crops, LOCALIZE=True), 5)
        t_sel = timeit(lambda: ops.crop_select(reg, cls, boxes, camd, pre, H1d, H2d, P1d, P2d))
        t0 = time.time()
        for _ in range(3):
            c = [t.cpu() for t in (reg, cls, boxes)]
            ocr.refine_from_detections(c[0], c[1], c[2], cam, pre_loc, H, H2, P, P2)
        t_cpu = (time.time() - t0) / 3 * 1e6
        print("n = %3d objects, A = %d anchors/crop:  state->image + crop boxes %6.1f us | roi_align 18x1080p -> %dx3x112x112 %6.1f us"
              " | LOCALIZE detector (ResNet-34, batch %d @112x112) %8.1f us | select (class max, top-50, homographies, best box) %6.1f us"
              " || reference's host path for the last stage (4 copies + torch CPU, %d threads) %9.1f us"
              % (n, reg.shape[1], t_geo, n, t_roi, n, t_det, t_sel, torch.get_num_threads(), t_cpu), flush=True)


if __name__ == "__main__":
    main()
