#!/usr/bin/env python3
"""Detection parsing after the MULTI_FRAME detector: rn_parse_detections with the detector
output resident on the GPU, against what the reference does with the same data -- four device->host copies
(MC3D_crop_tracker.py:1078-1083) followed by parse_detections on the CPU (oracle/tracker_post.py restates it; torch
CPU ops + the greedy NMS).  Latency-bound: microseconds per call, not a roofline.
    python tools/bench_parse.py"""
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
from oracle import tracker_post as otp     # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for n_obj in (120, 2000):
        scores, labels, boxes, cams, names, (P, H), (P2, H2) = gc.tracker_post_inputs(n_obj=n_obj, seed=77)

        def make_hg(Pm, Hm):
            hg = hgm.Homography(device="cuda:0")
            hg.correspondence = {n: {"P": Pm[i], "H": Hm[i], "H_inv": np.linalg.inv(Hm[i])} for i, n in enumerate(names)}
            hg.default_correspondence = names[0]
            return hg

        class Tracker(mc3d_post.DetectionParser):
            pass
        me = Tracker()
        me.sigma_d, me.phi_nms_im, me.phi_nms_space, me.cameras, me.est_ts = 0.1, 0.3, 0.2, list(names), False
        me.hg = hgm.Homography_Wrapper(hg1=make_hg(P, H), hg2=make_hg(P2, H2))
        g = [t.to(dev) for t in (scores, labels, boxes, cams)]
        for refine in (False, True):
            for _ in range(3):
                out = me.parse_detections(*g, refine_height=refine)
            torch.cuda.synchronize()
            ts = []
            for _ in range(50):                                              # each call ends in the count read-back (a sync)
                t0 = time.time()
                out = me.parse_detections(*g, refine_height=refine)
                ts.append(time.time() - t0)
            gpu_us = float(np.median(ts)) * 1e6
            t0 = time.time()
            reps = 3
            for _ in range(reps):
                c = [t.cpu() for t in g]
                ref = otp.parse_detections(*c, H, H2, P, P2, perform_nms=True, refine_height=refine)
            cpu_us = (time.time() - t0) / reps * 1e6
            same = np.array_equal(out[3].cpu().numpy(), ref[3].numpy()) and np.array_equal(out[2].cpu().numpy(), ref[2].numpy())
            print("d = %5d -> %4d kept   refine_height=%-5s   device %8.1f us/call (median of 50)   reference path on CPU (%d threads) %10.1f us"
                  "   x%.0f   identical survivors: %s" % (len(scores), len(out[0]), refine, gpu_us, torch.get_num_threads(),
                                                          cpu_us, cpu_us / gpu_us, same), flush=True)


if __name__ == "__main__":
    main()
