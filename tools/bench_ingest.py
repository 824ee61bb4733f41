#!/usr/bin/env python3
"""Frame ingest: rn_frame_ingest at 1080p against the HBM roof, and the reference's host-side
F.to_tensor + F.normalize (restated in oracle/ingest.py) timed on this box's CPU for one frame.
Algorithmic bytes per pixel: 3 read + 16 written (NHWC4) or 12 written (NCHW).
    python tools/bench_ingest.py"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
sys.path.insert(0, REPO)
from oracle import ingest as oing          # noqa: E402
from retinanet_mi355x import ops           # noqa: E402

PEAK = 8000.0


def main():
    dev = torch.device("cuda:0")
    H, W = 1080, 1920
    for B in (1, 3, 8, 18):
        f = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device=dev)
        for nhwc4 in (True, False):
            for _ in range(3):
                ops.frame_ingest(f, nhwc4=nhwc4)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.frame_ingest(f, nhwc4=nhwc4)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            nbytes = B * H * W * (3 + (16 if nhwc4 else 12))
            gbs = nbytes / us / 1e3
            print("B = %2d  %-6s  %7.1f us  %7.1f GB/s  %4.1f%% of 8 TB/s  (%.1f MB; includes the output allocation)"
                  % (B, "NHWC4" if nhwc4 else "NCHW", us, gbs, 100 * gbs / PEAK, nbytes / 1e6), flush=True)
    f1 = torch.randint(0, 256, (1, H, W, 3), dtype=torch.uint8)
    t0 = time.time()
    for _ in range(5):
        oing.to_tensor_normalize(f1)
    print("reference path on the host (to_tensor + normalize, %d threads): %.1f ms per 1080p frame, then a 24.9 MB fp32 upload"
          % (torch.get_num_threads(), (time.time() - t0) / 5 * 1e3))


if __name__ == "__main__":
    main()
