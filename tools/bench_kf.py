#!/usr/bin/env python3
"""Tracker Kalman filter: the drop-in Torch_KF on the GPU against the same algebra in torch CPU ops
(oracle/kf.py = what the reference's class runs; the tracker keeps its filter on the CPU).  Microseconds per call.
    python tools/bench_kf.py"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
import golden_cases as gc              # noqa: E402
from oracle import kf as okf           # noqa: E402
from util_track.kf import Torch_KF     # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for n in (40, 300, 2000):
        INIT, det, directions, times, speed, upd_ids, z, dts = gc.kf_inputs(n=n)
        kf = Torch_KF(dev, INIT=INIT, ADD_MEAN_R=True)
        ids = list(range(n))
        kf.add(det, ids, directions, times)
        upd = [ids[i] for i in upd_ids]
        zd, dtd = z.to(dev), dts.to(dev)

        def t_gpu(fn, it=50):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(it):
                fn()
            torch.cuda.synchronize()
            return (time.time() - t0) / it * 1e6
        g_pred = t_gpu(lambda: kf.predict(dt=dtd))
        g_view = t_gpu(lambda: kf.view(dt=dtd, with_direction=True))
        g_upd = t_gpu(lambda: kf.update(zd, upd))
        X, P, T = kf.X.cpu(), kf.P.cpu(), kf.T.cpu()
        F, H, Q, R, mu = (getattr(kf, k).cpu() for k in ("F", "H", "Q", "R", "mu_R"))

        def t_cpu(fn, it=20):
            fn()
            t0 = time.time()
            for _ in range(it):
                fn()
            return (time.time() - t0) / it * 1e6
        c_pred = t_cpu(lambda: okf.predict(X, P, directions, T, F, Q, dts))
        c_view = t_cpu(lambda: okf.view(X, directions, F, dts, True))
        c_upd = t_cpu(lambda: okf.update(X, P, upd_ids, z, H, R, mu))
        print("n = %4d objects   predict %6.1f us (CPU %7.1f)   view %6.1f us (CPU %7.1f)   update of %4d %6.1f us (CPU %8.1f)   [wall time per call incl. Python]"
              % (n, g_pred, c_pred, g_view, c_view, len(upd), g_upd, c_upd), flush=True)


if __name__ == "__main__":
    main()
