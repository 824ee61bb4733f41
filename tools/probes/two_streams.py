"""Probe: do two independent Winograd layer chains (the two head towers) overlap when launched on two streams?  A chain = four
3x3 256->256 layers over the five pyramid levels of a 1080p batch of 8 (input transform, 36 GEMMs, output transform each).
  python tools/probes/two_streams.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-playground_amd"))
import torch
from retinanet_mi355x import conv as cv

dev = torch.device("cuda:0")
B, C = 8, 256
levels = [(135, 240), (68, 120), (34, 60), (17, 30), (9, 15)]
xs = [torch.randn(B, h, w, C, device=dev) for h, w in levels]
Us = [[cv.wino_weights(torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5, 0) for _ in range(4)] for _ in range(2)]
bias = torch.randn(C, device=dev) * 0.1


def chain(t):
    ts = xs
    for i in range(4):
        ts = cv.wino_conv_group(ts, Us[t][i], shift=bias, act=cv.ACT_RELU)
    return ts


side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def sequential():
    chain(0)
    chain(1)


def concurrent():
    ev = torch.cuda.Event()
    ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        chain(1)
        done = torch.cuda.Event()
        done.record(side)
    chain(0)
    main.wait_event(done)


for name, fn in (("one stream", sequential), ("two streams", concurrent), ("one stream", sequential), ("two streams", concurrent)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print("%-12s %.3f ms for both chains" % (name, (time.time() - t0) / 10 * 1e3))
