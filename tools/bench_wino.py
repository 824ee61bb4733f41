"""Head-tower convolution (3x3, 256 -> 256, bias + ReLU) on the five pyramid levels of a 1080p batch of 8: the direct
grouped implicit-GEMM launch against the Winograd F(4x4,3x3) path (input transform, one batched GEMM, output transform);
forward and data gradient.   python tools/bench_wino.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "3d-playground_amd"))
import torch
from retinanet_mi355x import conv as cv, prof

dev = torch.device("cuda:0")
B, C = 8, 256
levels = [(135, 240), (68, 120), (34, 60), (17, 30), (9, 15)]
xs = [torch.randn(B, h, w, C, device=dev) for h, w in levels]
w = torch.randn(C, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
bias = torch.randn(C, device=dev) * 0.1
flops = sum(2.0 * B * h * w_ * C * C * 9 for h, w_ in levels)


def timeit(run, iters=10):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters


wp = cv.pack_weights(w, 0)
wd = cv.pack_weights(w, 1)
U = cv.wino_weights(w, 0)
Ud = cv.wino_weights(w, 1)
geo = lambda x: (x.shape[1], x.shape[2], C, 3, 3, 1, 1, -1, 0)
outs = [torch.empty_like(x) for x in xs]


def direct_fwd():
    cv.conv_igemm_grouped([{"x": x, "y": y, "geom": geo(x)} for x, y in zip(xs, outs)], wp, shift=bias, act=cv.ACT_RELU)


def wino_fwd():
    cv.wino_conv_group(xs, U, outs=outs, shift=bias, act=cv.ACT_RELU)


def direct_bwd():
    cv.conv_igemm_grouped([{"x": x, "y": y, "geom": (x.shape[1], x.shape[2], C, 3, 3, 1, -1, 1, 0), "mask": x}
                           for x, y in zip(xs, outs)], wd)


def wino_bwd():
    cv.wino_conv_group(xs, Ud, outs=outs, masks=xs, mask_mode=2)


for name, d, wi in (("forward (bias + ReLU)", direct_fwd, wino_fwd), ("data gradient (ReLU mask)", direct_bwd, wino_bwd)):
    td, tw = timeit(d), timeit(wi)
    print("%-28s direct %.3f ms (%.1f TF)   winograd %.3f ms (%.1f TF algorithmic)   x%.2f" %
          (name, td * 1e3, flops / td / 1e12, tw * 1e3, flops / tw / 1e12, td / tw))
# stage times of the Winograd path
prof.ACTIVE = prof.KernelTimer()
wino_fwd()
for k, v in prof.ACTIVE.summary().items():
    print("   %-16s %d launches  %.3f ms" % (k, v["launches"], v["ms_total"]))

# weight gradient: direct (one launch per level) against Winograd (transforms + one batched launch)
gs = [torch.randn_like(x) for x in xs]
dw = torch.zeros_like(wp)
cs = torch.zeros(C, device=dev)


def direct_wgrad():
    for g, x in zip(gs, xs):
        cv.wgrad(g, x, dw, C, 3, 1, 1, colsum=cs)


def wino_wgrad():
    cv.wino_wgrad_group(gs, xs, dw, cs)


td, tw = timeit(direct_wgrad), timeit(wino_wgrad)
print("%-28s direct %.3f ms (%.1f TF)   winograd %.3f ms (%.1f TF algorithmic)   x%.2f" %
      ("weight gradient", td * 1e3, flops / td / 1e12, tw * 1e3, flops / tw / 1e12, td / tw))
prof.ACTIVE = prof.KernelTimer()
wino_wgrad()
for k, v in prof.ACTIVE.summary().items():
    print("   %-16s %d launches  %.3f ms" % (k, v["launches"], v["ms_total"]))
