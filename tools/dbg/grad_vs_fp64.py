"""Parameter gradients of the engine against an fp64 run of the oracle model (the same torch program in double), both fp32
product modes and both kernel families, with the oracle's own fp32 run (= what the reference computes on a CPU) beside
them: how far is each fp32 computation from the exact gradients?   python tools/dbg/grad_vs_fp64.py [arch ...]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "3d-playground_amd"), os.path.join(REPO, "tests")]
import golden_cases as gc                                   # noqa: E402
from oracle import model as omodel                          # noqa: E402
from retinanet_mi355x import conv as cv, modules            # noqa: E402

dev = torch.device("cuda:0")
STATE = ("running_mean", "running_var", "num_batches_tracked")


def oracle_grads(arch, sd, img, ann, dtype):
    p = {k: (v.to(dtype).clone().requires_grad_(not k.endswith(STATE)) if v.is_floating_point() else v) for k, v in sd.items()}
    losses = omodel.train_forward(img.to(dtype), ann.to(dtype), p, arch)
    sum(l.sum() for l in losses).backward()
    return {k: v.grad.double() for k, v in p.items() if v.is_floating_point() and v.grad is not None}


def engine_grads(arch, sd, img, ann, wino, mode):
    cv.set_fp32_mfma(mode)
    net = getattr(modules, arch)(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    net.freeze_bn()
    net._engine.use_wino = wino
    sum(l.sum() for l in net([img.to(dev), ann.to(dev)])).backward()
    return {n: p.grad.double().cpu() for n, p in net.named_parameters()}


def summary(g, ref):
    e = sorted(float((g[k] - ref[k]).norm() / ref[k].norm()) for k in ref if k in g)
    return "median %.1e  p90 %.1e  max %.1e" % (e[len(e) // 2], e[int(len(e) * 0.9) - 1], e[-1])


for arch in sys.argv[1:] or ["resnet18", "resnet50"]:
    fn, sd, img, ann = gc.model_case(arch, True)
    ref = oracle_grads(arch, sd, img, ann, torch.float64)
    print("== %s (%d gradient tensors), L2 error per tensor against the fp64 gradients" % (arch, len(ref)))
    print("   %-34s %s" % ("oracle in fp32 (torch CPU)", summary(oracle_grads(arch, sd, img, ann, torch.float32), ref)))
    for wino in (False, True):
        for mode in ("native", "split"):
            print("   %-34s %s" % ("engine %s, %s products" % ("Winograd" if wino else "direct", mode),
                                   summary(engine_grads(arch, sd, img, ann, wino, mode), ref)))
cv.set_fp32_mfma("split")
