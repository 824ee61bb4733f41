"""Diagnostic: where does the 3.8e-4 error of fpn.P4_2.bias (ResNet-18, batch 2, 72x104, Winograd on) come from?
Captures the gradient tensor the layer's weight-gradient pass receives in both modes and separates the error of that
tensor from the error of the column sum taken from it."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "3d-playground_amd"), os.path.join(REPO, "tests")]
import golden_cases as gc                                  # noqa: E402
from retinanet_mi355x import modules                       # noqa: E402

dev = torch.device("cuda:0")
z = np.load(os.path.join(REPO, "tests", "golden", "model.npz"))


def run(arch, wino, batch=None):
    fn, sd, img, ann = gc.model_case(arch, True)
    if batch is not None:
        img, ann = img[:batch], ann[:batch]
    net = getattr(modules, arch)(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    net._engine.use_wino = wino
    cap = {}
    for lname in ("fpn.P4_2", "fpn.P3_2", "fpn.P5_2"):
        L = net._engine.layers[lname]
        orig = L.bwd_params

        def hook(g, x, in_relu=False, _o=orig, _n=lname):
            cap[_n] = g.detach().clone()
            return _o(g, x, in_relu=in_relu)
        L.bwd_params = hook
    losses = net([img.to(dev), ann.to(dev)])
    sum(l.sum() for l in losses).backward()
    grads = {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()}
    return cap, grads


for arch, batch in (("resnet18", None), ("resnet18", 1), ("resnet50", None)):
    cw, gw = run(arch, True, batch)
    cd, gd = run(arch, False, batch)
    print("==", arch, "batch", batch)
    for lname in ("fpn.P3_2", "fpn.P4_2", "fpn.P5_2"):
        a, b = cw[lname].double().cpu(), cd[lname].double().cpu()
        e_t = float((a - b).norm() / b.norm())
        sa, sb = a.sum(dim=(0, 1, 2)), b.sum(dim=(0, 1, 2))
        cancel = float(sb.abs().sum() / b.abs().sum(dim=(0, 1, 2)).sum())
        e_s = float((sa - sb).norm() / sb.norm())
        e_gw = float((gw[lname + ".bias"] - sa).norm() / sa.norm())
        e_gd = float((gd[lname + ".bias"] - sb).norm() / sb.norm())
        key = "%s_dir_g_%s.bias" % (arch, lname)
        ref = torch.from_numpy(z[key]).double() if (batch is None and key in z.files) else None
        e_ref = [float((g[lname + ".bias"] - ref).norm() / ref.norm()) for g in (gw, gd)] if ref is not None else None
        print("%-9s shape %-18s dY err(wino vs direct) %.2e | |sum|/sum|.| %.3f | fp64 colsum err %.2e | engine colsum vs fp64 "
              "sum of its own dY: wino %.2e direct %.2e | vs reference golden (wino, direct) %s"
              % (lname, tuple(a.shape), e_t, cancel, e_s, e_gw, e_gd, e_ref))


# ---- are the outliers ReLU-mask flips?  Tower activations of both modes, per level: sign differences and how close to zero
def tower_acts(arch, wino):
    fn, sd, img, ann = gc.model_case(arch, True)
    net = getattr(modules, arch)(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    eng = net._engine
    eng.use_wino = wino
    reg, cls, S = eng.forward(net._tensor_dict(), img.to(dev), save=True)
    return S["towers"]


tw, td = tower_acts("resnet18", True), tower_acts("resnet18", False)
for prefix in ("regressionModel", "classificationModel"):
    for li in range(5):
        for i in range(4):
            a, b = tw[prefix][li][i], td[prefix][li][i]
            flips = (a > 0) != (b > 0)
            n = int(flips.sum())
            if n:
                mag = torch.maximum(a, b)[flips]
                print("%s level %d conv%d: %d of %d masks differ; activation there <= %.2e (max activation %.2e)"
                      % (prefix, li, i + 1, n, a.numel(), float(mag.max()), float(b.max())))
