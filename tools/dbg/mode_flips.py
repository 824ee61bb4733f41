"""Diagnostic: are the gradient differences between the two fp32 product modes (native fp32 MFMA / split bf16x3 MFMA) ReLU-mask
flips or arithmetic error?  Small golden cases (tests/golden/model.npz): per mode the gradient errors against the
reference golden, the mode-to-mode difference, and for every saved activation of the forward pass the elements whose zero /
non-zero pattern differs between the modes with their magnitude relative to the tensor's maximum."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "3d-playground_amd"), os.path.join(REPO, "tests")]
import golden_cases as gc                                  # noqa: E402
from retinanet_mi355x import conv as cv, modules            # noqa: E402

dev = torch.device("cuda:0")
z = np.load(os.path.join(REPO, "tests", "golden", "model.npz"))


def walk(o, path, out):
    if isinstance(o, torch.Tensor):
        if o.dtype == torch.float32 and o.numel() > 0:
            out[path] = o
    elif isinstance(o, dict):
        for k, v in o.items():
            walk(v, "%s/%s" % (path, k), out)
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o):
            walk(v, "%s/%d" % (path, i), out)


def run(arch, wino, mode):
    cv.set_fp32_mfma(mode)
    fn, sd, img, ann = gc.model_case(arch, True)
    net = getattr(modules, arch)(num_classes=4)
    net.load_state_dict(sd)
    net = net.to(dev).train()
    net.freeze_bn()
    net._engine.use_wino = wino
    reg, cls, S = net._engine.forward(net._tensor_dict(), img.to(dev), save=True)
    acts = {}
    walk(S, "S", acts)
    acts = {k: v.detach().clone() for k, v in acts.items()}
    losses = net([img.to(dev), ann.to(dev)])
    sum(l.sum() for l in losses).backward()
    grads = {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()}
    return [float(l) for l in losses], grads, acts


ARCHS = sys.argv[1:] or ["resnet18", "resnet50"]
for arch in ARCHS:
    z = np.load(os.path.join(REPO, "tests", "golden", gc.MODEL_CASES[arch][0] + ".npz"))
    for wino in (False, True):
        ln, gn, an = run(arch, wino, "native")
        ls, gs, as_ = run(arch, wino, "split")
        print("== %s  winograd %s   losses native %s split %s golden %s" % (arch, wino, ln, ls, z["%s_dir_losses" % arch]))
        rows = []
        for name in gn:
            key = "%s_dir_g_%s" % (arch, name)
            if key not in z.files:
                continue
            ref = torch.from_numpy(z[key]).double()
            en = float((gn[name] - ref).norm() / ref.norm())
            es = float((gs[name] - ref).norm() / ref.norm())
            d = float((gn[name] - gs[name]).norm() / ref.norm())
            rows.append((name, en, es, d))
        for name, en, es, d in rows:
            if max(en, es) > 3e-5:
                print("   %-44s vs golden: native %.2e split %.2e   native vs split %.2e" % (name, en, es, d))
        print("   tensors within 3e-5 of the golden: native %d, split %d of %d" % (sum(r[1] <= 3e-5 for r in rows), sum(r[2] <= 3e-5 for r in rows), len(rows)))
        for k in an:
            if k not in as_ or an[k].shape != as_[k].shape or "wino_v" in k:
                continue
            a, b = an[k], as_[k]
            flips = (a > 0) != (b > 0)
            n = int(flips.sum())
            if n:
                mag = float(torch.maximum(a.abs(), b.abs())[flips].max() / a.abs().max())
                print("   %-40s %d element(s) zero in one mode only; largest of them %.1e of the tensor's max; tensor diff %.1e" % (
                    k, n, mag, float((a - b).abs().max() / a.abs().max())))
cv.set_fp32_mfma("native")
