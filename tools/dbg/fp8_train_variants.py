"""Loss deviation of the fp8-forward training step from the bf16 step at 1080p (ResNet-101, batch 2) for a few choices of which layers
stay out of e4m3 (debug aid for the engine's fp8 policy)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import modules, synth  # noqa: E402

dev = torch.device("cuda:0")
H, W, B = 1080, 1920, 2
sd = synth.state_dict("resnet101", 8, 12, seed=2)
img = synth.frames(B, H, W, seed=0).to(dev)
ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)


def build():
    net = modules.resnet101(num_classes=8)
    net.load_state_dict(sd)
    return net.to(dev)


def losses(net):
    with torch.no_grad():
        return np.array([float(l) for l in net([img, ann])])


ref32 = build()
ref32.train(); ref32.freeze_bn()
l32 = losses(ref32)
del ref32
ref = build()
ref.set_compute_dtype("bf16")
ref.train(); ref.freeze_bn()
l16 = losses(ref)
del ref
torch.cuda.empty_cache()
print("fp32 losses", l32, "bf16 losses", l16, "rel", np.abs(l16 - l32) / l32)
net = build().eval()
net.calibrate_fp8(torch.cat([synth.frames(1, H, W, seed=123), synth.frames(1, H, W, seed=124)]).to(dev), margin=1.25)
net.train(); net.freeze_bn()
stream = lambda n: n.endswith(".conv3") or ".downsample." in n or n.startswith("fpn.")
tower = lambda p: (lambda n: n.startswith(p) and not n.endswith(".output"))
variants = {
    "default (stream + fpn bf16)": stream,
    "+ regression tower": lambda n: stream(n) or tower("regressionModel.")(n),
    "+ regression conv4": lambda n: stream(n) or n == "regressionModel.conv4",
    "+ both towers": lambda n: stream(n) or tower("regressionModel.")(n) or tower("classificationModel.")(n),
    "+ head outputs": lambda n: stream(n) or n.endswith(".output"),
    "+ regression tower + its output": lambda n: stream(n) or n.startswith("regressionModel."),
}
for name, keep in variants.items():
    net._engine.set_fp8_layers(lambda n, keep=keep: not keep(n), other="bf16")
    l8 = losses(net)
    print("%-36s %s  vs bf16 %s  vs fp32 %s" % (name, l8, np.round(np.abs(l8 - l16) / l16, 4), np.round(np.abs(l8 - l32) / l32, 4)))
