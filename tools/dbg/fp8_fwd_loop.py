"""fp8 forward of ResNet-101 at 16 x 1080p, a few passes (for rocprofv3 --kernel-trace --stats): RN_FP8_POLICY picks the layer formats."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import modules, synth  # noqa: E402

dev = torch.device("cuda:0")
net = modules.resnet101(num_classes=8)
net.load_state_dict(synth.state_dict("resnet101", 8, 12, seed=2))
net = net.to(dev).eval()
img = synth.frames(16, 1080, 1920, seed=0).to(dev)
net.calibrate_fp8(img[:2])
P = net._tensor_dict()
with torch.no_grad():
    for _ in range(6):
        net._engine.forward(P, img, save=False)
torch.cuda.synchronize()
