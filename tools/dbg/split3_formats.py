"""Which weight form every convolution launch of a training step takes in split3 mode (debug aid)."""
import collections
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import conv as cv, modules, synth  # noqa: E402

cv.set_fp32_mfma("split3")
dev = torch.device("cuda:0")
net = modules.resnet50(num_classes=8)
net.load_state_dict(synth.state_dict("resnet50", 8, 12, seed=2))
net = net.to(dev)
net.train()
net.freeze_bn()
B, H, W = 2, 256, 384
img = synth.frames(B, H, W, seed=0).to(dev)
ann = synth.labels_dir(B, 10, H, W, 8, seed=1).to(dev)
count = collections.Counter()
orig = cv._w_operand


def spy(w_packed, d=None):
    r = orig(w_packed, d)
    lib = cv._hip.load()
    import ctypes
    wants = lib.rn_conv_igemm_wants_f16(ctypes.byref(d)) if d is not None else -1
    count[(r[1], wants, hasattr(w_packed, "_rn_split"), hasattr(w_packed, "_rn_split16"),
           None if d is None else (d.Cin, d.Cout, d.kh, d.in_relu, d.div_shift, int(d.w_batch_stride != 0)))] += 1
    return r


cv._w_operand = spy
l = net([img, ann])
(l[0].mean() + l[1].mean() + l[2].mean()).backward()
torch.cuda.synchronize()
for k, v in sorted(count.items(), key=lambda kv: -kv[1]):
    print(v, k)
