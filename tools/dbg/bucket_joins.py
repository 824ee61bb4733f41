#!/usr/bin/env python3
"""Does the number of gradient buckets matter on ONE GPU?  Every bucket boundary joins the weight-gradient stream into the main one (the
bucket's unpack launch reads the accumulators).  32 MB buckets (the default: five joins per step) against one bucket (one join)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
os.environ.setdefault("RN_DEFERRED_LABEL_CHECK", "1")
from retinanet_mi355x import modules, optim, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    net = modules.resnet50(num_classes=8)
    net.load_state_dict(synth.state_dict("resnet50", 8, 12, seed=2))
    net = net.to(dev)
    net.train()
    net.freeze_bn()
    img = synth.frames(8, 1080, 1920, seed=0).to(dev)
    ann = synth.labels_dir(8, 10, 1080, 1920, 8, seed=1).to(dev)
    for mb in [int(a) for a in sys.argv[1:]] or [32, 1 << 20, 32, 1 << 20]:
        net._engine.set_flat_grads(mb << 20)
        opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = sum(l.mean() for l in net([img, ann]))
            loss.backward()
            opt.step()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t = (time.time() - t0) / 10
        print("bucket %7d MB (%d buckets): %.2f ms per step, %.1f images/s" % (mb, len(net._engine._flat["buckets"]), 1e3 * t, 8 / t), flush=True)


if __name__ == "__main__":
    main()
