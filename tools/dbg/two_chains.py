#!/usr/bin/env python3
"""What would two half-batch chains be worth?  An upper-bound experiment before touching the engine: TWO networks (own weights, own optimizer)
each train on 4 images, their steps enqueued one after the other on two plain streams (RN_TOWER_STREAMS=0 RN_WGRAD_STREAMS=0: one stream
per chain, two in all) -- every kernel of one chain can run beside any kernel of the other.  Against the same network on 8 images
(a) with everything on one stream and (b) as shipped (towers + weight-gradient stream).  The two-network form does the optimizer and the
weight preparation twice (~1.5 ms per step more work than a real two-chain step would)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
os.environ.setdefault("RN_DEFERRED_LABEL_CHECK", "1")
from retinanet_mi355x import modules, optim, synth  # noqa: E402


def make(dev, B, seed):
    net = modules.resnet50(num_classes=8)
    net.load_state_dict(synth.state_dict("resnet50", 8, 12, seed=2))
    net = net.to(dev)
    net.train()
    net.freeze_bn()
    net.use_flat_gradients()
    img = synth.frames(B, 1080, 1920, seed=seed).to(dev)
    ann = synth.labels_dir(B, 10, 1080, 1920, 8, seed=1 + seed).to(dev)
    opt = optim.ClipAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, max_norm=0.1)
    return net, opt, img, ann


def step(net, opt, img, ann):
    opt.zero_grad(set_to_none=True)
    loss = sum(l.mean() for l in net([img, ann]))
    loss.backward()
    opt.step()


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


def main():
    dev = torch.device("cuda:0")
    mode = sys.argv[1] if len(sys.argv) > 1 else "two"
    if mode == "one":
        a = make(dev, 8, 0)
        t = timeit(lambda: step(*a))
        print("one network, 8 images: %.2f ms per step, %.1f images/s  (RN_TOWER_STREAMS=%s RN_WGRAD_STREAMS=%s)"
              % (1e3 * t, 8 / t, os.environ.get("RN_TOWER_STREAMS", "1"), os.environ.get("RN_WGRAD_STREAMS", "1")))
        return
    a, b = make(dev, 4, 0), make(dev, 4, 1)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def both():
        with torch.cuda.stream(sa):
            step(*a)
        with torch.cuda.stream(sb):
            step(*b)
    t = timeit(both)
    print("two networks, 4 images each, two streams: %.2f ms per pair of steps, %.1f images/s  (RN_TOWER_STREAMS=%s RN_WGRAD_STREAMS=%s)"
          % (1e3 * t, 8 / t, os.environ.get("RN_TOWER_STREAMS", "1"), os.environ.get("RN_WGRAD_STREAMS", "1")))
    t1 = timeit(lambda: step(*a))
    print("one of them alone, 4 images: %.2f ms per step, %.1f images/s" % (1e3 * t1, 4 / t1))


if __name__ == "__main__":
    main()
