#!/usr/bin/env python3
"""Where the fp8 forward's end-to-end error comes from: BASELINE configs[4]'s network (ResNet-101 FPN) at
1920x1080 with ONE group of layers at a time in e4m3 and everything else in fp32, against the all-fp32 forward of the same engine on the
same GPU (which the parity tests hold within 1e-4 of the oracle).  Scales calibrated on other frames (margin 1.25), as in
tests/test_gpu_full_size_lowp.py.  Also: all groups in fp8 (the shipping configuration), and all-but-one.

  python tools/fp8_error_budget.py [--batch 2] [--arch resnet101] [--other fp32|bf16]
"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
from retinanet_mi355x import modules, synth  # noqa: E402

GROUPS = [
    ("layer1", lambda n: n.startswith("layer1.")),
    ("layer2", lambda n: n.startswith("layer2.")),
    ("layer3", lambda n: n.startswith("layer3.")),
    ("layer4", lambda n: n.startswith("layer4.")),
    ("fpn", lambda n: n.startswith("fpn.")),
    ("regression tower", lambda n: n.startswith("regressionModel.") and not n.endswith(".output")),
    ("classification tower", lambda n: n.startswith("classificationModel.") and not n.endswith(".output")),
    ("head outputs (e4m3 operands, fp32 result)", lambda n: n.endswith(".output")),
]


def rel(a, b):
    d = (a.double() - b.double()).abs()
    return float(d.max() / b.abs().max()), float((d ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--other", default="fp32", choices=["fp32", "bf16"], help="format of the layers that are NOT in fp8")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H, W, B = args.height, args.width, args.batch
    net = getattr(modules, args.arch)(num_classes=8)
    net.load_state_dict(synth.state_dict(args.arch, 8, 12, seed=2))
    net = net.to(dev).eval()
    img = synth.frames(B, H, W, seed=0).to(dev)
    with torch.no_grad():
        ref_boxes, ref_cls = net(img, LOCALIZE=True)                              # the fp32 engine
        calib = torch.cat([synth.frames(1, H, W, seed=123), synth.frames(1, H, W, seed=124)]).to(dev)
        net.calibrate_fp8(calib, margin=1.25)
        eng = net._engine
        print("%s, %d x %dx%d, scales from two other frames (margin 1.25); error against the all-fp32 forward: max |err| / max |ref|, rms err / rms ref"
              % (args.arch, B, W, H))
        print("%-58s %5s   %-21s %-21s" % ("layers in e4m3 (the rest %s)" % args.other, "n", "scores", "boxes"))

        def run(label, pick):
            on = eng.set_fp8_layers(pick, other=args.other)
            boxes, cls = net(img, LOCALIZE=True)
            s, b = rel(cls, ref_cls), rel(boxes, ref_boxes)
            print("%-58s %5d   %.3e / %.3e   %.3e / %.3e" % (label, len(on), s[0], s[1], b[0], b[1]), flush=True)
            return s

        run("none (only the pooled stem output passes through e4m3)", lambda n: False)
        single = {}
        for name, pick in GROUPS:
            single[name] = run(name, pick)
        run("all (the shipping configuration)", None)
        for name, pick in GROUPS:
            run("all but: " + name, lambda n, pick=pick: not pick(n))
        # the residual stream: the last convolution of every bottleneck (conv3, and the shortcut's 1x1) writes the tensor that the NEXT
        # block adds to its own result -- as e4m3 it is re-quantised 33 times on the way through ResNet-101.  Those layers in bf16 (their
        # result stays bf16; the 1x1 / 3x3 convolutions inside the blocks keep e4m3 operands and results):
        stream = lambda n: n.endswith(".conv3") or ".downsample." in n
        other = args.other
        args.other = "bf16"
        run("all, residual stream (conv3 + shortcut) of layer3 in bf16", lambda n: not (n.startswith("layer3.") and stream(n)))
        run("all, residual stream of every layer in bf16", lambda n: not stream(n))
        run("all, residual stream + fpn in bf16", lambda n: not (stream(n) or n.startswith("fpn.")))
        run("all, residual stream + fpn + classification tower in bf16",
            lambda n: not (stream(n) or n.startswith("fpn.") or (n.startswith("classificationModel.") and not n.endswith(".output"))))
        run("all, layer3 + fpn in bf16", lambda n: not (n.startswith("layer3.") or n.startswith("fpn.")))
        run("all, residual stream + fpn + classification conv4 in bf16",
            lambda n: not (stream(n) or n.startswith("fpn.") or n == "classificationModel.conv4"))
        run("all, residual stream + fpn + classification conv3, conv4 in bf16",
            lambda n: not (stream(n) or n.startswith("fpn.") or n in ("classificationModel.conv3", "classificationModel.conv4")))
        args.other = other
        eng.set_fp8_layers(None)
        eng._fp8_explicit = None                                                   # back to the engine's own policy
        for pol in ("stream_bf16", "all"):
            eng.fp8_policy = pol
            boxes, cls = net(img, LOCALIZE=True)
            s_, b_ = rel(cls, ref_cls), rel(boxes, ref_boxes)
            print("%-58s %5d   %.3e / %.3e   %.3e / %.3e" % ("engine policy RN_FP8_POLICY=%s%s" % (pol, " (default)" if pol == "stream_bf16" else ""),
                                                               sum(L.mode() == "fp8" for L in eng.layers.values()), s_[0], s_[1], b_[0], b_[1]))
        eng.fp8_policy = "stream_bf16"
    worst = max(single, key=lambda k: single[k][1])
    print("most sensitive single group by rms: %s" % worst)


if __name__ == "__main__":
    main()
