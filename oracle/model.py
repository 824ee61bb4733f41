"""ResNet + FPN + heads forward on CPU from a plain state_dict
(test infrastructure -- see oracle/__init__.py).

  backbone              <- ResNet stem + layer1-4      D/model.py:208-298, blocks D/utils.py:12-80
  pyramid               <- PyramidFeatures.forward     D/model.py:84-117 (crop-to-min on shape mismatch :92-108)
  head_tower            <- Regression/ClassificationModel.forward   D/model.py:139-157, :182-205
  forward_heads         <- ResNet.forward up to the cat D/model.py:284-306
  train_forward / eval  <- ResNet.forward               D/model.py:308-397 ; R/model.py:243-311

Convolution, batch-norm (always eval: frozen, D/model.py:278-282), max-pool,
nearest upsample and sigmoid are the third-party torch CPU kernels the
reference itself calls (torch is not part of the reference tree; version
unpinned there, 2.10 CPU here).  Only the wiring is restated.  Weights arrive as
a dict with the reference's own state_dict keys, so one seeded dict can be loaded
into the reference model, this oracle and the HIP engine alike.
"""
import torch
import torch.nn.functional as F

from . import anchors as oanchors
from . import boxes as oboxes
from . import losses as olosses

LAYERS = {                                     # D/model.py:401-453
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}
BN_EPS = 1e-5


def _bn(x, sd, key):
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"],
                        sd[key + ".weight"], sd[key + ".bias"], False, 0.0, BN_EPS)


def _conv(x, sd, key, stride=1, padding=0):
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride, padding)


def _tap(taps, name, t):
    """taps: optional dict the caller passes to collect every ReLU OUTPUT of the network under the name of the convolution
    that produced it (tests/test_gpu_model.py locates activations whose sign differs between this CPU run and the GPU run)."""
    if taps is not None:
        taps[name] = t.detach()
    return t


def _block(x, sd, pre, kind, stride, taps=None):
    res = x
    if kind == "basic":                                            # D/utils.py:25-43
        out = _tap(taps, pre + ".conv1", F.relu(_bn(_conv(x, sd, pre + ".conv1", stride, 1), sd, pre + ".bn1")))
        out = _bn(_conv(out, sd, pre + ".conv2", 1, 1), sd, pre + ".bn2")
        last = pre + ".conv2"
    else:                                                          # D/utils.py:60-80, stride on conv2
        out = _tap(taps, pre + ".conv1", F.relu(_bn(_conv(x, sd, pre + ".conv1"), sd, pre + ".bn1")))
        out = _tap(taps, pre + ".conv2", F.relu(_bn(_conv(out, sd, pre + ".conv2", stride, 1), sd, pre + ".bn2")))
        out = _bn(_conv(out, sd, pre + ".conv3"), sd, pre + ".bn3")
        last = pre + ".conv3"
    if (pre + ".downsample.0.weight") in sd:
        res = _bn(_conv(x, sd, pre + ".downsample.0", stride), sd, pre + ".downsample.1")
    return _tap(taps, last, F.relu(out + res))


def backbone(img, sd, arch, taps=None):
    kind, counts = LAYERS[arch]
    x = _tap(taps, "conv1", F.relu(_bn(_conv(img, sd, "conv1", 2, 3), sd, "bn1")))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, n in enumerate(counts, start=1):
        for b in range(n):
            x = _block(x, sd, "layer%d.%d" % (li, b), kind, 2 if (li > 1 and b == 0) else 1, taps)
        feats.append(x)
    return feats[1], feats[2], feats[3]


def _add_cropped(up, lat):
    """D/model.py:91-97: plain add, or both cropped to the common top-left window when shapes differ."""
    if up.shape == lat.shape:
        return up + lat
    h = min(up.shape[2], lat.shape[2])
    w = min(up.shape[3], lat.shape[3])
    return up[:, :, :h, :w] + lat[:, :, :h, :w]


def pyramid(c3, c4, c5, sd, taps=None):
    p5 = _conv(c5, sd, "fpn.P5_1")
    p5_up = F.interpolate(p5, scale_factor=2, mode="nearest")
    p5 = _conv(p5, sd, "fpn.P5_2", 1, 1)
    p4 = _add_cropped(p5_up, _conv(c4, sd, "fpn.P4_1"))
    p4_up = F.interpolate(p4, scale_factor=2, mode="nearest")
    p4 = _conv(p4, sd, "fpn.P4_2", 1, 1)
    p3 = _add_cropped(p4_up, _conv(c3, sd, "fpn.P3_1"))
    p3 = _conv(p3, sd, "fpn.P3_2", 1, 1)
    p6 = _conv(c5, sd, "fpn.P6", 2, 1)
    p7 = _conv(_tap(taps, "fpn.P6", F.relu(p6)), sd, "fpn.P7_2", 2, 1)
    return [p3, p4, p5, p6, p7]


def head_tower(x, sd, pre, width, sigmoid, taps=None, level=0):
    for i in range(1, 5):
        x = _tap(taps, "%s.conv%d@%d" % (pre, i, level), F.relu(_conv(x, sd, "%s.conv%d" % (pre, i), 1, 1)))
    x = _conv(x, sd, pre + ".output", 1, 1)
    if sigmoid:
        x = torch.sigmoid(x)
    return x.permute(0, 2, 3, 1).contiguous().view(x.shape[0], -1, width)


def forward_heads(img, sd, arch, taps=None):
    """-> (regression [B,A,n], classification [B,A,C], anchors [1,A,4])."""
    n_reg = sd["regressionModel.output.weight"].shape[0] // 9
    n_cls = sd["classificationModel.output.weight"].shape[0] // 9
    feats = pyramid(*backbone(img, sd, arch, taps), sd, taps)
    reg = torch.cat([head_tower(f, sd, "regressionModel", n_reg, False, taps, li) for li, f in enumerate(feats)], dim=1)
    cls = torch.cat([head_tower(f, sd, "classificationModel", n_cls, True, taps, li) for li, f in enumerate(feats)], dim=1)
    anc = torch.from_numpy(oanchors.anchors_for_image(img.shape[2], img.shape[3])).to(img.device)
    return reg, cls, anc


def train_forward(img, ann, sd, arch, taps=None):
    reg, cls, anc = forward_heads(img, sd, arch, taps)
    if reg.shape[2] == 12:
        return olosses.focal_loss_dir(cls, reg, anc, ann)
    return olosses.focal_loss_2d(cls, reg, anc, ann)


def eval_forward(img, sd, arch, LOCALIZE=False, MULTI_FRAME=False):
    reg, cls, anc = forward_heads(img, sd, arch)
    if reg.shape[2] == 12:
        boxes = oboxes.decode_dir(anc, reg)
        if MULTI_FRAME:
            return oboxes.postprocess_multi(cls, boxes)
        if LOCALIZE:
            return boxes, cls
        return list(oboxes.postprocess_single(cls, boxes))
    boxes = oboxes.clip_boxes(oboxes.decode_2d(anc, reg), img.shape[2], img.shape[3])
    if LOCALIZE:
        return boxes, cls
    return list(oboxes.postprocess_2d(cls, boxes))
