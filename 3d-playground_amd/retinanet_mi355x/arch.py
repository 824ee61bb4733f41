"""Static description of the detector: layer list, parameter names and shapes.

The names are the reference's ``state_dict`` keys (torchvision-style ResNet +
``fpn.*`` + ``regressionModel.*`` / ``classificationModel.*``; SURVEY.md 8b,
D/model.py:208-240) so a checkpoint of the reference loads unchanged.

A ``ConvSpec`` is one convolution with everything the HIP engine fuses into it:
the frozen batch-norm that follows (D/model.py:278-282), the residual add and the
activation.  ``build_plan`` returns the convolutions in execution order.
"""
from collections import OrderedDict, namedtuple

LAYERS = {                                     # D/model.py:401-453
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}
NUM_ANCHORS = 9
FEATURE = 256
BN_EPS = 1e-5

# name: state_dict prefix of the conv; bn: prefix of its batch-norm or None; bias: conv has a bias
ConvSpec = namedtuple("ConvSpec", "name cin cout k stride pad bn bias")


def backbone_convs(arch):
    """[(ConvSpec, role)] for stem + layer1..4 in forward order; role in
    {'stem','conv1','conv2','conv3','down'} plus the block prefix."""
    kind, counts = LAYERS[arch]
    exp = 1 if kind == "basic" else 4
    out = [(ConvSpec("conv1", 3, 64, 7, 2, 3, "bn1", False), "stem", None)]
    inplanes = 64
    for li, n in enumerate(counts, start=1):
        planes = 64 * 2 ** (li - 1)
        for b in range(n):
            stride = 2 if (li > 1 and b == 0) else 1
            pre = "layer%d.%d" % (li, b)
            need_down = b == 0 and (stride != 1 or inplanes != planes * exp)
            if kind == "basic":
                out.append((ConvSpec(pre + ".conv1", inplanes, planes, 3, stride, 1, pre + ".bn1", False), "conv1", pre))
                out.append((ConvSpec(pre + ".conv2", planes, planes, 3, 1, 1, pre + ".bn2", False), "conv2", pre))
            else:
                out.append((ConvSpec(pre + ".conv1", inplanes, planes, 1, 1, 0, pre + ".bn1", False), "conv1", pre))
                out.append((ConvSpec(pre + ".conv2", planes, planes, 3, stride, 1, pre + ".bn2", False), "conv2", pre))
                out.append((ConvSpec(pre + ".conv3", planes, planes * 4, 1, 1, 0, pre + ".bn3", False), "conv3", pre))
            if need_down:
                out.append((ConvSpec(pre + ".downsample.0", inplanes, planes * exp, 1, stride, 0,
                                     pre + ".downsample.1", False), "down", pre))
            inplanes = planes * exp
    return out


def fpn_sizes(arch):
    kind, _ = LAYERS[arch]
    exp = 1 if kind == "basic" else 4
    return 128 * exp, 256 * exp, 512 * exp


def fpn_convs(arch):
    c3, c4, c5 = fpn_sizes(arch)
    f = FEATURE
    return [ConvSpec("fpn.P5_1", c5, f, 1, 1, 0, None, True), ConvSpec("fpn.P5_2", f, f, 3, 1, 1, None, True),
            ConvSpec("fpn.P4_1", c4, f, 1, 1, 0, None, True), ConvSpec("fpn.P4_2", f, f, 3, 1, 1, None, True),
            ConvSpec("fpn.P3_1", c3, f, 1, 1, 0, None, True), ConvSpec("fpn.P3_2", f, f, 3, 1, 1, None, True),
            ConvSpec("fpn.P6", c5, f, 3, 2, 1, None, True), ConvSpec("fpn.P7_2", f, f, 3, 2, 1, None, True)]


def head_convs(prefix, n_out):
    f = FEATURE
    return [ConvSpec("%s.conv%d" % (prefix, i), f, f, 3, 1, 1, None, True) for i in range(1, 5)] + \
           [ConvSpec(prefix + ".output", f, NUM_ANCHORS * n_out, 3, 1, 1, None, True)]


def all_convs(arch, num_classes, n_reg):
    return [c for c, _, _ in backbone_convs(arch)] + fpn_convs(arch) + \
        head_convs("regressionModel", n_reg) + head_convs("classificationModel", num_classes)


def state_dict_shapes(arch, num_classes, n_reg=12):
    """OrderedDict key -> shape, in the order torch's ``state_dict()`` of the reference model yields
    (module registration order: conv1, bn1, layer1..4, fpn, regressionModel, classificationModel)."""
    sd = OrderedDict()

    def bn(pre, c):
        sd[pre + ".weight"] = (c,)
        sd[pre + ".bias"] = (c,)
        sd[pre + ".running_mean"] = (c,)
        sd[pre + ".running_var"] = (c,)
        sd[pre + ".num_batches_tracked"] = ()

    def conv(c):
        sd[c.name + ".weight"] = (c.cout, c.cin, c.k, c.k)
        if c.bias:
            sd[c.name + ".bias"] = (c.cout,)

    # backbone: within a block torch registers conv1,bn1,conv2,bn2,(conv3,bn3),downsample
    blocks = OrderedDict()
    for c, role, pre in backbone_convs(arch):
        blocks.setdefault(pre, []).append(c)
    for pre, convs in blocks.items():
        for c in convs:
            conv(c)
            bn(c.bn, c.cout)
    for c in fpn_convs(arch) + head_convs("regressionModel", n_reg) + head_convs("classificationModel", num_classes):
        conv(c)
    return sd


def pyramid_shapes(height, width):
    """Feature-map sizes of P3..P7 as the network produces them (conv arithmetic, not the anchors' ceil):
    stem s2 p3 k7, maxpool s2 p1 k3, then three s2 p1 k3 stages, then P6/P7 s2 p1 k3.  For every size the
    result equals ceil(H / 2^l) -- D/anchors.py:25 relies on that."""
    def down(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    h, w = down(height, 7, 2, 3), down(width, 7, 2, 3)
    h, w = down(h, 3, 2, 1), down(w, 3, 2, 1)          # maxpool -> layer1 (stride 4)
    out = []
    for _ in range(3):                                  # layer2..4 -> strides 8,16,32
        h, w = down(h, 3, 2, 1), down(w, 3, 2, 1)
        out.append((h, w))
    for _ in range(2):                                  # P6, P7
        h, w = down(h, 3, 2, 1), down(w, 3, 2, 1)
        out.append((h, w))
    return out
