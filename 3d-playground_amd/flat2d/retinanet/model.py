"""Same names as the reference's R/model.py (regression head 9*4, two losses, ClipBoxes, 0.05 threshold)."""
import functools

from retinanet_mi355x import modules as _m
from retinanet_mi355x.modules import BasicBlock, Bottleneck, ClassificationModel, PyramidFeatures  # noqa: F401
from retinanet.anchors import Anchors  # noqa: F401
from retinanet.utils import BBoxTransform, ClipBoxes  # noqa: F401
from retinanet import losses  # noqa: F401


class RegressionModel(_m.RegressionModel):            # R/model.py:80-96
    def __init__(self, num_features_in, num_anchors=9, feature_size=256):
        super().__init__(num_features_in, num_anchors, feature_size, n_outputs=4)


class ResNet(_m.ResNet):                              # R/model.py:167
    def __init__(self, num_classes, block, layers):
        super().__init__(num_classes, block, layers, directional=False)


resnet18 = functools.partial(_m.resnet18, directional=False)
resnet34 = functools.partial(_m.resnet34, directional=False)
resnet50 = functools.partial(_m.resnet50, directional=False)
resnet101 = functools.partial(_m.resnet101, directional=False)
resnet152 = functools.partial(_m.resnet152, directional=False)
