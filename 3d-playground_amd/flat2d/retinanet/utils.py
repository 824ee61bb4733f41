"""Same names as the reference's R/utils.py: BasicBlock, Bottleneck, BBoxTransform (std .1,.1,.2,.2), ClipBoxes."""
from retinanet_mi355x import modules as _m
from retinanet_mi355x.modules import BasicBlock, Bottleneck, ClipBoxes  # noqa: F401


class BBoxTransform(_m.BBoxTransform):
    def __init__(self, mean=None, std=None):
        super().__init__(mean, std, directional=False)
