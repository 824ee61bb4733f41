"""Same names as the reference's D/utils.py: BasicBlock, Bottleneck, BBoxTransform (12 -> 20), ClipBoxes."""
from retinanet_mi355x.modules import BasicBlock, BBoxTransform, Bottleneck, ClipBoxes  # noqa: F401
