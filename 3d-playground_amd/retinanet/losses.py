"""Same names as the reference's D/losses.py: calc_iou, FocalLoss (directional: 3 losses)."""
from retinanet_mi355x.modules import FocalLoss, calc_iou  # noqa: F401
