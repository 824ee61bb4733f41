"""Same names as the reference's R/losses.py: calc_iou, FocalLoss (2D: two losses)."""
from retinanet_mi355x import modules as _m
from retinanet_mi355x.modules import calc_iou  # noqa: F401


class FocalLoss(_m.FocalLoss):
    def __init__(self):
        super().__init__(directional=False)
