"""Same names as the reference's D/model.py: ResNet, resnet18..152, PyramidFeatures, RegressionModel,
ClassificationModel, batched_nms, nms."""
import torch

from retinanet_mi355x import ops as _ops
from retinanet_mi355x.modules import (BasicBlock, Bottleneck, ClassificationModel, PyramidFeatures,  # noqa: F401
                                      RegressionModel, ResNet, resnet18, resnet34, resnet50, resnet101, resnet152)
from retinanet.anchors import Anchors  # noqa: F401
from retinanet.utils import BBoxTransform, ClipBoxes  # noqa: F401
from retinanet import losses  # noqa: F401


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms contract on device (the reference imports it at D/model.py:5): int64 keep indices,
    decreasing score."""
    return _ops.nms(boxes, scores, iou_threshold)


def batched_nms(boxes, scores, idxs, iou_threshold):
    """D/model.py:19-57."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    return _ops.nms(boxes, scores, iou_threshold, idxs)
