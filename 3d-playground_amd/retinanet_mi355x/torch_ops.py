"""The HIP entry points as PyTorch custom operators: ``torch.ops.retinanet_mi355x.*`` (north_star: "exposed to Python through
PyTorch-ROCm custom ops").

Registered with ``torch.library.custom_op`` over the SAME C ABI the rest of the package binds (``_hip.py`` /
``include/retinanet_mi355x.h``): an operator is a schema + a CUDA(HIP)-device implementation that launches the library's
kernel on the current stream + a fake (meta) implementation that only derives output shapes, so the ops show up in
``torch.ops``, carry dispatcher-level schemas and device checks, are visible to ``torch.compile`` / export tracing as
opaque nodes, and ``focal_loss`` carries its hand-written backward through ``register_autograd``.  There is no CPU
kernel registered: calling an op with CPU tensors fails in the dispatcher ("no kernel for CPU"), which is this package's
no-fallback rule at the operator level.

Operator                                         reference code it stands for
  anchors(H, W, device)                          Anchors.forward                       D/anchors.py:21-40
  pairwise_iou(a, b)                             calc_iou                              D/losses.py:5-22
  focal_loss(cls, reg, anchors, ann, dir)        FocalLoss.forward (+ autograd)        D/losses.py:27-362, R/losses.py:27-177
  decode_dir(anchors, reg) / decode_2d(...)      BBoxTransform.forward                 D/utils.py:102-149, R/utils.py:102-126
  clip_boxes_(boxes, H, W)                       ClipBoxes.forward (in place)          R/utils.py:134-144
  nms(boxes, scores, thr)                        torchvision.ops.nms as the path uses it   D/model.py:383
  state_to_space / state_to_im / im_to_state     Homography transforms                 homography.py:305-320, 479-500
  frame_ingest(frames_u8, swap_rb, nhwc4)        to_tensor + normalize of the loaders  util_track/mp_loader.py:239-243

The whole-network training call stays one ``torch.autograd.Function`` (modules._NetFn): its inputs are the module's ~200
parameters and its saved state is a Python structure of activations, which is a scheduler, not an operator.
"""
from typing import Optional, Tuple

import torch

from . import ops

NS = "retinanet_mi355x"
_lib = torch.library


@_lib.custom_op(NS + "::anchors", mutates_args=(), device_types="cuda")
def anchors(height: int, width: int, device: torch.device) -> torch.Tensor:
    return ops.anchors(height, width, device)


@anchors.register_fake
def _(height, width, device):
    return torch.empty((1, ops.anchor_count(height, width), 4), dtype=torch.float32, device=device)


@_lib.custom_op(NS + "::pairwise_iou", mutates_args=(), device_types="cuda")
def pairwise_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return ops.pairwise_iou(a, b)


@pairwise_iou.register_fake
def _(a, b):
    return a.new_empty((a.shape[0], b.shape[0]), dtype=torch.float32)


# ---- focal loss: forward returns the three losses and the workspace its backward reads
@_lib.custom_op(NS + "::focal_loss_fwd", mutates_args=(), device_types="cuda")
def focal_loss_fwd(cls: torch.Tensor, reg: torch.Tensor, anchor_boxes: torch.Tensor, ann: torch.Tensor,
                   directional: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    return ops.focal_loss_forward_raw(cls, reg, anchor_boxes, ann, directional)


@focal_loss_fwd.register_fake
def _(cls, reg, anchor_boxes, ann, directional):
    return cls.new_empty(3, dtype=torch.float32), cls.new_empty(ops.focal_workspace_bytes(cls.shape[0], cls.shape[1]), dtype=torch.uint8)


@_lib.custom_op(NS + "::focal_loss_bwd", mutates_args=(), device_types="cuda")
def focal_loss_bwd(cls: torch.Tensor, reg: torch.Tensor, anchor_boxes: torch.Tensor, ann: torch.Tensor, directional: bool,
                   ws: torch.Tensor, grad_losses: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return ops.focal_loss_backward_raw(cls, reg, anchor_boxes, ann, directional, ws, grad_losses)


@focal_loss_bwd.register_fake
def _(cls, reg, anchor_boxes, ann, directional, ws, grad_losses):
    return torch.empty_like(cls), torch.empty_like(reg)


def _focal_setup(ctx, inputs, output):
    cls, reg, anchor_boxes, ann, directional = inputs
    ctx.save_for_backward(cls, reg, anchor_boxes, ann, output[1])
    ctx.directional = directional


def _focal_backward(ctx, g_losses, g_ws):
    cls, reg, anchor_boxes, ann, ws = ctx.saved_tensors
    dcls, dreg = torch.ops.retinanet_mi355x.focal_loss_bwd(cls, reg, anchor_boxes, ann, ctx.directional, ws, g_losses.contiguous())
    return dcls, dreg, None, None, None


focal_loss_fwd.register_autograd(_focal_backward, setup_context=_focal_setup)


def focal_loss(cls, reg, anchor_boxes, ann, directional=True):
    """(cls_loss[1], reg_loss[1], vp_loss[1]) -- or the first two for the 2D variant -- through the registered operator."""
    ops.check_labels(ann, directional, eager=True)
    losses, _ = torch.ops.retinanet_mi355x.focal_loss_fwd(cls, reg, anchor_boxes, ann, bool(directional))
    out = (losses[0:1], losses[1:2], losses[2:3])
    return out if directional else out[:2]


# ---- decode / clip / nms
@_lib.custom_op(NS + "::decode_dir", mutates_args=(), device_types="cuda")
def decode_dir(anchor_boxes: torch.Tensor, reg: torch.Tensor) -> torch.Tensor:
    return ops.decode_dir(anchor_boxes, reg)


@decode_dir.register_fake
def _(anchor_boxes, reg):
    return reg.new_empty((reg.shape[0], reg.shape[1], 20), dtype=torch.float32)


@_lib.custom_op(NS + "::decode_2d", mutates_args=(), device_types="cuda")
def decode_2d(anchor_boxes: torch.Tensor, deltas: torch.Tensor, clip: bool, height: int, width: int) -> torch.Tensor:
    return ops.decode_2d(anchor_boxes, deltas, clip_hw=(height, width) if clip else None)


@decode_2d.register_fake
def _(anchor_boxes, deltas, clip, height, width):
    return deltas.new_empty(tuple(deltas.shape), dtype=torch.float32)


@_lib.custom_op(NS + "::clip_boxes_", mutates_args=("boxes",), device_types="cuda")
def clip_boxes_(boxes: torch.Tensor, height: int, width: int) -> None:
    ops.clip_boxes_(boxes, height, width)


@_lib.custom_op(NS + "::nms", mutates_args=(), device_types="cuda")
def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    return ops.nms(boxes, scores, iou_threshold)


@nms.register_fake
def _(boxes, scores, iou_threshold):
    n = torch.library.get_ctx().new_dynamic_size()
    return boxes.new_empty((n,), dtype=torch.int64)


# ---- homography
@_lib.custom_op(NS + "::state_to_space", mutates_args=(), device_types="cuda")
def state_to_space(state: torch.Tensor) -> torch.Tensor:
    return ops.hg_state_to_space(state)


@state_to_space.register_fake
def _(state):
    return state.new_empty((state.shape[0], 8, 3), dtype=torch.float32)


@_lib.custom_op(NS + "::state_to_im", mutates_args=(), device_types="cuda")
def state_to_im(state: torch.Tensor, P: torch.Tensor, P2: Optional[torch.Tensor], mat_index: Optional[torch.Tensor]) -> torch.Tensor:
    return ops.hg_to_im(state, P, P2, mat_index, from_state=True)


@state_to_im.register_fake
def _(state, P, P2, mat_index):
    return state.new_empty((state.shape[0], 8, 2), dtype=torch.float64)


@_lib.custom_op(NS + "::im_to_state", mutates_args=(), device_types="cuda")
def im_to_state(im: torch.Tensor, heights: torch.Tensor, H: torch.Tensor, H2: Optional[torch.Tensor],
                mat_index: Optional[torch.Tensor]) -> torch.Tensor:
    return ops.hg_from_im(im, heights, H, H2, mat_index, to_state=True)


@im_to_state.register_fake
def _(im, heights, H, H2, mat_index):
    return im.new_empty((im.shape[0], 6), dtype=torch.float32)


# ---- frame ingest
@_lib.custom_op(NS + "::frame_ingest", mutates_args=(), device_types="cuda")
def frame_ingest(frames_u8: torch.Tensor, swap_rb: bool, nhwc4: bool) -> torch.Tensor:
    return ops.frame_ingest(frames_u8, swap_rb=swap_rb, nhwc4=nhwc4)


@frame_ingest.register_fake
def _(frames_u8, swap_rb, nhwc4):
    B, H, W, _ = frames_u8.shape
    return frames_u8.new_empty((B, H, W, 4) if nhwc4 else (B, 3, H, W), dtype=torch.float32)


OPERATORS = ("anchors", "pairwise_iou", "focal_loss_fwd", "focal_loss_bwd", "decode_dir", "decode_2d", "clip_boxes_", "nms",
             "state_to_space", "state_to_im", "im_to_state", "frame_ingest")
