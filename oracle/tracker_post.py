"""Detection parsing of the multi-camera tracker, restated on CPU (test infrastructure -- see oracle/__init__.py).

  md_iou            <- MC_Crop_Tracker.md_iou            MC3D_crop_tracker.py:1030-1049
  im_nms            <- MC_Crop_Tracker.im_nms            MC3D_crop_tracker.py:592-616
  space_nms         <- MC_Crop_Tracker.space_nms         MC3D_crop_tracker.py:618-636
  parse_detections  <- MC_Crop_Tracker.parse_detections  MC3D_crop_tracker.py:319-383

Quirks kept:
  * im_nms computes a per-camera offset but never uses it: every box is shifted by the SAME 10 000
    (MC3D_crop_tracker.py:610-613), so detections of different cameras do suppress each other, and the IoU is
    evaluated on the shifted fp32 coordinates;
  * heights come from Homography.guess_heights(labels) with labels an integer tensor: every lookup misses the
    string-keyed table and falls back to "other" = 5 ft (homography.py:502-517);
  * survivors come back in NMS order (decreasing score), twice;
  * an empty input or nothing above sigma_d returns four empty lists (MC3D_crop_tracker.py:334-348).
``nms`` is torchvision.ops.nms in the reference (a third-party dependency, not under /root/reference, version
unpinned); oracle/boxes.greedy_nms restates its published contract (parity unpinned for NMS itself, see there).
The timestamp-bias side effect (estimate_ts_bias, MC3D_crop_tracker.py:373-374) is tracker state, not part of the
returned values, and is left to the caller.
"""
import numpy as np
import torch

from . import boxes as oboxes
from . import homography as ohg

LARGE_OFFSET = 10000                     # MC3D_crop_tracker.py:610


def md_iou(a, b):
    """[B,N,4] x [B,N,4] -> [B,N] float64-promoted IoU, no clamps on the union.  MC3D_crop_tracker.py:1030-1049."""
    area_a = (a[:, :, 2] - a[:, :, 0]) * (a[:, :, 3] - a[:, :, 1])
    area_b = (b[:, :, 2] - b[:, :, 0]) * (b[:, :, 3] - b[:, :, 1])
    minx = torch.max(a[:, :, 0], b[:, :, 0])
    maxx = torch.min(a[:, :, 2], b[:, :, 2])
    miny = torch.max(a[:, :, 1], b[:, :, 1])
    maxy = torch.min(a[:, :, 3], b[:, :, 3])
    zeros = torch.zeros(minx.shape, dtype=torch.float64)
    inter = torch.max(zeros, maxx - minx) * torch.max(zeros, maxy - miny)
    return inter / (area_a + area_b - inter)


def im_boxes(detections, groups=None):
    """[d,8,2] -> [d,4] envelope, shifted by the constant offset when groups are given (the quirk above)."""
    b = torch.stack((detections[:, :, 0].min(1).values, detections[:, :, 1].min(1).values,
                     detections[:, :, 0].max(1).values, detections[:, :, 1].max(1).values), dim=1)
    if groups is not None:
        b = b + LARGE_OFFSET
    return b


def im_nms(detections, scores, threshold=0.8, groups=None):
    return oboxes.greedy_nms(im_boxes(detections, groups), scores, threshold)


def space_boxes(state):
    """[d,6] -> [d,4] footprint (min/max of the four bottom corners) in fp32.  MC3D_crop_tracker.py:626-633."""
    sp = torch.from_numpy(ohg.state_to_space(state.numpy() if isinstance(state, torch.Tensor) else state))
    out = torch.zeros((sp.shape[0], 4))
    out[:, 0] = sp[:, 0:4, 0].min(1).values
    out[:, 2] = sp[:, 0:4, 0].max(1).values
    out[:, 1] = sp[:, 0:4, 1].min(1).values
    out[:, 3] = sp[:, 0:4, 1].max(1).values
    return out


def space_nms(state, scores, threshold=0.1):
    return oboxes.greedy_nms(space_boxes(state), scores, threshold)


def parse_detections(scores, labels, boxes, camera_idxs, H1, H2, P1, P2, sigma_d=0.1, phi_nms_im=0.3,
                     phi_nms_space=0.2, perform_nms=True, refine_height=False):
    """scores [d] f32, labels [d] i64, boxes [d,20] f32, camera_idxs [d] i64; H*/P*: per-CAMERA matrices
    ([n_cam,3,3] / [n_cam,3,4], index = camera index) of the wrapper's two homographies.
    -> (state [k,6] f32, labels [k], scores [k], camera_idxs [k]) or four empty lists."""
    if len(scores) == 0:
        return [], [], [], []
    keep = torch.where(scores > torch.ones(scores.shape) * sigma_d)        # :338-339
    labels, det, scores, camera_idxs = labels[keep], boxes[keep], scores[keep], camera_idxs[keep]
    if len(det) == 0:
        return [], [], [], []
    det = det.reshape(-1, 10, 2)[:, :8, :]                                  # :349-350
    if perform_nms:
        idxs = im_nms(det, scores, groups=camera_idxs, threshold=phi_nms_im)
        labels, det, scores, camera_idxs = labels[idxs], det[idxs], scores[idxs], camera_idxs[idxs]
    cam = camera_idxs.numpy()
    heights = ohg.guess_heights(list(labels))                               # tensor keys: all "other"
    dn = det.numpy()

    def to_state(h):
        return ohg.space_to_state(ohg.wrapper_im_to_space(dn, H1[cam], H2[cam], h))
    state = to_state(heights)
    if refine_height:                                                       # :366-370
        repro = ohg.wrapper_space_to_im(ohg.state_to_space(state), P1[cam], P2[cam])
        state = to_state(ohg.height_from_template(repro, heights, dn))
    state = torch.from_numpy(np.asarray(state, dtype=np.float32))
    if perform_nms:
        idxs = space_nms(state, scores, threshold=phi_nms_space)
        labels, state, scores, camera_idxs = labels[idxs], state[idxs], scores[idxs], camera_idxs[idxs]
    return state, labels, scores, camera_idxs
