"""Drop-in for the detection-parsing methods of the reference's multi-camera tracker (``MC3D_crop_tracker.py``):
``parse_detections`` (:319-383), ``im_nms`` (:592-616), ``space_nms`` (:618-636) and ``md_iou`` (:1030-1049), and for
its crop-refinement path: ``get_crop_boxes`` (:920-944), ``local_to_global`` (:946-969), ``select_best_box`` (:972-1028)
plus ``crop_refine``, the whole measurement block of ``track`` (:1172-1226) fused on the device.

The functions take ``self`` exactly like the methods they replace and read the same attributes (``sigma_d``,
``phi_nms_im``, ``phi_nms_space``, ``cameras``, ``hg`` = a ``Homography_Wrapper``, ``est_ts`` /
``estimate_ts_bias``), so a maintainer binds them into the reference class unchanged::

    import mc3d_post
    MC_Crop_Tracker.parse_detections = mc3d_post.parse_detections
    MC_Crop_Tracker.im_nms, MC_Crop_Tracker.space_nms = mc3d_post.im_nms, mc3d_post.space_nms
    MC_Crop_Tracker.md_iou = mc3d_post.md_iou

or inherits ``DetectionParser``.  Inputs may stay on the GPU (drop the four ``.cpu()`` copies at
MC3D_crop_tracker.py:1078-1083): the whole chain -- confidence filter, image NMS, per-camera homographies with the
optional height refinement, road-plane NMS, gathers -- runs in libretinanet_mi355x.so (``rn_parse_detections``)
with one device->host word (the survivor count) at the end.  CPU inputs are accepted and give CPU outputs, as the
reference's callers expect.  Reference quirks kept: see include/retinanet_mi355x.h and oracle/tracker_post.py.
"""
import numpy as np
import torch

from retinanet_mi355x import ops as _ops

LARGE_OFFSET = 10000                     # MC3D_crop_tracker.py:610


def _device(self, *tensors):
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    hg1 = getattr(getattr(self, "hg", None), "hg1", None)
    return torch.device(getattr(hg1, "device", "cuda:0"))


def _camera_matrices(self, dev):
    """Per-camera H / P of both wrapper homographies, stacked in ``self.cameras`` order, resident on the device.
    Rebuilt only when the camera list or a correspondence dict object changes (the tracker overwrites
    ``hg.correspondence`` wholesale, MC3D_crop_tracker.py:1561)."""
    hg1, hg2 = self.hg.hg1, self.hg.hg2
    key = (tuple(self.cameras), id(hg1.correspondence), id(hg2.correspondence), str(dev))
    cache = getattr(self, "_rn_parse_cache", None)
    if cache is None or cache[0] != key:
        def stack(hg, k):
            m = np.stack([np.asarray(hg.correspondence[c][k], dtype=np.float64) for c in self.cameras])
            return torch.from_numpy(np.ascontiguousarray(m)).to(dev)
        cache = (key, stack(hg1, "H"), stack(hg2, "H"), stack(hg1, "P"), stack(hg2, "P"))
        self._rn_parse_cache = cache
    return cache[1:]


def _heights(self, labels, dev):
    """guess_heights(labels) (homography.py:502-517).  Tensor labels never hit the string-keyed table: all "other"."""
    if isinstance(labels, torch.Tensor):
        return None                                             # kernel default = class_heights["other"] = 5
    return self.hg.guess_heights(labels).to(dev)


def md_iou(self, a, b):
    dev = _device(self, a, b)
    out = _ops.md_iou(a.to(dev), b.to(dev))
    return out if a.is_cuda else out.cpu()


def _envelope(detections, offset):
    b = torch.stack((detections[:, :, 0].min(1).values, detections[:, :, 1].min(1).values,
                     detections[:, :, 0].max(1).values, detections[:, :, 1].max(1).values), dim=1)
    return b + offset if offset else b


def im_nms(self, detections, scores, threshold=0.8, groups=None):
    """Indices kept, decreasing score.  ``groups`` only switches on the constant 10 000 shift (the reference's
    per-camera offset is computed and dropped, MC3D_crop_tracker.py:610-613)."""
    dev = _device(self, detections, scores)
    det = detections.to(dev).float()
    idx = _ops.nms(_envelope(det, LARGE_OFFSET if groups is not None else 0), scores.to(dev), threshold)
    return idx if detections.is_cuda else idx.cpu()


def space_nms(self, detections, scores, threshold=0.1):
    dev = _device(self, detections, scores)
    sp = _ops.hg_state_to_space(detections.to(dev))
    boxes = torch.stack((sp[:, 0:4, 0].min(1).values, sp[:, 0:4, 1].min(1).values,
                         sp[:, 0:4, 0].max(1).values, sp[:, 0:4, 1].max(1).values), dim=1)
    idx = _ops.nms(boxes, scores.to(dev), threshold)
    return idx if detections.is_cuda else idx.cpu()


def parse_detections(self, scores, labels, boxes, camera_idxs, n_best=200, perform_nms=True, refine_height=False):
    """-> (boxes [k,6] state, labels [k], scores [k], camera_idxs [k]); four empty lists when nothing survives the
    confidence cutoff (MC3D_crop_tracker.py:334-348).  ``n_best`` is accepted and unused, as in the reference."""
    if len(scores) == 0:
        return [], [], [], []
    dev = _device(self, scores, boxes)
    on_gpu = scores.is_cuda
    H1, H2, P1, P2 = _camera_matrices(self, dev)
    est_ts = bool(getattr(self, "est_ts", False))
    flags = 0
    if perform_nms:
        flags = _ops.NMS_IM if est_ts else (_ops.NMS_IM | _ops.NMS_SPACE)
    label_t = labels.to(dev) if isinstance(labels, torch.Tensor) else torch.as_tensor(np.asarray(labels)).to(dev)
    st, lb, sc, cm, count = _ops.parse_detections(
        scores.to(dev), label_t, boxes.to(dev), camera_idxs.to(dev), H1, H2, P1, P2, self.sigma_d, self.phi_nms_im,
        self.phi_nms_space, nms_flags=flags, refine_height=refine_height, heights=_heights(self, labels, dev))
    k = int(count)                                               # the one device->host word
    if k == 0:
        return [], [], [], []
    st, lb, sc, cm = st[:k], lb[:k], sc[:k], cm[:k]
    if est_ts:                                                   # tracker state: looks at the boxes before the space NMS
        self.estimate_ts_bias(st.clone(), cm)
        if perform_nms:
            idxs = space_nms(self, st, sc, threshold=self.phi_nms_space)
            st, lb, sc, cm = st[idxs], lb[idxs], sc[idxs], cm[idxs]
    if not on_gpu:
        st, lb, sc, cm = st.cpu(), lb.cpu(), sc.cpu(), cm.cpu()
    return st, lb, sc, cm


# ------------------------------------------------------------------------------------------- crop refinement
def get_crop_boxes(self, objects):
    """MC3D_crop_tracker.py:920-944: [n,8,2] image corners -> [n,4] square crops (x1,y1,x2,y2), on ``self.device``."""
    dev = _device(self, objects)
    return _ops.crop_boxes(objects.to(dev), b=self.b).to(objects.dtype)


def local_to_global(self, preds, crop_boxes):
    """MC3D_crop_tracker.py:946-969: [n,d,20] crop pixels -> [n,d,8,2] frame pixels (dtype promotion as in torch)."""
    n, d = preds.shape[0], preds.shape[1]
    p = preds.reshape(n, d, 10, 2)[:, :, :8, :]
    scales = torch.max(torch.stack([crop_boxes[:, 2] - crop_boxes[:, 0], crop_boxes[:, 3] - crop_boxes[:, 1]]), dim=0)[0]
    p = (p * scales[:, None, None, None] / self.cs).clone()
    p[:, :, :, 0] += crop_boxes[:, 0][:, None, None]
    p[:, :, :, 1] += crop_boxes[:, 1][:, None, None]
    return p


def select_best_box(self, a_priori, preds, confs, classes, n_objs):
    """MC3D_crop_tracker.py:972-1028 from already transformed candidates: preds [n*d,6] state."""
    dev = _device(self, preds, a_priori)

    def foot(st):
        sp = _ops.hg_state_to_space(st.to(dev))
        return torch.stack((sp[:, 0:4, 0].min(1).values, sp[:, 0:4, 1].min(1).values,
                            sp[:, 0:4, 0].max(1).values, sp[:, 0:4, 1].max(1).values), dim=1)
    fp = foot(preds).reshape(n_objs, -1, 4)
    d = fp.shape[1]
    prior = foot(a_priori)[:, None, :].repeat(1, d, 1)
    ious = _ops.md_iou(fp.double(), prior.double())
    scores = (1 - self.W) * ious + self.W * confs.to(dev)
    keep = torch.argmax(scores, dim=1)
    idx = torch.arange(n_objs, device=dev)
    out = (preds.to(dev).reshape(n_objs, -1, 6)[idx, keep, :], classes.to(dev)[idx, keep], confs.to(dev)[idx, keep])
    return out if preds.is_cuda else tuple(t.cpu() for t in out)


def crop_refine(self, frames, pre_loc, cam_idxs, detector=None):
    """The measurement block of MC_Crop_Tracker.track (MC3D_crop_tracker.py:1172-1226) without leaving the device:
    priors -> image -> crop boxes -> roi_align -> LOCALIZE detector -> best refined box per object.
    frames [n_cam,3,H,W] float32 on the GPU, pre_loc [n,6], cam_idxs [n]; detector defaults to ``self.crop_detector``.
    -> (detections [n,6], classes [n], confs [n], crop_boxes [n,4] float64) as device tensors."""
    dev = frames.device
    det = detector if detector is not None else self.crop_detector
    H1, H2, P1, P2 = _camera_matrices(self, dev)
    cam = cam_idxs.to(dev).long()
    pre = pre_loc.to(dev).float()
    im_objs = _ops.hg_to_im(pre, P1, P2, cam.int(), from_state=True)          # self.hg.state_to_im(pre_loc, name=cam_names)
    boxes, rois = _ops.crop_boxes(im_objs, cam, b=self.b)
    crops = _ops.roi_align(frames, rois, (self.cs, self.cs))
    with torch.no_grad():
        reg_boxes, classes = det(crops, LOCALIZE=True)
    st, cl, cf = _ops.crop_select(reg_boxes, classes, boxes, cam, pre, H1, H2, P1, P2, cs=self.cs, cd_max=self.cd_max, W=self.W)
    return st, cl, cf, boxes


class DetectionParser:
    """Mixin carrying the methods; the host class provides the attributes listed in the module docstring (and, for the
    crop path, ``b``, ``cs``, ``cd_max``, ``W``, ``crop_detector``)."""
    parse_detections = parse_detections
    im_nms = im_nms
    space_nms = space_nms
    md_iou = md_iou
    get_crop_boxes = get_crop_boxes
    local_to_global = local_to_global
    select_best_box = select_best_box
    crop_refine = crop_refine
