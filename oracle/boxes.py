"""Box decode, clipping, NMS and the detector's post-process modes, restated on CPU
(test infrastructure -- see oracle/__init__.py).

  decode_dir            <- BBoxTransform.forward   D/utils.py:102-149
  decode_2d             <- BBoxTransform.forward   R/utils.py:102-126
  clip_boxes            <- ClipBoxes.forward       R/utils.py:134-144
  greedy_nms            <- torchvision.ops.nms     (third party, NOT in the reference tree, version unpinned;
                                                    restated from its documented contract: order by score
                                                    descending, drop a box whose IoU with an already kept box
                                                    is > thr, return kept indices in score order.
                                                    PARITY UNPINNED: no reference test or fixture covers it.
                                                    Ties in score are broken by lower index first.)
  batched_nms           <- batched_nms             D/model.py:19-57 (offset trick in fp32)
  adaptive_threshold    <- the while-loops         D/model.py:322-328 (MULTI_FRAME), :368-374 (single)
  postprocess_single    <- ResNet.forward eval     D/model.py:346-397
  postprocess_multi     <- ResNet.forward eval     D/model.py:311-344
  postprocess_2d        <- ResNet.forward eval     R/model.py:269-311
"""
import torch

from .losses import corners_from_regression


def decode_dir(anchors, reg):
    """anchors [1,A,4], reg [B,A,12] -> [B,A,20]; x cols * w + cx, y cols * h + cy (two roundings).
    D/utils.py:104-135."""
    w = anchors[:, :, 2] - anchors[:, :, 0]
    h = anchors[:, :, 3] - anchors[:, :, 1]
    cx = anchors[:, :, 0] + 0.5 * w
    cy = anchors[:, :, 1] + 0.5 * h
    out = corners_from_regression(reg).clone()
    out[:, :, 0::2] = out[:, :, 0::2] * w[:, :, None] + cx[:, :, None]
    out[:, :, 1::2] = out[:, :, 1::2] * h[:, :, None] + cy[:, :, None]
    return out


def decode_2d(anchors, deltas):
    """anchors [1,A,4], deltas [B,A,4] -> [B,A,4] (x1,y1,x2,y2); std (0.1,0.1,0.2,0.2), mean 0.
    R/utils.py:104-126."""
    w = anchors[:, :, 2] - anchors[:, :, 0]
    h = anchors[:, :, 3] - anchors[:, :, 1]
    cx = anchors[:, :, 0] + 0.5 * w
    cy = anchors[:, :, 1] + 0.5 * h
    std = torch.tensor([0.1, 0.1, 0.2, 0.2], dtype=deltas.dtype)
    dx = deltas[:, :, 0] * std[0] + 0.0
    dy = deltas[:, :, 1] * std[1] + 0.0
    dw = deltas[:, :, 2] * std[2] + 0.0
    dh = deltas[:, :, 3] * std[3] + 0.0
    pcx = cx + dx * w
    pcy = cy + dy * h
    pw = torch.exp(dw) * w
    ph = torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=2)


def clip_boxes(boxes, height, width):
    """In place: x1,y1 >= 0; x2 <= W; y2 <= H.  Returns the same tensor.  R/utils.py:134-144."""
    boxes[:, :, 0].clamp_(min=0)
    boxes[:, :, 1].clamp_(min=0)
    boxes[:, :, 2].clamp_(max=width)
    boxes[:, :, 3].clamp_(max=height)
    return boxes


def greedy_nms(boxes, scores, thr):
    """int64 keep indices, decreasing score.  See module docstring (parity unpinned)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        if i + 1 == n:
            break
        r = b[i + 1:]
        iw = (torch.min(r[:, 2], b[i, 2]) - torch.max(r[:, 0], b[i, 0])).clamp(min=0)
        ih = (torch.min(r[:, 3], b[i, 3]) - torch.max(r[:, 1], b[i, 1])).clamp(min=0)
        inter = iw * ih
        iou = inter / (area[i] + area[i + 1:] - inter)
        dead[i + 1:] |= iou > thr
    return order[torch.tensor(keep, dtype=torch.int64)]


def batched_nms(boxes, scores, idxs, thr):
    """Per-category NMS through the fp32 offset trick.  D/model.py:47-57."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    offs = idxs.to(boxes) * (boxes.max() + 1)
    return greedy_nms(boxes + offs[:, None], scores, thr)


def adaptive_threshold(scores, start, keep=10000):
    """Mask chosen by the reference's loop: t = start; repeat {mask = s > t; n = mask.sum(); t *= 10**0.2}
    until n <= keep; the mask is the one computed with the pre-multiplied t.
    The comparison is done in fp32 (python scalar against an fp32 tensor).  D/model.py:368-374, :322-328."""
    t = float(start)
    while True:
        mask = scores > t
        n = int(mask.sum())
        t *= (10 ** .2)
        if n <= keep:
            return mask


def postprocess_single(cls, boxes20):
    """cls [1,A,C], boxes20 [1,A,20] -> (scores[K], class_idx[K] int64, boxes[K,20]).  D/model.py:346-397.
    Batch must be 1 (the reference squeezes the batch dimension away, :366,381)."""
    assert cls.shape[0] == 1, "single-frame post-process assumes batch 1 (D/model.py:366)"
    out_s, out_c, out_b = [], [], []
    for c in range(cls.shape[2]):
        s = cls[0, :, c]
        mask = adaptive_threshold(s, 1e-25)
        if mask.sum() == 0:
            continue
        s = s[mask]
        b = boxes20[0][mask]
        k = greedy_nms(b[:, 16:20], s, 0.5)
        out_s.append(s[k])
        out_c.append(torch.full((k.numel(),), c, dtype=torch.int64))
        out_b.append(b[k])
    if not out_s:
        return torch.zeros(0), torch.zeros(0, dtype=torch.int64), torch.zeros(0)
    return torch.cat(out_s), torch.cat(out_c), torch.cat(out_b)


def postprocess_multi(cls, boxes20):
    """cls [B,A,C], boxes20 [B,A,20] -> (scores, classes, boxes[K,20], im_index).  D/model.py:311-344."""
    B, A, C = cls.shape
    im = torch.arange(B).unsqueeze(1).repeat(1, A).reshape(-1)
    flat_b = boxes20.reshape(-1, 20)
    flat_c = cls.reshape(-1, C)
    s, k = flat_c.max(dim=1)
    mask = adaptive_threshold(s, 1e-7)
    s, k, flat_b, im = s[mask], k[mask], flat_b[mask], im[mask]
    keep = batched_nms(flat_b[:, 16:20], s, im, 0.5)
    return s[keep], k[keep], flat_b[keep], im[keep]


def postprocess_2d(cls, boxes4):
    """cls [1,A,C], boxes4 [1,A,4] (already clipped) -> (scores, class_idx, boxes[K,4]).  R/model.py:283-311."""
    assert cls.shape[0] == 1
    out_s, out_c, out_b = [], [], []
    for c in range(cls.shape[2]):
        s = cls[0, :, c]
        mask = s > 0.05
        if mask.sum() == 0:
            continue
        s = s[mask]
        b = boxes4[0][mask]
        k = greedy_nms(b, s, 0.5)
        out_s.append(s[k])
        out_c.append(torch.full((k.numel(),), c, dtype=torch.int64))
        out_b.append(b[k])
    if not out_s:
        return torch.zeros(0), torch.zeros(0, dtype=torch.int64), torch.zeros(0)
    return torch.cat(out_s), torch.cat(out_c), torch.cat(out_b)
