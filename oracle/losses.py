"""IoU matching + focal / smooth-L1 / vanishing-point losses, restated on CPU
(test infrastructure -- see oracle/__init__.py).

  pairwise_iou      <- calc_iou                D/losses.py:5-22  (R/losses.py:5-22 identical)
  assign            <- IoU max/argmax + bands  D/losses.py:93-124, R/losses.py:82-104
  focal_loss_dir    <- FocalLoss.forward       D/losses.py:27-362   (directional 3D, 27-col labels)
  focal_loss_2d     <- FocalLoss.forward       R/losses.py:27-177   (2D, 5-col labels)

Every fp32 operation is a separate torch op in the same order as the reference,
so IoU values -- and therefore the integer assignment -- are bit-identical to
the reference on CPU (no fused multiply-add anywhere).  Autograd works through
these functions; the tests use that for the gradient goldens.
"""
import torch

ALPHA = 0.25        # D/losses.py:29
GAMMA = 2.0         # D/losses.py:30
TOP_WEIGHT = 0.5    # D/losses.py:28
IOU_NEG = 0.4       # D/losses.py:121
IOU_POS = 0.5       # D/losses.py:124
BETA = 1.0 / 9.0    # D/losses.py:346

# corner synthesis: pred[2j+a] = r[a] + sl*r[2+a] + sw*r[4+a] + sh*r[6+a], j = 0..7
# (fbl fbr bbl bbr ftl ftr btl btr), D/losses.py:310-327 and D/utils.py:113-130.
CORNER_SIGNS = (
    (-1, -1, +1), (-1, +1, +1), (+1, -1, +1), (+1, +1, +1),
    (-1, -1, -1), (-1, +1, -1), (+1, -1, -1), (+1, +1, -1),
)


def pairwise_iou(a, b):
    """[A,4] x [N,4] -> [A,N]; no +1, clamp(iw,ih >= 0), union clamp 1e-8.  D/losses.py:5-22."""
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = torch.min(a[:, 2:3], b[:, 2]) - torch.max(a[:, 0:1], b[:, 0])
    ih = torch.min(a[:, 3:4], b[:, 3]) - torch.max(a[:, 1:2], b[:, 1])
    iw = iw.clamp(min=0)
    ih = ih.clamp(min=0)
    area_a = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None]
    union = (area_a + area_b - iw * ih).clamp(min=1e-8)
    return (iw * ih) / union


def envelope_boxes(corners16):
    """2D box used for matching in the directional loss: min/max envelope of the
    8 corners (cols 0..15), NOT label cols 16:20.  D/losses.py:93-107."""
    xs = corners16[:, 0:16:2]
    ys = corners16[:, 1:16:2]
    return torch.stack((xs.min(1).values, ys.min(1).values, xs.max(1).values, ys.max(1).values), dim=1)


def assign(anchors, gt_boxes):
    """Per-anchor (iou_max, argmax, state) with state 0 = negative (IoU < 0.4),
    1 = positive (IoU >= 0.5), -1 = ignored.  Ties in argmax resolve to the
    lowest GT index (torch CPU ``max`` keeps the first maximum).
    D/losses.py:109-124."""
    iou = pairwise_iou(anchors, gt_boxes)
    iou_max, arg = iou.max(dim=1)
    state = torch.full_like(arg, -1)
    state[iou_max < IOU_NEG] = 0
    state[iou_max >= IOU_POS] = 1
    return iou_max, arg, state


def _focal_terms(p, target):
    """Element-wise focal BCE for targets in {-1,0,1}; p already clamped.  D/losses.py:137-152."""
    is_pos = target == 1.0
    alpha_w = torch.where(is_pos, torch.full_like(p, ALPHA), torch.full_like(p, 1.0 - ALPHA))
    mod = torch.where(is_pos, 1.0 - p, p)
    weight = alpha_w * mod.pow(GAMMA)
    bce = -(target * torch.log(p) + (1.0 - target) * torch.log(1.0 - p))
    loss = weight * bce
    return torch.where(target != -1.0, loss, torch.zeros_like(loss))


def _empty_image_cls(p):
    """All-negative focal SUM (not normalised) for an image without labels.  D/losses.py:58-87."""
    return ((1.0 - ALPHA) * p.pow(GAMMA) * (-torch.log(1.0 - p))).sum()


def _cls_targets(p, state, cls_idx):
    tgt = torch.full_like(p, -1.0)
    tgt[state == 0, :] = 0.0
    pos = state == 1
    tgt[pos, :] = 0.0
    tgt[pos, cls_idx[pos]] = 1.0
    return tgt


def _smooth_l1(diff):
    return torch.where(diff <= BETA, 0.5 * 9.0 * diff.pow(2), diff - 0.5 / 9.0)


def _anchor_geometry(anchors):
    w = anchors[:, 2] - anchors[:, 0]
    h = anchors[:, 3] - anchors[:, 1]
    return w, h, anchors[:, 0] + 0.5 * w, anchors[:, 1] + 0.5 * h   # D/losses.py:37-40


def corners_from_regression(r):
    """[...,12] -> [...,20]: 8 corners (x,y interleaved) + the 2D box copied.  D/losses.py:310-328."""
    cols = []
    for sl, sw, sh in CORNER_SIGNS:
        for a in (0, 1):
            v = r[..., a]
            v = v + r[..., 2 + a] if sl > 0 else v - r[..., 2 + a]
            v = v + r[..., 4 + a] if sw > 0 else v - r[..., 4 + a]
            v = v + r[..., 6 + a] if sh > 0 else v - r[..., 6 + a]
            cols.append(v)
    cols.extend(r[..., 8 + k] for k in range(4))
    return torch.stack(cols, dim=-1)


def _vp_targets(t):
    """Three mean-corner direction vectors in raw pixels.  D/losses.py:217-283.
    t: [P,>=16].  k=0 back-front, k=1 right-left, k=2 bottom-top."""
    def grp(plus, minus, a):
        sp = ((t[:, 2 * plus[0] + a] + t[:, 2 * plus[1] + a]) + t[:, 2 * plus[2] + a]) + t[:, 2 * plus[3] + a]
        sm = ((t[:, 2 * minus[0] + a] + t[:, 2 * minus[1] + a]) + t[:, 2 * minus[2] + a]) + t[:, 2 * minus[3] + a]
        return (sp - sm) / 4.0
    groups = (((2, 3, 6, 7), (0, 1, 4, 5)),      # back - front   D/losses.py:221-222
              ((1, 3, 5, 7), (0, 2, 4, 6)),      # right - left   D/losses.py:250-251
              ((0, 1, 2, 3), (4, 5, 6, 7)))      # bottom - top   D/losses.py:277-278
    return [(grp(p, m, 0), grp(p, m, 1)) for p, m in groups]


def focal_loss_dir(cls, reg, anchors, ann):
    """Directional 3D loss.  cls [B,A,C] (post-sigmoid), reg [B,A,12], anchors [1,A,4],
    ann [B,N,27] -> (cls_loss[1], reg_loss[1], vp_loss[1]).  D/losses.py:27-362.

    Quirks kept: labels valid iff col 20 != -1 (:54); matching on the corner envelope (:93-107);
    VP targets from un-normalised corners (:217-283) computed before the target normalisation (:330-331);
    top-corner 0.5 weight applied to |diff| before the beta test (:343-349); image without labels
    contributes an un-normalised cls SUM and a 0 to reg but nothing to vp (:58-87); image with labels
    but no positive contributes 0 to reg and vp (:351-357).
    """
    anc = anchors[0]
    aw, ah, acx, acy = _anchor_geometry(anc)
    cls_terms, reg_terms, vp_terms = [], [], []
    for j in range(cls.shape[0]):
        p = cls[j].clamp(1e-4, 1.0 - 1e-4)
        lab = ann[j, :, :21]
        lab = lab[lab[:, 20] != -1]
        if lab.shape[0] == 0:
            cls_terms.append(_empty_image_cls(p))
            reg_terms.append(torch.zeros((), dtype=cls.dtype, device=cls.device))
            continue
        _, arg, state = assign(anc, envelope_boxes(lab[:, :16]))
        pos = state == 1
        npos = pos.sum()
        picked = lab[arg]
        tgt = _cls_targets(p, state, picked[:, 20].long())
        cls_terms.append(_focal_terms(p, tgt).sum() / npos.float().clamp(min=1.0))
        if npos == 0:
            reg_terms.append(torch.zeros((), dtype=cls.dtype, device=cls.device))
            vp_terms.append(torch.zeros((), dtype=cls.dtype, device=cls.device))
            continue
        t = picked[pos, :20]
        r = reg[j][pos]
        # vanishing-point direction term
        vp = 0.0
        for k, (tx, ty) in enumerate(_vp_targets(t)):
            vx, vy = r[:, 2 + 2 * k], r[:, 3 + 2 * k]
            cos = (vx * tx + vy * ty) / (torch.sqrt(vx.pow(2) + vy.pow(2)) * torch.sqrt(tx.pow(2) + ty.pow(2)))
            vp = vp + (1 - cos)
        vp_terms.append((vp / 3.0).mean())
        # smooth-L1 over 20 normalised values
        pred = corners_from_regression(r)
        tn = t.clone()
        tn[:, 0::2] = (t[:, 0::2] - acx[pos, None]) / aw[pos, None]
        tn[:, 1::2] = (t[:, 1::2] - acy[pos, None]) / ah[pos, None]
        diff = (tn - pred).abs()
        wts = torch.ones(20, dtype=diff.dtype, device=diff.device)
        wts[8:16] = TOP_WEIGHT
        reg_terms.append(_smooth_l1(diff * wts).mean())
    return (torch.stack(cls_terms).mean(0, keepdim=True),
            torch.stack(reg_terms).mean(0, keepdim=True),
            torch.stack(vp_terms).mean(0, keepdim=True))


def focal_loss_2d(cls, reg, anchors, ann):
    """2D loss.  cls [B,A,C], reg [B,A,4], anchors [1,A,4], ann [B,N,5] -> (cls_loss[1], reg_loss[1]).
    R/losses.py:27-177.  Targets (dx,dy,log dw,log dh)/(0.1,0.1,0.2,0.2), gt w,h clamped to >= 1 (:137-157)."""
    anc = anchors[0]
    aw, ah, acx, acy = _anchor_geometry(anc)
    cls_terms, reg_terms = [], []
    for j in range(cls.shape[0]):
        p = cls[j].clamp(1e-4, 1.0 - 1e-4)
        lab = ann[j]
        lab = lab[lab[:, 4] != -1]
        if lab.shape[0] == 0:
            cls_terms.append(_empty_image_cls(p))
            reg_terms.append(torch.zeros((), dtype=cls.dtype, device=cls.device))
            continue
        _, arg, state = assign(anc, lab[:, :4])
        pos = state == 1
        npos = pos.sum()
        picked = lab[arg]
        tgt = _cls_targets(p, state, picked[:, 4].long())
        cls_terms.append(_focal_terms(p, tgt).sum() / npos.float().clamp(min=1.0))
        if npos == 0:
            reg_terms.append(torch.zeros((), dtype=cls.dtype, device=cls.device))
            continue
        g = picked[pos]
        gw = g[:, 2] - g[:, 0]
        gh = g[:, 3] - g[:, 1]
        gcx = g[:, 0] + 0.5 * gw
        gcy = g[:, 1] + 0.5 * gh
        gw = gw.clamp(min=1)
        gh = gh.clamp(min=1)
        tgt4 = torch.stack(((gcx - acx[pos]) / aw[pos], (gcy - acy[pos]) / ah[pos],
                            torch.log(gw / aw[pos]), torch.log(gh / ah[pos])), dim=1)
        tgt4 = tgt4 / torch.tensor([[0.1, 0.1, 0.2, 0.2]], dtype=tgt4.dtype, device=tgt4.device)
        reg_terms.append(_smooth_l1((tgt4 - reg[j][pos]).abs()).mean())
    return (torch.stack(cls_terms).mean(0, keepdim=True),
            torch.stack(reg_terms).mean(0, keepdim=True))
