"""Crop-refinement path of the multi-camera tracker, restated on CPU (test infrastructure -- see oracle/__init__.py).

  get_crop_boxes    <- MC_Crop_Tracker.get_crop_boxes     MC3D_crop_tracker.py:920-944
  local_to_global   <- MC_Crop_Tracker.local_to_global    MC3D_crop_tracker.py:946-969
  select_best_box   <- MC_Crop_Tracker.select_best_box    MC3D_crop_tracker.py:972-1028
  refine            <- the block of MC_Crop_Tracker.track  MC3D_crop_tracker.py:1172-1226
                       (state_to_im of the priors, crop boxes, roi_align, LOCALIZE detector, class max, local_to_global,
                       top-k, image -> state with height refinement, best box per object)
  roi_align         <- torchvision.ops.roi_align (third-party, not under /root/reference and not installed here, version
                       unpinned): its published algorithm restated -- spatial_scale 1, sampling_ratio -1 (adaptive:
                       ceil(roi size / output size) samples per bin and axis), aligned=False, fp32 -- "parity unpinned".
The first three are pinned by reference-generated goldens (tests/golden/crop_refine.npz).
Quirks kept: guess_heights on an integer class tensor gives the "other" height (5 ft) for every box; the final
score mixes a float64 IoU with a float32 confidence; torch.topk order = decreasing confidence.
"""
import numpy as np
import torch

from . import homography as ohg
from . import tracker_post as otp


def get_crop_boxes(objects, b=1.25):
    """[n,8,2] image corners -> [n,4] (xmin, ymin, xmax, ymax) square crops of side max(w,h)*b.  :920-944."""
    minx, miny = objects[:, :, 0].min(1).values, objects[:, :, 1].min(1).values
    maxx, maxy = objects[:, :, 0].max(1).values, objects[:, :, 1].max(1).values
    w, h = maxx - minx, maxy - miny
    scale = torch.max(torch.stack([w, h]), dim=0)[0] * b
    minx2 = (minx + maxx) / 2.0 - scale / 2.0
    maxx2 = (minx + maxx) / 2.0 + scale / 2.0
    miny2 = (miny + maxy) / 2.0 - scale / 2.0
    maxy2 = (miny + maxy) / 2.0 + scale / 2.0
    return torch.stack([minx2, miny2, maxx2, maxy2]).transpose(0, 1)


def _bilinear(img, y, x):
    """torchvision's bilinear_interpolate for one [C,H,W] image and sample arrays y, x (float32)."""
    C, H, W = img.shape
    out_of_range = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = np.where(y <= 0, np.float32(0), y)
    x = np.where(x <= 0, np.float32(0), x)
    y_low, x_low = y.astype(np.int32), x.astype(np.int32)
    ty, tx = y_low >= H - 1, x_low >= W - 1
    y_low, x_low = np.where(ty, H - 1, y_low), np.where(tx, W - 1, x_low)
    y_high, x_high = np.where(ty, H - 1, y_low + 1), np.where(tx, W - 1, x_low + 1)
    y = np.where(ty, y_low.astype(np.float32), y)
    x = np.where(tx, x_low.astype(np.float32), x)
    ly, lx = (y - y_low.astype(np.float32)).astype(np.float32), (x - x_low.astype(np.float32)).astype(np.float32)
    hy, hx = (np.float32(1) - ly).astype(np.float32), (np.float32(1) - lx).astype(np.float32)
    w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
    v = ((w1 * img[:, y_low, x_low] + w2 * img[:, y_low, x_high]) + w3 * img[:, y_high, x_low]) + w4 * img[:, y_high, x_high]
    return np.where(out_of_range, np.float32(0), v).astype(np.float32)


def roi_align(frames, rois, output_size):
    """frames [N,C,H,W] f32, rois [n,5] (batch index, x1, y1, x2, y2) f32 -> [n,C,ph,pw] f32."""
    frames = np.asarray(frames, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    ph, pw = output_size
    out = np.zeros((rois.shape[0], frames.shape[1], ph, pw), dtype=np.float32)
    for n, r in enumerate(rois):
        img = frames[int(r[0])]
        x1, y1, x2, y2 = r[1], r[2], r[3], r[4]
        rw, rh = max(np.float32(x2 - x1), np.float32(1)), max(np.float32(y2 - y1), np.float32(1))
        bh, bw = np.float32(rh / np.float32(ph)), np.float32(rw / np.float32(pw))
        gh, gw = int(np.ceil(rh / np.float32(ph))), int(np.ceil(rw / np.float32(pw)))
        count = np.float32(max(gh * gw, 1))
        acc = np.zeros((frames.shape[1], ph, pw), dtype=np.float32)
        pys = np.arange(ph, dtype=np.float32)[:, None]
        pxs = np.arange(pw, dtype=np.float32)[None, :]
        for iy in range(gh):
            yy = (y1 + pys * bh) + np.float32(iy + 0.5) * bh / np.float32(gh)
            for ix in range(gw):
                xx = (x1 + pxs * bw) + np.float32(ix + 0.5) * bw / np.float32(gw)
                Y, X = np.broadcast_arrays(yy.astype(np.float32), xx.astype(np.float32))
                acc = (acc + _bilinear(img, Y, X)).astype(np.float32)
        out[n] = acc / count
    return out


def local_to_global(preds, crop_boxes, cs=112):
    """[n,d,20] crop coordinates -> [n,d,8,2] frame coordinates.  :946-969."""
    n, d = preds.shape[0], preds.shape[1]
    p = preds.reshape(n, d, 10, 2)[:, :, :8, :]
    scales = torch.max(torch.stack([crop_boxes[:, 2] - crop_boxes[:, 0], crop_boxes[:, 3] - crop_boxes[:, 1]]), dim=0)[0]
    p = p * scales[:, None, None, None] / cs
    p = p.clone()
    p[:, :, :, 0] += crop_boxes[:, 0][:, None, None]
    p[:, :, :, 1] += crop_boxes[:, 1][:, None, None]
    return p


def select_best_box(a_priori, preds, confs, classes, n_objs, W=0.5):
    """a_priori [n,6], preds [n*d,6] state, confs / classes [n,d] -> (best [n,6], class [n], conf [n]).  :972-1028."""
    foot = otp.space_boxes(preds).reshape(n_objs, -1, 4)
    preds = preds.reshape(n_objs, -1, 6)
    d = foot.shape[1]
    prior = otp.space_boxes(a_priori)[:, None, :].repeat(1, d, 1)
    ious = otp.md_iou(foot.double(), prior.double())
    scores = (1 - W) * ious + W * confs
    keep = torch.argmax(scores, dim=1)
    idx = torch.arange(n_objs)
    return preds[idx, keep, :], classes[idx, keep], confs[idx, keep]


def refine_from_detections(reg_boxes, cls, crop_boxes, cam_idxs, pre_loc, H1, H2, P1, P2, cs=112, cd_max=50, W=0.5):
    """Everything after the LOCALIZE detector (MC3D_crop_tracker.py:1192-1226): reg_boxes [n,A,20], cls [n,A,C],
    crop_boxes [n,4], cam_idxs [n] i64, pre_loc [n,6] -> (detections [n,6], classes [n], confs [n])."""
    confs, classes = torch.max(cls, dim=2)
    g = local_to_global(reg_boxes, crop_boxes, cs)
    top = torch.topk(confs, cd_max, dim=1)[1]
    rows = torch.arange(g.shape[0])[:, None].repeat(1, top.shape[1])
    g, confs, classes = g[rows, top, :, :], confs[rows, top], classes[rows, top]
    n_objs = g.shape[0]
    cam = cam_idxs.numpy().repeat(g.shape[1])
    pts = g.reshape(-1, 8, 2).numpy()
    heights = ohg.guess_heights(list(classes.reshape(-1)))

    def to_state(h):
        return ohg.space_to_state(ohg.wrapper_im_to_space(pts, H1[cam], H2[cam], h))
    st = to_state(heights)
    repro = ohg.wrapper_space_to_im(ohg.state_to_space(st), P1[cam], P2[cam])
    st = to_state(ohg.height_from_template(repro, heights, pts))
    st = torch.from_numpy(np.asarray(st, dtype=np.float32))
    return select_best_box(pre_loc, st, confs, classes, n_objs, W)
