"""State <-> space <-> image transforms of the I-24 homography, restated on CPU
(test infrastructure -- see oracle/__init__.py).

  state_to_space        <- Homography.i24_state_to_space   homography.py:305-320
  space_to_state        <- Homography.i24_space_to_state   homography.py:274-303
  space_to_im           <- Homography.space_to_im          homography.py:438-476
  im_to_space           <- Homography.im_to_space          homography.py:388-435
  state_to_im/im_to_state                                  homography.py:479-500
  guess_heights         <- Homography.guess_heights        homography.py:502-517 (+ class_heights :191-202)
  height_from_template  <- Homography.height_from_template homography.py:519-551
  wrapper_*             <- Homography_Wrapper              homography.py:840-862 (switch on space y of corner 0 > 60)

dtype contract kept from the reference: state and state_to_space are float32;
everything that touched ``.double()`` (image points, im_to_space output) is
float64; im_to_state computes in float64 and stores into a float32 [d,6].

Matrices are passed in as plain numpy float64 (H 3x3 image->space, P 3x4
space->image), per object when ``per_object`` arrays of shape [d,3,3]/[d,3,4]
are given (the reference's ``name`` = list of camera names) or one matrix for
all objects (``name`` = str).
"""
import numpy as np

CLASS_HEIGHTS = {                       # homography.py:191-202
    "sedan": 4, "midsize": 5, "van": 6, "pickup": 5, "semi": 12, "truck (other)": 12,
    "truck": 12, "motorcycle": 4, "trailer": 3, "other": 5,
}


def state_to_space(state):
    """[d,6] (x_rear, y_ctr, l, w, h, dir) f32 -> [d,8,3] f32.  homography.py:305-320.
    Corner order fbr fbl bbr bbl ftr ftl btr btl: x = x+dir*l for {0,1,4,5}; y-/+ dir*w/2 for even/odd; z=-h top."""
    s = np.asarray(state, dtype=np.float32)
    d = s.shape[0]
    out = np.zeros((d, 8, 3), dtype=np.float32)
    xf = s[:, 0] + s[:, 5] * s[:, 2]
    half = s[:, 5] * s[:, 3] / np.float32(2.0)
    out[:, [0, 1, 4, 5], 0] = xf[:, None]
    out[:, [2, 3, 6, 7], 0] = s[:, 0][:, None]
    out[:, [0, 2, 4, 6], 1] = (s[:, 1] - half)[:, None]
    out[:, [1, 3, 5, 7], 1] = (s[:, 1] + half)[:, None]
    out[:, 4:8, 2] = -s[:, 4][:, None]
    return out


def space_to_state(pts):
    """[d,8,3] (any float) -> [d,6] f32.  homography.py:274-303."""
    p = np.asarray(pts)
    d = p.shape[0]
    out = np.zeros((d, 6), dtype=np.float32)
    fx = p[:, 0, 0] + p[:, 1, 0]
    rx = p[:, 2, 0] + p[:, 3, 0]
    out[:, 0] = rx / 2.0
    out[:, 1] = (p[:, 0, 1] + p[:, 1, 1] + p[:, 2, 1] + p[:, 3, 1]) / 4.0
    out[:, 2] = np.abs((fx - rx) / 2.0)
    out[:, 3] = np.abs(((p[:, 0, 1] + p[:, 2, 1]) - (p[:, 1, 1] + p[:, 3, 1])) / 2.0)
    out[:, 4] = np.mean(np.abs(p[:, 0:4, 2] - p[:, 4:8, 2]), axis=1)
    out[:, 5] = np.sign((fx - rx) / 2.0)
    return out


def _per_object(mat, d, m):
    mat = np.asarray(mat, dtype=np.float64)
    if mat.ndim == 2:
        return np.broadcast_to(mat, (d * m,) + mat.shape)
    assert m == 8, "per-object matrices hard-code 8 points/object (homography.py:405,459)"
    return np.repeat(mat, m, axis=0)


def space_to_im(pts, P):
    """[d,m,3] -> [d,m,2] f64 through P (3x4 or [d,3,4]).  homography.py:438-476."""
    p = np.asarray(pts)
    d, m = p.shape[0], p.shape[1]
    hom = np.concatenate((p.reshape(-1, 3).astype(np.float64), np.ones((d * m, 1))), axis=1)
    Pm = _per_object(P, d, m)
    proj = np.einsum("nij,nj->ni", Pm, hom)
    return np.stack((proj[:, 0] / proj[:, 2], proj[:, 1] / proj[:, 2]), axis=1).reshape(d, m, 2)


def im_to_space(pts, H, heights):
    """[d,m,2] image points + heights[d] -> [d,m,3] f64 through H (3x3 or [d,3,3]); z = 0 for points 0-3,
    heights for points 4-7.  homography.py:388-435."""
    p = np.asarray(pts)
    d, m = p.shape[0], p.shape[1]
    hom = np.concatenate((p.reshape(-1, 2).astype(np.float64), np.ones((d * m, 1))), axis=1)
    Hm = _per_object(H, d, m)
    proj = np.einsum("nij,nj->ni", Hm, hom)
    xy = np.stack((proj[:, 0] / proj[:, 2], proj[:, 1] / proj[:, 2]), axis=1).reshape(d, m, 2)
    out = np.concatenate((xy, np.zeros((d, m, 1))), axis=2)
    out[:, 4:8, 2] = np.asarray(heights, dtype=np.float64)[:, None]
    return out


def state_to_im(state, P):
    return space_to_im(state_to_space(state), P)           # homography.py:479-488


def im_to_state(pts, H, heights):
    return space_to_state(im_to_space(pts, H, heights))    # homography.py:491-500


def guess_heights(classes):
    """String class names -> f32 heights; any key miss (ints included) -> "other" = 5.  homography.py:502-517."""
    out = np.zeros(len(classes), dtype=np.float32)
    for i, c in enumerate(classes):
        try:
            out[i] = CLASS_HEIGHTS[c]
        except (KeyError, TypeError):
            out[i] = CLASS_HEIGHTS["other"]
    return out


def height_from_template(template_boxes, template_heights, boxes):
    """homography.py:519-551: space height = image height / (template image height / template space height),
    image height = sum over (x,y) of |mean(top 4) - mean(bottom 4)|."""
    def im_h(b):
        b = np.asarray(b)
        return np.sum(np.sqrt(np.power(b[:, 4:8].mean(1) - b[:, 0:4].mean(1), 2)), axis=1)
    ratio = im_h(template_boxes) / np.asarray(template_heights)
    return im_h(boxes) / ratio


def wrapper_space_to_im(pts, P1, P2):
    """Homography_Wrapper.space_to_im (homography.py:849-856): rows whose corner-0 space y > 60 come from hg2."""
    a = space_to_im(pts, P1)
    b = space_to_im(pts, P2)
    sel = np.asarray(pts)[:, 0, 1] > 60
    a[sel] = b[sel]
    return a


def wrapper_im_to_space(pts, H1, H2, heights):
    """Homography_Wrapper.im_to_space (homography.py:840-847): switch on hg1's result, corner-0 y > 60."""
    a = im_to_space(pts, H1, heights)
    b = im_to_space(pts, H2, heights)
    sel = a[:, 0, 1] > 60
    a[sel] = b[sel]
    return a
