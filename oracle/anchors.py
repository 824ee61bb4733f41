"""Anchor generation, restated (test infrastructure -- see oracle/__init__.py).

Follows D/anchors.py:6-40 (Anchors), :42-73 (generate_anchors), :109-129 (shift);
R/anchors.py is identical on every path that runs.

The reference does all arithmetic in numpy float64 and casts to float32 once at
the very end (D/anchors.py:28-38): ``np.append`` of a float32 seed with float64
rows promotes to float64.  This restatement keeps exactly that: fp64 base boxes,
fp64 cell centres, one fp64 add, one cast.
"""
import numpy as np

PYRAMID_LEVELS = (3, 4, 5, 6, 7)                     # D/anchors.py:11
RATIOS = (0.5, 1.0, 2.0)                             # D/anchors.py:17
SCALES = (2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0))  # D/anchors.py:19


def level_grid(height, width, level):
    """ceil(H / 2^l), ceil(W / 2^l) -- D/anchors.py:25."""
    s = 2 ** level
    return (height + s - 1) // s, (width + s - 1) // s


def base_boxes(size):
    """9 zero-centred boxes of one pyramid level, ratio-major / scale-minor,
    float64 [9,4] as (x1,y1,x2,y2).  D/anchors.py:42-73.

    side = size*scale; area = side*side; w = sqrt(area/ratio); h = w*ratio;
    box = (0 - w/2, 0 - h/2, w - w/2, h - h/2).
    """
    out = np.zeros((len(RATIOS) * len(SCALES), 4), dtype=np.float64)
    i = 0
    for r in RATIOS:
        for s in SCALES:
            side = np.float64(size) * np.float64(s)
            area = side * side
            w = np.sqrt(area / np.float64(r))
            h = w * np.float64(r)
            out[i] = (0.0 - w * 0.5, 0.0 - h * 0.5, w - w * 0.5, h - h * 0.5)
            i += 1
    return out


def anchors_for_image(height, width):
    """All anchors of an H x W image: float32 [1, A, 4].

    Order: level -> row -> col -> base box (D/anchors.py:109-129 ``shift`` puts
    the cell index outermost and the 9 boxes innermost), which is the order of
    the heads' ``permute(0,2,3,1).view`` (D/model.py:155-157).
    """
    rows = []
    for lvl in PYRAMID_LEVELS:
        stride = 2 ** lvl
        size = 2 ** (lvl + 2)
        gh, gw = level_grid(height, width, lvl)
        base = base_boxes(size)                                 # [9,4] f64
        cx = (np.arange(gw, dtype=np.float64) + 0.5) * stride   # D/anchors.py:110
        cy = (np.arange(gh, dtype=np.float64) + 0.5) * stride   # D/anchors.py:111
        cell = np.zeros((gh, gw, 1, 4), dtype=np.float64)
        cell[..., 0, 0] = cx[None, :]
        cell[..., 0, 1] = cy[:, None]
        cell[..., 0, 2] = cx[None, :]
        cell[..., 0, 3] = cy[:, None]
        rows.append((cell + base[None, None, :, :]).reshape(-1, 4))
    return np.concatenate(rows, axis=0).astype(np.float32)[None]


def num_anchors(height, width):
    return sum(9 * gh * gw for gh, gw in (level_grid(height, width, l) for l in PYRAMID_LEVELS))
