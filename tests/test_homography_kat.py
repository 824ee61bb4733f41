"""CPU: known-answer rows from the reference's own result CSVs (3D_tracking_results.csv,
working_3D_tracking_data.csv -- written by MC3D_crop_tracker.py:1318-1453) pin the oracle's
i24_state_to_space and space_to_im.

The CSVs hold, per tracked vehicle: its state (rear x, centre y, length, width, height, direction), the
four bottom corners in space, and all eight corners in the image.  The camera matrices themselves are
not in the reference tree (git-ignored pickles), so P is recovered per (camera, side of y = 60) by a DLT
fit over the same rows; a single 3x4 matrix reproducing every image coordinate of every row to a few
milli-pixels is only possible if the oracle's corner construction and projection are the reference's.
"""
import numpy as np

from oracle import homography as ohg


def _cols(z):
    names = [str(c) for c in z["columns"]]
    return {n: i for i, n in enumerate(names)}, z["rows"]


def _state(rows, c):
    return np.stack([rows[:, c["veh rear x"]], rows[:, c["veh center y"]], rows[:, c["length"]],
                     rows[:, c["width"]], rows[:, c["height"]], rows[:, c["direction"]]], 1).astype(np.float32)


def test_state_to_space_matches_csv(golden):
    c, rows = _cols(golden("csv_kat"))
    space = ohg.state_to_space(_state(rows, c))
    want = rows[:, [c[k] for k in ("fbr_x", "fbr_y", "fbl_x", "fbl_y", "bbr_x", "bbr_y", "bbl_x", "bbl_y")]]
    got = space[:, 0:4, 0:2].reshape(-1, 8)
    assert rows.shape[0] > 500
    assert np.abs(got - want).max() <= 5e-4          # feet; the CSV prints fp32 values


def _dlt(space_pts, im_pts):
    """P (3x4) minimising the algebraic error, with Hartley normalisation."""
    def norm(p):
        m = p.mean(0)
        s = np.sqrt(p.shape[1]) / np.sqrt(((p - m) ** 2).sum(1)).mean()
        T = np.eye(p.shape[1] + 1)
        T[:-1, :-1] *= s
        T[:-1, -1] = -s * m
        return T
    Ts, Ti = norm(space_pts), norm(im_pts)
    X = (Ts @ np.c_[space_pts, np.ones(len(space_pts))].T).T
    x = (Ti @ np.c_[im_pts, np.ones(len(im_pts))].T).T
    rows = []
    for Xi, xi in zip(X, x):
        rows.append(np.r_[Xi, np.zeros(4), -xi[0] * Xi])
        rows.append(np.r_[np.zeros(4), Xi, -xi[1] * Xi])
    _, _, vt = np.linalg.svd(np.asarray(rows))
    P = np.linalg.inv(Ti) @ vt[-1].reshape(3, 4) @ Ts
    return P / P[2, 3]


def test_space_to_im_matches_csv(golden):
    c, rows = _cols(golden("csv_kat"))
    im_cols = [c[k] for k in ("fbrx", "fbry", "fblx", "fbly", "bbrx", "bbry", "bblx", "bbly",
                              "ftrx", "ftry", "ftlx", "ftly", "btrx", "btry", "btlx", "btly")]
    state = _state(rows, c)
    space = ohg.state_to_space(state).astype(np.float64)
    checked = 0
    for cam in np.unique(rows[:, 0]):
        for side in (False, True):                    # Homography_Wrapper switches at corner-0 space y > 60
            sel = (rows[:, 0] == cam) & ((space[:, 0, 1] > 60) == side)
            if sel.sum() < 12:
                continue
            im = rows[sel][:, im_cols].reshape(-1, 8, 2)
            P = _dlt(space[sel].reshape(-1, 3), im.reshape(-1, 2))
            proj = ohg.space_to_im(space[sel], P)
            err = np.abs(proj - im)
            # image coordinates run to +-13000 px for boxes behind the horizon: relative tolerance
            assert np.median(err) < 1e-3, (cam, side, np.median(err))
            assert (err / (1.0 + np.abs(im))).max() < 1e-4, (cam, side)
            checked += int(sel.sum())
    assert checked > 500
