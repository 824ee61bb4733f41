"""GPU parity of the tracker's detection parsing: rn_parse_detections / rn_md_iou through the
reference-shaped functions of mc3d_post.py, against the reference-generated goldens (tests/golden/tracker_post.npz)
and the CPU oracle.  Index-type outputs (labels, cameras, kept order) and scores exact; states within 1e-4
(fp32 results of fp64 projections; the oracle itself matches the reference to 1e-5)."""
import types

import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import tracker_post as otp

pytestmark = pytest.mark.gpu


def _tracker(dev, est_ts=False):
    import homography as hgm
    import mc3d_post
    scores, labels, boxes, cams, names, (P, H), (P2, H2) = gc.tracker_post_inputs()

    def make_hg(Pm, Hm):
        hg = hgm.Homography(device=str(dev))
        hg.correspondence = {n: {"P": Pm[i], "H": Hm[i], "H_inv": np.linalg.inv(Hm[i])} for i, n in enumerate(names)}
        hg.default_correspondence = names[0]
        return hg

    class Tracker(mc3d_post.DetectionParser):
        pass
    me = Tracker()
    me.sigma_d, me.phi_nms_im, me.phi_nms_space = 0.1, 0.3, 0.2
    me.cameras, me.est_ts = list(names), est_ts
    me.hg = hgm.Homography_Wrapper(hg1=make_hg(P, H), hg2=make_hg(P2, H2))
    return me, (scores, labels, boxes, cams), (P, H, P2, H2)


@pytest.mark.parametrize("on_gpu", [True, False])
def test_parse_detections_golden(dev, golden, on_gpu):
    z = golden("tracker_post")
    me, inp, _ = _tracker(dev)
    args = [t.to(dev) if on_gpu else t for t in inp]
    for tag, kw in (("nms", dict(perform_nms=True, refine_height=False)),
                    ("nms_refine", dict(perform_nms=True, refine_height=True)),
                    ("plain", dict(perform_nms=False, refine_height=False))):
        st, lb, sc, cm = me.parse_detections(*args, **kw)
        assert st.is_cuda == on_gpu and lb.dtype == torch.int64 and cm.dtype == torch.int64 and st.dtype == torch.float32
        assert np.array_equal(lb.cpu().numpy(), z[tag + "_labels"]), tag
        assert np.array_equal(cm.cpu().numpy(), z[tag + "_cams"]), tag
        assert np.array_equal(sc.cpu().numpy(), z[tag + "_scores"]), tag
        assert np.allclose(st.cpu().numpy(), z[tag + "_state"], rtol=1e-5, atol=1e-4), tag


def test_parse_detections_empty(dev):
    me, (scores, labels, boxes, cams), _ = _tracker(dev)
    assert me.parse_detections(scores[:0], labels[:0], boxes[:0], cams[:0]) == ([], [], [], [])
    assert me.parse_detections((scores * 0.01).to(dev), labels.to(dev), boxes.to(dev), cams.to(dev)) == ([], [], [], [])


def test_parse_detections_est_ts_sees_boxes_before_space_nms(dev):
    """estimate_ts_bias (tracker state) is called between the transforms and the space NMS (MC3D_crop_tracker.py:373)."""
    me, inp, (P, H, P2, H2) = _tracker(dev, est_ts=True)
    seen = {}
    me.estimate_ts_bias = types.MethodType(lambda self, b, c: seen.update(boxes=b.cpu(), cams=c.cpu()), me)
    st, lb, sc, cm = me.parse_detections(*[t.to(dev) for t in inp], refine_height=True)
    scores, labels, boxes, cams = inp
    keep = scores > 0.1
    det = boxes[keep].reshape(-1, 10, 2)[:, :8, :]
    idx = otp.im_nms(det, scores[keep], threshold=0.3, groups=cams[keep])
    assert np.array_equal(seen["cams"].numpy(), cams[keep][idx].numpy())
    ref = otp.parse_detections(*inp, H, H2, P, P2, perform_nms=True, refine_height=True)
    assert np.array_equal(cm.cpu().numpy(), ref[3].numpy()) and np.array_equal(sc.cpu().numpy(), ref[2].numpy())
    assert np.allclose(st.cpu().numpy(), ref[0].numpy(), rtol=1e-5, atol=1e-4)


def test_nms_pieces_and_md_iou_golden(dev, golden):
    z = golden("tracker_post")
    me, (scores, labels, boxes, cams), _ = _tracker(dev)
    keep = scores > 0.1
    det = boxes[keep].reshape(-1, 10, 2)[:, :8, :]
    assert np.array_equal(me.im_nms(det.to(dev), scores[keep].to(dev), threshold=0.3, groups=cams[keep].to(dev)).cpu().numpy(),
                          z["im_nms_idx"])
    assert np.array_equal(me.im_nms(det, scores[keep], threshold=0.3).numpy(), z["im_nms_idx_nogroups"])
    st = torch.from_numpy(z["plain_state"])
    assert np.array_equal(me.space_nms(st.to(dev), torch.from_numpy(z["plain_scores"]).to(dev), threshold=0.2).cpu().numpy(),
                          z["space_nms_idx"])
    b4 = boxes[:64, 16:20].double()
    got = me.md_iou(b4[None].repeat(64, 1, 1).to(dev), b4[:, None].repeat(1, 64, 1).to(dev))
    assert got.dtype == torch.float64
    assert np.allclose(got.cpu().numpy(), z["md_iou"], rtol=1e-12, equal_nan=True)


def test_parse_detections_large_vs_oracle(dev):
    """A detector-sized input (d ~ 9 000 of the 10 000 the MULTI_FRAME branch can return) against the oracle."""
    scores, labels, boxes, cams, names, (P, H), (P2, H2) = gc.tracker_post_inputs(n_obj=2000, seed=77)
    me, _, _ = _tracker(dev)
    for refine in (False, True):
        ref = otp.parse_detections(scores, labels, boxes, cams, H, H2, P, P2, perform_nms=True, refine_height=refine)
        st, lb, sc, cm = me.parse_detections(scores.to(dev), labels.to(dev), boxes.to(dev), cams.to(dev), refine_height=refine)
        assert np.array_equal(lb.cpu().numpy(), ref[1].numpy())
        assert np.array_equal(cm.cpu().numpy(), ref[3].numpy())
        assert np.array_equal(sc.cpu().numpy(), ref[2].numpy())
        assert np.allclose(st.cpu().numpy(), ref[0].numpy(), rtol=1e-5, atol=1e-4)


def test_parse_detections_too_many(dev):
    me, _, _ = _tracker(dev)
    d = 16385
    with pytest.raises(RuntimeError, match="at most"):
        me.parse_detections(torch.ones(d, device=dev), torch.zeros(d, dtype=torch.int64, device=dev),
                            torch.zeros(d, 20, device=dev), torch.zeros(d, dtype=torch.int64, device=dev))
