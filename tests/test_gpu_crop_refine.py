"""GPU parity of the tracker's crop-refinement path: rn_crop_boxes / rn_roi_align / rn_crop_select
and the reference-shaped functions of mc3d_post.py against the reference-generated goldens
(tests/golden/crop_refine.npz) and the CPU oracle.  Index-type results exact; crop boxes and roi_align bit for bit
against the oracle (same fp operation order); states within 1e-4."""
import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import crop_refine as ocr
from retinanet_mi355x import modules, ops, synth

pytestmark = pytest.mark.gpu


def _tracker(dev):
    import homography as hgm
    import mc3d_post
    pre_loc, cam, im_objs, names, (P, H), (P2, H2) = gc.crop_refine_inputs()

    def make_hg(Pm, Hm):
        hg = hgm.Homography(device=str(dev))
        hg.correspondence = {n: {"P": Pm[i], "H": Hm[i], "H_inv": np.linalg.inv(Hm[i])} for i, n in enumerate(names)}
        hg.default_correspondence = names[0]
        return hg

    class Tracker(mc3d_post.DetectionParser):
        pass
    me = Tracker()
    me.b, me.cs, me.cd_max, me.W = 1.25, 112, 50, 0.5
    me.cameras, me.device = list(names), dev
    me.hg = hgm.Homography_Wrapper(hg1=make_hg(P, H), hg2=make_hg(P2, H2))
    return me, (pre_loc, cam, im_objs), (P, H, P2, H2)


def test_crop_boxes_and_local_to_global_golden(dev, golden):
    z = golden("crop_refine")
    me, (pre_loc, cam, im_objs), _ = _tracker(dev)
    boxes = me.get_crop_boxes(im_objs.to(dev))
    assert boxes.dtype == torch.float64
    assert np.array_equal(boxes.cpu().numpy(), z["crop_boxes"])
    b2, rois = ops.crop_boxes(im_objs.to(dev), cam.to(dev))
    want = torch.cat((cam.double()[:, None], torch.from_numpy(z["crop_boxes"])), dim=1).float()
    assert torch.equal(rois.cpu(), want)
    reg_boxes, _ = gc.crop_detections(im_objs, torch.from_numpy(z["crop_boxes"]))
    glob = me.local_to_global(reg_boxes.to(dev), boxes)
    # torch float64 ops on the device: the division may round differently from the CPU in the last bit
    assert glob.dtype == torch.float64 and np.allclose(glob.cpu().numpy(), z["local_to_global"], rtol=1e-12, atol=0)


def test_select_best_box_golden(dev, golden):
    z = golden("crop_refine")
    me, (pre_loc, cam, im_objs), _ = _tracker(dev)
    best, cls, conf = me.select_best_box(pre_loc.to(dev), torch.from_numpy(z["cand_state"]).to(dev),
                                         torch.from_numpy(z["cand_confs"]).to(dev), torch.from_numpy(z["cand_classes"]).to(dev),
                                         pre_loc.shape[0])
    assert np.array_equal(best.cpu().numpy(), z["best_state"])
    assert np.array_equal(cls.cpu().numpy(), z["best_classes"]) and np.array_equal(conf.cpu().numpy(), z["best_confs"])


def test_crop_select_fused_golden(dev, golden):
    """rn_crop_select = class max + local_to_global + top-k + homographies with refinement + best box, one launch."""
    z = golden("crop_refine")
    me, (pre_loc, cam, im_objs), (P, H, P2, H2) = _tracker(dev)
    boxes = torch.from_numpy(z["crop_boxes"])
    reg_boxes, cls = gc.crop_detections(im_objs, boxes)
    import mc3d_post
    H1d, H2d, P1d, P2d = mc3d_post._camera_matrices(me, dev)
    st, oc, of = ops.crop_select(reg_boxes.to(dev), cls.to(dev), boxes.to(dev), cam.to(dev), pre_loc.to(dev), H1d, H2d, P1d, P2d)
    assert np.array_equal(oc.cpu().numpy(), z["best_classes"])
    assert np.array_equal(of.cpu().numpy(), z["best_confs"])
    assert np.allclose(st.cpu().numpy(), z["best_state"], rtol=1e-5, atol=1e-4)
    with pytest.raises(RuntimeError, match="at most"):
        ops.crop_select(torch.zeros(1, 5000, 20, device=dev), torch.zeros(1, 5000, 8, device=dev), boxes[:1].to(dev),
                        cam[:1].to(dev), pre_loc[:1].to(dev), H1d, H2d, P1d, P2d)


@pytest.mark.parametrize("nhwc4", [False, True])
def test_roi_align_matches_oracle_bitwise(dev, nhwc4):
    """Against the restated torchvision algorithm: boxes inside, partly outside and far outside the frame, sub-pixel
    corners, adaptive grids of 1..4 samples per bin."""
    N, C, H, W = 3, 3, 90, 140
    frames = torch.from_numpy(synth.uniform((N, C, H, W), 3).astype(np.float32))
    u = synth.uniform((12, 4), 4)
    rois = np.zeros((12, 5), dtype=np.float32)
    rois[:, 0] = np.arange(12) % N
    rois[:, 1] = -20 + 120 * u[:, 0]
    rois[:, 2] = -15 + 70 * u[:, 1]
    rois[:, 3] = rois[:, 1] + 8 + 90 * u[:, 2]
    rois[:, 4] = rois[:, 2] + 8 + 90 * u[:, 3]
    rois[11, 1:] = (-300, -300, -200, -200)                                  # nothing but zeros
    want = ocr.roi_align(frames.numpy(), rois, (28, 28))
    got = ops.roi_align(frames.to(dev), torch.from_numpy(rois).to(dev), (28, 28), nhwc4=nhwc4)
    if nhwc4:
        assert got.shape == (12, 28, 28, 4) and float(got[..., 3].abs().max()) == 0.0
        got = got[..., :3].permute(0, 3, 1, 2)
    assert np.array_equal(got.cpu().numpy(), want)
    bad = torch.tensor([[0, float("nan"), 0, 50, 50], [0, 0, 0, float("inf"), 50]], dtype=torch.float32, device=dev)
    out = ops.roi_align(frames.to(dev), bad, (4, 4))                          # must terminate and stay in bounds
    assert out.shape == (2, 3, 4, 4)


def test_crop_refine_end_to_end_vs_oracle(dev):
    """The fused measurement block with a real LOCALIZE detector (ResNet-18, 4 classes, 112x112 crops) against the
    same chain computed by the oracle from the device detector's outputs."""
    me, (pre_loc, cam, im_objs), (P, H, P2, H2) = _tracker(dev)
    sd, _, _ = gc.model_inputs("resnet18", True)
    det = modules.resnet18(num_classes=4)
    det.load_state_dict(sd)
    det = det.to(dev).eval()
    frames = torch.from_numpy(synth.uniform((18, 3, 270, 480), 9).astype(np.float32) - 0.5)
    me.crop_detector = det
    # priors scaled into this small synthetic frame do not matter: the chain is geometric
    st, cl, cf, boxes = me.crop_refine(frames.to(dev), pre_loc, cam)
    assert st.shape == (pre_loc.shape[0], 6) and cl.dtype == torch.int64
    crops = ops.roi_align(frames.to(dev), torch.cat((cam.double()[:, None].to(dev), boxes), 1).float(), (112, 112))
    reg, cls = det(crops, LOCALIZE=True)
    ref = ocr.refine_from_detections(reg.cpu(), cls.cpu(), boxes.cpu(), cam, pre_loc, H, H2, P, P2)
    assert np.array_equal(cl.cpu().numpy(), ref[1].numpy()) and np.array_equal(cf.cpu().numpy(), ref[2].numpy())
    assert np.allclose(st.cpu().numpy(), ref[0].numpy(), rtol=1e-5, atol=1e-4, equal_nan=True)
