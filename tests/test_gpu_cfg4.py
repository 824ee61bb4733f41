"""BASELINE configs[3] as ONE chained test at its real size: three cameras' raw 1080p uint8 frames -> fused ingest ->
ResNet-50 detector -> MULTI_FRAME post-process (decode of the survivors, batched NMS) -> the tracker's parse_detections
(confidence cut, image NMS, image->state through each detection's camera, road-plane NMS) -> state_to_im.
(util_track/mp_loader.py:239-243; D/model.py:311-344; MC3D_crop_tracker.py:197-215, 1074-1088, 319-383; homography.py:479-488)

A detector with random weights has no margin between neighbouring scores, so "same survivors as the CPU pipeline end to
end" is not a meaningful question (a 1e-7 score difference reorders an NMS); the chain is checked link by link instead,
each device stage against the oracle applied to THAT stage's device input:
  1. ingest + network: class scores / decoded boxes of all 3 x 389 205 anchors vs the oracle's CPU forward (1e-4);
  2. post-process: the oracle's MULTI_FRAME branch run on the device's own scores / boxes -> identical survivors, bit for bit;
  3. parse_detections on those survivors vs the oracle's: identical labels / cameras / scores / order, states 1e-4;
  4. state_to_im of the parsed states vs the oracle (1e-9 relative).
"""
import numpy as np
import pytest
import torch

import golden_cases as gc
from oracle import boxes as oboxes, homography as ohg, ingest as oingest, model as omodel, tracker_post as otp

pytestmark = pytest.mark.gpu
H, W, CAMS = 1080, 1920, 3


def _detector(dev):
    from retinanet_mi355x import modules, synth
    sd = synth.state_dict("resnet50", 8, 12, seed=2, head_scale=2e-3)
    # heads that behave like a trained detector's: spread class scores, boxes of about the anchor's size
    w = sd["classificationModel.output.weight"]
    sd["classificationModel.output.weight"] = torch.from_numpy((synth.uniform(tuple(w.shape), 901) - 0.5).astype(np.float32) * 0.01)
    sd["classificationModel.output.bias"] = torch.full_like(sd["classificationModel.output.bias"], -5.5)
    one = torch.tensor([0.0, 0.0, 0.30, 0.05, 0.05, 0.15, 0.0, 0.20, -0.5, -0.5, 0.5, 0.5])
    sd["regressionModel.output.bias"] = one.repeat(9)
    net = modules.resnet50(num_classes=8)
    net.load_state_dict(sd)
    return net.to(dev).eval(), sd


def test_three_camera_pipeline_link_by_link(dev):
    import homography as hgm
    import mc3d_post
    from retinanet_mi355x import ops, synth
    net, sd = _detector(dev)
    g = torch.Generator().manual_seed(1234)
    frames = torch.randint(0, 256, (CAMS, H, W, 3), generator=g, dtype=torch.uint8)
    frames[:, ::7, ::5] //= 3                                            # some structure besides noise
    fd = frames.to(dev)

    # ---- link 1: ingest + network
    boxes_d, cls_d = net(fd, LOCALIZE=True)
    with torch.no_grad():
        x = oingest.to_tensor_normalize(frames)
        boxes_o, cls_o = omodel.eval_forward(x, sd, "resnet50", LOCALIZE=True)
    assert float((cls_d.cpu() - cls_o).abs().max()) <= 1e-4 * float(cls_o.abs().max())
    assert float((boxes_d.cpu() - boxes_o).abs().max()) <= 1e-4 * float(boxes_o.abs().max())

    # ---- link 2: the model's MULTI_FRAME call vs the oracle's branch on the device's own scores / boxes
    s, c, b, im = net(fd, MULTI_FRAME=True)
    so, co, bo, imo = oboxes.postprocess_multi(cls_d.cpu(), boxes_d.cpu())
    assert 50 < s.numel() <= 10000, s.numel()
    assert np.array_equal(s.cpu().numpy(), so.numpy()) and np.array_equal(c.cpu().numpy(), co.numpy())
    assert np.array_equal(im.cpu().numpy(), imo.numpy()) and np.array_equal(b.cpu().numpy(), bo.numpy())

    # ---- link 3: parse_detections, one camera per frame (MC3D_crop_tracker.py:1088)
    names, _, _, (P, Hm), (P2, H2) = gc.homography_inputs()
    cams = [names[0], names[7], names[13]]
    sel = [names.index(n) for n in cams]

    def make_hg(Pm, Hh):
        hg = hgm.Homography(device=str(dev))
        hg.correspondence = {n: {"P": Pm[i], "H": Hh[i], "H_inv": np.linalg.inv(Hh[i])} for n, i in zip(cams, sel)}
        hg.default_correspondence = cams[0]
        return hg

    class Tracker(mc3d_post.DetectionParser):
        pass
    me = Tracker()
    me.sigma_d, me.phi_nms_im, me.phi_nms_space = 0.3, 0.8, 0.6          # the tracker's parameters (MC3D_crop_tracker.py:62-87); loose
    # NMS thresholds so that more than a handful of these synthetic, heavily overlapping detections reach the transforms
    me.cameras, me.est_ts = cams, False
    me.hg = hgm.Homography_Wrapper(hg1=make_hg(P, Hm), hg2=make_hg(P2, H2))
    st, lb, sc, cm = me.parse_detections(s, c, b, im)                    # device tensors straight in: no .cpu() copies
    ref = otp.parse_detections(s.cpu(), c.cpu(), b.cpu(), im.cpu(), Hm[sel], H2[sel], P[sel], P2[sel], sigma_d=0.3,
                               phi_nms_im=0.8, phi_nms_space=0.6, perform_nms=True, refine_height=False)
    assert st.is_cuda and 20 < st.shape[0] <= s.numel(), st.shape
    assert np.array_equal(lb.cpu().numpy(), ref[1].numpy()) and np.array_equal(cm.cpu().numpy(), ref[3].numpy())
    assert np.array_equal(sc.cpu().numpy(), ref[2].numpy())
    assert np.allclose(st.cpu().numpy(), ref[0].numpy(), rtol=1e-5, atol=1e-4, equal_nan=True)

    # ---- link 4: back to the image through each state's camera (homography.py:479-488)
    im_d = me.hg.state_to_im(st, name=[cams[i] for i in cm.cpu().tolist()])
    sp = ohg.state_to_space(st.cpu().numpy())
    idx = np.array(sel)[cm.cpu().numpy()]
    im_o = ohg.wrapper_space_to_im(sp, P[idx], P2[idx])
    assert im_d.dtype == torch.float64
    assert np.allclose(im_d.cpu().numpy(), im_o, rtol=1e-9, atol=1e-9, equal_nan=True)
