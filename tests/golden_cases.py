"""Input builders shared by tools/make_golden.py (which feeds them to the reference) and the tests
(which feed them to the oracle and to the HIP path).  Built only from the portable generators of
``retinanet_mi355x.synth``, so fixtures hold outputs only."""
import numpy as np
import torch

from retinanet_mi355x import synth

LOSS_HW = (96, 128)          # A = 2313
POST_HW = (256, 320)         # A = 15354 > 10000: the adaptive-threshold loops iterate
MODEL_HW = (72, 104)         # odd pyramid sizes: both FPN crop fall-backs are taken (D/model.py:92-108)


def level_counts(height, width):
    return [9 * ((height + 2 ** l - 1) // 2 ** l) * ((width + 2 ** l - 1) // 2 ** l) for l in (3, 4, 5, 6, 7)]


def num_anchors(height, width):
    return sum(level_counts(height, width))


def loss_labels_dir():
    """5 images: normal / ragged / no labels / duplicate+outside+degenerate / labels but no positives."""
    H, W = LOSS_HW
    ann = synth.labels_dir(5, 6, H, W, num_classes=4, seed=11, size_px=(28, 70))
    ann[1, 3:] = -1
    ann[2] = -1
    ann[3, 1] = ann[3, 0]
    ann[3, 1, 20] = (ann[3, 0, 20] + 1) % 4            # same box, other class: tie -> lowest index wins
    ann[3, 2, 0:20:2] += 400.0                         # fully outside the image
    ann[3, 3, :20] = 50.0                              # degenerate zero-size box
    ann[4] = synth.labels_dir(1, 6, H, W, num_classes=4, seed=12, size_px=(3, 6))[0]
    return ann


def loss_labels_2d():
    H, W = LOSS_HW
    ann = synth.labels_2d(5, 6, H, W, num_classes=4, seed=13, size_px=(28, 70))
    ann[1, 3:] = -1
    ann[2] = -1
    ann[3, 1] = ann[3, 0]
    ann[3, 1, 4] = (ann[3, 0, 4] + 1) % 4
    ann[3, 2, 0:4:2] += 400.0
    ann[3, 3, 2:4] = ann[3, 3, 0:2] + 0.25             # w,h < 1 -> clamp(min=1) path (R/losses.py:143-144)
    ann[4] = synth.labels_2d(1, 6, H, W, num_classes=4, seed=14, size_px=(3, 6))[0]
    return ann


def loss_heads(n_reg, seed):
    """cls = u^6 (reaches below 1e-4 and close to 1: both clamp edges), reg ~ N(0, 0.3)."""
    A = num_anchors(*LOSS_HW)
    cls = synth.scores_portable(5, A, 4, seed, power=6)
    reg = torch.from_numpy(synth.normal((5, A, n_reg), seed + 1, 0.3))
    return cls, reg


def post_reg_dir(batch, seed):
    """Regression whose 2D-box part (cols 8:12) decodes to a jittered copy of the anchor, so NMS has overlaps."""
    A = num_anchors(*POST_HW)
    reg = synth.normal((batch, A, 12), seed, 0.4)
    jit = synth.normal((batch, A, 4), seed + 1, 0.12)
    reg[:, :, 8:12] = jit + np.array([-0.5, -0.5, 0.5, 0.5], dtype=np.float32)
    return torch.from_numpy(reg)


def post_single_inputs():
    A = num_anchors(*POST_HW)
    return synth.scores_portable(1, A, 2, 41, power=3), post_reg_dir(1, 42)


def post_multi_inputs():
    A = num_anchors(*POST_HW)
    return synth.scores_portable(3, A, 3, 43, power=6), post_reg_dir(3, 44)


def post_2d_inputs():
    A = num_anchors(*POST_HW)
    return synth.scores_portable(1, A, 3, 45, power=6), torch.from_numpy(synth.normal((1, A, 4), 46, 1.0))


CROP_HW = (112, 112)         # the tracker's crop detector input (train_crop_detector.py CROP=112, MC3D_crop_tracker.py:1185)


# architecture -> (golden file, input size, batch) of the whole-model fixtures (tools/make_golden.py gen_model / gen_model_deep)
MODEL_CASES = {"resnet18": ("model", None, None), "resnet50": ("model", None, None),
               "resnet34": ("model_deep", CROP_HW, 2), "resnet101": ("model_deep", MODEL_HW, 1)}


def model_case(arch, directional=True):
    fn, hw, batch = MODEL_CASES[arch]
    return (fn,) + model_inputs(arch, directional, hw=hw, batch=batch)


def model_inputs(arch, directional=True, hw=None, batch=None):
    H, W = MODEL_HW if hw is None else hw
    B = (2 if arch == "resnet18" else 1) if batch is None else batch
    sd = synth.state_dict(arch, num_classes=4, n_reg=12 if directional else 4, seed=7, head_scale=3e-4)
    img = synth.frames(B, H, W, seed=8)
    if directional:
        ann = synth.labels_dir(B, 5, H, W, num_classes=4, seed=9, size_px=(24, 60))
        ann[0, 4] = -1
    else:
        ann = synth.labels_2d(B, 5, H, W, num_classes=4, seed=9, size_px=(24, 60))
        ann[0, 4] = -1
    return sd, img, ann


def homography_inputs():
    Ps, Hs = synth.camera_matrices(18, seed=5)
    Ps2, Hs2 = synth.camera_matrices(18, seed=55)
    names = ["p%dc%d" % (p, c) for p in (1, 2, 3) for c in range(1, 7)]
    state = synth.vehicle_states(90, seed=6)
    cam_index = np.arange(90) % 18
    return names, state, cam_index, (Ps, Hs), (Ps2, Hs2)


def tracker_post_inputs(n_obj=120, seed=21):
    """Detector output as the tracker receives it (MC3D_crop_tracker.py:1074): scores [d], class labels [d] i64,
    boxes [d,20] (16 image corner coords + 2D box) and camera index [d], d = 4 * n_obj + low-score clutter.  Each
    vehicle is seen by its own camera three times (pixel jitter: im_nms fodder) and once by a neighbouring camera
    (space_nms fodder); clutter sits below sigma_d.  Built from the portable generators + the fixture's matrices."""
    from oracle import homography as ohg        # only to build inputs; tests and the golden script both run on CPU
    names, _, _, (Ps, Hs), (Ps2, Hs2) = homography_inputs()
    n_cam = len(names)
    state = synth.vehicle_states(n_obj, seed=seed).numpy()
    cam = (synth.uniform((n_obj,), seed + 1) * n_cam).astype(np.int64) % n_cam
    dets, cams, scores = [], [], []
    u = synth.uniform((n_obj, 4, 16), seed + 2).astype(np.float64)
    sc = synth.uniform((n_obj, 4), seed + 3)
    for k in range(4):
        c = cam if k < 3 else (cam + 1) % n_cam
        im = ohg.wrapper_space_to_im(ohg.state_to_space(state), Ps[c], Ps2[c]).reshape(n_obj, 16)
        dets.append(im + (u[:, k] - 0.5) * (6.0 if k < 3 else 2.0))
        cams.append(c)
        scores.append(0.15 + 0.84 * sc[:, k])
    det = np.concatenate(dets).astype(np.float32)
    cams = np.concatenate(cams)
    scores = np.concatenate(scores).astype(np.float32)
    n_cl = n_obj // 2                                                        # clutter below the confidence cutoff
    det = np.concatenate((det, det[:n_cl] + 40.0))
    cams = np.concatenate((cams, cams[:n_cl]))
    scores = np.concatenate((scores, (0.09 * synth.uniform((n_cl,), seed + 4)).astype(np.float32)))
    perm = np.argsort(synth.uniform((len(scores),), seed + 5), kind="stable")   # detector order is not grouped
    det, cams, scores = det[perm], cams[perm], scores[perm]
    xs, ys = det[:, 0:16:2], det[:, 1:16:2]
    box2d = np.stack((xs.min(1), ys.min(1), xs.max(1), ys.max(1)), 1)
    boxes = np.concatenate((det, box2d), 1).astype(np.float32)
    labels = (synth.uniform((len(scores),), seed + 6) * 8).astype(np.int64) % 8
    return (torch.from_numpy(scores), torch.from_numpy(labels), torch.from_numpy(boxes), torch.from_numpy(cams),
            names, (Ps, Hs), (Ps2, Hs2))


def crop_refine_inputs(n_obj=24, n_det=64, seed=41, cs=112):
    """Inputs of the tracker's crop-refinement path (MC3D_crop_tracker.py:1172-1226): priors pre_loc [n,6] in state
    space with their cameras, the image corners of the priors (state_to_im), and a stand-in for the LOCALIZE
    detector's output inside each crop: reg_boxes [n,d,20] (crop pixels: the prior's own box plus jitter) and class
    scores cls [n,d,8]."""
    from oracle import homography as ohg
    names, _, _, (Ps, Hs), (Ps2, Hs2) = homography_inputs()
    n_cam = len(names)
    pre_loc = synth.vehicle_states(n_obj, seed=seed)
    cam = torch.from_numpy((synth.uniform((n_obj,), seed + 1) * n_cam).astype(np.int64) % n_cam)
    im_objs = torch.from_numpy(ohg.wrapper_space_to_im(ohg.state_to_space(pre_loc.numpy()), Ps[cam.numpy()], Ps2[cam.numpy()]))
    return pre_loc, cam, im_objs, names, (Ps, Hs), (Ps2, Hs2)


def crop_detections(im_objs, crop_boxes, n_det=64, seed=43, cs=112):
    """Synthetic LOCALIZE output for each crop: the object's own corners mapped into crop pixels, jittered per detection."""
    n = im_objs.shape[0]
    scale = (crop_boxes[:, 2] - crop_boxes[:, 0]).double()
    local = (im_objs.double() - crop_boxes[:, None, 0:2].double()) / scale[:, None, None] * cs        # [n,8,2]
    jit = torch.from_numpy(synth.uniform((n, n_det, 8, 2), seed).astype(np.float64) - 0.5) * 10.0
    det = (local[:, None] + jit).float()                                                              # [n,d,8,2]
    xs, ys = det[..., 0], det[..., 1]
    box2d = torch.stack((xs.min(2).values, ys.min(2).values, xs.max(2).values, ys.max(2).values), dim=2)
    reg_boxes = torch.cat((det.reshape(n, n_det, 16), box2d), dim=2).contiguous()                       # [n,d,20]
    cls = torch.from_numpy(synth.uniform((n, n_det, 8), seed + 1).astype(np.float32))
    cls = cls * cls                                                                                    # skewed scores, few ties
    return reg_boxes, cls


def kf_inputs(n=40, seed=61):
    """A populated tracker filter (util_track/kf.py): INIT matrices (F, H, SPD Q / R / P0, offsets), n objects with
    direction and time stamps, a subset to update with measurements, per-object dt."""
    def spd(k, sd, scale):
        a = synth.uniform((k, k), sd).astype(np.float64) - 0.5
        return torch.from_numpy((a @ a.T * scale + np.eye(k) * scale).astype(np.float32))
    F = torch.eye(6)
    H = torch.zeros(5, 6)
    H[:5, :5] = torch.eye(5)
    H[0, 5] = 0.03                                             # a measurement model that also sees the speed a little
    INIT = {"P": spd(6, seed, 20.0), "F": F, "H": H, "Q": spd(6, seed + 1, 0.5), "R": spd(5, seed + 2, 2.0),
            "mu_Q": torch.zeros(6), "mu_R": torch.from_numpy((synth.uniform((5,), seed + 3) - 0.5).astype(np.float32))}
    st = synth.vehicle_states(n, seed=seed + 4)
    det = st[:, :5].clone()                                    # x, y, l, w, h
    directions = st[:, 5].clone()
    times = torch.from_numpy(synth.uniform((n,), seed + 5).astype(np.float64) * 0.2 + 10.0)
    speed = torch.from_numpy((60 + 60 * synth.uniform((n,), seed + 6)).astype(np.float32))
    upd_ids = [i for i in range(n) if i % 3 != 1]
    z = det[upd_ids] + torch.from_numpy((synth.uniform((len(upd_ids), 5), seed + 7) - 0.5).astype(np.float32)) * 3.0
    dts = torch.from_numpy(synth.uniform((n,), seed + 8).astype(np.float64) * 0.08 + 0.01)
    return INIT, det, directions, times, speed, upd_ids, z, dts
