#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference (/root/reference, read-only, never copied) is imported unmodified
behind three harness-side shims:

  1. a stub ``torchvision.ops.nms`` (torchvision is not installed) -- the build's own greedy NMS
     (oracle/boxes.py); NMS results are therefore "parity unpinned" and only the surrounding
     reference logic (threshold loops, per-class loops, offset trick) is pinned by these vectors;
  2. an empty stub ``cv2`` (only fitting/plotting code touches it);
  3. ``Tensor.cuda`` -> identity, except a leaf that requires grad returns a clone (a real ``.cuda()``
     returns a non-leaf copy; D/losses.py:310 writes into such a tensor in place).

Inputs come from ``retinanet_mi355x.synth`` (portable integer-hash generators), so the fixtures
hold OUTPUTS (and the few inputs built with libm/LAPACK).  This script refuses to run without
/root/reference and is never executed on the GPU box.

    python tools/make_golden.py                     # writes tests/golden/*.npz
    python tools/make_golden.py --out DIR [names]   # elsewhere (tests/test_golden_regen.py compares with the committed set)
"""
import hashlib
import importlib
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(REPO, "3d-playground_amd"))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from retinanet_mi355x import synth          # noqa: E402
from oracle import boxes as oboxes          # noqa: E402  (only its greedy_nms, as shim 1)


def install_shims():
    tv = types.ModuleType("torchvision")
    tv_ops = types.ModuleType("torchvision.ops")
    tv_ops.nms = oboxes.greedy_nms
    tv.ops = tv_ops
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tv_ops
    sys.modules["cv2"] = types.ModuleType("cv2")
    torch.Tensor.cuda = lambda self, *a, **k: self.clone() if (self.requires_grad and self.is_leaf) else self


def import_variant(which):
    """Import the reference's top-level ``retinanet`` package from D/ or R/ (same package name)."""
    for k in [k for k in sys.modules if k == "retinanet" or k.startswith("retinanet.")]:
        del sys.modules[k]
    root = os.path.join(REF, "pytorch_retinanet_detector_directional") if which == "dir" else REF
    sys.path.insert(0, root)
    try:
        model = importlib.import_module("retinanet.model")
        losses = importlib.import_module("retinanet.losses")
        utils = importlib.import_module("retinanet.utils")
        anchors = importlib.import_module("retinanet.anchors")
    finally:
        sys.path.remove(root)
    return model, losses, utils, anchors


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def t2n(t):
    return t.detach().cpu().numpy()


import golden_cases as gc                   # noqa: E402  (tests/golden_cases.py)


# ----------------------------------------------------------------------------- generators
def gen_anchors(anchors_mod):
    out = {}
    A = anchors_mod.Anchors()
    for (h, w) in [(64, 96), (72, 104), (112, 112), (96, 128), (512, 512)]:
        out["full_%dx%d" % (h, w)] = t2n(A(torch.zeros(1, 3, h, w)))
    for (h, w) in [(540, 960), (1080, 1920), (1081, 1917)]:
        a = t2n(A(torch.zeros(1, 3, h, w)))
        out["sha_%dx%d" % (h, w)] = np.array(sha(a))
        out["count_%dx%d" % (h, w)] = np.array(a.shape[1])
        out["sample_%dx%d" % (h, w)] = a[0, ::997].copy()
    np.savez_compressed(os.path.join(OUT, "anchors.npz"), **out)


def gen_losses(losses_dir_mod, losses_2d_mod, anchors_mod):
    H, W = gc.LOSS_HW
    anc = anchors_mod.Anchors()(torch.zeros(1, 3, H, W))
    out = {}
    # pairwise IoU + assignment (integers must be reproduced bit-exactly)
    ann = gc.loss_labels_dir()
    for j in range(ann.shape[0]):
        lab = ann[j][ann[j, :, 20] != -1]
        if lab.shape[0] == 0:
            continue
        xs, ys = lab[:, 0:16:2], lab[:, 1:16:2]
        env = torch.stack((xs.min(1).values, ys.min(1).values, xs.max(1).values, ys.max(1).values), 1)
        iou = losses_dir_mod.calc_iou(anc[0], env)
        m, a = torch.max(iou, dim=1)
        out["dir_iou_max_%d" % j] = t2n(m)
        out["dir_iou_arg_%d" % j] = t2n(a)
    # directional loss, forward + input gradients, all images and each image alone
    cls, reg = gc.loss_heads(12, 21)
    cls.requires_grad_(True)
    reg.requires_grad_(True)
    fl = losses_dir_mod.FocalLoss()
    l = fl(cls, reg, anc, ann.clone())
    (l[0] + 2.0 * l[1] + 3.0 * l[2]).sum().backward()
    out["dir_losses"] = np.array([float(x.detach()) for x in l], dtype=np.float32)
    out["dir_dcls"] = t2n(cls.grad)
    out["dir_dreg"] = t2n(reg.grad)
    for j in (0, 1, 3, 4):
        lj = fl(cls[j:j + 1].detach(), reg[j:j + 1].detach(), anc, ann[j:j + 1].clone())
        out["dir_losses_img%d" % j] = np.array([float(x) for x in lj], dtype=np.float32)
    # 2D loss
    ann2 = gc.loss_labels_2d()
    cls2, reg2 = gc.loss_heads(4, 31)
    cls2.requires_grad_(True)
    reg2.requires_grad_(True)
    fl2 = losses_2d_mod.FocalLoss()
    l2 = fl2(cls2, reg2, anc, ann2.clone())
    (l2[0] + 2.0 * l2[1]).sum().backward()
    out["2d_losses"] = np.array([float(x.detach()) for x in l2], dtype=np.float32)
    out["2d_dcls"] = t2n(cls2.grad)
    out["2d_dreg"] = t2n(reg2.grad)
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)


class _Stub(torch.nn.Module):
    """Stands in for a head: returns preset per-level slices so the reference's own post-process code
    runs on designed inputs."""

    def __init__(self, full, level_counts):
        super().__init__()
        self.chunks = list(torch.split(full, level_counts, dim=1))
        self.i = 0

    def forward(self, x):
        c = self.chunks[self.i % len(self.chunks)]
        self.i += 1
        return c


def gen_boxes(model_dir, utils_dir, model_2d, utils_2d):
    out = {}
    H, W = gc.POST_HW
    counts = gc.level_counts(H, W)
    img = torch.zeros(1, 3, H, W)

    def pack(prefix, s, c, b, im=None):
        """scores/classes in full; boxes as sha + every 8th row (NMS itself is parity-unpinned: what these
        vectors pin is the reference's threshold loop, per-class loop, offset trick and gather order)."""
        out[prefix + "_scores"], out[prefix + "_classes"] = t2n(s), t2n(c)
        out[prefix + "_boxes_sha"] = np.array(sha(t2n(b)))
        out[prefix + "_boxes_sample"] = t2n(b)[::8].copy()
        if im is not None:
            out[prefix + "_im"] = t2n(im)
    # --- directional single-frame and LOCALIZE
    net = model_dir.resnet18(num_classes=2)
    net.eval()
    cls, reg = gc.post_single_inputs()
    net.regressionModel = _Stub(reg, counts)
    net.classificationModel = _Stub(cls, counts)
    with torch.no_grad():
        boxes, cls_back = net(img, LOCALIZE=True)
        s, c, b = net(img)
    out["dir_decode_sha"] = np.array(sha(t2n(boxes)))
    out["dir_decode_sample"] = t2n(boxes)[0, ::53].copy()
    pack("dir_single", s, c, b)
    # --- MULTI_FRAME, B = 3
    cls3, reg3 = gc.post_multi_inputs()
    net = model_dir.resnet18(num_classes=3)
    net.eval()
    net.regressionModel = _Stub(reg3, counts)
    net.classificationModel = _Stub(cls3, counts)
    with torch.no_grad():
        s, c, b, im = net(torch.zeros(3, 3, H, W), MULTI_FRAME=True)
    pack("dir_multi", s, c, b, im)
    # --- 2D: decode + clip + per-class 0.05 threshold + NMS
    net2 = model_2d.resnet18(num_classes=3)
    net2.eval()
    cls2, reg2 = gc.post_2d_inputs()
    net2.regressionModel = _Stub(reg2, counts)
    net2.classificationModel = _Stub(cls2, counts)
    with torch.no_grad():
        boxes2, _ = net2(img, LOCALIZE=True)
        s, c, b = net2(img)
    out["2d_decode_clip_sample"] = t2n(boxes2)[0, ::7].copy()
    out["2d_decode_clip_sha"] = np.array(sha(t2n(boxes2)))
    pack("2d", s, c, b)
    np.savez_compressed(os.path.join(OUT, "boxes.npz"), **out)


def _grad_summary(model, out, tag, full_limit=4096):
    for name, p in model.named_parameters():
        g = p.grad
        if g is None:
            continue
        g = t2n(g)
        out["%s_gsum_%s" % (tag, name)] = np.array([g.sum(dtype=np.float64), np.abs(g).sum(dtype=np.float64),
                                                    np.sqrt((g.astype(np.float64) ** 2).sum())])
        if g.size <= full_limit:
            out["%s_g_%s" % (tag, name)] = g


def gen_model(model_dir, model_2d):
    out = {}
    for arch in ("resnet18", "resnet50"):
        sd, img, ann = gc.model_inputs(arch, directional=True)
        net = getattr(model_dir, arch)(num_classes=4)
        net.load_state_dict(sd)
        net.train()
        net.freeze_bn()
        l = net([img, ann.clone()])
        (l[0] + l[1] + l[2]).sum().backward()
        out["%s_dir_losses" % arch] = np.array([float(x.detach()) for x in l], dtype=np.float32)
        _grad_summary(net, out, "%s_dir" % arch)
        net.eval()
        with torch.no_grad():
            boxes, cls = net(img, LOCALIZE=True)
        out["%s_dir_boxes" % arch] = t2n(boxes)
        out["%s_dir_cls" % arch] = t2n(cls)
    # 2D twin, cfg1-like plumbing at small size
    sd, img, ann = gc.model_inputs("resnet18", directional=False)
    net = model_2d.resnet18(num_classes=4)
    net.load_state_dict(sd)
    net.train()
    net.freeze_bn()
    l = net([img, ann.clone()])
    (l[0] + l[1]).sum().backward()
    out["resnet18_2d_losses"] = np.array([float(x.detach()) for x in l], dtype=np.float32)
    _grad_summary(net, out, "resnet18_2d")
    net.eval()
    with torch.no_grad():
        boxes, cls = net(img, LOCALIZE=True)
    out["resnet18_2d_boxes"] = t2n(boxes)
    out["resnet18_2d_cls"] = t2n(cls)
    np.savez_compressed(os.path.join(OUT, "model.npz"), **out)


def gen_model_deep(model_dir):
    """The other two architectures callers build: resnet34 = the tracker's crop detector (MC3D_crop_tracker.py:1535, at its
    112x112 crop size, batch 2) and resnet101 (BASELINE configs[4]).  Own file so that model.npz stays as it was."""
    out = {}
    for arch, hw, batch in (("resnet34", gc.CROP_HW, 2), ("resnet101", gc.MODEL_HW, 1)):
        sd, img, ann = gc.model_inputs(arch, directional=True, hw=hw, batch=batch)
        net = getattr(model_dir, arch)(num_classes=4)
        net.load_state_dict(sd)
        net.train()
        net.freeze_bn()
        l = net([img, ann.clone()])
        (l[0] + l[1] + l[2]).sum().backward()
        out["%s_dir_losses" % arch] = np.array([float(x.detach()) for x in l], dtype=np.float32)
        _grad_summary(net, out, "%s_dir" % arch, full_limit=2048)
        net.eval()
        with torch.no_grad():
            boxes, cls = net(img, LOCALIZE=True)
        out["%s_dir_boxes" % arch] = t2n(boxes)
        out["%s_dir_cls" % arch] = t2n(cls)
    np.savez_compressed(os.path.join(OUT, "model_deep.npz"), **out)


def gen_homography():
    hgmod = ref_module_from_file("_reference_homography", "homography.py")
    out = {}
    names, state, cam_index, (Ps, Hs), (Ps2, Hs2) = gc.homography_inputs()
    out["P"], out["H"] = Ps, Hs

    def make_hg(P, H):
        hg = hgmod.Homography()
        hg.correspondence = {n: {"P": P[i], "H": H[i], "H_inv": np.linalg.inv(H[i])} for i, n in enumerate(names)}
        hg.default_correspondence = names[0]
        return hg
    hg = make_hg(Ps, Hs)
    cams = [names[i] for i in cam_index]
    out["state"] = t2n(state)
    out["cam_index"] = cam_index
    space = hg.state_to_space(state)
    out["space"] = t2n(space)
    out["space_to_state"] = t2n(hg.space_to_state(space))
    im_list = hg.state_to_im(state, name=cams)
    im_one = hg.state_to_im(state, name="p1c3")
    im_default = hg.state_to_im(state)
    out["im_list"], out["im_one"], out["im_default"] = t2n(im_list), t2n(im_one), t2n(im_default)
    heights = hg.guess_heights(["sedan", "semi", 3, "nonsense", "trailer", "truck (other)"])
    out["guess_heights"] = t2n(heights)
    h = state[:, 4]
    out["back_space_list"] = t2n(hg.im_to_space(im_list, name=cams, heights=h))
    out["back_state_list"] = t2n(hg.im_to_state(im_list, name=cams, heights=h))
    out["back_state_one"] = t2n(hg.im_to_state(im_one, name="p1c3", heights=h))
    out["height_from_template"] = t2n(hg.height_from_template(im_list, h, im_list * 1.07 + 3.0))
    # wrapper: second homography = perturbed cameras
    out["P2"], out["H2"] = Ps2, Hs2
    wr = hgmod.Homography_Wrapper(hg1=hg, hg2=make_hg(Ps2, Hs2))
    out["wr_im_list"] = t2n(wr.state_to_im(state, name=cams))
    out["wr_back_state_list"] = t2n(wr.im_to_state(t2n_t(out["wr_im_list"]), name=cams, heights=h))
    out["wr_im_one"] = t2n(wr.state_to_im(state, name="p2c4"))
    np.savez_compressed(os.path.join(OUT, "homography.npz"), **out)


def t2n_t(a):
    return torch.from_numpy(np.asarray(a))


def gen_csv_kat():
    """Known-answer rows from the reference's own result CSVs (data files): state columns ->
    space-corner columns (i24_state_to_space) and image-corner columns (space_to_im with the camera's P).
    A strided sample of rows is kept as a fixture; P is recovered per (camera, side of y=60) by DLT from the
    SAME rows inside the test (tests/test_homography_kat.py), so nothing but data is stored."""
    import csv
    rows = []
    for fi, fn in enumerate(("3D_tracking_results.csv", "working_3D_tracking_data.csv")):
        with open(os.path.join(REF, fn)) as f:
            rd = csv.reader(f)
            hdr = next(rd)
            col = {n: i for i, n in enumerate(hdr)}
            keep = ["fbrx", "fbry", "fblx", "fbly", "bbrx", "bbry", "bblx", "bbly", "ftrx", "ftry", "ftlx", "ftly",
                    "btrx", "btry", "btlx", "btly", "fbr_x", "fbr_y", "fbl_x", "fbl_y", "bbr_x", "bbr_y", "bbl_x",
                    "bbl_y", "direction", "veh rear x", "veh center y", "width", "length", "height"]
            n = 0
            for r in rd:
                if len(r) < len(hdr) - 1 or r[col["fbrx"]] == "":
                    continue
                n += 1
                if fn.startswith("3D_tracking") and n % 12:
                    continue
                cam = r[col["camera"]]
                rows.append([100.0 * fi + float(cam[1]) * 10 + float(cam[3])] + [float(r[col[k]]) for k in keep])
    arr = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "csv_kat.npz"), rows=arr,
                        columns=np.array(["cam_code"] + keep))


def gen_csv_rows():
    """A strided sample of the reference's shipped result files AS TEXT (data files, not source): the byte-level fixture
    for the row formatting of write_results_csv (MC3D_crop_tracker.py:1318-1453).  One header + sample per file."""
    for fn, stride, out_name in (("3D_tracking_results.csv", 59, "results_rows_3D_tracking_results.csv"),
                                 ("working_3D_tracking_data.csv", 3, "results_rows_working_3D_tracking_data.csv")):
        with open(os.path.join(REF, fn), newline="") as f:
            lines = f.read().split("\r\n")
        keep = [lines[0]] + [l for i, l in enumerate(lines[1:]) if l and i % stride == 0]
        with open(os.path.join(OUT, out_name), "w", newline="") as f:
            f.write("\r\n".join(keep) + "\r\n")


def tracker_import_shims():
    """Two more stub attributes the tracker module needs at import time."""
    tv = sys.modules["torchvision"]
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvt.functional = tvf
    tv.transforms = tvt
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.functional"] = tvf

    def _no_roi_align(*a, **k):
        raise NotImplementedError("roi_align is not on this path")
    sys.modules["torchvision.ops"].roi_align = _no_roi_align
    matplotlib_stub()


def matplotlib_stub():
    """util_track/kf.py imports matplotlib.pyplot at the top (unused by the class, absent here)."""
    if "matplotlib" not in sys.modules:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"], sys.modules["matplotlib.pyplot"] = mpl, mpl.pyplot


def gen_tracker_post():
    """MC_Crop_Tracker.parse_detections / im_nms / space_nms / md_iou (MC3D_crop_tracker.py:319-383, 592-636,
    1030-1049) run UNBOUND on a stand-in ``self`` that carries only the attributes those methods read (sigma_d,
    phi_nms_*, cameras, est_ts, hg = the reference's own Homography_Wrapper filled with the fixture's matrices).
    The module imports behind two more stub attributes (torchvision.transforms.functional, torchvision.ops.roi_align);
    constructing the tracker itself needs videos and checkpoints the reference does not ship."""
    trk, hgmod = import_reference_tracker()
    T = trk.MC_Crop_Tracker
    scores, labels, boxes, cams, names, (Ps, Hs), (Ps2, Hs2) = gc.tracker_post_inputs()

    def make_hg(P, H):
        hg = hgmod.Homography()
        hg.correspondence = {n: {"P": P[i], "H": H[i], "H_inv": np.linalg.inv(H[i])} for i, n in enumerate(names)}
        hg.default_correspondence = names[0]
        return hg
    me = types.SimpleNamespace(sigma_d=0.1, phi_nms_im=0.3, phi_nms_space=0.2, cameras=list(names), est_ts=False,
                               hg=hgmod.Homography_Wrapper(hg1=make_hg(Ps, Hs), hg2=make_hg(Ps2, Hs2)))
    me.im_nms = types.MethodType(T.im_nms, me)
    me.space_nms = types.MethodType(T.space_nms, me)
    out = {}
    for tag, kw in (("nms", dict(perform_nms=True, refine_height=False)),
                    ("nms_refine", dict(perform_nms=True, refine_height=True)),
                    ("plain", dict(perform_nms=False, refine_height=False))):
        st, lb, sc, cm = T.parse_detections(me, scores.clone(), labels.clone(), boxes.clone(), cams.clone(), **kw)
        out[tag + "_state"], out[tag + "_labels"] = t2n(st), t2n(lb)
        out[tag + "_scores"], out[tag + "_cams"] = t2n(sc), t2n(cm)
    keep = scores > 0.1
    det = boxes[keep].reshape(-1, 10, 2)[:, :8, :]
    out["im_nms_idx"] = t2n(T.im_nms(me, det, scores[keep], groups=cams[keep], threshold=0.3))
    out["im_nms_idx_nogroups"] = t2n(T.im_nms(me, det, scores[keep], threshold=0.3))
    st_plain = torch.from_numpy(out["plain_state"])
    out["space_nms_idx"] = t2n(T.space_nms(me, st_plain, torch.from_numpy(out["plain_scores"]), threshold=0.2))
    b4 = boxes[:64, 16:20].double()
    out["md_iou"] = t2n(T.md_iou(me, b4[None].repeat(64, 1, 1), b4[:, None].repeat(1, 64, 1)))
    empty = T.parse_detections(me, scores[:0], labels[:0], boxes[:0], cams[:0])
    low = T.parse_detections(me, scores * 0.01, labels, boxes, cams)
    out["empty_is_lists"] = np.array([all(isinstance(e, list) and len(e) == 0 for e in empty),
                                      all(isinstance(e, list) and len(e) == 0 for e in low)])
    np.savez_compressed(os.path.join(OUT, "tracker_post.npz"), **out)


def gen_crop_refine():
    """MC_Crop_Tracker.get_crop_boxes / local_to_global / select_best_box (MC3D_crop_tracker.py:920-1028) run unbound
    on a stand-in ``self`` (b, cs, W, device, hg, md_iou); same import shims as gen_tracker_post."""
    trk, hgmod = import_reference_tracker()
    T = trk.MC_Crop_Tracker
    from oracle import crop_refine as ocr          # only for the roi-free middle of the pipeline (top-k, homographies)
    pre_loc, cam, im_objs, names, (Ps, Hs), (Ps2, Hs2) = gc.crop_refine_inputs()

    def make_hg(P, H):
        hg = hgmod.Homography()
        hg.correspondence = {n: {"P": P[i], "H": H[i], "H_inv": np.linalg.inv(H[i])} for i, n in enumerate(names)}
        hg.default_correspondence = names[0]
        return hg
    me = types.SimpleNamespace(b=1.25, cs=112, W=0.5, device=torch.device("cpu"),
                               hg=hgmod.Homography_Wrapper(hg1=make_hg(Ps, Hs), hg2=make_hg(Ps2, Hs2)))
    me.md_iou = types.MethodType(T.md_iou, me)
    out = {}
    crop_boxes = T.get_crop_boxes(me, im_objs)
    out["crop_boxes"] = t2n(crop_boxes)
    reg_boxes, cls = gc.crop_detections(im_objs, crop_boxes)
    # as in track(): crop_boxes stays float64 (it comes from the float64 state_to_im), so the float32 detections are
    # promoted and the frame coordinates are float64 (MC3D_crop_tracker.py:1198, 1204)
    glob = T.local_to_global(me, reg_boxes.clone(), crop_boxes)
    out["local_to_global"] = t2n(glob)
    # the reference's own sequence between the detector and select_best_box (MC3D_crop_tracker.py:1188-1219), run with
    # the reference's Homography_Wrapper
    confs, classes = torch.max(cls, dim=2)
    top = torch.topk(confs, 50, dim=1)[1]
    rows = torch.arange(glob.shape[0]).unsqueeze(1).repeat(1, top.shape[1])
    g, confs, classes = glob[rows, top, :, :], confs[rows, top], classes[rows, top]
    n_objs = g.shape[0]
    cam_rep = [names[int(c)] for c in cam for _ in range(g.shape[1])]
    pts = g.reshape(-1, 8, 2)
    heights = me.hg.guess_heights(classes.reshape(-1))
    st = me.hg.im_to_state(pts, heights=heights, name=cam_rep)
    repro = me.hg.state_to_im(st, name=cam_rep)
    st = me.hg.im_to_state(pts, heights=me.hg.height_from_template(repro, heights, pts), name=cam_rep)
    out["cand_state"], out["cand_confs"], out["cand_classes"] = t2n(st), t2n(confs), t2n(classes)
    best, bcls, bconf = T.select_best_box(me, pre_loc.clone(), st.clone(), confs, classes, n_objs)
    out["best_state"], out["best_classes"], out["best_confs"] = t2n(best), t2n(bcls), t2n(bconf)
    np.savez_compressed(os.path.join(OUT, "crop_refine.npz"), **out)


def ref_module_from_file(alias, relpath):
    """A reference module loaded from its file, under a private name: this repository ships same-named drop-ins
    (util_track/kf.py, homography.py) that come first on sys.path, and a golden must come from the REFERENCE's code."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(alias, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REF) + os.sep)
    return mod


def import_reference_tracker():
    """MC3D_crop_tracker + homography with the reference checkout FIRST on sys.path, so that its own
    `from util_track.kf import ...` / `util_track.mp_loader` / `homography` resolve inside the reference (util_track is a
    namespace package there and a namespace portion here: whichever root comes first wins per submodule)."""
    for k in [k for k in sys.modules if k in ("homography", "MC3D_crop_tracker") or k == "util_track" or k.startswith("util_track.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        trk = importlib.import_module("MC3D_crop_tracker")
        hgmod = importlib.import_module("homography")
    finally:
        sys.path.remove(REF)
    for m in (trk, hgmod, sys.modules["util_track.kf"]):
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(REF) + os.sep), m.__file__
    for k in [k for k in sys.modules if k == "util_track" or k.startswith("util_track.") or k == "homography"]:
        del sys.modules[k]                          # later imports of these names are not to find the reference's
    return trk, hgmod


def gen_kf():
    """Torch_KF (util_track/kf.py) itself: add, predict with the default dt / a float dt / a per-object dt tensor, view,
    update.  The module imports matplotlib.pyplot at the top (unused by the class, absent here): stubbed."""
    matplotlib_stub()
    kfmod = ref_module_from_file("_reference_util_track_kf", "util_track/kf.py")
    INIT, det, directions, times, speed, upd_ids, z, dts = gc.kf_inputs()
    out = {}

    def snap(t):                                  # the class updates X, P and T in place: snapshots must be copies
        return t2n(t.clone())
    kf = kfmod.Torch_KF(torch.device("cpu"), INIT={k: v.clone() for k, v in INIT.items()}, ADD_MEAN_R=True)
    ids = list(range(100, 100 + len(det)))
    kf.add(det.clone(), ids, directions.clone(), times.clone())
    kf.X[:, 5] = speed
    out["X0"], out["P0"], out["T0"] = snap(kf.X), snap(kf.P), snap(kf.T)
    kf.predict()
    out["X1"], out["P1"], out["T1"] = snap(kf.X), snap(kf.P), snap(kf.T)
    kf.predict(dt=0.05)
    out["X2"], out["P2"], out["T2"] = snap(kf.X), snap(kf.P), snap(kf.T)
    kf.predict(dt=dts.clone())
    out["X3"], out["P3"], out["T3"] = snap(kf.X), snap(kf.P), snap(kf.T)
    _, v = kf.view(dt=dts.clone(), with_direction=True)
    out["view_dir"] = t2n(v)
    _, v = kf.view(dt=1 / 30.0)
    out["view_plain"] = t2n(v)
    kf.update(z.clone(), [ids[i] for i in upd_ids])
    out["X4"], out["P4"] = snap(kf.X), snap(kf.P)
    kf.remove([ids[0], ids[5]])
    out["X5"], out["T5"] = snap(kf.X), snap(kf.T)
    out["ids5"] = np.array(kf.view()[0])
    # the default constructor (diagonal matrices, H sees 4 of 5 measurements: kf.py:60-68)
    kd = kfmod.Torch_KF(torch.device("cpu"))
    kd.add(det.clone(), ids, directions.clone(), times.clone())
    kd.predict()
    kd.update(z.clone(), [ids[i] for i in upd_ids])
    out["Xd"], out["Pd"] = snap(kd.X), snap(kd.P)
    np.savez_compressed(os.path.join(OUT, "kf.npz"), **out)


def main():
    if not os.path.isdir(REF):
        sys.exit("make_golden.py needs the reference checkout at %s (build container only)" % REF)
    global OUT
    argv = sys.argv[1:]
    if "--out" in argv:
        i = argv.index("--out")
        OUT = os.path.abspath(argv[i + 1])
        del argv[i:i + 2]
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_shims()
    m_dir, l_dir, u_dir, a_dir = import_variant("dir")
    dir_mods = (m_dir, l_dir, u_dir, a_dir)
    m_2d, l_2d, u_2d, a_2d = import_variant("2d")
    which = set(argv) or {"anchors", "losses", "boxes", "model", "model_deep", "homography", "csv", "csv_rows", "tracker_post", "crop_refine", "kf"}
    if "anchors" in which:
        gen_anchors(a_dir)
    if "losses" in which:
        gen_losses(l_dir, l_2d, a_dir)
    if "boxes" in which:
        gen_boxes(m_dir, u_dir, m_2d, u_2d)
    if "model" in which:
        gen_model(m_dir, m_2d)
    if "model_deep" in which:
        gen_model_deep(m_dir)
    if "homography" in which:
        gen_homography()
    if "csv" in which:
        gen_csv_kat()
    if "csv_rows" in which:
        gen_csv_rows()
    if "tracker_post" in which or "crop_refine" in which:
        tracker_import_shims()
    if "tracker_post" in which:
        gen_tracker_post()
    if "crop_refine" in which:
        gen_crop_refine()
    if "kf" in which:
        gen_kf()
    for fn in sorted(os.listdir(OUT)):
        print("%-20s %8.1f KiB" % (fn, os.path.getsize(os.path.join(OUT, fn)) / 1024))
    del dir_mods


if __name__ == "__main__":
    main()
