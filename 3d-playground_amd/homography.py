"""Drop-in for the transform half of the reference's ``homography.py``: ``Homography`` and
``Homography_Wrapper`` with the same attributes (``correspondence[name] = {"H","H_inv","P",...}`` numpy float64,
``default_correspondence``, ``class_heights``) and the same methods for the per-frame hot path
(homography.py:274-333, 388-551, 793-862): state_to_space, space_to_state, space_to_im, im_to_space,
state_to_im, im_to_state, guess_heights, height_from_template; ``name`` may be None, a str or a list of names.

Objects are plain picklable classes (the reference pickles them, homography.py:22-23, 75-76) holding only
numpy matrices; inputs and outputs are CPU tensors exactly as every caller of the reference passes and expects
(state fp32, image points fp64), while the arithmetic runs in the HIP kernels of libretinanet_mi355x.so on
``device`` (default cuda:0).  A device tensor input stays on its device.

Out of scope here: correspondence fitting (``add_correspondence`` needs cv2.findHomography),
vanishing-point estimation, ``scale_Z``, CSV loading and plotting -- one-off set-up code that needs OpenCV and
data files the reference does not ship.  Populate ``correspondence`` directly, as the tracker does
(MC3D_crop_tracker.py:1561).
"""
import numpy as np
import torch

from retinanet_mi355x import ops as _ops


class Homography():
    def __init__(self, f1=None, f2=None, device="cuda:0"):
        if f1 is not None or f2 is not None:
            raise NotImplementedError("custom state<->space functions run as Python in the reference "
                                      "(homography.py:180-186); only the built-in I-24 formulation has kernels")
        self.device = device
        self.correspondence = {}
        self.default_correspondence = None
        self.class_heights = {                      # homography.py:191-202
            "sedan": 4, "midsize": 5, "van": 6, "pickup": 5, "semi": 12, "truck (other)": 12, "truck": 12,
            "motorcycle": 4, "trailer": 3, "other": 5,
        }
        self.class_dims = {                         # homography.py:205-216
            "sedan": [16, 6, 4], "midsize": [18, 6.5, 5], "van": [20, 6, 6.5], "pickup": [20, 6, 5],
            "semi": [55, 9, 12], "truck (other)": [25, 9, 12], "truck": [25, 9, 12], "motorcycle": [7, 3, 4],
            "trailer": [16, 7, 3], "other": [18, 6.5, 5],
        }
        names = ["sedan", "midsize", "van", "pickup", "semi", "truck (other)", "motorcycle", "trailer"]
        self.class_dict = {n: i for i, n in enumerate(names)}          # homography.py:218-235 (both directions)
        self.class_dict["truck"] = 5
        self.class_dict.update({i: n for i, n in enumerate(names)})

    # ---- plumbing
    def _dev(self, t):
        return t.device if t.is_cuda else torch.device(self.device)

    def _matrices(self, key, name, dev):
        """(stacked fp64 matrices on device, per-object int32 index or None)."""
        if name is None:
            name = self.default_correspondence
        if isinstance(name, list):
            uniq = sorted(set(name))
            pos = {n: i for i, n in enumerate(uniq)}
            mats = np.stack([np.asarray(self.correspondence[n][key], dtype=np.float64) for n in uniq])
            idx = torch.tensor([pos[n] for n in name], dtype=torch.int32, device=dev)
            return torch.from_numpy(mats).to(dev), idx
        mats = np.asarray(self.correspondence[name][key], dtype=np.float64)[None]
        return torch.from_numpy(np.ascontiguousarray(mats)).to(dev), None

    @staticmethod
    def _back(out, like):
        return out if like.is_cuda else out.cpu()

    # ---- state <-> space (homography.py:274-333)
    def i24_state_to_space(self, points):
        return self._back(_ops.hg_state_to_space(points.to(self._dev(points))), points)

    def i24_space_to_state(self, points):
        return self._back(_ops.hg_space_to_state(points.to(self._dev(points))), points)

    def state_to_space(self, points):
        return self.i24_state_to_space(points)

    def space_to_state(self, points):
        return self.i24_space_to_state(points)

    # ---- space <-> image (homography.py:388-476)
    def im_to_space(self, points, name=None, heights=None):
        if heights is None:
            print("No heights were input")              # homography.py:430-432
            return
        dev = self._dev(points)
        H, idx = self._matrices("H", name, dev)
        self._need8(points, idx)
        return self._back(_ops.hg_from_im(points.to(dev), heights.to(dev), H, None, idx, to_state=False), points)

    def space_to_im(self, points, name=None):
        dev = self._dev(points)
        P, idx = self._matrices("P", name, dev)
        self._need8(points, idx)
        return self._back(_ops.hg_to_im(points.to(dev), P, None, idx, from_state=False), points)

    def state_to_im(self, points, name=None):
        dev = self._dev(points)
        P, idx = self._matrices("P", name, dev)
        return self._back(_ops.hg_to_im(points.to(dev), P, None, idx, from_state=True), points)

    def im_to_state(self, points, name=None, heights=None):
        if heights is None:
            print("No heights were input")
            return self.space_to_state(None)            # the reference fails the same way (None.shape)
        dev = self._dev(points)
        H, idx = self._matrices("H", name, dev)
        self._need8(points, idx)
        return self._back(_ops.hg_from_im(points.to(dev), heights.to(dev), H, None, idx, to_state=True), points)

    @staticmethod
    def _need8(points, idx):
        if points.dim() != 3 or points.shape[1] != 8:
            raise RuntimeError("the box transforms take 8 points per object (homography.py:405, 459 hard-code 8); "
                               "got %s" % (tuple(points.shape),))

    # ---- heights (homography.py:502-551)
    def guess_heights(self, classes):
        heights = torch.zeros(len(classes))
        for i in range(len(classes)):
            try:
                heights[i] = self.class_heights[classes[i]]
            except (KeyError, TypeError):
                heights[i] = self.class_heights["other"]
        return heights

    def height_from_template(self, template_boxes, template_space_heights, boxes):
        def im_height(b):
            top = torch.mean(b[:, 4:8, :], dim=1)
            bottom = torch.mean(b[:, 0:4, :], dim=1)
            return torch.sum(torch.sqrt(torch.pow(top - bottom, 2)), dim=1)
        return im_height(boxes) / (im_height(template_boxes) / template_space_heights)


class Homography_Wrapper():
    """Two homographies, one per travel direction; objects whose corner-0 space y > 60 use the second
    (homography.py:793-862)."""

    def __init__(self, hg1=None, hg2=None):
        if hg1 is None or hg2 is None:
            raise RuntimeError("the default constructor unpickles EB_homography2.cpkl / WB_homography2.cpkl, which the "
                               "reference does not ship (homography.py:824-826): pass two populated Homography objects")
        self.hg1 = hg1
        self.hg2 = hg2

    def guess_heights(self, classes):
        return self.hg1.guess_heights(classes)

    def state_to_space(self, points):
        return self.hg1.state_to_space(points)

    def space_to_state(self, points):
        return self.hg1.space_to_state(points)

    def height_from_template(self, template_boxes, template_space_heights, boxes):
        return self.hg1.height_from_template(template_boxes, template_space_heights, boxes)

    def _pair(self, key, name, dev):
        m1, idx = self.hg1._matrices(key, name, dev)
        m2, idx2 = self.hg2._matrices(key, name, dev)
        if m1.shape != m2.shape:
            raise RuntimeError("hg1 and hg2 must hold the same set of correspondence names (homography.py:821)")
        return m1, m2, idx

    def im_to_space(self, points, name=None, heights=None):
        dev = self.hg1._dev(points)
        H1, H2, idx = self._pair("H", self.hg1.default_correspondence if name is None else name, dev)
        return Homography._back(_ops.hg_from_im(points.to(dev), heights.to(dev), H1, H2, idx, to_state=False), points)

    def space_to_im(self, points, name=None):
        dev = self.hg1._dev(points)
        P1, P2, idx = self._pair("P", self.hg1.default_correspondence if name is None else name, dev)
        return Homography._back(_ops.hg_to_im(points.to(dev), P1, P2, idx, from_state=False), points)

    def im_to_state(self, points, name=None, heights=None):
        dev = self.hg1._dev(points)
        H1, H2, idx = self._pair("H", self.hg1.default_correspondence if name is None else name, dev)
        return Homography._back(_ops.hg_from_im(points.to(dev), heights.to(dev), H1, H2, idx, to_state=True), points)

    def state_to_im(self, points, name=None):
        dev = self.hg1._dev(points)
        P1, P2, idx = self._pair("P", self.hg1.default_correspondence if name is None else name, dev)
        return Homography._back(_ops.hg_to_im(points.to(dev), P1, P2, idx, from_state=True), points)
