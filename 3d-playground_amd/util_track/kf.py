"""Drop-in for the reference's ``util_track/kf.py``: ``Torch_KF`` with the same constructor, attributes (``X``, ``P``,
``D``, ``T``, ``obj_idxs``, ``F``, ``H``, ``Q``, ``R``, ``mu_R``, ``P0``, ``dt_default`` ...) and methods (``get_dt``,
``add``, ``remove``, ``view``, ``predict``, ``update``, ``objs``), the per-object tensor algebra of view / predict /
update running in libretinanet_mi355x.so (``rn_kf_view / rn_kf_predict / rn_kf_update``, one lane per object) on the
filter's device instead of a chain of repeat / bmm / inverse calls.  Bookkeeping (the id -> row dictionary, add,
remove) is host Python exactly as in the reference.  The kernels are written for the tracker's filter: 6 states
(x, y, l, w, h, v), 5 measurements; other INIT shapes raise.

The reference keeps this filter on the CPU (MC3D_crop_tracker.py:103); here it is meant to live next to the detector's
outputs on the GPU.  CPU tensors handed to ``add`` / ``update`` are moved to the device.
"""
import numpy as np
import torch

from retinanet_mi355x import _hip


def _p(t):
    return t.data_ptr()


class Torch_KF(object):
    def __init__(self, device, state_err=10000, meas_err=1, mod_err=1, INIT=None, ADD_MEAN_Q=False, ADD_MEAN_R=False):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("this Torch_KF runs its algebra in HIP kernels: give it a GPU device (the reference's own "
                               "class is the CPU path)")
        self.meas_size, self.state_size = 5, 6
        self.dt_default = 1 / 30.0
        self.device = device
        self.X = self.P = self.D = self.T = None
        self.obj_idxs = {}
        if INIT is None:                                      # kf.py:60-68
            self.P0 = torch.eye(6).unsqueeze(0) * state_err
            self.F = torch.eye(6).float()
            self.H = torch.zeros(5, 6)
            self.H[:4, :4] = torch.eye(4)
            self.Q = torch.eye(6).unsqueeze(0) * mod_err
            self.R = torch.eye(5).unsqueeze(0) * meas_err
            self.R2 = torch.eye(5).unsqueeze(0) * meas_err
            self.mu_Q = torch.zeros([1, 6])
            self.mu_R = torch.zeros([1, 5])
        else:                                                 # kf.py:72-104
            self.P0 = INIT["P"].unsqueeze(0)
            self.F, self.H = INIT["F"], INIT["H"]
            self.Q, self.R = INIT["Q"].unsqueeze(0), INIT["R"].unsqueeze(0)
            self.mu_Q, self.mu_R = INIT["mu_Q"].unsqueeze(0), INIT["mu_R"].unsqueeze(0)
            for k in (2, 3):
                if "R%d" % k in INIT:
                    setattr(self, "R%d" % k, INIT["R%d" % k].unsqueeze(0).to(device).float())
                    setattr(self, "mu_R%d" % k, INIT["mu_R%d" % k].unsqueeze(0).to(device).float())
                    setattr(self, "H%d" % k, INIT["H%d" % k].to(device).float())
            for k in ("mu_v", "class_size", "class_covariance"):
                if k in INIT:
                    setattr(self, k, INIT[k])
            if tuple(self.F.shape) != (6, 6) or tuple(self.H.shape) != (5, 6):
                raise RuntimeError("the HIP filter is built for 6 states / 5 measurements, got F %s H %s"
                                   % (tuple(self.F.shape), tuple(self.H.shape)))
            if not ADD_MEAN_Q:
                self.mu_Q = torch.zeros([1, 6])
            if not ADD_MEAN_R:
                self.mu_R = torch.zeros([1, 5])
        for k in ("F", "H", "Q", "R", "P0", "mu_Q", "mu_R"):
            setattr(self, k, getattr(self, k).to(device).float().contiguous())
        _hip.load()

    # ---- host bookkeeping, as in the reference
    def get_dt(self, target_time, idxs=None, use_default=True):          # kf.py:120-155
        if self.X is None or len(self.X) == 0:
            return None
        if type(target_time) == float:
            return target_time - self.T
        if type(target_time) == list:
            target_time = torch.tensor(target_time, dtype=torch.double, device=self.device)
            if idxs is None:
                return target_time - self.T
            dt = torch.zeros(len(self.X), device=self.device)
            dt = dt + self.dt_default if use_default else dt
            ii = torch.as_tensor(list(idxs), dtype=torch.long, device=self.device)
            dt[ii] = (target_time[:len(idxs)] - self.T[ii]).to(dt.dtype)
            return dt
        return target_time.to(self.device) - self.T

    def add(self, detections, obj_ids, directions, times, init_speed=False, classes=None):   # kf.py:159-228
        dev = self.device

        def t(a):
            return (torch.from_numpy(a) if isinstance(a, np.ndarray) else a).to(dev)
        det = t(detections)
        newX = torch.zeros((len(det), 6), device=dev)
        if det.shape[1] == 5:
            newX[:, :5] = det
        else:
            newX = det.float()
        newD, newT = t(directions), t(times)
        if init_speed:
            newX[:, -1] = self.mu_v.repeat(len(det)).to(dev)
        newP = self.P0.repeat(len(obj_ids), 1, 1)
        if classes is not None:
            for i in range(len(newX)):
                newX[i, 2:5] = self.class_size[classes[i]]
                newP[i, 2:5, 2:5] = self.class_covariance[classes[i]]
        if self.X is not None and len(self.X) > 0:
            new_idx = len(self.X)
            self.X = torch.cat((self.X, newX.float()), dim=0)
            self.P = torch.cat((self.P, newP), dim=0)
            self.D = torch.cat((self.D, newD.float()), dim=0)
            self.T = torch.cat((self.T, newT.double()), dim=0)
        else:
            new_idx = 0
            self.X, self.P = newX.float(), newP.float()
            self.D, self.T = newD.float(), newT.double()
        for i, oid in enumerate(obj_ids):
            self.obj_idxs[oid] = new_idx + i

    def remove(self, obj_ids):                                            # kf.py:230-262
        if self.X is None:
            return
        keepers = list(range(len(self.X)))
        for oid in obj_ids:
            keepers.remove(self.obj_idxs[oid])
            self.obj_idxs[oid] = None
        keepers.sort()
        k = torch.as_tensor(keepers, dtype=torch.long, device=self.device)
        self.X, self.P, self.D, self.T = self.X[k], self.P[k], self.D[k], self.T[k]
        new_id, removals = 0, []
        for oid in self.obj_idxs:
            if self.obj_idxs[oid] is not None:
                self.obj_idxs[oid] = new_id
                new_id += 1
            else:
                removals.append(oid)
        for oid in removals:
            del self.obj_idxs[oid]

    # ---- the algebra, in HIP
    def _dt(self, dt):
        """-> (fp64 device tensor, is_tensor flag); a Python number keeps the reference's all-float32 path."""
        if isinstance(dt, torch.Tensor):
            if dt.numel() != len(self.X):
                raise RuntimeError("dt has %d entries for %d objects" % (dt.numel(), len(self.X)))
            return dt.to(self.device).double().contiguous(), 1
        return torch.tensor([float(dt)], dtype=torch.float64, device=self.device), 0

    def view(self, dt=None, with_direction=False):                         # kf.py:264-289
        if self.X is None or len(self.X) == 0:
            return [], []
        n = len(self.X)
        self.X, self.D = self.X.contiguous(), self.D.float().contiguous()
        out = torch.empty((n, 7 if with_direction else 6), dtype=torch.float32, device=self.device)
        dtt, flag = (None, 0) if dt is None else self._dt(dt)
        with torch.cuda.device(self.device):
            _hip.check(_hip.load().rn_kf_view(_p(self.X), _p(self.D), _p(self.F), _hip.ptr(dtt), flag, int(bool(with_direction)),
                                              _p(out), n, _hip.stream()), "rn_kf_view")
        inverted = dict([(self.obj_idxs[key], key) for key in self.obj_idxs.keys()])
        return [inverted[i] for i in range(n)], out

    def predict(self, dt=None):                                            # kf.py:292-332
        if self.X is None or len(self.X) == 0:
            return
        dtt, flag = self._dt(self.dt_default if dt is None else dt)
        self.X, self.P = self.X.float().contiguous(), self.P.float().contiguous()
        self.D, self.T = self.D.float().contiguous(), self.T.double().contiguous()
        with torch.cuda.device(self.device):
            _hip.check(_hip.load().rn_kf_predict(_p(self.X), _p(self.P), _p(self.D), _p(self.T), _p(self.F), _p(self.Q), _p(dtt),
                                                 flag, float(self.dt_default), len(self.X), _hip.stream()), "rn_kf_predict")

    def update(self, detections, obj_ids, measurement_idx=1):              # kf.py:335-403
        if measurement_idx == 1:
            mu_R, H, R = self.mu_R, self.H, self.R
        elif measurement_idx in (2, 3):
            mu_R, H, R = (getattr(self, "%s%d" % (k, measurement_idx)) for k in ("mu_R", "H", "R"))
        else:
            print("This measurement index does not exist in this filter")
            raise ValueError
        rows = [self.obj_idxs[oid] for oid in obj_ids]
        if len(set(rows)) != len(rows):
            raise RuntimeError("update: an object id appears twice (the reference lets the last write win; the kernel "
                               "updates rows in parallel)")
        z = (torch.from_numpy(detections) if isinstance(detections, np.ndarray) else detections).to(self.device).double().contiguous()
        if len(rows) == 0:
            return
        r = torch.as_tensor(rows, dtype=torch.int32, device=self.device)
        self.X, self.P = self.X.float().contiguous(), self.P.float().contiguous()
        with torch.cuda.device(self.device):
            _hip.check(_hip.load().rn_kf_update(_p(self.X), _p(self.P), _p(r), _p(z), _p(H.contiguous()), _p(R.contiguous()),
                                                _p(mu_R.contiguous()), len(rows), _hip.stream()), "rn_kf_update")

    def objs(self, with_direction=False, with_time=False):                 # kf.py:420-428
        return self.view(dt=None, with_direction=with_direction)
