"""Seeded synthetic inputs and weights (SURVEY.md 8d recipe).

There is no dataset, checkpoint or network here, so tests, ``bench.py`` and the
golden-vector tool all draw frames, labels and weights from these generators.

The generator is a counter-based integer hash (splitmix64 finaliser) evaluated
with numpy uint64 arithmetic, mapped to fp32 with exact operations only (a
24-bit integer divided by 2^24; "normal" = Irwin-Hall sum of four uniforms).
No libm call and no library RNG stream is involved, so the same (shape, seed)
gives the same bytes on every machine -- golden fixtures therefore store
outputs only, never inputs.
"""
import math

import numpy as np
import torch

from . import arch as _arch

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _hash(n, stream):
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(stream + 1) * _GOLD
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return x


def uniform(shape, seed, lo=0.0, hi=1.0):
    """fp32 U[lo,hi) (numpy array)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_hash(n, seed) >> np.uint64(40)).astype(np.float32) / np.float32(16777216.0)
    if lo != 0.0 or hi != 1.0:
        u = u * np.float32(hi - lo) + np.float32(lo)
    return u.reshape(shape)


def normal(shape, seed, std=1.0, mean=0.0):
    """fp32 approximately N(mean, std): (u0+u1)+(u2+u3)-2 has variance 1/3."""
    s = 4 * seed + 1000003
    v = (uniform(shape, s) + uniform(shape, s + 1)) + (uniform(shape, s + 2) + uniform(shape, s + 3))
    v = (v - np.float32(2.0)) * np.float32(1.7320508 * std)
    if mean != 0.0:
        v = v + np.float32(mean)
    return v


def randint(shape, seed, n):
    return (uniform(shape, seed) * np.float32(n)).astype(np.int64).clip(0, n - 1)


def frames(batch, height, width, seed=0):
    """ImageNet-normalised frames are ~N(0,1): [B,3,H,W] fp32."""
    return torch.from_numpy(normal((batch, 3, height, width), seed))


def _boxes(n, height, width, seed, size_px):
    x0 = uniform((n,), seed, 0.05 * width, 0.75 * width)
    y0 = uniform((n,), seed + 1, 0.05 * height, 0.65 * height)
    lo, hi = size_px if size_px is not None else (0.06 * height, 0.21 * height)
    w = uniform((n,), seed + 2, lo, hi)
    h = uniform((n,), seed + 3, lo, hi)
    return x0, y0, w, h


def labels_dir(batch, n_gt, height, width, num_classes=8, seed=1, size_px=None):
    """[B,N,27] directional labels: 16 corner coords (fbl fbr bbl bbr ftl ftr btl btr as x,y), 2D box = corner
    envelope in 16:20, class in 20, three vanishing points (unused by the loss) = 0 in 21:27.
    Layout: corrected_3D_dataset.py collate / D/losses.py:44-54."""
    out = np.zeros((batch, n_gt, 27), dtype=np.float32)
    for b in range(batch):
        x0, y0, w, h = _boxes(n_gt, height, width, 100 * seed + 10 * b, size_px)
        f32 = np.float32
        bottom = np.stack((x0, y0 + h, x0 + f32(.6) * w, y0 + f32(.9) * h,
                           x0 + f32(.4) * w, y0 + f32(.6) * h, x0 + w, y0 + f32(.5) * h), 1)
        top = bottom.copy()
        top[:, 1::2] -= f32(.5) * h[:, None]
        c = np.concatenate((bottom, top), 1)
        out[b, :, :16] = c
        out[b, :, 16] = c[:, 0::2].min(1)
        out[b, :, 17] = c[:, 1::2].min(1)
        out[b, :, 18] = c[:, 0::2].max(1)
        out[b, :, 19] = c[:, 1::2].max(1)
        out[b, :, 20] = randint((n_gt,), 100 * seed + 10 * b + 4, num_classes)
    return torch.from_numpy(out)


def labels_2d(batch, n_gt, height, width, num_classes=8, seed=1, size_px=None):
    """[B,N,5] = x1,y1,x2,y2,class (R/losses.py:46-47)."""
    out = np.zeros((batch, n_gt, 5), dtype=np.float32)
    for b in range(batch):
        x0, y0, w, h = _boxes(n_gt, height, width, 100 * seed + 10 * b, size_px)
        out[b, :, 0], out[b, :, 1], out[b, :, 2], out[b, :, 3] = x0, y0, x0 + w, y0 + h
        out[b, :, 4] = randint((n_gt,), 100 * seed + 10 * b + 4, num_classes)
    return torch.from_numpy(out)


def head_outputs(batch, n_anchors, num_classes=8, n_reg=12, seed=3, logit_mean=-4.6, reg_std=0.1):
    """Loss inputs: cls = sigmoid(N(-4.6,1)) (torch CPU sigmoid: callers that need bit-portable inputs pass
    the returned tensors around rather than regenerating), reg = N(0,0.1)."""
    cls = torch.sigmoid(torch.from_numpy(normal((batch, n_anchors, num_classes), seed, 1.0, logit_mean)))
    reg = torch.from_numpy(normal((batch, n_anchors, n_reg), seed + 1, reg_std))
    return cls, reg


def scores_portable(batch, n_anchors, num_classes, seed, power=6):
    """Well-separated scores in (0,1) built from exact ops only: u**power by repeated multiplication."""
    u = uniform((batch, n_anchors, num_classes), seed)
    v = u.copy()
    for _ in range(power - 1):
        v = v * u
    return torch.from_numpy(v)


def state_dict(arch, num_classes=8, n_reg=12, seed=2, plain_bn=False, head_scale=1e-4, law="balanced"):
    """Random weights under the reference's own keys.

    ``law="reference"`` is the reference's fan-out law; with randomised batch-norm it lets activations grow
    by orders of magnitude through a ResNet-50 (sigmoid saturates, gradients vanish), which makes a poor
    test signal.  ``law="balanced"`` (default) uses the fan-in form sqrt(2/(k*k*Cin)) and a small gamma
    U(.1,.4) on the last batch-norm of every residual block, keeping activations O(1) as in a trained net.

    Conv weights follow the reference init law N(0, sqrt(2/(k*k*Cout))) (D/model.py:244-247); head output
    weights ~U(0, head_scale) as the trainer re-initialises them (train_detector_3D_angle.py:290-291 --
    required, the stock zero init makes the VP loss 0/0); classification output bias -log(99)
    (D/model.py:252-255).  Unless ``plain_bn``, batch-norm affine and running statistics are randomised
    (gamma~U(.5,1.5), beta~N(0,.1), mean~N(0,.1), var~U(.5,1.5)) so the folded-BN epilogue is exercised.
    """
    sd = {}
    last_bn = "bn2" if _arch.LAYERS[arch][0] == "basic" else "bn3"
    for i, (key, shape) in enumerate(_arch.state_dict_shapes(arch, num_classes, n_reg).items()):
        s = seed * 100003 + i
        leaf = key.rsplit(".", 1)[1]
        is_bn = ".bn" in key or key.startswith("bn1") or ".downsample.1" in key
        if leaf == "num_batches_tracked":
            v = np.zeros((), dtype=np.int64)
        elif leaf == "running_mean":
            v = np.zeros(shape, np.float32) if plain_bn else normal(shape, s, 0.1)
        elif leaf == "running_var":
            v = np.ones(shape, np.float32) if plain_bn else uniform(shape, s, 0.5, 1.5)
        elif len(shape) == 4:
            cout, _, k, _ = shape
            if key.endswith("Model.output.weight"):
                v = uniform(shape, s, 0.0, head_scale)
            else:
                fan = cout if law == "reference" else shape[1]
                gain = 1.0 if (law == "balanced" and key.startswith("fpn.")) else 2.0   # no ReLU after FPN convs
                v = normal(shape, s, math.sqrt(gain / (k * k * fan)))
        elif is_bn and leaf == "weight":
            last = law == "balanced" and (key.endswith(last_bn + ".weight") and key.startswith("layer"))
            v = np.ones(shape, np.float32) if plain_bn else (uniform(shape, s, 0.1, 0.4) if last
                                                             else uniform(shape, s, 0.5, 1.5))
        elif is_bn:
            v = np.zeros(shape, np.float32) if plain_bn else normal(shape, s, 0.1)
        elif key == "classificationModel.output.bias":
            v = np.full(shape, -math.log((1.0 - 0.01) / 0.01), dtype=np.float32)
        elif key == "regressionModel.output.bias":
            v = np.zeros(shape, np.float32)
        else:                                   # conv biases of FPN and head towers
            v = normal(shape, s, 0.05)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v))
    return sd


def camera_matrices(n_cam=18, seed=5):
    """Plausible per-camera P (3x4, space->image) and H (3x3, image->space ground plane) as float64.
    A pinhole looking down at a roadway: feet in space, pixels in image.  H is the inverse of P's
    ground-plane columns, exactly how the reference builds P from H_inv (homography.py:357-370).
    (Uses libm/LAPACK: callers keep the returned matrices, fixtures store them.)"""
    Ps, Hs = [], []
    u = uniform((n_cam, 6), seed).astype(np.float64)
    for i in range(n_cam):
        f = 1200 + 1000 * u[i, 0]
        yaw, pitch = -0.6 + 1.2 * u[i, 1], 0.25 + 0.35 * u[i, 2]
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        R = np.array([[cy, -sy, 0], [sy * sp, cy * sp, -cp], [sy * cp, cy * cp, sp]])
        cam = np.array([200 + 700 * u[i, 3], -120 + 80 * u[i, 4], -60 + 30 * u[i, 5]])
        K = np.array([[f, 0, 960], [0, f, 540], [0, 0, 1.0]])
        P = K @ np.concatenate((R, (-R @ cam)[:, None]), 1)
        P = P / P[2, 3]
        Ps.append(P)
        Hs.append(np.linalg.inv(P[:, [0, 1, 3]]))
    return np.stack(Ps), np.stack(Hs)


def vehicle_states(n, seed=6):
    """[n,6] fp32 (x_rear, y_ctr, l, w, h, dir) in feet, both travel directions (y either side of 60)."""
    s = np.zeros((n, 6), dtype=np.float32)
    s[:, 0] = uniform((n,), seed, 100, 1200)
    s[:, 5] = np.where(uniform((n,), seed + 1) < 0.5, 1.0, -1.0)
    s[:, 1] = np.where(s[:, 5] > 0, uniform((n,), seed + 2, 5, 55), uniform((n,), seed + 3, 65, 115))
    s[:, 2] = uniform((n,), seed + 4, 12, 60)
    s[:, 3] = uniform((n,), seed + 5, 5, 9)
    s[:, 4] = uniform((n,), seed + 6, 4, 13)
    return torch.from_numpy(s)
