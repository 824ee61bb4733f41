"""Explicit forward / backward schedule of the detector over the HIP kernels.

The reference wires ResNet -> PyramidFeatures -> RegressionModel / ClassificationModel -> Anchors -> FocalLoss
out of torch modules (D/model.py:284-309) and lets torch autograd derive the backward pass.  Here the whole
network is ONE autograd node: ``forward`` runs a fixed kernel schedule (NHWC activations, batch-norm / bias /
residual / ReLU / sigmoid fused into conv epilogues, head outputs written straight into the concatenated
[B,A,n] tensors, fused IoU+focal loss), ``backward`` runs the hand-written reverse schedule and hands back one
gradient per parameter.  No per-layer autograd bookkeeping, no tracing compiler: plain launches on the current
HIP stream in a fixed order, so a training step can be captured into a hipGraph and gradient buckets can be
released to RCCL as soon as a layer's wgrad has retired (``grad_hook``).

Backward identities used (frozen batch-norm folded as y = s*c + t, s = gamma*rstd, t = beta - mean*s):
  g       = dL/dy masked by the ReLU of the layer output (mask fused into the producing dgrad's epilogue)
  dx      = dgrad(g, s*W)                      (scale folded into the re-packed dgrad weights)
  dW      = s * wgrad(g, x)
  dbeta   = colsum(g)
  dgamma  = (sum_k W[c,k] * wgrad(g,x)[c,k] - mean[c]*colsum(g)[c]) * rstd[c]
            (because sum_p g*c = sum_k W*dWraw: the pre-BN conv output never has to be stored)
"""
import os

import torch

from . import _hip, arch, conv as cv, ops, prof


# The two input-side transforms of an output gradient (for the weight gradient and for the data gradient of a Winograd layer) in
# one pass over it; RN_WINO_FUSE_DY=0 keeps the two separate launches (A/B).
FUSE_DY = os.environ.get("RN_WINO_FUSE_DY", "1") != "0"
# bf16 engine: the weight gradient of a head layer over its five pyramid levels as ONE launch (rn_conv_wgrad_bf16_grouped); 0: one per level (A/B)
GROUPED_WGRAD_BF16 = os.environ.get("RN_GROUPED_WGRAD_BF16", "1") != "0"
# bf16 engine: the data gradient of a 1x1 stride-2 shortcut computed on its output grid and stored at the even input positions; 0: the generic form (A/B)
S2_SHORTCUT_COMPACT = os.environ.get("RN_S2_SHORTCUT_COMPACT", "1") != "0"
# bf16 / fp8 engines: the fp32 stem's FORWARD products from the first bf16 terms only; 0: the fp32 path's split / native products (A/B)
STEM_BF16_PRODUCTS = os.environ.get("RN_STEM_BF16_PRODUCTS", "1") != "0"


# ---- mixed precision inside the fp8 engine (round 5).  Every layer of an fp8 engine can run its FORWARD in "fp8" (the default), "bf16"
# or "fp32" (Layer.fwd_mode; Engine.set_fp8_layers): tools/fp8_error_budget.py turns ONE group of layers to fp8 at a time to see where
# the end-to-end error comes from, and a group can be kept out of fp8 for good.  Tensors change format at the boundaries: every layer
# output carries the e4m3 scale calibrated for it (._rn_qscale; e4m3 tensors: ._rn_scale), so any consumer can quantise it.
def _qscale(t):
    return getattr(t, "_rn_scale", None) or getattr(t, "_rn_qscale", None)


def _as_fp8(t):
    if t is None or t.dtype == torch.uint8:
        return t
    q = getattr(t, "_rn_q8", None)                  # a tensor read by several e4m3 layers (block input: conv1 and the shortcut) is quantised once
    if q is None:
        sc = _qscale(t)
        assert sc, "a tensor entering an fp8 layer needs its calibrated scale (Layer.fwd tags its outputs)"
        q = t._rn_q8 = cv.fp8_quantize(t, sc) if t.dtype == torch.float32 else cv.bf16_to_fp8(t, sc)
    return q


def _as_f32(t):
    if t is None or t.dtype == torch.float32:
        return t
    y = cv.fp8_dequantize(t) if t.dtype == torch.uint8 else cv.to_f32(t)
    y._rn_qscale = _qscale(t)
    return y


def _as_bf16(t):
    if t is None or t.dtype == torch.bfloat16:
        return t
    y = cv.fp8_to_bf16(t) if t.dtype == torch.uint8 else cv.to_bf16(t)
    y._rn_qscale = _qscale(t)
    return y


_AS = {"fp8": _as_fp8, "bf16": _as_bf16, "fp32": _as_f32}


class Layer:
    """One convolution with its fused batch-norm / bias, the per-step packed weights and gradient accumulators."""

    def __init__(self, spec, kw_pad=None, cin_pad=None, bf16=False, fp8=False):
        self.spec = spec
        self.bf16 = bf16                               # bf16 activations / packed weights, fp32 accumulation (conv_bf16.hip)
        self.fp8 = fp8                                 # e4m3 activations / weights, forward only (conv_fp8.hip); see Engine
        self.out_scale = 1.0                           # fp8: per-tensor scale of this layer's output (Engine.calibration)
        self.calib = None                              # a dict while Engine.calibrate() records output magnitudes
        self.kw_pad = spec.k if kw_pad is None else kw_pad
        self.cin_pad = spec.cin if cin_pad is None else cin_pad
        # channel count of a padded head-output gradient = K of its data gradient: 16-byte chunks must not straddle taps
        self.cout_pad = (spec.cout + 31) // 32 * 32 if spec.cout % (8 if bf16 else 4) else spec.cout
        # Winograd F(4x4,3x3) is available for 3x3 / stride 1 / padding 1 layers with whole 16-byte channel chunks; the
        # engine turns it on for the head towers in training (conv_wino.hip: 2.1-2.3x on those layers, ~1e-5 accuracy)
        self.wino_ok = spec.k == 3 and spec.stride == 1 and spec.pad == 1 and spec.cin % 4 == 0 and spec.cout % 4 == 0
        # worth it from 128 channels on one side and 64 on the other (below, the 36 GEMMs have too short a K loop or too
        # few columns); set per step by the engine
        self.wino_layer = self.wino_ok and min(spec.cin, spec.cout) >= 64 and max(spec.cin, spec.cout) >= 128 and not bf16
        self.wino_active = False
        self.sign = False                              # training forward: ReLU outputs also leave their sign bits (conv.BITMASKS), the
                                                       # backward pass reads those instead of the activations (set per step by the engine)
        self.keep_v = True
        self._cache = None
        self.reset()

    def reset(self):
        self.wf = self.wd = self.scale = self.shift = self.rstd = self.mean = None
        self.dw = self.cs = None
        self.uf = self.ud = None
        self.wf16 = self.wd16 = None               # bf16 copies of the packed weights (bf16 mode)
        self.saved_v = None                        # Winograd input transform of the forward, kept for the weight gradient
        self.dy_v = None                           # B^T dy B of the step's output gradient, left by the weight gradient for the data gradient

    # ---- per-step preparation
    def prepare(self, P, cache):
        s = self.spec
        self._cache = cache
        w = P[s.name + ".weight"]
        self.weight = w
        self.wf = cache.get(("wf", s.name), w, lambda: cv.pack_weights(w, 0, kw_pad=self.kw_pad, c_pad=self.cin_pad))
        if s.bn:
            g, b = P[s.bn + ".weight"], P[s.bn + ".bias"]
            self.mean = P[s.bn + ".running_mean"]
            self.scale, self.shift, self.rstd = cache.get(
                ("bn", s.bn), (g, b), lambda: cv.bn_fold(g, b, self.mean, P[s.bn + ".running_var"], arch.BN_EPS))
        else:
            self.scale = self.rstd = self.mean = None
            self.shift = P[s.name + ".bias"].detach() if s.bias else None
        self.wd = None
        self.dw = self.cs = None
        self.uf = self.ud = None
        self.wd16 = None
        self.wf16 = cache.get(("wf16", s.name), w, lambda: cv.to_bf16(self.wf)) if self.bf16 else None
        if self.fp8:                                   # e4m3 rows + per-output-channel scale, times the folded batch-norm scale
            self.wq, sw = cache.get(("wq", s.name), w, lambda: cv.fp8_quantize_weights(self.wf))
            self.wscale = cache.get(("wqs", s.name), (w,) + ((P[s.bn + ".weight"], P[s.bn + ".running_var"]) if s.bn else ()),
                                    lambda: (sw * self.scale) if self.scale is not None else sw)
            self._fp8_scales = {}

    def adopt(self, P, cache, wf, bn, wd, wino=None, wf16=None, wd16=None):
        """Training step: take this step's packed weights / folded batch norm from the engine's batched preparation
        (persistent buffers refreshed by two launches for the whole net) instead of one launch per tensor."""
        s = self.spec
        self._cache = cache
        self.weight = P[s.name + ".weight"]
        self.wf = wf
        if s.bn:
            self.mean = P[s.bn + ".running_mean"]
            self.scale, self.shift, self.rstd = bn[0], bn[1], bn[2]
        else:
            self.scale = self.rstd = self.mean = None
            self.shift = P[s.name + ".bias"].detach() if s.bias else None
        self.wd = wd                                  # None: packed on demand (layers that normally take the Winograd path)
        self.dw = self.cs = None
        self.uf, self.ud = wino if wino is not None else (None, None)
        self.wd16 = wd16                              # bf16 twins from the batched preparation (launch 2), else on demand
        self.wf16 = wf16 if wf16 is not None else (cv.to_bf16(self.wf) if self.bf16 else None)
        if self.fp8:                                   # fp8-forward training: this step's weights as e4m3 rows + per-channel scales
            self.wq, sw = cv.fp8_quantize_weights(self.wf)
            self.wscale = (sw * self.scale) if self.scale is not None else sw
            self._fp8_scales = {}

    def wino_weights(self, mode):
        """Winograd-transformed weights of this step (mode 0 forward, 1 data gradient with the batch-norm scale folded in)."""
        if mode == 0:
            if self.uf is None:
                self.uf = self._cache.get(("uf", self.spec.name), self.weight, lambda: cv.wino_weights(self.weight, 0)) \
                    if self._cache is not None else cv.wino_weights(self.weight, 0)
            return self.uf
        if self.ud is None:
            self.ud = cv.wino_weights(self.weight, 1, scale=self.scale)
        return self.ud

    def dgrad_weights16(self):
        """bf16 copies of dgrad_weights()."""
        if self.wd16 is None:
            wd = self.dgrad_weights()
            self.wd16 = [cv.to_bf16(t) for t in wd] if isinstance(wd, list) else cv.to_bf16(wd)
        return self.wd16

    def dgrad_weights(self):
        """Packed dgrad weights (BN scale folded in); for a stride-2 k>1 layer: one tap subset per parity class."""
        if self.wd is None:
            s = self.spec
            if s.stride == 2 and s.k > 1:
                self.wd = [cv.pack_weights(self.weight, 1, scale=self.scale, c_pad=self.cout_pad, taps=c[2])
                           for c in cv.s2_classes(s.k, s.pad)]
            else:
                self.wd = cv.pack_weights(self.weight, 1, scale=self.scale, c_pad=self.cout_pad)
        return self.wd

    # ---- forward
    def fwd(self, x, act=cv.ACT_NONE, add=None, add_mode=0, add_hw=(0, 0), out=None, y_batch_stride=None, in_relu=False, sign=None):
        """sign: also write the result's sign bits (default: when the engine asked for them, self.sign, and the result is a ReLU output;
        True for a tensor whose sign masks a later gradient although it is stored before its ReLU: fpn.P6)."""
        s = self.spec
        sign = (self.sign and act == cv.ACT_RELU) if sign is None else (sign and self.sign)
        N, Hi, Wi, _ = x.shape
        Ho, Wo = cv.out_size(Hi, s.k, s.stride, s.pad), cv.out_size(Wi, s.k, s.stride, s.pad)
        mode = self.mode()
        if self.fp8:                                   # a layer of an fp8 engine: operands in the format this layer's forward runs in
            x, add = _AS[mode](x), _AS[mode](add)
        if mode == "fp8":
            assert not in_relu, "fp8: the caller applies the input ReLU"
            return self._fwd_fp8(x, (Ho, Wo), act, add, add_mode, add_hw, out, y_batch_stride)
        if mode == "bf16":
            assert not in_relu, "bf16: the caller applies the input ReLU (cv.relu_bf16)"
            if out is None:
                out = torch.empty((N, Ho, Wo, s.cout), dtype=torch.bfloat16, device=x.device)
            cv.conv_igemm_bf16(x, self.wf16, out, (Ho, Wo, s.cout, s.k, self.kw_pad, s.stride, 1, -s.pad, 0),
                               scale=self.scale, shift=self.shift, add=add, add_mode=add_mode, add_hw=add_hw, act=act,
                               y_batch_stride=y_batch_stride, flops=self.flops(N, Ho, Wo), sign=sign)
            out._rn_qscale = self.out_scale
            return out
        if self.wino_active and out is None and add is None and not in_relu and y_batch_stride is None \
                and act in (cv.ACT_NONE, cv.ACT_RELU) and x.is_contiguous():
            r = cv.wino_conv_group([x], self.wino_weights(0), scale=self.scale, shift=self.shift, act=act, keep_v=self.keep_v, sign=sign)
            ys, self.saved_v = r if self.keep_v else (r, None)
            return ys[0]
        if out is None:
            out = torch.empty((N, Ho, Wo, s.cout), dtype=torch.float32, device=x.device)
        cv.conv_igemm(x, self.wf, out, (Ho, Wo, s.cout, s.k, self.kw_pad, s.stride, 1, -s.pad, 0), scale=self.scale,
                      shift=self.shift, add=add, add_mode=add_mode, add_hw=add_hw, act=act,
                      y_batch_stride=y_batch_stride, in_relu=in_relu, flops=self.flops(N, Ho, Wo), sign=sign,
                      bf16_products=getattr(self, "bf16_products", False))
        self._tap(out if y_batch_stride is None else None)
        out._rn_qscale = self.out_scale
        return out

    def mode(self):
        """The format this layer's FORWARD runs in: "fp8" / "bf16" / "fp32" (an fp8 engine may keep single layers out of fp8: fwd_mode)."""
        return getattr(self, "fwd_mode", None) or ("fp8" if self.fp8 else ("bf16" if self.bf16 else "fp32"))

    def _tap(self, out):
        """Calibration of the fp8 path (Engine.calibrate): remember the largest magnitude this layer's output has taken."""
        if self.calib is not None and out is not None:
            self.calib[self.spec.name] = max(self.calib.get(self.spec.name, 0.0), float(out.abs().max()))

    def _fwd_fp8(self, x, out_hw, act, add, add_mode, add_hw, out, y_batch_stride):
        """One e4m3 convolution (conv_fp8.hip): x / add are uint8 tensors carrying their per-tensor scale (._rn_scale); the result
        is e4m3 with this layer's calibrated scale, or fp32 when `out` is an fp32 destination (the head outputs)."""
        s = self.spec
        Ho, Wo = out_hw
        sx = float(x._rn_scale)
        scale = self._fp8_scales.get(sx)
        if scale is None:                              # x_scale * weight row scale * folded batch-norm scale, per input scale seen
            scale = self._fp8_scales[sx] = (self.wscale * sx).contiguous()
        if out is None:
            out = torch.empty((x.shape[0], Ho, Wo, s.cout), dtype=torch.uint8, device=x.device)
        return cv.conv_igemm_fp8(x, self.wq, out, (Ho, Wo, s.cout, s.k, self.kw_pad, s.stride, 1, -s.pad, 0), scale, shift=self.shift,
                                 add=add, add_mode=add_mode, add_hw=add_hw, act=act, y_batch_stride=y_batch_stride,
                                 out_scale=self.out_scale, flops=self.flops(x.shape[0], Ho, Wo))

    def fwd_group(self, xs, act=cv.ACT_NONE, outs=None, y_batch_stride=None, wino=False, shared_v=None):
        """Same convolution on several inputs (pyramid levels) in one launch.  outs: destination tensors/views
        (with y_batch_stride) or None for fresh dense outputs.  wino: Winograd path (dense outputs only).  shared_v: the kept
        input transform of ANOTHER layer that read the same xs (both towers' conv1 read the pyramid): reused, not recomputed."""
        s = self.spec
        mode = self.mode()
        if self.fp8:
            xs = [_AS[mode](x) for x in xs]
        if mode == "fp8":                              # the levels of a head layer as one grouped e4m3 launch
            scales = {float(x._rn_scale) for x in xs}
            if len(scales) > 1 or len(xs) > _hip.RN_MAX_GROUP:      # different input scales cannot share one folded scale vector
                return [self._fwd_fp8(x, (x.shape[1], x.shape[2]), act, None, 0, (0, 0), None if outs is None else outs[i], y_batch_stride)
                        for i, x in enumerate(xs)]
            sx = scales.pop()
            scale = self._fp8_scales.get(sx)
            if scale is None:
                scale = self._fp8_scales[sx] = (self.wscale * sx).contiguous()
            probs, ys, fl = [], [], 0.0
            for i, x in enumerate(xs):
                N, Hi, Wi, _ = x.shape
                y = outs[i] if outs is not None else torch.empty((N, Hi, Wi, s.cout), dtype=torch.uint8, device=x.device)
                ys.append(y)
                fl += self.flops(N, Hi, Wi)
                probs.append({"x": x, "y": y, "geom": (Hi, Wi, s.cout, s.k, self.kw_pad, 1, 1, -s.pad, 0), "y_batch_stride": y_batch_stride})
            cv.conv_igemm_fp8_grouped(probs, self.wq, scale, shift=self.shift, act=act, out_scale=self.out_scale, flops=fl)
            return ys
        if mode == "bf16":                             # the levels of a head layer as one grouped bf16 launch
            probs, ys, fl = [], [], 0.0
            for i, x in enumerate(xs):
                N, Hi, Wi, _ = x.shape
                y = outs[i] if outs is not None else torch.empty((N, Hi, Wi, s.cout), dtype=torch.bfloat16, device=x.device)
                ys.append(y)
                fl += self.flops(N, Hi, Wi)
                probs.append({"x": x, "y": y, "geom": (Hi, Wi, s.cout, s.k, self.kw_pad, 1, 1, -s.pad, 0),
                              "y_batch_stride": y_batch_stride, "sign": self.sign and act == cv.ACT_RELU})
            cv.conv_igemm_bf16_grouped(probs, self.wf16, scale=self.scale, shift=self.shift, act=act, flops=fl)
            for y in ys:
                y._rn_qscale = self.out_scale
            return ys
        if (wino or self.wino_active) and self.wino_ok and (outs is None or y_batch_stride is not None):
            fl = sum(self.flops(x.shape[0], x.shape[1], x.shape[2]) for x in xs)
            r = cv.wino_conv_group(xs, self.wino_weights(0), outs=outs, scale=self.scale, shift=self.shift, act=act, flops=fl,
                                   keep_v=self.keep_v, y_batch_stride=y_batch_stride or 0, V_in=shared_v if self.keep_v else None,
                                   sign=self.sign and act == cv.ACT_RELU)
            ys, self.saved_v = r if self.keep_v else (r, None)
            return ys
        probs, ys, fl = [], [], 0.0
        for i, x in enumerate(xs):
            N, Hi, Wi, _ = x.shape
            Ho, Wo = cv.out_size(Hi, s.k, s.stride, s.pad), cv.out_size(Wi, s.k, s.stride, s.pad)
            y = outs[i] if outs is not None else torch.empty((N, Ho, Wo, s.cout), dtype=torch.float32, device=x.device)
            ys.append(y)
            fl += self.flops(N, Ho, Wo)
            probs.append({"x": x, "y": y, "geom": (Ho, Wo, s.cout, s.k, self.kw_pad, s.stride, 1, -s.pad, 0),
                          "y_batch_stride": y_batch_stride, "sign": self.sign and act == cv.ACT_RELU})
        cv.conv_igemm_grouped(probs, self.wf, scale=self.scale, shift=self.shift, act=act, flops=fl)
        if outs is None:
            for y in ys:
                self._tap(y)
                y._rn_qscale = self.out_scale
        return ys

    def bwd_data_group(self, gs, in_hws, adds=None, masks=None, wino=False):
        """Stride-1 data gradient of several problems in one launch; adds / masks: per-problem tensors or None."""
        s = self.spec
        assert s.stride == 1
        if self.bf16:
            probs, outs, fl = [], [], 0.0
            for i, g in enumerate(gs):
                Hi, Wi = in_hws[i]
                dx = torch.empty((g.shape[0], Hi, Wi, s.cin), dtype=torch.bfloat16, device=g.device)
                outs.append(dx)
                fl += self.flops(g.shape[0], g.shape[1], g.shape[2])
                probs.append({"x": g, "y": dx, "geom": (Hi, Wi, s.cin, s.k, s.k, 1, -1, s.pad, 0),
                              "add": None if adds is None else adds[i], "mask": None if masks is None else masks[i]})
            cv.conv_igemm_bf16_grouped(probs, self.dgrad_weights16(), flops=fl)
            return outs
        if (wino or self.wino_active) and self.wino_ok and all(g.shape[3] == s.cout for g in gs):
            fl = sum(self.flops(g.shape[0], g.shape[1], g.shape[2]) for g in gs)
            ready, self.dy_v = getattr(self, "dy_v", None), None
            return cv.wino_conv_group(gs, self.wino_weights(1), adds=adds, masks=masks, mask_mode=2, flops=fl, V_ready=ready)
        probs, outs, fl = [], [], 0.0
        for i, g in enumerate(gs):
            N = g.shape[0]
            Hi, Wi = in_hws[i]
            dx = torch.empty((N, Hi, Wi, s.cin), dtype=torch.float32, device=g.device)
            outs.append(dx)
            fl += self.flops(N, g.shape[1], g.shape[2])
            probs.append({"x": g, "y": dx, "geom": (Hi, Wi, s.cin, s.k, s.k, 1, -1, s.pad, 0),
                          "add": None if adds is None else adds[i], "mask": None if masks is None else masks[i]})
        cv.conv_igemm_grouped(probs, self.dgrad_weights(), flops=fl)
        return outs

    def flops(self, N, Ho, Wo):
        """Algorithmic FLOPs of this convolution on N images (2*MACs; the same count prices fprop, dgrad, wgrad)."""
        s = self.spec
        return 2.0 * N * Ho * Wo * s.cout * s.cin * s.k * s.k

    # ---- backward
    def bwd_params(self, g, x, in_relu=False, side=None):
        """Accumulate wgrad(g, x) and colsum(g).  g: [N,Ho,Wo,ld] with ld >= cout.  side: a stream for the reductions of the Winograd form
        (cv.wino_wgrad_group); the other forms run where they are called."""
        s = self.spec
        if self.dw is None:
            self.dw = torch.zeros_like(self.wf)
            self.cs = torch.zeros(s.cout, dtype=torch.float32, device=g.device)
        if self.bf16:
            assert not in_relu
            cv.wgrad_bf16(g, x, self.dw, s.cout, s.k, s.stride, s.pad, flops=self.flops(g.shape[0], g.shape[1], g.shape[2]),
                          colsum=self.cs)
            return
        if self.wino_active and not in_relu and g.shape[3] == s.cout and g.is_contiguous() and x.is_contiguous():
            du, self.du = getattr(self, "du", None), None          # the zeroed slot is good for one accumulation
            # (the data gradient of this layer follows: its input transform of g rides along -- bwd_data picks it up)
            self.dy_v = cv.wino_wgrad_group([g], [x], self.dw, self.cs, V=self.saved_v, dU=du, fuse_dgrad_input=FUSE_DY, side=side)
            self.saved_v = None
            return
        cv.wgrad(g, x, self.dw, s.cout, s.k, s.stride, s.pad, kw_pad=self.kw_pad, in_relu=in_relu,
                 flops=self.flops(g.shape[0], g.shape[1], g.shape[2]), colsum=self.cs)

    def bwd_params_group(self, gs, xs, wino=False, side=None):
        """bwd_params over several problems (pyramid levels); wino: one Winograd weight-gradient pass over all of them."""
        s = self.spec
        if (wino or self.wino_active) and self.wino_ok and all(g.shape[3] == s.cout and g.is_contiguous() for g in gs):
            if self.dw is None:
                self.dw = torch.zeros_like(self.wf)
                self.cs = torch.zeros(s.cout, dtype=torch.float32, device=gs[0].device)
            fl = sum(self.flops(g.shape[0], g.shape[1], g.shape[2]) for g in gs)
            du, self.du = getattr(self, "du", None), None
            self.dy_v = cv.wino_wgrad_group(gs, xs, self.dw, self.cs, flops=fl, V=self.saved_v, dU=du, fuse_dgrad_input=FUSE_DY, side=side)
            self.saved_v = None
            return
        if self.bf16 and GROUPED_WGRAD_BF16 and 1 < len(gs) <= 5 and all(g.is_contiguous() and x.is_contiguous() for g, x in zip(gs, xs)) \
                and len({g.shape[3] for g in gs}) == 1:
            if self.dw is None:
                self.dw = torch.zeros_like(self.wf)
                self.cs = torch.zeros(s.cout, dtype=torch.float32, device=gs[0].device)
            fl = sum(self.flops(g.shape[0], g.shape[1], g.shape[2]) for g in gs)
            cv.wgrad_bf16_grouped(gs, xs, self.dw, s.cout, s.k, s.stride, s.pad, flops=fl, colsum=self.cs)
            return
        for g, x in zip(gs, xs):
            self.bwd_params(g, x)

    def bwd_data(self, g, in_hw, add=None, mask=None, mask_mode=2, add2=None):
        s = self.spec
        if self.bf16:
            assert add2 is None
            kw = dict(add=add, add_mode=1 if add is not None else 0, mask=mask, mask_mode=mask_mode,
                      flops=self.flops(g.shape[0], g.shape[1], g.shape[2]))
            if s.stride == 2 and s.k > 1:
                return cv.dgrad_s2_classes_bf16(g, self.dgrad_weights16(), in_hw, s.cin, s.k, s.pad, **kw)
            if s.stride == 2 and s.k == 1 and s.pad == 0 and mask is None and S2_SHORTCUT_COMPACT:
                # 1x1 stride-2 shortcut (D/model.py:265-270): its gradient exists at the EVEN input positions only.  One launch on the
                # output grid, stored at those positions (out_map) of a zeroed tensor -- or of the addend itself, in place (the caller
                # hands over the lateral gradient and does not use it again): a quarter of the generic form's rows (which tried its one
                # tap at every input pixel and failed the divisibility test at three of four).
                Hi, Wi = in_hw
                N, Ho, Wo, _ = g.shape
                if add is None:
                    dx = torch.zeros((N, Hi, Wi, s.cin), dtype=torch.bfloat16, device=g.device)
                else:
                    assert add.shape == (N, Hi, Wi, s.cin) and add.dtype == torch.bfloat16 and add.is_contiguous()
                    dx = add
                return cv.conv_igemm_bf16(g, self.dgrad_weights16(), dx, (Ho, Wo, s.cin, 1, 1, 1, -1, 0, 0), out_map=(2, 0, 0, Hi, Wi),
                                          add=add, add_mode=1 if add is not None else 0, flops=kw["flops"])
            return cv.dgrad_any_bf16(g, self.dgrad_weights16(), in_hw, s.cin, s.k, s.stride, s.pad, **kw)
        if self.wino_active and add2 is None and g.shape[3] == s.cout and g.is_contiguous():
            ready, self.dy_v = getattr(self, "dy_v", None), None
            return cv.wino_conv_group([g], self.wino_weights(1), adds=None if add is None else [add],
                                      masks=None if mask is None else [mask], mask_mode=mask_mode, V_ready=ready)[0]
        kw = dict(add=add, add_mode=1 if add is not None else 0, mask=mask, mask_mode=mask_mode,
                  flops=self.flops(g.shape[0], g.shape[1], g.shape[2]))
        if s.stride == 2 and s.k > 1:
            assert add2 is None
            return cv.dgrad_s2_classes(g, self.dgrad_weights(), in_hw, s.cin, s.k, s.pad, **kw)
        return cv.dgrad(g, self.dgrad_weights(), in_hw, s.cin, s.k, s.stride, s.pad, add2=add2, **kw)

    def bwd_data_compact(self, g):
        """1x1 stride-2 shortcut: gradient on the OUTPUT grid only ([N,Ho,Wo,Cin]); it belongs at the even input
        positions and is added there by the consumer's epilogue (add2), the odd positions get nothing."""
        s = self.spec
        assert s.k == 1 and s.stride == 2
        N, Ho, Wo, _ = g.shape
        out = torch.empty((N, Ho, Wo, s.cin), dtype=torch.float32, device=g.device)
        return cv.conv_igemm(g, self.dgrad_weights(), out, (Ho, Wo, s.cin, 1, 1, 1, -1, 0, 0), flops=self.flops(N, Ho, Wo))

    def grad_names(self):
        """Names of the parameters this layer produces gradients for, in the order finish() reports them."""
        s = self.spec
        if s.bn:
            return [s.name + ".weight", s.bn + ".weight", s.bn + ".bias"]
        return [s.name + ".weight"] + ([s.name + ".bias"] if s.bias else [])

    def finish(self, out=None):
        """-> {param name: gradient} once every contribution has been accumulated.  out: {param name: destination
        tensor} (views of the engine's flat gradient buffer) or None for fresh tensors."""
        s = self.spec
        dst = None
        if out is not None:
            names = self.grad_names()
            dst = (out[names[0]], out[names[1]] if s.bn else None, out[names[-1]] if len(names) > 1 else None)
        dweight, dgamma, dbeta = cv.unpack_wgrad(
            self.dw, self.wf, tuple(self.weight.shape), kw_pad=self.kw_pad, c_pad=self.cin_pad, scale=self.scale,
            mean=self.mean, rstd=self.rstd, colsum=self.cs, want_bn=bool(s.bn), out=dst)
        out = {s.name + ".weight": dweight}
        if s.bn:
            out[s.bn + ".weight"] = dgamma
            out[s.bn + ".bias"] = dbeta
        elif s.bias:
            out[s.name + ".bias"] = dbeta
        self.dw = self.cs = self.wd = self.uf = self.ud = self.saved_v = self.dy_v = None
        return out


class _Cache:
    """Derived tensors (packed weights, folded batch-norm) keyed on the source parameters' identity and version,
    so eval frames reuse them and a replaced or updated Parameter invalidates them."""

    def __init__(self):
        self.store = {}

    def get(self, key, src, make):
        srcs = src if isinstance(src, tuple) else (src,)
        stamp = tuple((t.data_ptr(), t._version) for t in srcs)
        hit = self.store.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        val = make()
        self.store[key] = (stamp, val)
        return val


class Engine:
    def __init__(self, arch_name, num_classes, n_reg, dtype="fp32"):
        """dtype "fp32": the reference's arithmetic (default).  "bf16" (BASELINE configs[2]): bf16 activations and packed
        weights, fp32 accumulation / epilogues / parameters / gradients / loss; the 3-channel stem stays fp32 and the
        Winograd path is not used.  Bottleneck architectures only (resnet50 / 101 / 152).
        "fp8" (BASELINE configs[4], first cut, INFERENCE ONLY): e4m3 activations (one calibrated scale per tensor: calibrate())
        and weights (one scale per output channel) on the fp8 MFMA, fp32 accumulation and epilogues; the stem and the head
        outputs stay fp32; no backward pass."""
        if dtype not in ("fp32", "bf16", "fp8"):
            raise ValueError("dtype must be 'fp32', 'bf16' or 'fp8'")
        self.bf16 = dtype == "bf16"
        self.fp8 = dtype == "fp8"
        # the backward pass of the fp8 engine (round 5: fp8 forward, bf16 data / weight gradients) is the bf16 engine's: its layers are
        # built with both flags -- Layer.fwd / fwd_group look at fp8 first, every bwd_* at bf16 -- and carry both weight forms
        self.bwd16 = self.bf16 or self.fp8
        self.fp8_scales = None                     # {layer name / "pool": largest magnitude seen} from calibrate()
        # Which layers of the fp8 engine run in e4m3.  "stream_bf16" (default since round 5): all but the RESIDUAL STREAM -- the last
        # convolution of every bottleneck, the shortcuts -- and the FPN, which run in bf16 (bf16 result and addend): as e4m3 the stream
        # is re-quantised at every block, 33 times through ResNet-101, and together with the pyramid that is where the end-to-end error
        # comes from (tools/fp8_error_budget.py, profiles/r05_fp8_error_budget.txt: scores 14.7 % max / 5.7 % rms of the fp32 forward
        # with every layer in e4m3, 8.0 % / 2.0 % this way) -- and the regression tower's last layer, whose e4m3 result feeds an
        # unbounded regression output (training: the smooth-L1 loss 2.4 % off the bf16 step's with it in e4m3, 1.8 % without;
        # tools/dbg/fp8_train_variants.py).  "all": every layer in e4m3 (round 4's configuration, the fastest).
        self.fp8_policy = os.environ.get("RN_FP8_POLICY", "stream_bf16")
        if self.bf16 and arch.LAYERS[arch_name][0] != "bottleneck":
            raise NotImplementedError("the bf16 schedule is built for the bottleneck networks (resnet50/101/152)")
        self.fp8_trainable = self.fp8 and arch.LAYERS[arch_name][0] == "bottleneck"      # (the bf16 backward's condition)
        self.arch = arch_name
        self.num_classes = num_classes
        self.n_reg = n_reg
        self.kind = arch.LAYERS[arch_name][0]
        self.cache = _Cache()
        self.anchor_cache = {}
        self.grad_hook = None                      # callable({name: grad}) as soon as a layer's gradients are final
        # Persistent flat gradient buffer (set_flat_grads): every parameter gradient is written straight into its slot of
        # ONE device buffer laid out in the order the backward finishes layers, so (a) the all-reduce buckets are plain
        # slices of it (no packing copy) handed to bucket_hook the moment their last layer retires, and (b) gradient
        # pointers are the same every step (the optimizer's pointer table is uploaded once).
        self.flat_bucket_bytes = None
        self._tower_streams = {}
        self.flat_tail_bytes = None
        self.bucket_hook = None                    # callable(bucket index, flat slice) when a bucket is complete
        self._flat = None
        # Winograd for the head towers when activations are saved (= training); RN_WINOGRAD=0 keeps the direct kernels
        self.use_wino = os.environ.get("RN_WINOGRAD", "1") != "0"
        # opt-in for inference as well (default off: eval keeps the ~1e-7 behaviour of the direct kernels)
        self.wino_eval = os.environ.get("RN_WINOGRAD_EVAL", "0") == "1"
        # layers
        self.layers = {}
        self.blocks = []                            # [(prefix, [roles...])] in forward order
        cur = None
        for spec, role, pre in arch.backbone_convs(arch_name):
            if role == "stem":
                self.layers[spec.name] = Layer(spec, kw_pad=8, cin_pad=4)
                # bf16 / fp8 engines: the stem's tensors stay fp32 (3-channel input, max-pool boundary) but its products are formed from
                # the first bf16 terms of both operands (rn_conv_desc.w_format 2): the engine's arithmetic, one MFMA instead of six
                self.layers[spec.name].bf16_products = (self.bf16 or self.fp8) and STEM_BF16_PRODUCTS
                continue
            self.layers[spec.name] = Layer(spec, bf16=self.bwd16, fp8=self.fp8)
            if cur is None or cur[0] != pre:
                cur = (pre, {})
                self.blocks.append(cur)
            cur[1][role] = self.layers[spec.name]
        for spec in arch.fpn_convs(arch_name) + arch.head_convs("regressionModel", n_reg) + \
                arch.head_convs("classificationModel", num_classes):
            self.layers[spec.name] = Layer(spec, bf16=self.bwd16, fp8=self.fp8)
        self.param_names = [k for k, shp in arch.state_dict_shapes(arch_name, num_classes, n_reg).items()
                            if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]

    # ------------------------------------------------------------------------------------------- flat gradients
    def finish_order(self):
        """Layer names in the order backward() finishes them (heads -> FPN -> layer4 ... layer1 -> stem)."""
        order = []
        for prefix in ("regressionModel", "classificationModel"):
            order += [prefix + ".output"] + ["%s.conv%d" % (prefix, i) for i in (4, 3, 2, 1)]
        order += ["fpn." + n for n in ("P7_2", "P6", "P3_2", "P3_1", "P4_2", "P4_1", "P5_2", "P5_1")]
        for pre, roles in reversed(self.blocks):
            seq = ["conv2"] if self.kind == "basic" else ["conv3", "conv2"]
            seq += (["down"] if "down" in roles else []) + ["conv1"]
            order += [roles[r].spec.name for r in seq]
        order.append("conv1")
        assert sorted(order) == sorted(self.layers), "finish_order does not cover the layer set"
        return order

    # ------------------------------------------------------------------------------------------- ReLU outputs and their reach
    def relu_outputs(self, S):
        """{name: NHWC tensor} of every ReLU output among the saved activations S of forward(save=True), under the name of the
        convolution that produced it ("layer2.0.conv1", "fpn.P6", "regressionModel.conv3@<pyramid level>"): the naming of
        oracle.model's taps.  (fpn.P6 is stored before its ReLU -- the sign pattern is the same.)"""
        out = {"conv1": S["stem"]}
        for (pre, roles), (xin, t1, t2, x) in zip(self.blocks, S["blocks"]):
            out[pre + ".conv1"] = t1
            if t2 is not None:
                out[pre + ".conv2"] = t2
            out[pre + (".conv2" if self.kind == "basic" else ".conv3")] = x
        out["fpn.P6"] = S["fpn"][6]
        for prefix, acts in S["towers"].items():
            for level, per_conv in enumerate(acts):
                for i, t in enumerate(per_conv):
                    out["%s.conv%d@%d" % (prefix, i + 1, level)] = t
        return out

    def backward_cone(self, act_name):
        """Names of the parameters whose gradient the ReLU MASK of activation `act_name` (relu_outputs' naming) can reach: the
        layer that produced it and everything the backward pass runs after it along the data-gradient path.  An activation
        within rounding of zero that lands on the other side of the ReLU than in a reference run changes exactly these."""
        order = self.finish_order()
        backbone = [n for n in order if n.startswith("layer") or n == "conv1"]
        base, _, level = act_name.partition("@")
        if base.startswith(("regressionModel.", "classificationModel.")):
            prefix, conv = base.split(".")
            layers = ["%s.conv%d" % (prefix, j) for j in range(1, int(conv[-1]) + 1)]
            # pyramid level -> FPN layers its gradient flows through (top-down path: dP3sum -> dP4sum -> dP5lat)
            layers += {0: ["fpn.P3_2", "fpn.P3_1", "fpn.P4_1", "fpn.P5_1"], 1: ["fpn.P4_2", "fpn.P4_1", "fpn.P5_1"],
                       2: ["fpn.P5_2", "fpn.P5_1"], 3: ["fpn.P6"], 4: ["fpn.P7_2", "fpn.P6"]}[int(level)]
            layers += backbone                      # through C3 / C4 / C5 (every level reaches C5 or, for P3 / P4, layer2 / layer3 below it)
        elif base == "fpn.P6":
            layers = ["fpn.P6"] + backbone
        elif base == "conv1":
            layers = ["conv1"]
        else:
            layers = order[order.index(base):]      # the rest of the backbone's backward
            pre, _, role = base.rpartition(".")
            if role in ("conv1", "conv2") and pre + ".downsample.0" in layers and self.kind != "basic":
                layers = [n for n in layers if n != pre + ".downsample.0"]   # the shortcut of the same block does not see t1 / t2
        names = []
        for n in layers:
            names += self.layers[n].grad_names()
        return names

    def set_flat_grads(self, bucket_bytes=32 << 20, tail_bytes=6 << 20):
        """Turn the persistent flat gradient buffer on (bucket_bytes: all-reduce bucket size; None turns it off).
        tail_bytes: the LAST bucket is final only when the backward ends, so its all-reduce is fully exposed -- it is cut down
        to the last whole layers that fit in tail_bytes (ResNet-50: layer2 + layer1 + stem = 5.8 MB) and what was in front
        of them becomes a bucket of its own, released while those layers still compute."""
        self.flat_bucket_bytes = bucket_bytes
        self.flat_tail_bytes = tail_bytes
        self._flat = None

    def _flat_plan(self, device):
        """Slots (64-float aligned) of every parameter gradient in finish order + the buckets: [start, end, last layer]."""
        slots, buckets, off, start = {}, [], 0, 0
        ends = []                                   # (end offset, layer name) per layer, in finish order
        for lname in self.finish_order():
            L = self.layers[lname]
            shapes = [tuple(L.weight.shape)] + [(L.spec.cout,)] * (len(L.grad_names()) - 1)
            for pname, shp in zip(L.grad_names(), shapes):
                n = 1
                for d in shp:
                    n *= d
                slots[pname] = (off, n, shp)
                off += (n + 63) // 64 * 64
            ends.append((off, lname))
            if 4 * (off - start) >= self.flat_bucket_bytes:
                buckets.append((start, off, lname))
                start = off
        if off > start:
            buckets.append((start, off, self.finish_order()[-1]))
        tail = getattr(self, "flat_tail_bytes", None)
        if tail and buckets and 4 * (buckets[-1][1] - buckets[-1][0]) > tail:
            b0, b1, last = buckets[-1]
            cut = None
            for e, lname in ends:                   # first layer boundary inside the bucket from which the rest fits in `tail`
                if b0 < e < b1 and 4 * (b1 - e) <= tail:
                    cut = (e, lname)
                    break
            if cut is not None:
                buckets[-1:] = [(b0, cut[0], cut[1]), (cut[0], b1, last)]
        return {"slots": slots, "buckets": buckets, "total": off, "device": device, "arena": None}

    def _flat_views(self, device):
        """{param name: view} of the flat buffer for this backward.  The buffer is reused from step to step unless a
        parameter's .grad still lives in it (gradient accumulation over several backward calls, or zero_grad without
        set_to_none): then this call gets a fresh one and the old one stays with those .grad tensors."""
        f = self._flat
        if f is None or f["device"] != device:
            f = self._flat = self._flat_plan(device)
        arena = f["arena"]
        if arena is not None:
            base = arena.untyped_storage().data_ptr()
            for L in self.layers.values():
                g = L.weight.grad
                if g is not None and g.untyped_storage().data_ptr() == base:
                    arena = None
                    break
        if arena is None:
            arena = f["arena"] = torch.zeros(f["total"], dtype=torch.float32, device=device)   # alignment gaps stay zero
        return arena, {n: arena[o:o + k].view(shp) for n, (o, k, shp) in f["slots"].items()}

    # ------------------------------------------------------------------------------------------- helpers
    def _side_streams_ok(self, device):
        """Side streams at all?  Not on the CPU, not during a stream capture (a captured graph keeps the single-stream order), not under
        per-kernel timing (prof.ACTIVE: a kernel's duration means something only when nothing else shares the GPU)."""
        return device.type == "cuda" and prof.ACTIVE is None and not torch.cuda.is_current_stream_capturing()

    def tower_streams(self, device):
        """The two side streams the head towers run on (forward and backward), or None: RN_TOWER_STREAMS=0, where _side_streams_ok says no,
        and -- unless RN_TOWER_STREAMS=2 -- in a data-parallel run (bucket_hook set).  The process must stay within FOUR busy hardware
        queues: HIP maps streams onto that many, and beyond them the queues are time-sliced -- measured: the same step at 155 ms instead of
        64 with GPU_MAX_HW_QUEUES=8, and at 146 ms on one box with a fifth stream (profiles/r05_wgrad_stream.txt, 4 and 5).  Single GPU:
        main + two towers + weight gradients = 4.  Data-parallel: RCCL brings its own, so the towers give theirs up (they are worth
        +0.9 %, the weight-gradient stream +3.6 %): main + weight gradients + communication."""
        mode = os.environ.get("RN_TOWER_STREAMS", "1")
        if mode == "0" or not self._side_streams_ok(device) or (self.bucket_hook is not None and mode != "2"):
            return None
        st = self._tower_streams.get(device)
        if st is None:
            st = self._tower_streams[device] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
        return st

    def wgrad_streams(self, device):
        """The side stream(s) the backbone's and the pyramid's weight gradients run on (taken in turn), or None (RN_WGRAD_STREAMS=0, and
        where _side_streams_ok says no).  A weight gradient needs only its layer's output gradient and saved input -- nothing downstream
        waits for it but the gradient bookkeeping -- and every launch ends in a tail (the last round's workgroups draining tile-sized
        atomics: 14 % of the weight-gradient time, profiles/r05_wgrad_stream.txt) that the data gradients of the main stream run under."""
        n = int(os.environ.get("RN_WGRAD_STREAMS", "1"))
        if n <= 0 or not self._side_streams_ok(device):
            return None
        st = self._tower_streams.get(("wgrad", device, n))
        if st is None:
            st = self._tower_streams[("wgrad", device, n)] = [torch.cuda.Stream(device) for _ in range(n)]
        return st

    def anchors(self, H, W, device):
        key = (H, W, str(device))
        if key not in self.anchor_cache:
            self.anchor_cache[key] = ops.anchors(H, W, device)
        return self.anchor_cache[key]

    def _prepare(self, P):
        for L in self.layers.values():
            L.prepare(P, self.cache)

    def _prepare_training(self, P):
        """Per-step preparation of a TRAINING step in two launches (rn_prep_batched): every batch-norm fold and forward
        weight pack, then every data-gradient pack (they read the folded scale).  The job tables and the destination
        buffers persist; they are rebuilt when a parameter tensor has been replaced."""
        key = tuple(P[n].data_ptr() for n in sorted(P) if n.endswith((".weight", ".bias", "running_mean", "running_var"))) \
            + (cv.get_fp32_mfma(), cv.PRESPLIT)
        prep = getattr(self, "_prep", None)
        if prep is None or prep["key"] != key:
            prep = self._prep = self._build_prep(P, key)
        lib = _hip.load()
        for jobs, chunks, n in prep["launches"]:
            _hip.check(lib.rn_prep_batched(jobs.data_ptr(), chunks.data_ptr(), n, _hip.stream()), "rn_prep_batched")
        for name, L in self.layers.items():
            b = prep["bufs"][name]
            L.adopt(P, self.cache, b["wf"], b.get("bn"), b.get("wd"), b.get("wino"), wf16=b.get("wf16"), wd16=b.get("wd16"))

    def _build_prep(self, P, key):
        dev = next(iter(P.values())).device
        launches = [([], []), ([], []), ([], [])]     # (jobs, chunks) of launch 0 (folds + forward packs), 1 (dgrad packs) and
        bufs = {}                                     # 2 (bf16 mode: the bf16 copies of all of them)

        def cast_job(src):
            """bf16 mode: a bf16 twin of a packed fp32 buffer, filled by launch 2."""
            dst = torch.empty(src.shape, dtype=torch.bfloat16, device=dev)
            j = _hip.PrepJob()
            j.kind, j.rows, j.Kpad = 3, src.shape[0], src.shape[1]
            j.src, j.dst = src.data_ptr(), dst.data_ptr()
            add(2, j, src.numel())
            return dst

        fp32_mode = cv.get_fp32_mfma()                # ("mode" is a loop variable further down)
        presplit = cv.PRESPLIT and fp32_mode != "native" and not self.bwd16

        def split_job(src, force=False, f16=None):
            """split modes: the pre-split twin of a packed fp32 buffer (cv.split_weights' / split_weights_f16's attribute), filled by
            launch 2.  force: the stem of the bf16 / fp8 engines, whose one-term products read the twin's first plane in every mode.
            f16 = (rows, cin, taps) of the GEMM the buffer feeds: split3 mode writes the fp16 two-term form where the fp16 kernels
            take the layer (cv.f16_shape_ok), the three-term bf16 form elsewhere."""
            if not force and (not presplit or src.shape[-1] < _hip.load().rn_fp32_split_min_k()):
                return
            rows = src.numel() // src.shape[-1]
            j = _hip.PrepJob()
            j.rows, j.Kpad, j.src = rows, src.shape[-1], src.data_ptr()
            if not force and fp32_mode == "split3" and f16 is not None and cv.f16_shape_ok(*f16):
                dst = (torch.empty(src.numel() * 4, dtype=torch.uint8, device=dev), torch.empty(rows, dtype=torch.float32, device=dev))
                j.kind, j.dst, j.bn_scale = 5, dst[0].data_ptr(), dst[1].data_ptr()
                add(2, j, (rows + 3) // 4 * 256)
                src._rn_split16 = dst
                return
            dst = torch.empty(src.numel() * 6, dtype=torch.uint8, device=dev)
            j.kind, j.dst = 4, dst.data_ptr()
            add(2, j, src.numel() // 8)
            src._rn_split = dst

        def add(which, job, nelem):
            jobs, chunks = launches[which]
            jobs.append(job)
            chunks.extend((len(jobs) - 1, b) for b in range((nelem + 255) // 256))

        def pack_job(w, dst, mode, kw_pad, c_pad, scale, taps=(0, 0, 0, 0)):
            cout, cin, kh, kw = w.shape
            j = _hip.PrepJob()
            j.kind, j.Cout, j.Cin, j.kh, j.kw, j.kw_pad, j.c_pad, j.mode = 1, cout, cin, kh, kw, kw_pad, c_pad, mode
            j.r0, j.nr, j.s0, j.ns = taps
            j.rows, j.Kpad = dst.shape
            j.src, j.dst, j.scale = w.data_ptr(), dst.data_ptr(), None if scale is None else scale.data_ptr()
            return j

        for name, L in self.layers.items():
            s = L.spec
            w = P[s.name + ".weight"]
            assert w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda
            cout, cin, kh, kw = w.shape
            b = bufs[name] = {}
            b["wf"] = torch.empty((cout, (kh * L.kw_pad * L.cin_pad + 31) // 32 * 32), dtype=torch.float32, device=dev)
            add(0, pack_job(w, b["wf"], 0, L.kw_pad, L.cin_pad, None), b["wf"].numel())
            split_job(b["wf"], force=getattr(L, "bf16_products", False), f16=(cout, L.cin_pad, kh * L.kw_pad))
            if L.bf16:
                b["wf16"] = cast_job(b["wf"])
            scale = None
            if s.bn:
                bn = b["bn"] = torch.empty((3, cout), dtype=torch.float32, device=dev)
                j = _hip.PrepJob()
                j.kind, j.Cout, j.eps = 0, cout, arch.BN_EPS
                j.gamma, j.beta = P[s.bn + ".weight"].data_ptr(), P[s.bn + ".bias"].data_ptr()
                j.mean, j.var = P[s.bn + ".running_mean"].data_ptr(), P[s.bn + ".running_var"].data_ptr()
                j.bn_scale, j.bn_shift, j.bn_rstd = bn[0].data_ptr(), bn[1].data_ptr(), bn[2].data_ptr()
                add(0, j, cout)
                scale = bn[0]
            if self.use_wino and L.wino_layer:            # Winograd layers: transformed weights instead of dgrad packs
                uf = torch.empty((36, cout, (cin + 31) // 32 * 32), dtype=torch.float32, device=dev)
                ud = torch.empty((36, cin, (cout + 31) // 32 * 32), dtype=torch.float32, device=dev)
                for which, mode, dst in ((0, 0, uf), (1, 1, ud)):
                    j = _hip.PrepJob()
                    j.kind, j.Cout, j.Cin, j.mode = 2, cout, cin, mode
                    j.rows, j.Kpad = dst.shape[1], dst.shape[2]
                    j.src, j.dst, j.scale = w.data_ptr(), dst.data_ptr(), None if (mode == 0 or scale is None) else scale.data_ptr()
                    add(which, j, dst.shape[1] * dst.shape[2])
                    split_job(dst, f16=(dst.shape[1], dst.shape[2], 1))
                b["wino"] = (uf, ud)
                continue
            if name == "conv1":
                continue                              # the stem has no data gradient
            if s.stride == 2 and s.k > 1:
                b["wd"] = []
                for c in cv.s2_classes(s.k, s.pad):
                    r0, nr, s0, ns = c[2]
                    d = torch.empty((cin, (nr * ns * L.cout_pad + 31) // 32 * 32), dtype=torch.float32, device=dev)
                    add(1, pack_job(w, d, 2, kw, L.cout_pad, scale, c[2]), d.numel())
                    split_job(d, f16=(cin, L.cout_pad, nr * ns))
                    b["wd"].append(d)
                if L.bf16:
                    b["wd16"] = [cast_job(d) for d in b["wd"]]
            else:
                d = b["wd"] = torch.empty((cin, (kh * kw * L.cout_pad + 31) // 32 * 32), dtype=torch.float32, device=dev)
                add(1, pack_job(w, d, 1, kw, L.cout_pad, scale), d.numel())
                split_job(d, f16=(cin, L.cout_pad, kh * kw))
                if L.bf16:
                    b["wd16"] = cast_job(d)
        out = []
        for jobs, chunks in launches:
            if not jobs:
                continue
            arr = (_hip.PrepJob * len(jobs))(*jobs)
            jt = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            ct = torch.tensor(chunks, dtype=torch.int32).to(dev)
            out.append((jt, ct, len(chunks)))
        return {"key": key, "launches": out, "bufs": bufs}

    def _unpack_bucket(self, index, layers, views):
        """rn_unpack_batched over the layers of one bucket: accumulated packed gradients -> the parameters' slots of the flat
        buffer.  The job table lives on the device and is rebuilt only when one of the pointers in it has changed."""
        key = tuple((L.dw.data_ptr(), L.wf.data_ptr(), views[L.grad_names()[0]].data_ptr(), _hip.ptr(L.scale) or 0, _hip.ptr(L.mean) or 0)
                    for L in layers)
        cache = self.__dict__.setdefault("_unpack_tables", {})
        hit = cache.get(index)
        if hit is None or hit[0] != key:
            jobs, chunks = [], []
            for L in layers:
                s, names = L.spec, L.grad_names()
                cout, cin, kh, kw = L.weight.shape
                j = _hip.UnpackJob()
                j.dw, j.w_packed, j.dweight = L.dw.data_ptr(), L.wf.data_ptr(), views[names[0]].data_ptr()
                j.Cout, j.Cin, j.kh, j.kw, j.kw_pad, j.c_pad, j.Kpad = cout, cin, kh, kw, L.kw_pad, L.cin_pad, L.wf.shape[1]
                j.scale, j.mean, j.rstd, j.colsum = _hip.ptr(L.scale), _hip.ptr(L.mean), _hip.ptr(L.rstd), L.cs.data_ptr()
                j.dgamma = views[names[1]].data_ptr() if s.bn else None
                j.dbeta = views[names[-1]].data_ptr() if len(names) > 1 else None
                chunks.extend((len(jobs), co) for co in range(cout))
                jobs.append(j)
            arr = (_hip.UnpackJob * len(jobs))(*jobs)
            dev = layers[0].dw.device
            hit = cache[index] = (key, torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev),
                                  torch.tensor(chunks, dtype=torch.int32).to(dev), len(chunks))
        _hip.check(_hip.load().rn_unpack_batched(hit[1].data_ptr(), hit[2].data_ptr(), hit[3], _hip.stream()), "rn_unpack_batched")
        for L in layers:
            L.dw = L.cs = L.wd = L.uf = L.ud = L.saved_v = L.dy_v = None

    def _zero_grad_accumulators(self, device):
        """Weight-gradient and column-sum accumulators of every layer -- and the Winograd-domain accumulators dU of the layers that
        take that path -- as views of ONE buffer, zeroed by one fill per step (they are atomically accumulated into: ~160 separate
        zero-fills per step otherwise)."""
        wino = self.use_wino and not self.bwd16
        sizes = []
        for L in self.layers.values():
            s = L.spec
            du = 36 * s.cout * ((s.cin + 31) // 32 * 32) if (wino and L.wino_layer) else 0
            sizes.append((L, L.wf.numel(), (s.cout + 3) // 4 * 4, du))
        total = sum(a + b + c for _, a, b, c in sizes)
        arena = getattr(self, "_arena", None)
        if arena is None or arena.numel() != total or arena.device != device:
            arena = self._arena = torch.empty(total, dtype=torch.float32, device=device)
        arena.zero_()
        o = 0
        for L, a, b, c in sizes:
            L.dw = arena[o:o + a].view_as(L.wf)
            L.cs = arena[o + a:o + a + L.spec.cout]
            L.du = arena[o + a + b:o + a + b + c].view(36, L.spec.cout, -1) if c else None
            o += a + b + c

    # ------------------------------------------------------------------------------------------- fp8 calibration
    def calibrate(self, P, frames, margin=1.0):
        """Per-tensor activation scales of the fp8 path from THIS engine's own fp32 / bf16 forward on `frames` ([B,3,H,W],
        representative inputs): the largest magnitude every convolution output (and the pooled stem) takes, times margin.
        -> dict for Engine(dtype="fp8").fp8_scales.  One host read per layer: a set-up call, not a hot path."""
        if self.fp8 or self.bf16:
            raise RuntimeError("calibrate with the fp32 engine (ResNet.calibrate_fp8 switches to it): the magnitudes are recorded "
                               "by the fp32 direct kernels' path only")
        rec = {}
        wino_eval, self.wino_eval = self.wino_eval, False      # the Winograd branches return before the tap: direct kernels here
        for L in self.layers.values():
            L.calib = rec
        self.calib = rec
        try:
            with torch.no_grad():
                self.forward(P, frames, save=False)
        finally:
            for L in self.layers.values():
                L.calib = None
            self.calib = None
            self.wino_eval = wino_eval
        missing = [n for n in self.fp8_scale_names() if n not in rec]
        assert not missing, "calibration pass did not record %s" % missing
        return {k: v * margin for k, v in rec.items()}

    def fp8_scale_names(self):
        """Names the fp8 forward needs an activation scale for: the pooled stem and every convolution whose output is stored as
        e4m3 (all but the fp32 stem and the fp32 head outputs)."""
        return ["pool"] + [n for n in self.layers if n != "conv1" and not n.endswith(".output")]

    def set_fp8_layers(self, fp8=None, other="fp32"):
        """fp8 engine: which layers run their forward in fp8 (names, or a predicate on the name; None = all, the default), the others in
        `other` ("fp32" or "bf16").  The stem is fp32 and the head outputs are fp32 results in any case.  Formats change at the
        boundaries by themselves (Layer.fwd).  -> the names that run in fp8."""
        assert self.fp8 and other in ("fp32", "bf16")
        pick = (lambda n: True) if fp8 is None else (fp8 if callable(fp8) else (lambda n, names=set(fp8): n in names))
        self._fp8_explicit = True                           # from now on the caller's choice stands (not fp8_policy)
        on = []
        for n, L in self.layers.items():
            if n == "conv1":
                continue
            L.fwd_mode = "fp8" if pick(n) else other
            if L.fwd_mode == "fp8":
                on.append(n)
        return on

    # ------------------------------------------------------------------------------------------- forward
    def forward(self, P, img, save, x4=None):
        """img [B,3,H,W] on device -> (reg [B,A,n_reg], cls [B,A,C], saved activations or None).  x4: the input already
        in the stem's layout ([B,H,W,4] fp32, e.g. from ops.frame_ingest(nhwc4=True)); img is then ignored."""
        Ls = self.layers
        if save and os.environ.get("RN_BATCHED_PREP", "1") != "0":
            self._prepare_training(P)
        else:
            self._prepare(P)
        if self.fp8:
            if save and not self.fp8_trainable:
                raise RuntimeError("the fp8 training step (fp8 forward, bf16 gradients) is built for the bottleneck networks "
                                   "(resnet50/101/152), like the bf16 schedule")
            if not self.fp8_scales:
                raise RuntimeError("the fp8 engine needs activation scales: net.calibrate_fp8(frames) first")
            missing = [n_ for n_ in self.fp8_scale_names() if not self.fp8_scales.get(n_, 0.0) > 0.0]
            if missing:                                     # a scale of 0 would saturate every activation to +-448: finite garbage
                raise RuntimeError("fp8 activation scales are missing (or zero) for %d tensors, e.g. %s: run net.calibrate_fp8(frames) "
                                   "on representative frames" % (len(missing), ", ".join(missing[:4])))
            if getattr(self, "_fp8_explicit", None) is None:    # (set_fp8_layers overrides the policy)
                stream = (lambda n_: n_.endswith(".conv3") or ".downsample." in n_ or n_.startswith("fpn.") or n_ == "regressionModel.conv4") \
                    if self.fp8_policy == "stream_bf16" else (lambda n_: False)
                for n_, L in Ls.items():
                    if n_ != "conv1":
                        L.fwd_mode = "bf16" if stream(n_) else "fp8"
            if save and any(L.mode() == "fp32" for n_, L in Ls.items() if n_ != "conv1"):
                raise RuntimeError("the fp8 training step saves e4m3 / bf16 activations for the bf16 gradient kernels: no fp32-format layers")
            for n_, L in Ls.items():
                L.out_scale = max(self.fp8_scales.get(n_, 0.0), 1e-30) / cv.FP8_MAX
            # the five pyramid maps share ONE scale, so that the towers' first layer takes them in one grouped launch
            pyr = ("fpn.P3_2", "fpn.P4_2", "fpn.P5_2", "fpn.P6", "fpn.P7_2")
            common = max(Ls[n_].out_scale for n_ in pyr)
            for n_ in pyr:
                Ls[n_].out_scale = common
        for L in Ls.values():                              # Winograd where it pays; in inference only on request
            L.wino_active = bool((save or self.wino_eval) and self.use_wino and L.wino_layer and not self.bf16 and not self.fp8)
            L.keep_v = bool(save)                          # the input transform is kept only when a backward will follow
            # ReLU outputs leave their sign bits for the backward pass (fp32 engine; bf16: opt-in, measured neutral)
            L.sign = bool(save) and not L.fp8 and (not L.bf16 or cv.BITMASKS_BF16)
        if x4 is None:
            _hip.need_gpu(img)
            x4 = cv.nchw_to_nhwc4(img)
        else:
            _hip.need_gpu(x4)
        B, H, W, _ = x4.shape
        S = {} if save else None
        stem = Ls["conv1"].fwd(x4, act=cv.ACT_RELU)
        pool = cv.maxpool_fwd_bf16 if self.bf16 else cv.maxpool_fwd       # bf16 mode: the fp32 stem ends here
        if self.fp8 and not save and getattr(self, "calib", None) is None:
            # fp8 mode: the fp32 stem ends here -- max-pool and quantisation in one pass (no fp32 pooled tensor)
            x = cv.maxpool_fwd_fp8(stem, max(self.fp8_scales.get("pool", 0.0), 1e-30) / cv.FP8_MAX)
        else:
            if save:
                x, pool_arg = pool(stem, want_argmax=True)
                S["x4"], S["stem"], S["pool_arg"] = x4, stem, pool_arg
            else:
                x = pool(stem)
            if getattr(self, "calib", None) is not None:
                self.calib["pool"] = max(self.calib.get("pool", 0.0), float(x.abs().max()))
            if self.fp8:                                    # (a saving or calibrating pass: pool, then quantise)
                x = cv.fp8_quantize(x, max(self.fp8_scales.get("pool", 0.0), 1e-30) / cv.FP8_MAX)
        if save:
            S["blocks"] = []
        feats = {}
        for pre, roles in self.blocks:
            xin = x
            t1 = roles["conv1"].fwd(xin, act=cv.ACT_RELU)
            res = roles["down"].fwd(xin) if "down" in roles else xin
            if self.kind == "basic":
                t2 = None
                x = roles["conv2"].fwd(t1, act=cv.ACT_RELU, add=res, add_mode=1)
            else:
                t2 = roles["conv2"].fwd(t1, act=cv.ACT_RELU)
                x = roles["conv3"].fwd(t2, act=cv.ACT_RELU, add=res, add_mode=1)
            if save:
                S["blocks"].append((xin, t1, t2, x))
            feats[pre.split(".")[0]] = x
        c3, c4, c5 = feats["layer2"], feats["layer3"], feats["layer4"]
        # ---- FPN (D/model.py:84-117)
        p5lat = Ls["fpn.P5_1"].fwd(c5)
        p5 = Ls["fpn.P5_2"].fwd(p5lat)
        p4sum = Ls["fpn.P4_1"].fwd(c4, add=p5lat, add_mode=2, add_hw=(p5lat.shape[1], p5lat.shape[2]))
        p4 = Ls["fpn.P4_2"].fwd(p4sum)
        p3sum = Ls["fpn.P3_1"].fwd(c3, add=p4sum, add_mode=2, add_hw=(p4sum.shape[1], p4sum.shape[2]))
        p3 = Ls["fpn.P3_2"].fwd(p3sum)
        p6 = Ls["fpn.P6"].fwd(c5, sign=True)               # stored before its ReLU; P7_2's data gradient is masked by its sign
        p6r = cv.relu_bf16(p6) if self.bf16 else None                      # fp32: the ReLU rides on the fragments (in_relu)
        if self.fp8:                                                       # (a 17 x 30 map: through fp32, same scale; any format P6 ran in)
            p6r = torch.relu(_as_f32(p6))
            p6r._rn_qscale = _qscale(p6)
        p7 = Ls["fpn.P7_2"].fwd(p6r) if (self.bf16 or self.fp8) else Ls["fpn.P7_2"].fwd(p6, in_relu=True)
        pyramid = [p3, p4, p5, p6, p7]
        if save:
            S["fpn"] = (c3, c4, c5, p5lat, p4sum, p3sum, p6)
            S["p6r"] = p6r
            S["pyramid"] = pyramid
        # ---- heads: towers share weights across levels; outputs land in their slice of [B, A, n]
        counts = [f.shape[1] * f.shape[2] * arch.NUM_ANCHORS for f in pyramid]
        A = sum(counts)
        reg = torch.empty((B, A, self.n_reg), dtype=torch.float32, device=x4.device)
        cls = torch.empty((B, A, self.num_classes), dtype=torch.float32, device=x4.device)
        if save:
            S["towers"] = {"regressionModel": [], "classificationModel": []}
            S["counts"] = counts
        # The two towers are independent chains: each runs on a stream of its own (tower_streams), so that one tower's HBM-bound
        # Winograd transforms and store phases overlap the other's MFMA-bound GEMMs.  The classification tower's first layer reuses
        # the regression tower's input transform of the pyramid: it waits for that one launch, not for the tower.
        side = self.tower_streams(x4.device)
        main = torch.cuda.current_stream(x4.device)
        v_ready = None
        if side is not None:
            fork = main.record_event()
        for ti, (prefix, out, width, act) in enumerate((("regressionModel", reg, self.n_reg, cv.ACT_NONE),
                                                        ("classificationModel", cls, self.num_classes, cv.ACT_SIGMOID))):
            if side is not None:
                side[ti].wait_event(fork)
                if ti == 1 and v_ready is not None:
                    side[ti].wait_event(v_ready)
            with torch.cuda.stream(side[ti] if side is not None else main):
                ts = pyramid
                acts = []                                                 # acts[i][level]
                for i in range(1, 5):                                     # one launch per tower conv, all 5 levels
                    first = Ls["regressionModel.conv1"]                   # both towers' conv1 transform the same pyramid: once
                    shared = first.saved_v if (i == 1 and prefix == "classificationModel") else None
                    ts = Ls["%s.conv%d" % (prefix, i)].fwd_group(ts, act=cv.ACT_RELU, wino=save and self.use_wino and not self.bf16 and not self.fp8,
                                                                 shared_v=shared)
                    acts.append(ts)
                    if side is not None and ti == 0 and i == 1:
                        v_ready = side[0].record_event()
                views, off = [], 0
                for cnt in counts:
                    views.append(out.view(B, -1)[:, off * width:])
                    off += cnt
                Ls[prefix + ".output"].fwd_group(ts, act=act, outs=views, y_batch_stride=A * width)
            if save:
                S["towers"][prefix] = [[acts[i][li] for i in range(4)] for li in range(5)]
        if side is not None:
            main.wait_stream(side[0])
            main.wait_stream(side[1])
        if save:
            # The Winograd input transforms kept for the weight gradients belong to THIS call: a second forward before
            # this one's backward must not replace them (the Layer objects are shared between calls).
            S["wino_v"] = {n: L.saved_v for n, L in Ls.items() if L.saved_v is not None}
        for L in Ls.values():
            L.saved_v = None
        return reg, cls, S

    # ------------------------------------------------------------------------------------------- backward
    def backward(self, S, dreg, dcls, cls):
        """dreg [B,A,n_reg], dcls [B,A,C] (gradient w.r.t. the post-sigmoid classification) -> {param: grad}."""
        Ls = self.layers
        grads = {}
        self._zero_grad_accumulators(dreg.device)
        for L in Ls.values():
            L.saved_v = L.dy_v = None
        for n, v in S.pop("wino_v", {}).items():          # this call's Winograd input transforms (see forward)
            Ls[n].saved_v = v

        flat = views = None
        next_bucket = 0
        if self.flat_bucket_bytes:
            flat, views = self._flat_views(dreg.device)
            buckets = self._flat["buckets"]

        # Flat buffer and no per-layer hook: the layers of a bucket are unpacked by ONE launch when the bucket's last layer
        # retires (rn_unpack_batched) instead of one launch per layer.
        batched = flat is not None and self.grad_hook is None and os.environ.get("RN_BATCHED_UNPACK", "1") != "0"
        pending = []

        # Weight gradients on a stream of their own (wgrad_stream), joined where their results are first read: the bucket's unpack launch.
        # (Only with the batched unpack: a per-layer finish would join after every layer.)
        main0 = torch.cuda.current_stream(dreg.device) if dreg.is_cuda else None
        wss = self.wgrad_streams(dreg.device) if batched else None
        wg_count = 0
        held = []

        def join():
            """The main stream waits for the weight gradients queued so far; what they read may then be released."""
            for ws in wss or ():
                main0.wait_stream(ws)
                cv.side_release(ws)
            held.clear()

        def aside(fn, tensors):
            """fn() on the weight-gradient stream, after everything queued so far on the current one; `tensors`: what it reads."""
            nonlocal wg_count
            ws = wss[wg_count % len(wss)]
            wg_count += 1
            if cv.want_amax():
                # amax tables of the operands exist BEFORE the fork: computed lazily over there (a tensor whose producer left none), the
                # table would be cached on the tensor and read by this stream's next kernel with nothing ordering the two
                for t in tensors:
                    if t.dtype == torch.float32 and t.is_contiguous():
                        cv.amax_words(t)
            ws.wait_event(torch.cuda.current_stream(dreg.device).record_event())   # the operands and the zeroed accumulators are ready
            with torch.cuda.stream(ws):
                fn()
            # their memory is not to be reused before the side stream is through: held until the main stream has waited for it (join);
            # record_stream would leave that to the allocator's event polling -- see cv.SIDE_HELD
            held.append(tuple(tensors) + tuple(getattr(t, "_rn_amax", (None,))[0] for t in tensors))

        def wg(L, g, x, in_relu=False):
            """L.bwd_params(g, x) beside the data gradients.  A Winograd layer transforms g here (its data gradient wants one of the two
            transforms) and only the reductions go aside (cv.wino_wgrad_group(side=...))."""
            if wss is None:
                L.bwd_params(g, x, in_relu=in_relu)
            elif L.wino_active and not L.bf16:
                L.bwd_params(g, x, in_relu=in_relu, side=wss[0] if os.environ.get("RN_WGRAD_WINO", "1") != "0" else None)
            else:
                aside(lambda: L.bwd_params(g, x, in_relu=in_relu), (g, x))

        def wg_group(L, gs, xs, wino=False):
            """L.bwd_params_group likewise (the head towers' layers over the pyramid levels)."""
            if wss is None or os.environ.get("RN_WGRAD_TOWERS", "0") == "0":
                L.bwd_params_group(gs, xs, wino=wino)
            elif wino or L.wino_active:
                L.bwd_params_group(gs, xs, wino=wino, side=wss[0])
            else:
                aside(lambda: L.bwd_params_group(gs, xs), tuple(gs) + tuple(xs))

        def done(layer):
            nonlocal next_bucket
            if batched:
                pending.append(layer)
                grads.update({n: views[n] for n in layer.grad_names()})
            else:
                g = layer.finish(views)
                grads.update(g)
                if self.grad_hook is not None:
                    self.grad_hook(g)
            if flat is not None and layer.spec.name == buckets[next_bucket][2]:
                if batched:
                    join()
                    self._unpack_bucket(next_bucket, pending, views)
                    pending.clear()
                if self.bucket_hook is not None:             # every gradient of this slice is final: release it
                    self.bucket_hook(next_bucket, flat[buckets[next_bucket][0]:buckets[next_bucket][1]])
                next_bucket += 1

        # fp8-forward training: the saved activations are e4m3; the bf16 gradient kernels read bf16 (rn_fp8_to_bf16 at the point of use)
        ACT = _as_bf16 if self.fp8 else (lambda t: t)
        pyramid, counts = [ACT(t) for t in S["pyramid"]], S["counts"]
        B = dreg.shape[0]
        A = dreg.shape[1]
        dpyr = [None] * 5
        # ---- heads
        hws = [(f.shape[1], f.shape[2]) for f in pyramid]
        # Two streams as in forward.  The towers meet in the pyramid's gradient: the second tower's last data gradient adds the
        # first tower's, so it waits for that tower's end there; the gradient bookkeeping (done) runs on the main stream after both.
        side = self.tower_streams(dreg.device)
        main = torch.cuda.current_stream(dreg.device)
        if side is not None:
            fork = main.record_event()
        retired = []
        for ti, (prefix, dout, width, sig) in enumerate((("regressionModel", dreg, self.n_reg, None),
                                                         ("classificationModel", dcls, self.num_classes, cls))):
            Lout = Ls[prefix + ".output"]
            tower = [Ls["%s.conv%d" % (prefix, i)] for i in range(1, 5)]
            if side is not None:
                side[ti].wait_event(fork)
            with torch.cuda.stream(side[ti] if side is not None else main):
                acts = [[ACT(t) for t in lvl] for lvl in S["towers"][prefix]]   # acts[level][i] (fp8 training: as bf16)
                gs, off = [], 0
                for (Hh, Ww), cnt in zip(hws, counts):                    # head-output gradient slices -> dense, padded
                    byte_off = 4 * off * width
                    g = cv.sigmoid_bwd_pad(dout.data_ptr() + byte_off, None if sig is None else sig.data_ptr() + byte_off,
                                           B, Hh * Ww, arch.NUM_ANCHORS * width, Lout.cout_pad, A * width, dout.device,
                                           bf16=self.bwd16)
                    gs.append(cv.amax_carry(g.view(B, Hh, Ww, Lout.cout_pad), g))
                    off += cnt
                wg_group(Lout, gs, [acts[li][3] for li in range(5)])          # direct: one launch per level (K slices fill the GPU)
                gs = Lout.bwd_data_group(gs, hws, masks=[acts[li][3] for li in range(5)])
                for i in (3, 2, 1):
                    wg_group(tower[i], gs, [acts[li][i - 1] for li in range(5)], wino=self.use_wino and not self.bwd16)
                    gs = tower[i].bwd_data_group(gs, hws, masks=[acts[li][i - 1] for li in range(5)], wino=self.use_wino and not self.bwd16)
                wg_group(tower[0], gs, pyramid, wino=self.use_wino and not self.bwd16)
                first = dpyr[0] is None
                if side is not None and not first:
                    side[ti].wait_stream(side[0])                         # the other tower's pyramid gradient is complete
                dpyr = tower[0].bwd_data_group(gs, hws, adds=None if first else dpyr, wino=self.use_wino and not self.bwd16)
            S["towers"][prefix] = None
            retired.append(Lout)
            retired.extend(reversed(tower))
        if side is not None:
            main.wait_stream(side[0])
            main.wait_stream(side[1])
        for L in retired:
            done(L)
        # ---- FPN
        c3, c4, c5, p5lat, p4sum, p3sum, p6 = (ACT(t) for t in S["fpn"])
        dp3, dp4, dp5, dp6, dp7 = dpyr
        hw = lambda t: (t.shape[1], t.shape[2])
        L = Ls["fpn.P7_2"]
        if self.bwd16:
            wg(L, dp7, ACT(S["p6r"]))
        else:
            wg(L, dp7, p6, in_relu=True)
        dp6 = L.bwd_data(dp7, hw(p6), add=dp6, mask=p6, mask_mode=1)     # d relu(p6) masked, heads' part added raw
        done(L)
        L = Ls["fpn.P6"]
        wg(L, dp6, c5)
        dc5 = L.bwd_data(dp6, hw(c5))
        done(L)
        L = Ls["fpn.P3_2"]
        wg(L, dp3, p3sum)
        dp3sum = L.bwd_data(dp3, hw(p3sum))
        done(L)
        L = Ls["fpn.P3_1"]
        wg(L, dp3sum, c3)
        dc3 = L.bwd_data(dp3sum, hw(c3))
        done(L)
        L = Ls["fpn.P4_2"]
        wg(L, dp4, p4sum)
        dp4sum = L.bwd_data(dp4, hw(p4sum))
        cv.upsample_add_bwd(dp3sum, dp4sum)
        done(L)
        L = Ls["fpn.P4_1"]
        wg(L, dp4sum, c4)
        dc4 = L.bwd_data(dp4sum, hw(c4))
        done(L)
        L = Ls["fpn.P5_2"]
        wg(L, dp5, p5lat)
        dp5lat = L.bwd_data(dp5, hw(p5lat))
        cv.upsample_add_bwd(dp4sum, dp5lat)
        done(L)
        L = Ls["fpn.P5_1"]
        wg(L, dp5lat, c5)
        dc5 = L.bwd_data(dp5lat, hw(c5), add=dc5, mask=c5)                # both consumers in: ReLU mask of C5
        done(L)
        # ---- backbone, last block first.  `g` = gradient w.r.t. the block's pre-ReLU output, already masked.
        lateral = {"layer2": dc3, "layer3": dc4}            # extra consumer of these layers' outputs (FPN)
        g = dc5
        nblocks = len(self.blocks)
        for bi in range(nblocks - 1, -1, -1):
            pre, roles = self.blocks[bi]
            xin, t1, t2, z = (ACT(t) for t in S["blocks"][bi])
            S["blocks"][bi] = None
            in_hw = hw(xin)
            last = roles["conv2"] if self.kind == "basic" else roles["conv3"]
            tin = t1 if self.kind == "basic" else t2
            wg(last, g, tin)
            gt = last.bwd_data(g, hw(tin), mask=tin)
            done(last)
            if self.kind != "basic":
                wg(roles["conv2"], gt, t1)
                gt = roles["conv2"].bwd_data(gt, hw(t1), mask=t1)
                done(roles["conv2"])
            # gradient reaching the block input: conv1 path + residual path (+ FPN lateral on layer outputs)
            layer_name = pre.split(".")[0]
            first_of_layer = pre.endswith(".0")
            extra = None
            if first_of_layer and bi > 0:
                prev_layer = self.blocks[bi - 1][0].split(".")[0]
                extra = lateral.get(prev_layer)
            dcompact = None
            if "down" in roles:
                wg(roles["down"], g, xin)
                if roles["down"].spec.stride == 2 and roles["conv1"].spec.stride == 1 and not self.bwd16:
                    dcompact = roles["down"].bwd_data_compact(g)       # bottleneck: conv1 is 1x1 s1, takes add2
                    dres = extra
                else:
                    dres = roles["down"].bwd_data(g, in_hw, add=extra)
                done(roles["down"])
            else:
                dres = g if extra is None else cv.add_(extra, g)
            wg(roles["conv1"], gt, xin)
            # the block input is a ReLU output (previous block) except for the very first block (max-pool output)
            g = roles["conv1"].bwd_data(gt, in_hw, add=dres, mask=xin if bi > 0 else None, add2=dcompact)
            done(roles["conv1"])
            del layer_name
        # ---- stem
        gstem = (cv.maxpool_bwd_bf16 if self.bwd16 else cv.maxpool_bwd)(S["stem"], g, S["pool_arg"], relu_mask=True)
        wg(Ls["conv1"], gstem, S["x4"])
        done(Ls["conv1"])
        join()
        assert flat is None or next_bucket == len(buckets), "backward finished layers in an order finish_order() does not describe"
        return grads
