"""Thin wrappers over the convolution-engine entry points of the C ABI (NHWC fp32 device tensors).

These are the building blocks ``engine.py`` schedules; they do no autograd themselves.
"""
import ctypes
import os

import torch

from . import _hip, prof

ConvDesc = _hip.ConvDesc

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def kpad(kh, kw, c):
    return (kh * kw * c + 31) // 32 * 32


def out_size(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


def pack_weights(weight, mode=0, scale=None, kw_pad=None, c_pad=None, taps=None, presplit=True):
    """OIHW parameter -> packed GEMM rows (mode 0: [Cout][kh][kw][Cin]; mode 1 (dgrad): [Cin][kh][kw][Cout]*scale;
    taps=(r0, nr, s0, ns): dgrad layout restricted to taps r0+2i, s0+2j -- one parity class of a stride-2 dgrad)."""
    lib = _hip.load()
    w = _hip.f32c(weight.detach())
    _hip.need_gpu(w, scale)
    cout, cin, kh, kw = w.shape
    kw_pad = kw if kw_pad is None else kw_pad
    if c_pad is None:
        c = cin if mode == 0 else cout
        c_pad = (c + 3) // 4 * 4
    rows = cout if mode == 0 else cin
    r0, nr, s0, ns = (0, 0, 0, 0) if taps is None else taps
    if taps is not None:
        mode = 2
    ktaps = nr * ns if taps is not None else kh * kw_pad
    out = torch.empty((rows, (ktaps * c_pad + 31) // 32 * 32), dtype=torch.float32, device=w.device)
    _hip.check(lib.rn_pack_weights(w.data_ptr(), out.data_ptr(), cout, cin, kh, kw, kw_pad, c_pad, mode,
                                   _hip.ptr(scale), r0, nr, s0, ns, _hip.stream()), "rn_pack_weights")
    return _maybe_split(out, rows=rows, cin=c_pad, taps=ktaps) if presplit else out        # split modes: with the pre-split twin attached


# Sign bits of ReLU outputs (round 4; csrc/common.h: rn_sign_store).  A producer called with sign=True also writes one bit per
# element of its result (y > 0) and attaches the words as y._rn_sign; a consumer whose `mask` tensor carries that attribute reads the
# bits instead of the fp32 activation (1/32 of the bytes; same mask, bit-identical gradients).  RN_BITMASKS=0: fp32 masks (A/B).
BITMASKS = os.environ.get("RN_BITMASKS", "1") != "0"
# The bf16 engine's tensors can carry them too (conv_bf16.hip), but there a mask costs 2 bytes, not 4, and the step measured the same
# with and without (194.8 / 195.3 against 194.1 / 194.8 images/s, profiles/r04_bitmask_ab_step.txt): opt-in.
BITMASKS_BF16 = os.environ.get("RN_BITMASKS_BF16", "0") == "1"
MASK_BITS = 4                                    # RN_MASK_BITS of include/retinanet_mi355x.h


def _sign_words(y, want):
    """-> int32 tensor for the sign bits of the dense fp32 / bf16 tensor y (attached as y._rn_sign), or None.  Called by every producer
    for its result: words an EARLIER producer left on the same tensor object (a caller reusing it through ``out=``) are dropped first,
    so a consumer never reads bits of a tensor that has been rewritten since.  (In-place edits of a producer's output by anything else
    -- torch operations included -- invalidate the bits as well: drop ``._rn_sign`` or use a fresh tensor.)"""
    if hasattr(y, "_rn_sign"):
        del y._rn_sign
    if not (want and BITMASKS) or y.dtype not in (torch.float32, torch.bfloat16) or y.shape[-1] % 32 or not y.is_contiguous():
        return None
    bits = torch.empty(y.numel() // 32, dtype=torch.int32, device=y.device)
    y._rn_sign = bits
    return bits


def _mask_operand(mask, mask_mode):
    """(pointer, mask_mode) of a mask operand: the sign bits its producer left, if any, else the fp32 tensor itself."""
    if mask is None:
        return None, 0
    bits = getattr(mask, "_rn_sign", None) if BITMASKS else None
    if bits is not None:
        return bits.data_ptr(), mask_mode | MASK_BITS
    return mask.data_ptr(), mask_mode


def conv_igemm(x, w_packed, y, geom, scale=None, shift=None, add=None, add_mode=0, add_hw=(0, 0), mask=None,
               mask_mode=2, act=ACT_NONE, y_batch_stride=None, add_batch_stride=None, in_relu=False, flops=0.0,
               out_map=None, add2=None, w_batch_stride=0, kind=None, sign=False, bf16_products=False, x_amax=None, x_amax_rows=False,
               amax=None):
    """Launch rn_conv_igemm.  x [N,Hi,Wi,Cin]; y a tensor whose storage receives [N,Ho,Wo,Cout] at batch stride
    y_batch_stride; geom = (Ho, Wo, Cout, kh, kw, a, b, p, div_shift) with p an int or (p_rows, p_cols).
    out_map = (os, oo_h, oo_w, Hy, Wy) stores output pixel (oh,ow) at (oh*os+oo_h, ow*os+oo_w) of a [N,Hy,Wy,Cout]
    tensor.  add2: [N,Ha2,Wa2,Cout] added at even stored positions (1x1 stride-2 shortcut gradient).
    split3 mode: x_amax = the amax words of x when the caller holds them (default: amax_words(x), one per image; x_amax_rows: one per GEMM
    row instead -- the Winograd stage); a dense result gets words of its own (y._rn_amax), `amax` = the words several launches that fill
    ONE tensor share (the parity classes of a stride-2 data gradient)."""
    lib = _hip.load()
    N, Hi, Wi, Cin = x.shape
    Ho, Wo, Cout, kh, kw, a, b, p, ds = geom
    p_h, p_w = p if isinstance(p, tuple) else (p, p)
    os_, oo_h, oo_w, Hy, Wy = (1, 0, 0, Ho, Wo) if out_map is None else out_map
    ybs = Hy * Wy * Cout if y_batch_stride is None else y_batch_stride
    if add_batch_stride is None:
        add_batch_stride = ybs if add_mode == 1 else add_hw[0] * add_hw[1] * Cout
    a2 = (0, 0, 0, 0) if add2 is None else (3, add2.shape[1], add2.shape[2], add2.shape[1] * add2.shape[2] * Cout)
    mask_ptr, mask_mode = _mask_operand(mask, mask_mode)
    d = ConvDesc(N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, a, b, p_h, p_w, ds, act, add_mode, add_hw[0], add_hw[1],
                 mask_mode, int(in_relu), os_, oo_h, oo_w, Hy, Wy,
                 a2[0], a2[1], a2[2], a2[3], Hi * Wi * Cin, ybs, add_batch_stride, w_batch_stride)
    bits = _sign_words(y, sign and y_batch_stride is None and out_map is None)
    d.sign_out = None if bits is None else bits.data_ptr()
    yam = None
    if want_amax() and not w_batch_stride and y_batch_stride is None and (out_map is None or amax is not None):
        yam = amax if amax is not None else amax_slot(y.device, N)
        d.y_amax = yam.data_ptr()
    if kind is None:
        kind = "conv_igemm_4x1" if Cout <= 64 else "conv_igemm_2x2"
    if prof.BY_SHAPE:                                    # profiling aid (tools/profile_layers.py): one row per layer shape
        kind += " %dx%dx%d %d->%d k%d a%d b%d ds%d%s" % (N, Ho, Wo, Cin, Cout, kh, a, b, ds, " x36" if w_batch_stride else "")
    ws_bytes = 0 if w_batch_stride else lib.rn_conv_splitk_workspace_bytes(ctypes.byref(d))   # > 0: few output tiles, long K -> split-K
    if ws_bytes > 0:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        rc = prof.timed(kind, flops, lambda: lib.rn_conv_igemm_splitk(
            ctypes.byref(d), x.data_ptr(), w_packed.data_ptr(), y.data_ptr(), _hip.ptr(scale), _hip.ptr(shift),
            _hip.ptr(add), mask_ptr, _hip.ptr(add2), ws.data_ptr(), _hip.stream()))
        _hip.check(rc, "rn_conv_igemm_splitk")
        if yam is not None and amax is None:
            amax_attach(y, yam)
        return y
    wptr, d.w_format, d.w_unscale = _w_operand(w_packed, d)
    if d.w_format == 3:
        d.x_amax = (x_amax if x_amax is not None else amax_words(x)).data_ptr()
        d.x_amax_img_stride, d.x_amax_row_stride = (0, 1) if x_amax_rows else (1, 0)
    if bf16_products:                      # the fp32 stem of the bf16 / fp8 engines: products from the first bf16 terms (w_format 2)
        ws = getattr(w_packed, "_rn_split", None)
        if ws is None:
            ws = split_weights(w_packed)._rn_split
        wptr, d.w_format = ws.data_ptr(), 2
    nb = 0.0
    if prof.ACTIVE is not None:            # algorithmic bytes: every operand once (the split kernels read weights as 6-byte terms)
        out_el = N * Ho * Wo * Cout
        mask_el = 0.0 if mask is None else (1.0 / 32 if mask_mode & MASK_BITS else 1.0)
        nb = 4.0 * (x.numel() + out_el * (1 + (add is not None) + mask_el + (1.0 / 32 if bits is not None else 0.0))) \
            + {0: 4.0, 1: 6.0, 2: 6.0, 3: 4.0}[d.w_format] * w_packed.numel() + (4.0 * add2.numel() if add2 is not None else 0.0)
    rc = prof.timed(kind, flops, lambda: lib.rn_conv_igemm(
        ctypes.byref(d), x.data_ptr(), wptr, y.data_ptr(), _hip.ptr(scale), _hip.ptr(shift),
        _hip.ptr(add), mask_ptr, _hip.ptr(add2), _hip.stream()), nb)
    _hip.check(rc, "rn_conv_igemm")
    if yam is not None and amax is None:
        amax_attach(y, yam)
    return y


def _make_desc(x, geom, act, add_mode, add_hw, mask_mode, in_relu, out_map, y_batch_stride, add_batch_stride, add2):
    N, Hi, Wi, Cin = x.shape
    Ho, Wo, Cout, kh, kw, a, b, p, ds = geom
    p_h, p_w = p if isinstance(p, tuple) else (p, p)
    os_, oo_h, oo_w, Hy, Wy = (1, 0, 0, Ho, Wo) if out_map is None else out_map
    ybs = Hy * Wy * Cout if y_batch_stride is None else y_batch_stride
    if add_batch_stride is None:
        add_batch_stride = ybs if add_mode == 1 else add_hw[0] * add_hw[1] * Cout
    a2 = (0, 0, 0, 0) if add2 is None else (3, add2.shape[1], add2.shape[2], add2.shape[1] * add2.shape[2] * Cout)
    return ConvDesc(N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, a, b, p_h, p_w, ds, act, add_mode, add_hw[0], add_hw[1],
                    mask_mode, int(in_relu), os_, oo_h, oo_w, Hy, Wy, a2[0], a2[1], a2[2], a2[3],
                    Hi * Wi * Cin, ybs, add_batch_stride)


# ---------------------------------------------------------------------------------------------- Winograd F(4x4,3x3)
FP32_MFMA_MODES = ("native", "split", "split3")  # RN_FP32_NATIVE, RN_FP32_SPLIT, RN_FP32_SPLIT3 of include/retinanet_mi355x.h


def set_fp32_mfma(mode):
    """How the fp32 convolution kernels form their products: "native" (v_mfma_f32_32x32x2_f32), "split" (three-term
    bf16 splits of both fp32 operands on v_mfma_f32_32x32x16_bf16, six MFMAs per product) or "split3" (two-term fp16 splits of the
    operands scaled by powers of two, three MFMAs per product); fp32 accumulation in all (csrc/mfma_split.h).  Process-wide."""
    _hip.check(_hip.load().rn_set_fp32_mfma(FP32_MFMA_MODES.index(mode)), "rn_set_fp32_mfma")


def get_fp32_mfma():
    return FP32_MFMA_MODES[_hip.load().rn_get_fp32_mfma()]


PRESPLIT = os.environ.get("RN_FP32_PRESPLIT", "1") != "0"      # split modes: prepare the weights' terms once (rn_split_weights*)


def split_weights(w_packed):
    """Attach the pre-split twin of a packed fp32 weight tensor ([..., rows, Kpad] -> rows * Kpad * 6 bytes,
    include/retinanet_mi355x.h: rn_split_weights) as w_packed._rn_split; the convolution wrappers pass it instead of the
    fp32 tensor in split mode.  Whoever rewrites w_packed in place refreshes the twin (the engine's batched preparation does)."""
    lib = _hip.load()
    kpad = w_packed.shape[-1]
    ws = getattr(w_packed, "_rn_split", None)
    if ws is None:
        ws = torch.empty(w_packed.numel() * 6, dtype=torch.uint8, device=w_packed.device)
    _hip.check(lib.rn_split_weights(w_packed.data_ptr(), ws.data_ptr(), w_packed.numel() // kpad, kpad, _hip.stream()), "rn_split_weights")
    w_packed._rn_split = ws
    return w_packed


def split_weights_f16(w_packed):
    """split3 mode: attach the fp16 two-term twin of a packed fp32 weight tensor (rn_split_weights_f16: rows * Kpad * 4 bytes, every
    row written with its own power-of-two scale) as w_packed._rn_split16 = (terms, inverse row scales [rows])."""
    lib = _hip.load()
    kpad = w_packed.shape[-1]
    rows = w_packed.numel() // kpad
    tw = getattr(w_packed, "_rn_split16", None)
    if tw is None:
        tw = (torch.empty(w_packed.numel() * 4, dtype=torch.uint8, device=w_packed.device),
              torch.empty(rows, dtype=torch.float32, device=w_packed.device))
    _hip.check(lib.rn_split_weights_f16(w_packed.data_ptr(), tw[0].data_ptr(), tw[1].data_ptr(), rows, kpad, _hip.stream()),
               "rn_split_weights_f16")
    w_packed._rn_split16 = tw
    return w_packed


def f16_shape_ok(rows, cin, taps=1):
    """Which packed tensors get the fp16 twin in split3 mode: all of them (every split kernel has the two-term form since the second
    half of round 5; before, the layers with at most 64 output channels / odd channel counts kept the three-term twin)."""
    return True


def _maybe_split(w_packed, rows=None, cin=None, taps=1):
    """split modes: attach the pre-split twin -- for reductions long enough to run on the split kernels (rn_fp32_split_min_k;
    the packed row is the reduction, padded).  split3: the fp16 twin where the fp16 kernels take the layer (f16_shape_ok)."""
    mode = get_fp32_mfma()
    if not PRESPLIT or mode == "native" or w_packed.shape[-1] < _hip.load().rn_fp32_split_min_k():
        return w_packed
    if mode == "split3" and rows is not None and f16_shape_ok(rows, cin, taps):
        return split_weights_f16(w_packed)
    return split_weights(w_packed)


def _carry_split(src, view):
    """A view of a packed tensor keeps its pre-split twins."""
    for name in ("_rn_split", "_rn_split16"):
        ws = getattr(src, name, None)
        if ws is not None:
            setattr(view, name, ws)
    return view


def _w_operand(w_packed, d=None):
    """(pointer, rn_conv_desc.w_format, inverse-row-scale pointer) of the weight operand for rn_conv_igemm / _grouped; d: the problem's
    descriptor (split3: the library says whether it runs on an fp16-split kernel, rn_conv_igemm_wants_f16)."""
    if PRESPLIT:
        mode = get_fp32_mfma()
        if mode == "split3" and d is not None:
            tw = getattr(w_packed, "_rn_split16", None)
            if tw is not None and _hip.load().rn_conv_igemm_wants_f16(ctypes.byref(d)):
                return tw[0].data_ptr(), 3, tw[1].data_ptr()
        ws = getattr(w_packed, "_rn_split", None)
        if ws is not None and mode != "native":
            return ws.data_ptr(), 1, None
    return w_packed.data_ptr(), 0, None


# ---- amax words (split3; include/retinanet_mi355x.h: rn_conv_desc.x_amax / y_amax): ONE WORD PER IMAGE of a [N, ...] tensor.  A producer
# launched with amax words leaves in word n the exponent of the largest magnitude it stored in image n; the tensor object carries
# (words, tensor._version at that time) as ._rn_amax.  A consumer takes the words if the version still matches (an in-place torch
# operation since then invalidates them; this module's own in-place kernels drop the attribute), else one rn_amax pass computes them.
# Words are views of zeroed chunks that are never reused.  Per image, so that an image's scales -- and every bit of its results -- do not
# depend on what else is in the batch.
_AMAX_CHUNK = {}


AMAX_SUB = 64                                    # int32 words of one image's exponent table (csrc/mfma_split.h: RN_AMAX_BYTES = 256)


def _amax_alloc(device, nwords):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    nwords = (nwords + 7) // 8 * 8                    # blocks start on 32-byte boundaries
    c = _AMAX_CHUNK.get(key)
    if c is None or c[1] + nwords > c[0].numel():
        c = _AMAX_CHUNK[key] = [torch.zeros(max(1 << 18, nwords), dtype=torch.int32, device=device), 0]     # 1 MB: ~500 tensors of 8 images
    c[1] += nwords
    return c[0][c[1] - nwords:c[1]]


def amax_slot(device, n=1):
    """Fresh zeroed amax words for a tensor of n images (AMAX_SUB sub-words each) on `device`."""
    return _amax_alloc(device, n * AMAX_SUB)


def amax_single(device):
    """One fresh zeroed plain word (a Winograd-domain tensor's, for the weight gradient)."""
    return _amax_alloc(device, 1)[:1]


def amax_attach(t, words):
    t._rn_amax = (words, t._version)
    return words


def amax_words(t):
    """The amax words [N] of a dense fp32 tensor [N, ...]: its producer's (see above) or computed now (cached on the tensor object)."""
    a = getattr(t, "_rn_amax", None)
    if a is not None and a[1] == t._version and a[0].numel() == t.shape[0] * AMAX_SUB:
        return a[0]
    assert t.is_contiguous() and t.dtype == torch.float32
    words = amax_slot(t.device, t.shape[0])
    if t.numel():
        _hip.check(_hip.load().rn_amax(t.data_ptr(), t.numel() // t.shape[0], t.shape[0], words.data_ptr(), _hip.stream()), "rn_amax")
    return amax_attach(t, words)


def amax_carry(view, src):
    """A reshaped view of a tensor whose producer left amax words (the padded head-gradient slices) takes them along."""
    w = getattr(src, "_rn_amax_words", None)
    if w is not None and w.numel() == view.shape[0] * AMAX_SUB:
        amax_attach(view, w)
    return view


def amax_drop(t):
    """After an in-place kernel of this module rewrote t."""
    if hasattr(t, "_rn_amax"):
        del t._rn_amax


def want_amax():
    """Producers leave amax words in split3 mode (their consumers need them)."""
    return get_fp32_mfma() == "split3"


# RN_OPT_* of include/retinanet_mi355x.h
OPT_SPLITK, OPT_DETERMINISTIC, OPT_MF16, OPT_MF16_MIN, OPT_WGRAD_ONCE, OPT_BF16_P8, OPT_FP8_P8 = range(7)


def set_option(option, value):
    """Process-wide run-time option of the library (0 / 1 switches take a bool; the tile-count threshold and the P8 selectors an int)."""
    _hip.check(_hip.load().rn_set_option(option, int(value)), "rn_set_option")


def get_option(option):
    return int(_hip.load().rn_get_option(option))


def set_deterministic(on=True):
    """Fixed-order weight-gradient reductions (RN_OPT_DETERMINISTIC; environment RN_DETERMINISTIC=1): two runs of a training
    step give bit-identical gradients.  fp32 engine (the bf16 weight gradient keeps its atomics)."""
    set_option(OPT_DETERMINISTIC, on)


def _wgrad_call(lib, dev, dy_ptr, ldy, x_ptr, dw_ptr, cs_ptr, nbatch, dy_bs, x_bs, dw_bs, cs_batch, geom, amax=(None, None)):
    """rn_conv_wgrad_batched, or its fixed-order form with a slab workspace when the deterministic option is on.
    geom = (N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, stride, pad, in_relu); amax = the amax words of (dy, x) (split3 mode) or Nones."""
    cnt = lambda a: 0 if a is None else (-1 if a.numel() == 1 else a.numel() // AMAX_SUB)      # tables of n images, or one plain word
    am = (_hip.ptr(amax[0]), cnt(amax[0]), _hip.ptr(amax[1]), cnt(amax[1]))
    if not lib.rn_get_option(OPT_DETERMINISTIC):
        return lib.rn_conv_wgrad_batched(dy_ptr, ldy, x_ptr, dw_ptr, cs_ptr, nbatch, dy_bs, x_bs, dw_bs, cs_batch, *geom, *am, _hip.stream())
    N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw = geom[:9]
    nb = lib.rn_conv_wgrad_det_workspace_bytes(ldy, nbatch, dw_bs, N, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw)
    if nb < 0:
        return 1
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)      # caching allocator: same stream, reused by the next layer
    return lib.rn_conv_wgrad_batched_det(dy_ptr, ldy, x_ptr, dw_ptr, cs_ptr, nbatch, dy_bs, x_bs, dw_bs, cs_batch, *geom, *am,
                                         ws.data_ptr(), nb, _hip.stream())


def wino_weights(weight, mode=0, scale=None):
    """OIHW 3x3 parameter -> U [36, rows, Kpad] (mode 0: forward; mode 1: data gradient, batch-norm scale folded in)."""
    lib = _hip.load()
    w = _hip.f32c(weight.detach())
    _hip.need_gpu(w, scale)
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    rows, k = (cout, cin) if mode == 0 else (cin, cout)
    U = torch.empty((36, rows, (k + 31) // 32 * 32), dtype=torch.float32, device=w.device)
    _hip.check(lib.rn_wino_weights(w.data_ptr(), U.data_ptr(), cout, cin, mode, _hip.ptr(scale), _hip.stream()),
               "rn_wino_weights")
    return _maybe_split(U, rows=rows, cin=U.shape[2], taps=1)


_WINO_WS = {}
_WINO_ODD_PLANES = os.environ.get("RN_WINO_ODD_PLANES", "1") != "0"


def wino_tpad(T):
    """Rows of one of the 36 planes of V / M / Z for T tiles: a multiple of 256 (no GEMM tile straddles two planes) -- and an ODD one.
    The 36 values a thread of the output transform combines lie one plane apart; with an even count the planes of the head towers are
    43 x 512 KiB apart, the same low 19 address bits 36 times, and the transform reads at 4.6 instead of 5.0 TB/s
    (tools/dbg/wino_stride.py, profiles/r05_wino_plane_stride.txt).  RN_WINO_ODD_PLANES=0: the plain round-up (A/B)."""
    n = (T + 255) // 256
    if _WINO_ODD_PLANES and n % 2 == 0:
        n += 1
    return n * 256


def _wino_group(xs, srcs=None, dsts=None, adds=None, masks=None, mask_bits=False, signs=None, amaxs=None):
    """rn_wino_group for problems shaped like xs ([N,H,W,.]); at most RN_MAX_GROUP of them.  mask_bits: the masks' sign-bit words
    (tensor._rn_sign) instead of the fp32 tensors; signs: per problem the words that receive the result's sign bits, or None;
    amaxs: per problem the amax words (of the source for the input transforms, of the result for the output transform), or None."""
    g = _hip.WinoGroup()
    g.n = len(xs)
    for i, x in enumerate(xs):
        g.N[i], g.H[i], g.W[i] = x.shape[0], x.shape[1], x.shape[2]
        g.src[i] = None if srcs is None else srcs[i].data_ptr()
        g.dst[i] = None if dsts is None else dsts[i].data_ptr()
        g.add[i] = None if adds is None or adds[i] is None else adds[i].data_ptr()
        if masks is None or masks[i] is None:
            g.mask[i] = None
        else:
            g.mask[i] = masks[i]._rn_sign.data_ptr() if mask_bits else masks[i].data_ptr()
        g.sign[i] = None if signs is None or signs[i] is None else signs[i].data_ptr()
        g.amax[i] = None if amaxs is None or amaxs[i] is None else amaxs[i].data_ptr()
    return g


def _wino_transform_in(xs, V, C, Tpad, dy_form, want_rows=False, want_tensor=False):
    """Input-side transform of every problem of xs into V, RN_MAX_GROUP problems per launch.  split3 (want_rows / want_tensor): also the
    transformed tensor's amax words, derived from the sources' per-image words -- returns (row words [Tpad] or None, tensor word or None)."""
    lib = _hip.load()
    off = 0
    am = want_amax() and (want_rows or want_tensor)
    rows = torch.zeros(Tpad, dtype=torch.int32, device=V.device) if am and want_rows else None
    tword = amax_single(V.device) if am and want_tensor else None
    for k in range(0, len(xs), _hip.RN_MAX_GROUP):
        part = xs[k:k + _hip.RN_MAX_GROUP]
        t = sum(x.shape[0] * ((x.shape[1] + 3) // 4) * ((x.shape[2] + 3) // 4) for x in part)
        g = _wino_group(part, srcs=part, amaxs=[amax_words(x) for x in part] if am else None)
        nb = 4.0 * (sum(x.numel() for x in part) + 36 * t * C)
        _hip.check(prof.timed("wino_input", nb, lambda: lib.rn_wino_input_group(
            ctypes.byref(g), V.data_ptr(), C, off, Tpad, dy_form, _hip.ptr(rows), _hip.ptr(tword), _hip.stream())), "rn_wino_input_group")
        off += t
    return rows, tword


def _wino_workspace(device, floats_v, floats_m):
    """V and M of the Winograd path: two scratch tensors per device AND stream (layers launched on different streams run
    concurrently), grown on demand and reused by every layer of that stream."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WINO_WS.get(key)
    if ws is None or ws[0].numel() < floats_v or ws[1].numel() < floats_m:
        ws = (torch.empty(max(floats_v, ws[0].numel() if ws else 0), dtype=torch.float32, device=device),
              torch.empty(max(floats_m, ws[1].numel() if ws else 0), dtype=torch.float32, device=device))
        _WINO_WS[key] = ws
    return ws


def wino_conv_group(xs, U, outs=None, scale=None, shift=None, act=ACT_NONE, adds=None, masks=None, mask_mode=2,
                    flops=0.0, keep_v=False, y_batch_stride=0, V_in=None, V_ready=None, sign=False):
    """3x3 / stride 1 / padding 1 convolution of several inputs [N,H,W,C] with the same (transformed) weights U
    [36, Cout, Kpad]: one grouped input transform into V, ONE batched GEMM launch, one grouped output transform with
    the epilogue (outs: dense tensors, or slices with y_batch_stride).  Returns the outputs (and, with keep_v, the
    (V, shapes) pair wino_wgrad_group accepts).  flops is informational (the GEMM is priced by its executed FLOPs)."""
    lib = _hip.load()
    dev = xs[0].device
    C = xs[0].shape[3]
    cout = U.shape[1]                                # rows of the transformed weights = output channels
    tiles = [x.shape[0] * ((x.shape[1] + 3) // 4) * ((x.shape[2] + 3) // 4) for x in xs]
    T = sum(tiles)
    Tpad = wino_tpad(T)
    V, M = _wino_workspace(dev, 0 if keep_v else 36 * Tpad * C, 36 * Tpad * cout)
    shapes = tuple(tuple(x.shape) for x in xs)
    reuse = keep_v and V_in is not None and V_in[1] == shapes and V_in[0].numel() == 36 * Tpad * C   # (V, shapes) of the same inputs
    # V_ready: (V, shapes) of exactly these inputs computed a moment ago by wino_wgrad_group(fuse_dgrad_input=True) -- the
    # output gradient's two transforms in one pass -- so this call (the data gradient) starts at the GEMM
    ready = not keep_v and isinstance(V_ready, tuple) and V_ready[1] == shapes and V_ready[0].numel() >= 36 * Tpad * C
    if ready:
        V, reuse = V_ready[0], True
    elif reuse:
        V = V_in[0]
    elif keep_v:                                     # the caller keeps B^T d B of the inputs for the weight gradient
        V = torch.empty(36 * Tpad * C, dtype=torch.float32, device=dev)
    assert all(x.is_contiguous() for x in xs)
    v_am = (None, None)                              # split3: V's amax words (one per tile row for this GEMM, one for the weight gradient)
    if reuse:                                        # a kept / handed-over V comes with them
        src = V_ready if ready else V_in
        v_am = src[2] if len(src) > 2 and src[2] is not None else (None, None)
    else:
        v_am = _wino_transform_in(xs, V, C, Tpad, 0, want_rows=True, want_tensor=keep_v)
    # rows past T hold whatever the scratch tensor held: they produce rows of M nobody reads
    Vv = V[:36 * Tpad * C].view(36, 1, Tpad, C)
    Mv = M[:36 * Tpad * cout].view(36, 1, Tpad, cout)
    Uv = _carry_split(U, U.view(36 * cout, U.shape[2]))
    if want_amax() and v_am[0] is None:              # a V somebody else made without words: the three-term kernels for this launch
        Uv = U.view(36 * cout, U.shape[2])
    conv_igemm(Vv, Uv, Mv, (1, Tpad, cout, 1, 1, 1, 1, 0, 0), flops=2.0 * 36 * T * cout * C, w_batch_stride=cout * U.shape[2],
               x_amax=v_am[0], x_amax_rows=True)    # executed FLOPs: same kernel, same family
    if outs is None:
        outs = [torch.empty((x.shape[0], x.shape[1], x.shape[2], cout), dtype=torch.float32, device=dev) for x in xs]
    off = 0
    for k in range(0, len(xs), _hip.RN_MAX_GROUP):
        sl = slice(k, k + _hip.RN_MAX_GROUP)
        part, t = xs[sl], sum(tiles[sl])
        pa = None if adds is None else adds[sl]
        pm = None if masks is None else masks[sl]
        has_mask = pm is not None and pm[0] is not None
        # sign bits: read instead of the fp32 masks when every mask of the launch carries them; written for dense results on request
        mbits = has_mask and BITMASKS and all(getattr(m, "_rn_sign", None) is not None for m in pm)
        signs = [_sign_words(o, sign and not y_batch_stride) for o in outs[sl]]
        yams = [amax_slot(dev, o.shape[0]) for o in outs[sl]] if want_amax() and not y_batch_stride else None
        g = _wino_group(part, dsts=outs[sl], adds=pa, masks=pm, mask_bits=mbits, signs=signs, amaxs=yams)
        nops = 1 + (pa is not None and pa[0] is not None) + (1.0 / 32 if mbits else 1.0) * has_mask + (1.0 / 32 if signs[0] is not None else 0.0)
        nb = 4.0 * (36 * t * cout + nops * sum(x.shape[0] * x.shape[1] * x.shape[2] for x in part) * cout)
        mm = (mask_mode | (MASK_BITS if mbits else 0)) if has_mask else 0
        _hip.check(prof.timed("wino_output", nb, lambda: lib.rn_wino_output_group(
            ctypes.byref(g), M.data_ptr(), cout, off, Tpad, _hip.ptr(scale), _hip.ptr(shift), mm,
            act, y_batch_stride, _hip.stream())), "rn_wino_output_group")
        if yams is not None:
            for o, w_ in zip(outs[sl], yams):
                amax_attach(o, w_)
        off += t
    return (outs, (V, tuple(tuple(x.shape) for x in xs), v_am)) if keep_v else outs   # V + the shapes it belongs to (+ its amax word)


_WINO_Z = {}
SIDE_HELD = {}                                        # side stream -> operands of its launches (wino_wgrad_group(side=...)) until side_release()


def side_release(side=None):
    """After the allocating stream has waited for `side` (None: for every side stream): the operands held for its launches may be
    freed / reused.  Keyed by stream: two engines in one process release only their own."""
    if side is None:
        SIDE_HELD.clear()
    else:
        SIDE_HELD.pop(side.cuda_stream, None)


def _wino_z_buffer(device, floats):
    """A dedicated buffer for A dy A^T when the weight gradient that reads it runs on a SIDE stream (wino_wgrad_group(side=...)): the shared
    workspace's M half is overwritten by the very next GEMM of the calling stream.  Two per (device, calling stream), taken in turn; the
    caller's stream first waits for the side-stream launch that last read the one it gets.  -> (state, index)."""
    cur = torch.cuda.current_stream(device)
    st = _WINO_Z.setdefault((device, cur.cuda_stream), {"bufs": [None, None], "ev": [None, None], "i": 0})
    k = st["i"]
    st["i"] ^= 1
    if st["bufs"][k] is None or st["bufs"][k].numel() < floats:
        st["bufs"][k] = torch.empty(floats, dtype=torch.float32, device=device)
    if st["ev"][k] is not None:
        cur.wait_event(st["ev"][k])
    return st, k


def wino_wgrad_group(gs, xs, dw, colsum, flops=0.0, V=None, dU=None, fuse_dgrad_input=False, side=None):
    """Weight gradient of a 3x3 / stride 1 / padding 1 convolution over several problems (gs[i] = dY [N,H,W,Cout],
    xs[i] = its input [N,H,W,Cin]) by Winograd F(4x4,3x3): dw (packed [Cout][Kpad], accumulated into) and colsum.
    Returns None, or with fuse_dgrad_input the (B^T dy B, shapes) pair for wino_conv_group(V_ready=...) of the same layer's data gradient.
    side: a stream for the 36 reductions and the back-transform of dU (the fused form only: the transforms of dy stay on the caller's
    stream, which goes on to the data gradient while the reductions run beside it)."""
    lib = _hip.load()
    dev = xs[0].device
    C, cout = xs[0].shape[3], gs[0].shape[3]
    zst = None
    tiles = [x.shape[0] * ((x.shape[1] + 3) // 4) * ((x.shape[2] + 3) // 4) for x in xs]
    T = sum(tiles)
    Tpad = wino_tpad(T)
    # V: (B^T d B of the forward's inputs, their shapes) kept by wino_conv_group(keep_v=True); used only for the same grouping
    have_v = V is not None and V[1] == tuple(tuple(x.shape) for x in xs) and V[0].numel() == 36 * Tpad * C
    v_tw = V[2][1] if have_v and len(V) > 2 and V[2] is not None else None      # split3: the kept transform's tensor word
    V = V[0] if have_v else None
    Vw, Z = _wino_workspace(dev, 0 if have_v else 36 * Tpad * C, 36 * Tpad * cout)
    if not have_v:
        V = Vw
    assert all(g.shape[:3] == x.shape[:3] and g.is_contiguous() and x.is_contiguous() for g, x in zip(gs, xs))
    if not have_v:
        v_tw = _wino_transform_in(xs, V, C, Tpad, 0, want_tensor=True)[1]
    v_dy, z_tw = None, None
    if fuse_dgrad_input and have_v and len(gs) <= _hip.RN_MAX_GROUP:
        # The data gradient of the same layer follows and needs B^T dy B of the same gs: both transforms in ONE pass over dy
        # (rn_wino_input_both_group).  B^T dy B goes to the V half of the workspace (idle here: the forward's V was kept), and
        # the (tensor, shapes) pair goes back to the caller for wino_conv_group(V_ready=...).
        Vd, Z = _wino_workspace(dev, 36 * Tpad * cout, 36 * Tpad * cout)
        if side is not None:
            zst, zk = _wino_z_buffer(dev, 36 * Tpad * cout)
            Z = zst["bufs"][zk]
        am_on = want_amax()
        g = _wino_group(gs, srcs=gs, amaxs=[amax_words(t) for t in gs] if am_on else None)
        vd_rows = torch.zeros(Tpad, dtype=torch.int32, device=dev) if am_on else None
        z_tw = amax_single(dev) if am_on else None
        nb = 4.0 * (sum(t.numel() for t in gs) + 2 * 36 * T * cout)
        _hip.check(prof.timed("wino_input", nb, lambda: lib.rn_wino_input_both_group(
            ctypes.byref(g), Vd.data_ptr(), Z.data_ptr(), cout, 0, Tpad, _hip.ptr(vd_rows), _hip.ptr(z_tw), _hip.stream())),
            "rn_wino_input_both_group")
        v_dy = (Vd, tuple(tuple(t.shape) for t in gs), (vd_rows, None))
    else:
        z_tw = _wino_transform_in(gs, Z, cout, Tpad, 1, want_tensor=True)[1]
    # split3: both operands of the 36 reductions are Winograd-domain tensors, one word each (the reduction runs over all images)
    am = (z_tw, v_tw) if (want_amax() and z_tw is not None and v_tw is not None) else (None, None)
    ku = (C + 31) // 32 * 32
    if dU is None or tuple(dU.shape) != (36, cout, ku):      # dU: a ZEROED [36, cout, ku] accumulator of the caller (used once)
        dU = torch.zeros((36, cout, ku), dtype=torch.float32, device=dev)
    def reductions():
        rc = prof.timed("conv_wgrad" + (" winograd T%d %d->%d" % (T, C, cout) if prof.BY_SHAPE else ""), 2.0 * 36 * T * cout * C, lambda: _wgrad_call(    # executed FLOPs
            lib, dev, Z.data_ptr(), cout, V.data_ptr(), dU.data_ptr(), _hip.ptr(colsum), 36, Tpad * cout, Tpad * C, cout * ku, 7,
            (1, 1, T, C, 1, T, cout, 1, 1, 1, 0, 0), amax=am))
        _hip.check(rc, "rn_conv_wgrad_batched")
        _hip.check(lib.rn_wino_dw(dU.data_ptr(), dw.data_ptr(), cout, C, _hip.stream()), "rn_wino_dw")
    if zst is None:
        reductions()
    else:
        side.wait_event(torch.cuda.current_stream(dev).record_event())
        with torch.cuda.stream(side):
            reductions()
            zst["ev"][zk] = side.record_event()
        # What the side launches read must not be handed out again before they are through.  NOT record_stream: with some sixty side
        # launches per step it makes the caching allocator's reuse depend on how far the host runs ahead of the GPU -- blocks whose
        # events have not completed are replaced by fresh device allocations -- and the step time bimodal (64 or 90-140 ms from run to
        # run, profiles/r05_wgrad_stream.txt 7).  The tensors are held here instead and released by the caller (side_release) once the
        # stream they were allocated on has waited for the side stream.
        SIDE_HELD.setdefault(side.cuda_stream, []).append((V, dU) + tuple(am))
    return v_dy                                      # (B^T dy B, shapes) for the data gradient that follows, or None when not fused


def conv_igemm_grouped(problems, w_packed, scale=None, shift=None, act=ACT_NONE, flops=0.0):
    """One launch for up to 5 problems sharing weights / epilogue scalars (the pyramid levels of a head tower).
    problems: list of dicts with x, y, geom and optional add, mask, mask_mode, y_batch_stride, sign (write y's sign bits)."""
    lib = _hip.load()
    g = _hip.ConvGroup()
    g.n = len(problems)
    total = 0
    wptr, wfmt, wus = None, None, None
    yams = []
    for i, pr in enumerate(problems):
        x, geom = pr["x"], pr["geom"]
        add, mask = pr.get("add"), pr.get("mask")
        mask_ptr, mmode = _mask_operand(mask, pr.get("mask_mode", 2))
        d = _make_desc(x, geom, act, 1 if add is not None else 0, (0, 0), mmode, False, None, pr.get("y_batch_stride"), None, None)
        if wfmt is None:                       # the problems share Cin / Cout / taps: one kernel family, one weight form for all
            wptr, wfmt, wus = _w_operand(w_packed, d)
        d.w_format, d.w_unscale = wfmt, wus
        if wfmt == 3:
            d.x_amax, d.x_amax_img_stride, d.x_amax_row_stride = amax_words(x).data_ptr(), 1, 0
        bits = _sign_words(pr["y"], pr.get("sign", False) and pr.get("y_batch_stride") is None)
        d.sign_out = None if bits is None else bits.data_ptr()
        yam = amax_slot(x.device, x.shape[0]) if want_amax() and pr.get("y_batch_stride") is None else None
        yams.append(yam)
        d.y_amax = None if yam is None else yam.data_ptr()
        g.d[i] = d
        M = d.N * d.Ho * d.Wo
        total += (M + 255) // 256 if d.Cout <= 64 else ((M + 127) // 128) * ((d.Cout + 127) // 128)
        g.tile_end[i] = total
        g.x[i], g.y[i], g.add[i], g.mask[i] = x.data_ptr(), pr["y"].data_ptr(), _hip.ptr(add), mask_ptr
    kind = "conv_igemm_4x1" if problems[0]["geom"][2] <= 64 else "conv_igemm_2x2"
    if prof.BY_SHAPE:
        kind += " grouped %d->%d k%d" % (g.d[0].Cin, g.d[0].Cout, g.d[0].kh)
    nb = 0.0
    if prof.ACTIVE is not None:
        nb = {0: 4.0, 1: 6.0, 3: 4.0}[wfmt] * w_packed.numel()
        for pr in problems:
            dd = pr["geom"]
            out_el = pr["x"].shape[0] * dd[0] * dd[1] * dd[2]
            nb += 4.0 * (pr["x"].numel() + out_el * (1 + (pr.get("add") is not None) + (pr.get("mask") is not None)))
    rc = prof.timed(kind, flops, lambda: lib.rn_conv_igemm_grouped(
        ctypes.byref(g), wptr, _hip.ptr(scale), _hip.ptr(shift), _hip.stream()), nb)
    _hip.check(rc, "rn_conv_igemm_grouped")
    for pr, yam in zip(problems, yams):
        if yam is not None:
            amax_attach(pr["y"], yam)


def fprop(x, w_packed, cout, k, stride, pad, kw_pad=None, **kw):
    """Forward convolution, NHWC in -> new NHWC out."""
    N, Hi, Wi, _ = x.shape
    Ho, Wo = out_size(Hi, k, stride, pad), out_size(Wi, k, stride, pad)
    y = torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    return conv_igemm(x, w_packed, y, (Ho, Wo, cout, k, k if kw_pad is None else kw_pad, stride, 1, -pad, 0), **kw)


def dgrad(dy, w_packed_dgrad, in_hw, cin, k, stride, pad, **kw):
    """Data gradient, generic form: dy [N,Ho,Wo,Cout(_pad)] -> dx [N,Hi,Wi,Cin].  For stride 2 every tap is tried
    at every input pixel and 3 of 4 fail the divisibility test (wasted MFMAs): prefer dgrad_s2_classes."""
    N = dy.shape[0]
    Hi, Wi = in_hw
    dx = torch.empty((N, Hi, Wi, cin), dtype=torch.float32, device=dy.device)
    return conv_igemm(dy, w_packed_dgrad, dx, (Hi, Wi, cin, k, k, 1, -1, pad, stride.bit_length() - 1), **kw)


def s2_classes(k, pad):
    """Parity classes of a stride-2 data gradient: [(ph, pw, (r0, nr, s0, ns), (dh0, dw0))] with taps r = r0+2i
    contributing input row a + dh0 - i to output row 2a + ph."""
    out = []
    for ph in (0, 1):
        r0 = (ph + pad) % 2
        nr = len(range(r0, k, 2))
        for pw in (0, 1):
            s0 = (pw + pad) % 2
            ns = len(range(s0, k, 2))
            if nr and ns:
                out.append((ph, pw, (r0, nr, s0, ns), ((ph + pad - r0) // 2, (pw + pad - s0) // 2)))
    return out


def dgrad_s2_classes(dy, class_weights, in_hw, cin, k, pad, flops=0.0, **kw):
    """Stride-2 data gradient without wasted taps: one launch per output-parity class, each a stride-1 convolution
    of dy with that class's tap subset (class_weights[i] from pack_weights(..., taps=...)), stored at the class's
    strided positions.  add / mask (same geometry as dx) apply per class."""
    N, Ho, Wo, _ = dy.shape
    Hi, Wi = in_hw
    dx = torch.empty((N, Hi, Wi, cin), dtype=torch.float32, device=dy.device)
    classes = s2_classes(k, pad)
    if len(classes) < 4:
        dx.zero_()                                  # classes without taps (k = 1) receive no gradient
    total_taps = sum(c[2][1] * c[2][3] for c in classes)
    slot = amax_slot(dy.device, N) if want_amax() else None   # split3: the classes fill ONE tensor: one set of amax words for all of them
    for (ph, pw, (r0, nr, s0, ns), (dh0, dw0)), wc in zip(classes, class_weights):
        gh, gw = (Hi - ph + 1) // 2, (Wi - pw + 1) // 2
        if gh <= 0 or gw <= 0:
            continue
        # input row = a + dh0 - i, input column = b + dw0 - j  (a = 1, b = -1, p = dh0, p_w = dw0)
        conv_igemm(dy, wc, dx, (gh, gw, cin, nr, ns, 1, -1, (dh0, dw0), 0), out_map=(2, ph, pw, Hi, Wi),
                   flops=flops * nr * ns / total_taps, amax=slot, **kw)
    if slot is not None:
        amax_attach(dx, slot)
    return dx


def wgrad(dy, x, dw, cout, k, stride, pad, kw_pad=None, in_relu=False, flops=0.0, colsum=None):
    """dw[Cout][Kpad] += wgrad(dy, x) and, when given, colsum[Cout] += sum over pixels of dy.
    dy [N,Ho,Wo,ldy] (ldy >= cout), x [N,Hi,Wi,Cin]."""
    lib = _hip.load()
    N, Ho, Wo, ldy = dy.shape
    _, Hi, Wi, Cin = x.shape
    kind = "conv_wgrad" + (" %dx%dx%d %d->%d k%d s%d" % (N, Ho, Wo, Cin, cout, k, stride) if prof.BY_SHAPE else "")
    # split3: the operands' power-of-two scales
    am = (amax_words(dy), amax_words(x)) if want_amax() and dy.is_contiguous() and x.is_contiguous() else (None, None)
    rc = prof.timed(kind, flops, lambda: _wgrad_call(
        lib, dy.device, dy.data_ptr(), ldy, x.data_ptr(), dw.data_ptr(), _hip.ptr(colsum), 1, 0, 0, 0, 0,
        (N, Hi, Wi, Cin, Ho, Wo, cout, k, k if kw_pad is None else kw_pad, stride, pad, int(in_relu)), amax=am))
    _hip.check(rc, "rn_conv_wgrad")
    return dw


def unpack_wgrad(dw, w_packed, weight_shape, kw_pad=None, c_pad=None, scale=None, mean=None, rstd=None, colsum=None,
                 want_bn=False, out=None):
    """-> (dweight OIHW, dgamma or None, dbeta or None); out = the same triple as destination tensors (contiguous fp32
    of the right sizes, e.g. views of a flat gradient buffer) instead of fresh ones."""
    lib = _hip.load()
    cout, cin, kh, kw = weight_shape
    kw_pad = kw if kw_pad is None else kw_pad
    c_pad = (cin + 3) // 4 * 4 if c_pad is None else c_pad
    if out is not None:
        dweight, dgamma, dbeta = out
        assert tuple(dweight.shape) == tuple(weight_shape) and dweight.is_contiguous() and dweight.dtype == torch.float32
        assert (dgamma is not None) == bool(want_bn) and (dbeta is not None) == (colsum is not None)
    else:
        dweight = torch.empty(weight_shape, dtype=torch.float32, device=dw.device)
        dgamma = torch.empty(cout, dtype=torch.float32, device=dw.device) if want_bn else None
        dbeta = torch.empty(cout, dtype=torch.float32, device=dw.device) if colsum is not None else None
    _hip.check(lib.rn_unpack_wgrad(dw.data_ptr(), _hip.ptr(w_packed), dweight.data_ptr(), cout, cin, kh, kw, kw_pad,
                                   c_pad, _hip.ptr(scale), _hip.ptr(mean), _hip.ptr(rstd), _hip.ptr(colsum),
                                   _hip.ptr(dgamma), _hip.ptr(dbeta), _hip.stream()), "rn_unpack_wgrad")
    return dweight, dgamma, dbeta


def bn_fold(gamma, beta, mean, var, eps=1e-5):
    lib = _hip.load()
    C = gamma.numel()
    out = torch.empty((3, C), dtype=torch.float32, device=gamma.device)
    _hip.check(lib.rn_bn_fold(_hip.f32c(gamma.detach()).data_ptr(), _hip.f32c(beta.detach()).data_ptr(),
                              _hip.f32c(mean).data_ptr(), _hip.f32c(var).data_ptr(), eps, C, out[0].data_ptr(),
                              out[1].data_ptr(), out[2].data_ptr(), _hip.stream()), "rn_bn_fold")
    return out[0], out[1], out[2]


def nchw_to_nhwc4(img):
    lib = _hip.load()
    img = _hip.f32c(img)
    N, C, H, W = img.shape
    if C != 3:
        raise RuntimeError("the stem takes 3-channel images (D/model.py:213), got %d" % C)
    out = torch.empty((N, H, W, 4), dtype=torch.float32, device=img.device)
    am = amax_slot(img.device, N) if want_amax() else None
    _hip.check(lib.rn_nchw_to_nhwc4(img.data_ptr(), out.data_ptr(), N, H, W, _hip.ptr(am), _hip.stream()), "rn_nchw_to_nhwc4")
    if am is not None:
        amax_attach(out, am)
    return out


def maxpool_fwd(x, want_argmax=False):
    """-> y, or (y, argmax uint8 [N,Ho,Wo,C]) when the backward pass will need it."""
    lib = _hip.load()
    N, H, W, C = x.shape
    Ho, Wo = out_size(H, 3, 2, 1), out_size(W, 3, 2, 1)
    y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
    arg = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device) if want_argmax else None
    _hip.check(lib.rn_maxpool_fwd(x.data_ptr(), y.data_ptr(), _hip.ptr(arg), N, H, W, C, Ho, Wo, _hip.stream()),
               "rn_maxpool_fwd")
    a = getattr(x, "_rn_amax", None)
    if a is not None and a[1] == x._version:         # every input pixel lies in a window: for the ReLU outputs it pools, the pooled
        amax_attach(y, a[0])                         # tensor's largest magnitude per image is the input's (in general: at most it)
    return (y, arg) if want_argmax else y


def maxpool_bwd(x, dy, argmax, relu_mask=True):
    lib = _hip.load()
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    bits = getattr(x, "_rn_sign", None) if (relu_mask and BITMASKS) else None      # the stem's sign bits instead of re-reading it
    am = amax_slot(x.device, N) if want_amax() else None
    _hip.check(lib.rn_maxpool_bwd(x.data_ptr() if bits is None else bits.data_ptr(), dy.data_ptr(), argmax.data_ptr(), dx.data_ptr(),
                                  N, H, W, C, dy.shape[1], dy.shape[2], (2 if bits is not None else int(relu_mask)), _hip.ptr(am),
                                  _hip.stream()), "rn_maxpool_bwd")
    if am is not None:
        amax_attach(dx, am)
    return dx


def colsum(g, C=None, out=None):
    """Column sums of a [..., ld] tensor over all leading dims, first C columns; ``out`` given = accumulate."""
    lib = _hip.load()
    ld = g.shape[-1]
    C = ld if C is None else C
    rows = g.numel() // ld
    ws = torch.empty(lib.rn_colsum_workspace_bytes(rows, C), dtype=torch.uint8, device=g.device)
    acc = out is not None
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=g.device)
    _hip.check(lib.rn_colsum(g.data_ptr(), rows, C, ld, out.data_ptr(), int(acc), ws.data_ptr(), _hip.stream()),
               "rn_colsum")
    return out


def upsample_add_bwd(src, dst):
    """dst[n,h,w,:] += sum of the (in-bounds) 2x2 children in src."""
    lib = _hip.load()
    N, Hs, Ws, C = src.shape
    assert src.dtype == dst.dtype
    amax_drop(dst)
    if src.dtype == torch.bfloat16:
        _hip.check(lib.rn_upsample_add_bwd_bf16(src.data_ptr(), dst.data_ptr(), N, Hs, Ws, dst.shape[1], dst.shape[2], C, _hip.stream()),
                   "rn_upsample_add_bwd_bf16")
        return dst
    am = amax_slot(dst.device, N) if want_amax() else None      # the sums' own amax words (the old ones no longer bound them)
    _hip.check(lib.rn_upsample_add_bwd(src.data_ptr(), dst.data_ptr(), N, Hs, Ws, dst.shape[1], dst.shape[2], C, _hip.ptr(am), _hip.stream()),
               "rn_upsample_add_bwd")
    if am is not None:
        amax_attach(dst, am)
    return dst


def relu_mask_(g, z):
    _hip.check(_hip.load().rn_relu_mask(g.data_ptr(), z.data_ptr(), g.numel(), _hip.stream()), "rn_relu_mask")
    return g                                       # (a mask only lowers magnitudes: an amax word stays a valid bound)


def sigmoid_bwd_pad(dy_ptr, s_ptr, B, rows_per_image, C, ld, src_batch_stride, device, bf16=False):
    """Gradient slice of a head output (B images, rows_per_image pixels of C channels each, images
    src_batch_stride floats apart; raw device pointers) -> dense [B*rows_per_image, ld], zero-padded channels,
    multiplied by s(1-s) when the sigmoid output pointer is given."""
    lib = _hip.load()
    out = torch.empty((B * rows_per_image, ld), dtype=torch.bfloat16 if bf16 else torch.float32, device=device)
    if bf16:
        _hip.check(lib.rn_sigmoid_bwd_pad_bf16(dy_ptr, s_ptr, out.data_ptr(), B, rows_per_image, C, ld, src_batch_stride, _hip.stream()),
                   "rn_sigmoid_bwd_pad_bf16")
        return out
    am = amax_slot(device, B) if want_amax() else None
    _hip.check(lib.rn_sigmoid_bwd_pad(dy_ptr, s_ptr, out.data_ptr(), B, rows_per_image, C, ld, src_batch_stride, _hip.ptr(am), _hip.stream()),
               "rn_sigmoid_bwd_pad")
    if am is not None:
        out._rn_amax_words = am                      # the caller reshapes to [B, H, W, ld]: amax_carry(view, out) hands the words on
    return out


def add_(dst, src):
    amax_drop(dst)
    am = amax_slot(dst.device, dst.shape[0]) if want_amax() and dst.dim() == 4 else None
    _hip.check(_hip.load().rn_add_inplace(dst.data_ptr(), src.data_ptr(), dst.numel(), dst.numel() // max(dst.shape[0], 1), _hip.ptr(am),
                                          _hip.stream()), "rn_add_inplace")
    if am is not None:
        amax_attach(dst, am)
    return dst


# ---------------------------------------------------------------------------------------------- bf16 engine (BASELINE configs[2])
def to_bf16(t):
    """fp32 device tensor -> bf16 (round to nearest even, rn_f32_to_bf16)."""
    t = _hip.f32c(t)
    _hip.need_gpu(t)
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    if t.numel():
        _hip.check(_hip.load().rn_f32_to_bf16(t.data_ptr(), out.data_ptr(), t.numel(), _hip.stream()), "rn_f32_to_bf16")
    return out


def to_f32(t):
    _hip.need_gpu(t)
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    if t.numel():
        _hip.check(_hip.load().rn_bf16_to_f32(t.data_ptr(), out.data_ptr(), t.numel(), _hip.stream()), "rn_bf16_to_f32")
    return out


def pack_weights_bf16(weight, mode=0, scale=None, c_pad=None, taps=None):
    """fp32 OIHW master weights -> packed bf16 rows (pack_weights' layout; rows are multiples of 32 elements)."""
    cin, cout = weight.shape[1], weight.shape[0]
    if c_pad is None:
        c_pad = ((cin if mode == 0 else cout) + 7) // 8 * 8                  # 16-byte chunks hold 8 channels
    return to_bf16(pack_weights(weight, mode, scale=scale, c_pad=c_pad, taps=taps, presplit=False))


def _p8_suffix(fn_name, d, yf32):
    """"_p8" when a single launch takes the eight-wave 256 x 256 kernel for this problem (profiling only: the kernel families are timed apart)."""
    t = getattr(_hip.load(), fn_name)(ctypes.byref(d), int(yf32))
    return "_p8" if t == 256256 and fn_name.endswith("fp8_tile") or t >= 1000000 else ""


def conv_igemm_bf16(x, w_packed, y, geom, scale=None, shift=None, add=None, add_mode=0, add_hw=(0, 0), mask=None,
                    mask_mode=2, act=ACT_NONE, y_batch_stride=None, add_batch_stride=None, flops=0.0, out_map=None, sign=False):
    """rn_conv_igemm_bf16: x [N,Hi,Wi,Cin] bf16, w_packed bf16, y bf16 or fp32 (its dtype decides); geom as conv_igemm.
    sign / a mask that carries ._rn_sign: the sign bits of conv_igemm, here of the bf16 tensors."""
    lib = _hip.load()
    assert x.dtype == torch.bfloat16 and w_packed.dtype == torch.bfloat16 and y.dtype in (torch.bfloat16, torch.float32)
    mask_ptr, mask_mode = _mask_operand(mask, mask_mode)
    d = _make_desc(x, geom, act, add_mode, add_hw, mask_mode, False, out_map, y_batch_stride, add_batch_stride, None)
    bits = _sign_words(y, sign and y_batch_stride is None and out_map is None and y.dtype == torch.bfloat16)
    d.sign_out = None if bits is None else bits.data_ptr()
    kind = "conv_igemm_bf16"
    if prof.ACTIVE is not None:
        kind += _p8_suffix("rn_conv_igemm_bf16_tile", d, y.dtype == torch.float32)
    if prof.BY_SHAPE:                                    # profiling aid (tools/profile_layers.py): one row per layer shape
        kind += " %dx%dx%d %d->%d k%d a%d b%d ds%d" % (d.N, d.Ho, d.Wo, d.Cin, d.Cout, d.kh, d.a, d.b, d.div_shift)
    rc = prof.timed(kind, flops, lambda: lib.rn_conv_igemm_bf16(
        ctypes.byref(d), x.data_ptr(), w_packed.data_ptr(), y.data_ptr(), int(y.dtype == torch.float32), _hip.ptr(scale),
        _hip.ptr(shift), _hip.ptr(add), mask_ptr, _hip.stream()))
    _hip.check(rc, "rn_conv_igemm_bf16")
    return y


def fprop_bf16(x, w_packed, cout, k, stride, pad, out_dtype=torch.bfloat16, **kw):
    N, Hi, Wi, _ = x.shape
    Ho, Wo = out_size(Hi, k, stride, pad), out_size(Wi, k, stride, pad)
    y = torch.empty((N, Ho, Wo, cout), dtype=out_dtype, device=x.device)
    return conv_igemm_bf16(x, w_packed, y, (Ho, Wo, cout, k, k, stride, 1, -pad, 0), **kw)


def dgrad_bf16(dy, w_packed_dgrad, in_hw, cin, k, pad, **kw):
    """Stride-1 data gradient in bf16 (dy [N,Ho,Wo,Cout(_pad)] bf16 -> dx [N,Hi,Wi,Cin] bf16)."""
    N = dy.shape[0]
    Hi, Wi = in_hw
    dx = torch.empty((N, Hi, Wi, cin), dtype=torch.bfloat16, device=dy.device)
    return conv_igemm_bf16(dy, w_packed_dgrad, dx, (Hi, Wi, cin, k, k, 1, -1, pad, 0), **kw)


def wgrad_bf16(dy, x, dw, cout, k, stride, pad, flops=0.0, colsum=None):
    """dw[Cout][Kpad] (fp32) += wgrad(dy, x) with bf16 operands; colsum[Cout] += column sums of dy."""
    lib = _hip.load()
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32
    N, Ho, Wo, ldy = dy.shape
    _, Hi, Wi, Cin = x.shape
    kind = "conv_wgrad_bf16"
    if prof.BY_SHAPE:
        kind += " %dx%dx%d %d->%d k%d s%d" % (N, Ho, Wo, Cin, cout, k, stride)
    rc = prof.timed(kind, flops, lambda: lib.rn_conv_wgrad_bf16(
        dy.data_ptr(), ldy, x.data_ptr(), dw.data_ptr(), _hip.ptr(colsum), N, Hi, Wi, Cin, Ho, Wo, cout, k, k, stride, pad,
        _hip.stream()))
    _hip.check(rc, "rn_conv_wgrad_bf16")
    return dw


def wgrad_bf16_grouped(dys, xs, dw, cout, k, stride, pad, flops=0.0, colsum=None):
    """rn_conv_wgrad_bf16_grouped: dw[Cout][Kpad] (fp32) += the SUM over several problems that share one weight tensor (the pyramid
    levels of a head layer) of wgrad(dy_i, x_i), as ONE launch; colsum[Cout] += the column sums of every dy_i."""
    lib = _hip.load()
    n = len(dys)
    assert n == len(xs) and 1 <= n <= 5 and dw.dtype == torch.float32
    ld, N, cin = dys[0].shape[3], xs[0].shape[0], xs[0].shape[3]
    for g, x in zip(dys, xs):
        assert g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and g.is_contiguous() and x.is_contiguous()
        assert g.shape[3] == ld and x.shape[0] == N and g.shape[0] == N and x.shape[3] == cin
        _hip.need_gpu(g, x)
    PA = ctypes.c_void_p * n
    IA = ctypes.c_int * n
    dy_p, x_p = PA(*[g.data_ptr() for g in dys]), PA(*[x.data_ptr() for x in xs])
    hi, wi = IA(*[x.shape[1] for x in xs]), IA(*[x.shape[2] for x in xs])
    kind = "conv_wgrad_bf16" + (" grouped %d->%d k%d" % (cin, cout, k) if prof.BY_SHAPE else "")
    rc = prof.timed(kind, flops, lambda: lib.rn_conv_wgrad_bf16_grouped(
        n, dy_p, ld, x_p, dw.data_ptr(), _hip.ptr(colsum), N, hi, wi, cin, cout, k, k, stride, pad, _hip.stream()))
    _hip.check(rc, "rn_conv_wgrad_bf16_grouped")


def dgrad_any_bf16(dy, w_packed_dgrad, in_hw, cin, k, stride, pad, **kw):
    """Data gradient, generic form (every tap tried at every input pixel; for stride 2 three of four fail the divisibility
    test): used for the 1x1 stride-2 shortcuts, whose single tap makes the waste irrelevant."""
    N = dy.shape[0]
    Hi, Wi = in_hw
    dx = torch.empty((N, Hi, Wi, cin), dtype=torch.bfloat16, device=dy.device)
    return conv_igemm_bf16(dy, w_packed_dgrad, dx, (Hi, Wi, cin, k, k, 1, -1, pad, stride.bit_length() - 1), **kw)


def dgrad_s2_classes_bf16(dy, class_weights, in_hw, cin, k, pad, flops=0.0, **kw):
    """dgrad_s2_classes with bf16 operands: one launch per output-parity class, stored at the class's strided positions."""
    N, Ho, Wo, _ = dy.shape
    Hi, Wi = in_hw
    dx = torch.empty((N, Hi, Wi, cin), dtype=torch.bfloat16, device=dy.device)
    classes = s2_classes(k, pad)
    if len(classes) < 4:
        dx.zero_()
    total_taps = sum(c[2][1] * c[2][3] for c in classes)
    for (ph, pw, (r0, nr, s0, ns), (dh0, dw0)), wc in zip(classes, class_weights):
        gh, gw = (Hi - ph + 1) // 2, (Wi - pw + 1) // 2
        if gh <= 0 or gw <= 0:
            continue
        conv_igemm_bf16(dy, wc, dx, (gh, gw, cin, nr, ns, 1, -1, (dh0, dw0), 0), out_map=(2, ph, pw, Hi, Wi),
                        flops=flops * nr * ns / total_taps, **kw)
    return dx


def maxpool_fwd_bf16(x, want_argmax=False):
    """fp32 stem output -> bf16 pooled activations (+ uint8 argmax)."""
    lib = _hip.load()
    N, H, W, C = x.shape
    Ho, Wo = out_size(H, 3, 2, 1), out_size(W, 3, 2, 1)
    y = torch.empty((N, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    arg = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device) if want_argmax else None
    _hip.check(lib.rn_maxpool_fwd_bf16out(x.data_ptr(), y.data_ptr(), _hip.ptr(arg), N, H, W, C, Ho, Wo, _hip.stream()),
               "rn_maxpool_fwd_bf16out")
    return (y, arg) if want_argmax else y


def maxpool_bwd_bf16(x, dy, argmax, relu_mask=True):
    """bf16 gradient of the pooled activations -> fp32 gradient of the (fp32) stem output."""
    lib = _hip.load()
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    bits = getattr(x, "_rn_sign", None) if (relu_mask and BITMASKS) else None      # the stem's sign bits instead of re-reading it
    _hip.check(lib.rn_maxpool_bwd_bf16in(x.data_ptr() if bits is None else bits.data_ptr(), dy.data_ptr(), argmax.data_ptr(),
                                         dx.data_ptr(), N, H, W, C, dy.shape[1], dy.shape[2],
                                         (2 if bits is not None else int(relu_mask)), _hip.stream()), "rn_maxpool_bwd_bf16in")
    return dx


def relu_bf16(x):
    out = torch.empty_like(x)
    _hip.check(_hip.load().rn_relu_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _hip.stream()), "rn_relu_bf16")
    return out


def conv_igemm_bf16_grouped(problems, w_packed, scale=None, shift=None, act=ACT_NONE, flops=0.0):
    """conv_igemm_grouped with bf16 operands: one launch for up to 5 problems sharing weights / epilogue scalars (the
    pyramid levels of a head layer).  All results bf16, or all fp32 (the dtype of the first y decides)."""
    lib = _hip.load()
    g = _hip.ConvGroup()
    g.n = len(problems)
    total = 0
    yf32 = problems[0]["y"].dtype == torch.float32
    for i, pr in enumerate(problems):
        x, geom = pr["x"], pr["geom"]
        add, mask = pr.get("add"), pr.get("mask")
        assert x.dtype == torch.bfloat16 and (pr["y"].dtype == torch.float32) == yf32
        mask_ptr, mmode = _mask_operand(mask, pr.get("mask_mode", 2))
        d = _make_desc(x, geom, act, 1 if add is not None else 0, (0, 0), mmode, False, None, pr.get("y_batch_stride"), None, None)
        bits = _sign_words(pr["y"], pr.get("sign", False) and pr.get("y_batch_stride") is None and not yf32)
        d.sign_out = None if bits is None else bits.data_ptr()
        g.d[i] = d
        g.x[i], g.y[i], g.add[i], g.mask[i] = x.data_ptr(), pr["y"].data_ptr(), _hip.ptr(add), mask_ptr
    trm, trn = divmod(lib.rn_conv_igemm_bf16_tile_rows(ctypes.byref(g), int(yf32)), 1000)   # the launcher's tile for this group
    for i in range(g.n):
        d = g.d[i]
        M = d.N * d.Ho * d.Wo
        total += ((M + trm - 1) // trm) * ((d.Cout + trn - 1) // trn)
        g.tile_end[i] = total
    kind = "conv_igemm_bf16" + ("_p8" if (trm, trn) == (256, 256) else "") \
        + (" grouped %d->%d k%d" % (g.d[0].Cin, g.d[0].Cout, g.d[0].kh) if prof.BY_SHAPE else "")
    rc = prof.timed(kind, flops, lambda: lib.rn_conv_igemm_bf16_grouped(
        ctypes.byref(g), w_packed.data_ptr(), int(yf32), _hip.ptr(scale), _hip.ptr(shift), _hip.stream()))
    _hip.check(rc, "rn_conv_igemm_bf16_grouped")


# ---------------------------------------------------------------------------------------------- fp8 forward (BASELINE configs[4])
FP8_MAX = 448.0                                  # largest finite e4m3fn


def fp8_quantize(t, scale):
    """fp32 device tensor -> e4m3 bytes (uint8 tensor of the same shape) with ONE scale: q = fp8(t / scale), saturating.
    The scale rides on the result as ``._rn_scale`` (x ~= q * scale)."""
    t = _hip.f32c(t)
    _hip.need_gpu(t)
    out = torch.empty(t.shape, dtype=torch.uint8, device=t.device)
    _hip.check(_hip.load().rn_fp8_quantize(t.data_ptr(), out.data_ptr(), t.numel(), 1.0 / float(scale), _hip.stream()), "rn_fp8_quantize")
    out._rn_scale = float(scale)
    return out


def maxpool_fwd_fp8(x, scale):
    """fp32 stem output [N,H,W,C] -> e4m3 pooled activations (3x3 / 2 / pad 1 max-pool, then q = fp8(max / scale)): one pass instead of
    maxpool_fwd + fp8_quantize, the same bytes."""
    x = _hip.f32c(x)
    _hip.need_gpu(x)
    N, H, W, C = x.shape
    Ho, Wo = out_size(H, 3, 2, 1), out_size(W, 3, 2, 1)
    out = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    _hip.check(_hip.load().rn_maxpool_fwd_fp8out(x.data_ptr(), out.data_ptr(), N, H, W, C, Ho, Wo, 1.0 / float(scale), _hip.stream()),
               "rn_maxpool_fwd_fp8out")
    out._rn_scale = float(scale)
    return out


def fp8_dequantize(q, scale=None):
    scale = q._rn_scale if scale is None else scale
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    _hip.check(_hip.load().rn_fp8_dequantize(q.data_ptr(), out.data_ptr(), q.numel(), float(scale), _hip.stream()), "rn_fp8_dequantize")
    return out


def bf16_to_fp8(t, scale):
    """bf16 device tensor -> e4m3 bytes with ONE scale (q = fp8(t / scale), saturating): fp8_quantize for a bf16 source, one pass."""
    _hip.need_gpu(t)
    t = t.contiguous()
    n = t.numel()
    if n % 16 or not n:
        return fp8_quantize(to_f32(t), scale)
    out = torch.empty(t.shape, dtype=torch.uint8, device=t.device)
    _hip.check(_hip.load().rn_bf16_to_fp8(t.data_ptr(), out.data_ptr(), n, 1.0 / float(scale), _hip.stream()), "rn_bf16_to_fp8")
    out._rn_scale = float(scale)
    return out


def fp8_to_bf16(q, scale=None):
    """e4m3 tensor (uint8, ._rn_scale) -> bf16 tensor of the same shape: the fp8-forward training step's hand-over to the bf16 gradients."""
    scale = q._rn_scale if scale is None else scale
    out = torch.empty(q.shape, dtype=torch.bfloat16, device=q.device)
    n = q.numel()
    if n % 16 == 0 and n:
        _hip.check(_hip.load().rn_fp8_to_bf16(q.data_ptr(), out.data_ptr(), n, float(scale), _hip.stream()), "rn_fp8_to_bf16")
    elif n:                                          # (odd sizes: tiny tensors of the test networks)
        out.copy_(fp8_dequantize(q, scale))
    return out


def fp8_quantize_weights(w_packed):
    """Packed fp32 weight rows [rows][Kpad] -> (e4m3 rows [rows][round64(Kpad)], per-row scale [rows] = max|row| / 448)."""
    rows, kp = w_packed.shape
    kp64 = (kp + 63) // 64 * 64
    wq = torch.empty((rows, kp64), dtype=torch.uint8, device=w_packed.device)
    sc = torch.empty(rows, dtype=torch.float32, device=w_packed.device)
    _hip.check(_hip.load().rn_fp8_quantize_rows(w_packed.data_ptr(), wq.data_ptr(), sc.data_ptr(), rows, kp, _hip.stream()), "rn_fp8_quantize_rows")
    return wq, sc


def conv_igemm_fp8(xq, wq, y, geom, scale, shift=None, add=None, add_mode=0, add_hw=(0, 0), act=ACT_NONE, y_batch_stride=None,
                   out_scale=1.0, flops=0.0):
    """rn_conv_igemm_fp8: xq [N,Hi,Wi,Cin] e4m3 (uint8), wq from fp8_quantize_weights, y uint8 (e4m3, scale out_scale) or fp32;
    scale [Cout] = x_scale * weight row scale * folded batch-norm scale; add: e4m3 with ``._rn_scale``; geom as conv_igemm."""
    lib = _hip.load()
    assert xq.dtype == torch.uint8 and wq.dtype == torch.uint8 and y.dtype in (torch.uint8, torch.float32)
    d = _make_desc(xq, geom, act, add_mode, add_hw, 0, False, None, y_batch_stride, None, None)
    kind = "conv_igemm_fp8"
    if prof.ACTIVE is not None:
        kind += _p8_suffix("rn_conv_igemm_fp8_tile", d, y.dtype == torch.float32)
    if prof.BY_SHAPE:
        kind += " %dx%dx%d %d->%d k%d" % (d.N, d.Ho, d.Wo, d.Cin, d.Cout, d.kh)
    rc = prof.timed(kind, flops, lambda: lib.rn_conv_igemm_fp8(
        ctypes.byref(d), xq.data_ptr(), wq.data_ptr(), y.data_ptr(), int(y.dtype == torch.float32), _hip.ptr(scale), _hip.ptr(shift),
        _hip.ptr(add), float(add._rn_scale) if add is not None else 1.0, 1.0 / float(out_scale), _hip.stream()))
    _hip.check(rc, "rn_conv_igemm_fp8")
    if y.dtype == torch.uint8:
        y._rn_scale = float(out_scale)
    return y


def conv_igemm_fp8_grouped(problems, wq, scale, shift=None, act=ACT_NONE, out_scale=1.0, flops=0.0):
    """rn_conv_igemm_fp8_grouped: up to RN_MAX_GROUP problems (dicts with x, y, geom, optional y_batch_stride) that share the
    e4m3 weights, the folded scale vector (so: ONE input scale for all of them) and the activation; results all e4m3 or all fp32."""
    lib = _hip.load()
    g = _hip.ConvGroup()
    g.n = len(problems)
    yf32 = problems[0]["y"].dtype == torch.float32
    total = 0
    for i, pr in enumerate(problems):
        x = pr["x"]
        assert x.dtype == torch.uint8 and (pr["y"].dtype == torch.float32) == yf32
        g.d[i] = _make_desc(x, pr["geom"], act, 0, (0, 0), 0, False, None, pr.get("y_batch_stride"), None, None)
        g.x[i], g.y[i], g.add[i], g.mask[i] = x.data_ptr(), pr["y"].data_ptr(), None, None
    trm, trn = divmod(lib.rn_conv_igemm_fp8_tile_rows(ctypes.byref(g), int(yf32)), 1000)    # the launcher's tile for this group
    for i in range(g.n):
        d = g.d[i]
        total += ((d.N * d.Ho * d.Wo + trm - 1) // trm) * ((d.Cout + trn - 1) // trn)
        g.tile_end[i] = total
    kind = "conv_igemm_fp8" + ("_p8" if (trm, trn) == (256, 256) else "") + (" grouped %d->%d k%d" % (g.d[0].Cin, g.d[0].Cout, g.d[0].kh) if prof.BY_SHAPE else "")
    rc = prof.timed(kind, flops, lambda: lib.rn_conv_igemm_fp8_grouped(
        ctypes.byref(g), wq.data_ptr(), int(yf32), _hip.ptr(scale), _hip.ptr(shift), 1.0, 1.0 / float(out_scale), _hip.stream()))
    _hip.check(rc, "rn_conv_igemm_fp8_grouped")
    if not yf32:
        for pr in problems:
            pr["y"]._rn_scale = float(out_scale)
