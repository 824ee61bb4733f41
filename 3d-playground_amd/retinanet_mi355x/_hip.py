"""ctypes binding of libretinanet_mi355x.so (include/retinanet_mi355x.h).

The library is built in-tree by ``csrc/Makefile`` (``__graft_entry__.build()``).  There is no fallback: if the
shared object is missing or a tensor is not on a HIP device every op raises -- the product path never runs on
the CPU and never touches ``oracle/``.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RN_LIB_PATH") or os.path.join(_HERE, "lib", "libretinanet_mi355x.so")   # env: kernel A/B builds
_lib = None

c_i32, c_i64, c_f32, c_f64, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); one entry per symbol declared in include/retinanet_mi355x.h
SIGNATURES = {
    "rn_version": (ctypes.c_char_p, []),
    "rn_check_device": (c_i32, []),
    "rn_get_fp32_mfma": (c_i32, []),
    "rn_set_fp32_mfma": (c_i32, [c_i32]),
    "rn_get_option": (c_i32, [c_i32]),
    "rn_set_option": (c_i32, [c_i32, c_i32]),
    "rn_fp32_split_min_k": (c_i32, []),
    "rn_split_weights": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "rn_split_weights_f16": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "rn_amax": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "rn_anchor_count": (c_i64, [c_i32, c_i32]),
    "rn_anchor_base_boxes": (None, [c_vp]),
    "rn_anchors_fwd": (c_i32, [c_vp, c_i32, c_i32, c_vp]),
    "rn_pairwise_iou": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "rn_focal_workspace_bytes": (c_i64, [c_i32, c_i64]),
    "rn_focal_workspace_zero_bytes": (c_i64, [c_i32]),
    "rn_focal_loss_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rn_focal_loss_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rn_assign": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rn_decode_dir": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp]),
    "rn_decode_dir_select": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rn_decode_2d": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_f32, c_f32, c_vp]),
    "rn_clip_boxes": (c_i32, [c_vp, c_i64, c_f32, c_f32, c_vp]),
    "rn_post_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "rn_rowmax": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "rn_threshold_select": (c_i32, [c_vp, c_i64, c_i64, c_f64, c_i32, c_f64, c_vp, c_vp, c_vp, c_vp]),
    "rn_nms": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "rn_state_to_space": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "rn_space_to_state": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "rn_state_to_im": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rn_space_to_im": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rn_im_to_space": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rn_im_to_state": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rn_parse_workspace_bytes": (c_i64, [c_i64]),
    "rn_parse_detections": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_f32, c_f32, c_f32,
                                    c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rn_md_iou": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rn_crop_boxes": (c_i32, [c_vp, c_vp, c_i32, c_f64, c_vp, c_vp, c_vp]),
    "rn_roi_align": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "rn_crop_select": (c_i32, [c_vp] * 9 + [c_i32] * 4 + [c_f64, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "rn_kf_view": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "rn_kf_predict": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_f64, c_i32, c_vp]),
    "rn_kf_update": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "rn_frame_ingest": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32] + [c_f32] * 6 + [c_i32, c_vp, c_vp]),
}

class ConvDesc(ctypes.Structure):
    """rn_conv_desc of include/retinanet_mi355x.h."""
    _fields_ = [("N", c_i32), ("Hi", c_i32), ("Wi", c_i32), ("Cin", c_i32),
                ("Ho", c_i32), ("Wo", c_i32), ("Cout", c_i32),
                ("kh", c_i32), ("kw", c_i32),
                ("a", c_i32), ("b", c_i32), ("p", c_i32), ("p_w", c_i32), ("div_shift", c_i32),
                ("act", c_i32), ("add_mode", c_i32), ("Ha", c_i32), ("Wa", c_i32),
                ("mask_mode", c_i32), ("in_relu", c_i32),
                ("os", c_i32), ("oo_h", c_i32), ("oo_w", c_i32), ("Hy", c_i32), ("Wy", c_i32),
                ("add2_mode", c_i32), ("Ha2", c_i32), ("Wa2", c_i32), ("add2_batch_stride", c_i64),
                ("x_batch_stride", c_i64), ("y_batch_stride", c_i64), ("add_batch_stride", c_i64),
                ("w_batch_stride", c_i64), ("w_format", c_i32), ("sign_out", c_vp),
                ("x_amax", c_vp), ("y_amax", c_vp), ("w_unscale", c_vp), ("x_amax_img_stride", c_i32), ("x_amax_row_stride", c_i32)]


RN_MAX_GROUP = 5


class ConvGroup(ctypes.Structure):
    """rn_conv_group of include/retinanet_mi355x.h."""
    _fields_ = [("n", c_i32), ("tile_end", c_i32 * RN_MAX_GROUP), ("d", ConvDesc * RN_MAX_GROUP),
                ("x", c_vp * RN_MAX_GROUP), ("y", c_vp * RN_MAX_GROUP), ("add", c_vp * RN_MAX_GROUP),
                ("mask", c_vp * RN_MAX_GROUP)]


class PrepJob(ctypes.Structure):
    """rn_prep_job of include/retinanet_mi355x.h."""
    _fields_ = [("kind", c_i32)] + [(n, c_i32) for n in ("Cout", "Cin", "kh", "kw", "kw_pad", "c_pad", "mode", "r0", "nr", "s0",
                                                          "ns", "rows", "Kpad")] + \
               [("eps", c_f32), ("src", c_vp), ("dst", c_vp), ("scale", c_vp), ("gamma", c_vp), ("beta", c_vp), ("mean", c_vp),
                ("var", c_vp), ("bn_scale", c_vp), ("bn_shift", c_vp), ("bn_rstd", c_vp)]


class UnpackJob(ctypes.Structure):
    """rn_unpack_job of include/retinanet_mi355x.h."""
    _fields_ = [("dw", c_vp), ("w_packed", c_vp), ("dweight", c_vp)] + \
               [(n, c_i32) for n in ("Cout", "Cin", "kh", "kw", "kw_pad", "c_pad", "Kpad")] + \
               [("scale", c_vp), ("mean", c_vp), ("rstd", c_vp), ("colsum", c_vp), ("dgamma", c_vp), ("dbeta", c_vp)]


class WinoGroup(ctypes.Structure):
    """rn_wino_group of include/retinanet_mi355x.h."""
    _fields_ = [("n", c_i32), ("N", c_i32 * RN_MAX_GROUP), ("H", c_i32 * RN_MAX_GROUP), ("W", c_i32 * RN_MAX_GROUP),
                ("src", c_vp * RN_MAX_GROUP), ("dst", c_vp * RN_MAX_GROUP), ("add", c_vp * RN_MAX_GROUP),
                ("mask", c_vp * RN_MAX_GROUP), ("sign", c_vp * RN_MAX_GROUP), ("amax", c_vp * RN_MAX_GROUP)]


SIGNATURES.update({
    "rn_unpack_batched": (c_i32, [c_vp, c_vp, c_i32, c_vp]),
    "rn_prep_batched": (c_i32, [c_vp, c_vp, c_i32, c_vp]),
    "rn_wino_input_group": (c_i32, [ctypes.POINTER(WinoGroup), c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "rn_wino_input_both_group": (c_i32, [ctypes.POINTER(WinoGroup), c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "rn_wino_output_group": (c_i32, [ctypes.POINTER(WinoGroup), c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp]),
    "rn_conv_igemm_grouped": (c_i32, [ctypes.POINTER(ConvGroup), c_vp, c_vp, c_vp, c_vp]),
    "rn_conv_igemm": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rn_conv_splitk_workspace_bytes": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "rn_conv_igemm_wants_f16": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "rn_conv_igemm_splitk": (c_i32, [ctypes.POINTER(ConvDesc)] + [c_vp] * 10),
    "rn_conv_wgrad": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp] + [c_i32] * 12 + [c_vp]),
    "rn_pack_weights": (c_i32, [c_vp, c_vp] + [c_i32] * 7 + [c_vp] + [c_i32] * 4 + [c_vp]),
    "rn_unpack_wgrad": (c_i32, [c_vp, c_vp, c_vp] + [c_i32] * 6 + [c_vp] * 7),
    "rn_bn_fold": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rn_nchw_to_nhwc4": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "rn_maxpool_fwd": (c_i32, [c_vp, c_vp, c_vp] + [c_i32] * 6 + [c_vp]),
    "rn_maxpool_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp] + [c_i32] * 7 + [c_vp, c_vp]),
    "rn_colsum": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "rn_colsum_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "rn_upsample_add_bwd": (c_i32, [c_vp, c_vp] + [c_i32] * 6 + [c_vp, c_vp]),
    "rn_relu_mask": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "rn_sigmoid_bwd_pad": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i64, c_vp, c_vp]),
    "rn_add_inplace": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "rn_wino_input": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_vp]),
    "rn_wino_output": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp]),
    "rn_wino_weights": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "rn_wino_dy": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_vp]),
    "rn_wino_dw": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "rn_conv_wgrad_batched": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32] + [c_i32] * 12 + [c_vp, c_i32, c_vp, c_i32, c_vp]),
    "rn_conv_wgrad_batched_det": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_i64, c_i32] + [c_i32] * 12 + [c_vp, c_i32, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "rn_conv_wgrad_det_workspace_bytes": (c_i64, [c_i32, c_i32, c_i64] + [c_i32] * 9),
    "rn_f32_to_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "rn_bf16_to_f32": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "rn_conv_igemm_bf16": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rn_conv_igemm_bf16_grouped": (c_i32, [ctypes.POINTER(ConvGroup), c_vp, c_i32, c_vp, c_vp, c_vp]),
    "rn_conv_igemm_bf16_tile_rows": (c_i32, [ctypes.POINTER(ConvGroup), c_i32]),
    "rn_maxpool_fwd_fp8out": (c_i32, [c_vp, c_vp] + [c_i32] * 6 + [c_f32, c_vp]),
    "rn_conv_igemm_fp8_tile_rows": (c_i32, [ctypes.POINTER(ConvGroup), c_i32]),
    "rn_conv_igemm_fp8_tile": (c_i32, [ctypes.POINTER(ConvDesc), c_i32]),
    "rn_conv_igemm_bf16_tile": (c_i32, [ctypes.POINTER(ConvDesc), c_i32]),
    "rn_fp8_quantize": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    "rn_fp8_dequantize": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    "rn_fp8_to_bf16": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    "rn_bf16_to_fp8": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp]),
    "rn_fp8_quantize_rows": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "rn_conv_igemm_fp8": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp]),
    "rn_conv_igemm_fp8_grouped": (c_i32, [ctypes.POINTER(ConvGroup), c_vp, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp]),
    "rn_conv_wgrad_bf16": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp] + [c_i32] * 11 + [c_vp]),
    "rn_conv_wgrad_bf16_grouped": (c_i32, [c_i32, ctypes.POINTER(c_vp), c_i32, ctypes.POINTER(c_vp), c_vp, c_vp, c_i32,
                                           ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)] + [c_i32] * 6 + [c_vp]),
    "rn_maxpool_fwd_bf16out": (c_i32, [c_vp, c_vp, c_vp] + [c_i32] * 6 + [c_vp]),
    "rn_maxpool_bwd_bf16in": (c_i32, [c_vp, c_vp, c_vp, c_vp] + [c_i32] * 7 + [c_vp]),
    "rn_upsample_add_bwd_bf16": (c_i32, [c_vp, c_vp] + [c_i32] * 6 + [c_vp]),
    "rn_sigmoid_bwd_pad_bf16": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i64, c_vp]),
    "rn_relu_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "rn_opt_workspace_bytes": (c_i64, [c_i32]),
    "rn_opt_clip_adam": (c_i32, [c_vp, c_vp, c_i32, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rn_opt_clip_adam_dev": (c_i32, [c_vp, c_vp, c_i32, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "rn_opt_clip_adam_hp": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
})


RN_ERRORS = {10001: "RN_EINVAL (bad size / unsupported shape)",
             10002: "RN_ETOOMANY (more than RN_MAX_GT=256 label rows per image)"}


def load():
    """Load the shared library once and attach the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libretinanet_mi355x.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C 3d-playground_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, RN_ERRORS.get(rc, "hipError_t %d" % rc)))


def need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("retinanet_mi355x ops run on an MI355X only: got a %s tensor (no CPU fallback)" % t.device)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def f32c(t):
    """Contiguous fp32 view/copy of a tensor (plumbing: layout only)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
