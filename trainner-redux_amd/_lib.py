"""ctypes binding of libtrainner_hip.so (include/trainner_hip.h).

The library is mandatory: there is no eager / PyTorch / CPU fallback anywhere in this package.
`lib()` raises `RuntimeError` when the shared object is missing or its ABI version differs.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtrainner_hip.so")
ABI_VERSION = 4

F32, BF16, F16, BF16X3 = 0, 1, 2, 3
F32_WINOGRAD43 = 101    # ... tr_conv3x3_winograd43_f32 (F(4x4, 3x3), weights from tr_pack_winograd43)
F32_WINOGRAD = 100      # ops.conv3x3 routes this pseudo compute code to tr_conv3x3_winograd_f32 (fp32 views, Winograd-packed weights)
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_GELU, ACT_SILU, ACT_MISH = 0, 1, 2, 3, 4, 5
PAD_ZEROS, PAD_REPLICATE = 0, 1
IN_DIRECT, IN_NEAREST2X, IN_FULL, IN_VALID = 0, 1, 2, 3
LOSS_L1, LOSS_MSE, LOSS_CHARBONNIER = 0, 1, 2
GAN_TYPES = {"vanilla": 0, "lsgan": 1, "wgan": 2, "wgan_softplus": 3, "hinge": 4}


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int32), ("cstride", C.c_int32), ("coff", C.c_int32), ("c", C.c_int32)]


class Conv3x3Desc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32),
        ("compute", C.c_int32), ("in_mode", C.c_int32), ("pad_mode", C.c_int32), ("act", C.c_int32),
        ("slope", C.c_float), ("scale1", C.c_float), ("scale2", C.c_float), ("res1_scale", C.c_float), ("res2_scale", C.c_float),
        ("mask_slope", C.c_float), ("out2_scale", C.c_float),
        ("inp", View), ("out", View), ("out2", View), ("res1", View), ("res2", View), ("mask", View),
        ("wpacked", C.c_void_p), ("bias", C.c_void_p),
        ("ksize", C.c_int32), ("mask_mode", C.c_int32), ("sample_scale", C.c_void_p),
    ]


class PackSeg(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p),
        ("src_o", C.c_int32), ("src_i", C.c_int32),
        ("o_off", C.c_int32), ("o_cnt", C.c_int32), ("i_off", C.c_int32), ("i_cnt", C.c_int32),
        ("dst_cin", C.c_int32), ("dst_cout", C.c_int32),
        ("dst_o_off", C.c_int32), ("dst_i_off", C.c_int32),
        ("transpose", C.c_int32), ("compute", C.c_int32), ("ksize", C.c_int32), ("reserved", C.c_int32),
        ("first", C.c_int64),
    ]


class Wgrad3x3Desc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32),
        ("compute", C.c_int32), ("in_mode", C.c_int32), ("pad_mode", C.c_int32), ("accumulate", C.c_int32),
        ("inp", View), ("gy", View),
        ("dw", C.c_void_p), ("db", C.c_void_p), ("partials", C.c_void_p), ("partials_bytes", C.c_size_t),
        ("ksize", C.c_int32), ("dw_ld", C.c_int32),
    ]


class PixelLossDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dtype", C.c_int32), ("grad_dtype", C.c_int32), ("accumulate_grad", C.c_int32),
        ("pred", C.c_void_p), ("target", C.c_void_p), ("numel", C.c_int64),
        ("eps", C.c_float), ("value_weight", C.c_float), ("total_weight", C.c_float), ("grad_weight", C.c_float),
        ("value_out", C.c_void_p), ("total_out", C.c_void_p), ("grad", C.c_void_p), ("partials", C.c_void_p),
    ]


class GanLossDesc(C.Structure):
    _fields_ = [
        ("gan_type", C.c_int32), ("target_is_real", C.c_int32), ("is_disc", C.c_int32), ("reserved", C.c_int32),
        ("logits", C.c_void_p), ("numel", C.c_int64),
        ("real_label", C.c_float), ("fake_label", C.c_float), ("value_weight", C.c_float), ("total_weight", C.c_float), ("grad_weight", C.c_float),
        ("value_out", C.c_void_p), ("total_out", C.c_void_p), ("grad", C.c_void_p), ("partials", C.c_void_p),
    ]


class WindowAttnDesc(C.Structure):
    _fields_ = [
        ("qkv", View), ("out", View), ("dout", View), ("dqkv", View),
        ("table", C.c_void_p), ("dtable", C.c_void_p), ("dbias_ws", C.c_void_p),
        ("b", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("nh", C.c_int32), ("window", C.c_int32), ("shift", C.c_int32),
        ("scale", C.c_float), ("accumulate_dtable", C.c_int32),
    ]


class AdamWDesc(C.Structure):
    _fields_ = [
        ("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p),
        ("numel", C.c_int64),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
        ("bias_corr1", C.c_float), ("bias_corr2", C.c_float), ("ema_decay", C.c_float), ("clip", C.c_float),
        ("grad_norm", C.c_void_p), ("zero_grad", C.c_int32), ("inv_scale", C.c_float),
    ]


# every symbol include/trainner_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "tr_abi_version": (C.c_int, []),
    "tr_last_error": (C.c_char_p, []),
    "tr_tune_set": (C.c_int, [C.c_char_p, C.c_longlong]),
    "tr_tune_get": (C.c_longlong, [C.c_char_p]),
    "tr_conv3x3_packed_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "tr_conv_packed_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_conv3x3": (C.c_int, [C.c_void_p, C.POINTER(Conv3x3Desc)]),
    "tr_winograd_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "tr_pack_winograd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tr_conv3x3_winograd_f32": (C.c_int, [C.c_void_p, C.POINTER(Conv3x3Desc)]),
    "tr_winograd43_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "tr_pack_winograd43": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tr_conv3x3_winograd43_f32": (C.c_int, [C.c_void_p, C.POINTER(Conv3x3Desc)]),
    "tr_conv3x3_dgrad_image": (C.c_int, [C.c_void_p, View, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_float), C.c_void_p, C.c_int]),
    "tr_conv3x3_stem": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, View]),
    "tr_fold_replicate_nchw": (C.c_int, [C.c_void_p, View, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "tr_pack_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64]),
    "tr_wgrad3x3_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_wgrad3x3": (C.c_int, [C.c_void_p, C.POINTER(Wgrad3x3Desc)]),
    "tr_wgrad3x3_multi": (C.c_int, [C.c_void_p, C.POINTER(Wgrad3x3Desc), C.c_int]),
    "tr_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, View, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tr_nhwc_to_nchw": (C.c_int, [C.c_void_p, View, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_nhwc_to_nchw_scaled": (C.c_int, [C.c_void_p, View, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "tr_nhwc_to_nchw_affine": (C.c_int, [C.c_void_p, View, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tr_layernorm": (C.c_int, [C.c_void_p, View, View, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "tr_layernorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int]),
    "tr_layernorm_bwd": (C.c_int, [C.c_void_p, View, View, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, View, View, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_size_t, View, C.c_void_p, C.c_int64]),
    "tr_window_attention": (C.c_int, [C.c_void_p, C.POINTER(WindowAttnDesc)]),
    "tr_window_attention_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_window_attention_bwd": (C.c_int, [C.c_void_p, C.POINTER(WindowAttnDesc)]),
    "tr_pixel_shuffle": (C.c_int, [C.c_void_p, View, View, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_axpby": (C.c_int, [C.c_void_p, View, C.c_float, View, C.c_float, View, C.c_int64, View, C.c_float, C.c_void_p, C.c_int64]),
    "tr_u8hwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tr_metric_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "tr_metric_sqerr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int)]),
    "tr_metric_ssim": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int)]),
    "tr_tile_blend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_tile_normalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "tr_filter2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "tr_quantize_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "tr_resample_axis": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "tr_crop_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_gain_noise_clamp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_float), C.c_float]),
    "tr_filter2d_zeropad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "tr_axpby_clamp01": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_int64]),
    "tr_warp_bilinear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "tr_color601": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int]),
    "tr_bayer_round_trip": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tr_span_gate": (C.c_int, [C.c_void_p, View, View, View, View, C.c_int64]),
    "tr_span_gate_bwd": (C.c_int, [C.c_void_p, View, View, View, View, View, C.c_int, C.c_int64]),
    "tr_dwconv": (C.c_int, [C.c_void_p, View, View, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_dwconv_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_dwconv_wgrad": (C.c_int, [C.c_void_p, View, View, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "tr_gate_mish": (C.c_int, [C.c_void_p, View, View, View, View, C.c_int64]),
    "tr_gate_mish_bwd": (C.c_int, [C.c_void_p, View, View, View, View, View, View, View, C.c_int64]),
    "tr_act_grad": (C.c_int, [C.c_void_p, C.c_int, View, View, View, C.c_int64, C.c_void_p, C.c_int64]),
    "tr_space_to_depth2": (C.c_int, [C.c_void_p, View, View, C.c_int, C.c_int, C.c_int]),
    "tr_depth_to_space2": (C.c_int, [C.c_void_p, View, View, C.c_int, C.c_int, C.c_int]),
    "tr_expand_w4x4s2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "tr_collapse_g3x3s2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tr_bilinear_up2x": (C.c_int, [C.c_void_p, View, View, C.c_int, C.c_int, C.c_int]),
    "tr_bilinear_up2x_bwd": (C.c_int, [C.c_void_p, View, View, C.c_int, C.c_int, C.c_int]),
    "tr_spectral_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "tr_spectral_norm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tr_gan_loss": (C.c_int, [C.c_void_p, C.POINTER(GanLossDesc)]),
    "tr_sumpool2x2": (C.c_int, [C.c_void_p, View, View, View, C.c_float, C.c_int, C.c_int, C.c_int]),
    "tr_maxpool2x2": (C.c_int, [C.c_void_p, View, View, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_maxpool2x2_bwd": (C.c_int, [C.c_void_p, View, View, View, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_pixel_loss": (C.c_int, [C.c_void_p, C.POINTER(PixelLossDesc)]),
    "tr_pixel_loss_pool_bwd": (C.c_int, [C.c_void_p, C.POINTER(PixelLossDesc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "tr_batchnorm_workspace_bytes": (C.c_size_t, [C.c_int]),
    "tr_batchnorm_fwd": (C.c_int, [C.c_void_p, View, View, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "tr_batchnorm_bwd": (C.c_int, [C.c_void_p, View, View, View, View, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_size_t]),
    "tr_sumsq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "tr_adamw_ema": (C.c_int, [C.c_void_p, C.POINTER(AdamWDesc)]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the C-ABI library.  Fails loudly; never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
            "This package has no non-HIP execution path."
        )
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and libtorch_hip asks for it as "libamdhip64.so":
    # when THIS library is loaded first, its NEEDED libamdhip64.so.7 resolves to /opt/rocm's copy, torch then maps its own next to it, and the process
    # holds two HIP runtimes -- every launch from here fails with "no ROCm-capable device is detected" (seen with build() + smoke() in one process).
    # Loaded after torch, the same NEEDED entry binds to the runtime torch already mapped: one runtime, shared streams.  A host without torch is
    # not affected (it has one runtime to begin with).
    import torch  # noqa: F401

    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:  # pragma: no cover
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = handle.tr_abi_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
    _lib = handle
    return handle


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().tr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"trainner_hip {what} failed (rc={rc}): {msg}")
