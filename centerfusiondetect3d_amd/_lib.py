"""ctypes binding of libcfhip.so (include/cf_hip.h).  No CPU fallback: if the library is missing
or a call fails, this raises - the product path never silently degrades."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libcfhip.so")

ABI_VERSION = 6      # CF_ABI_VERSION of include/cf_hip.h this binding was written against
CF_MAX_SRC = 4
ACT_NONE, ACT_RELU, ACT_SIGMOID_CLAMP, ACT_RAW_AND_SIGDEPTH = 0, 1, 2, 3
LAYOUT_NHWC, LAYOUT_NCHW, LAYOUT_NHWC_SPLIT_BF16 = 0, 1, 2

_f = C.c_void_p  # device pointers travel as integers


class ConvArgs(C.Structure):
    _fields_ = [("src", _f * CF_MAX_SRC), ("src_c", C.c_int32 * CF_MAX_SRC), ("n_src", C.c_int32),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Ho", C.c_int32),
                ("Wo", C.c_int32), ("stride", C.c_int32), ("weight", _f), ("slots", _f),
                ("bias", _f), ("K_pad", C.c_int32), ("N", C.c_int32), ("N_pad", C.c_int32),
                ("residual", _f), ("res_stride", C.c_int32), ("out", _f), ("out2", _f),
                ("out_stride", C.c_int32), ("out_layout", C.c_int32), ("act", C.c_int32),
                ("precise", C.c_int32), ("out_scale", C.c_float), ("in_scale", C.c_float)]


class DcnArgs(C.Structure):
    _fields_ = [("x", _f), ("offmask", _f), ("om_stride", C.c_int32), ("B", C.c_int32),
                ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("weight", _f), ("bias", _f),
                ("N", C.c_int32), ("N_pad", C.c_int32), ("out", _f), ("out_stride", C.c_int32),
                ("act", C.c_int32), ("precise", C.c_int32), ("out_scale", C.c_float),
                ("out_split_bf16", _f), ("split_stride", C.c_int32), ("workspace", _f),
                ("workspace_bytes", C.c_size_t), ("mask_activated", C.c_int32), ("out_mx", _f), ("in_scale", C.c_float),
                ("mx_scale", C.c_float)]


CF_MAX_HEADS = 12


class HeadTailArgs(C.Structure):
    _fields_ = [("x", _f), ("x_stride", C.c_int32), ("B", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("n_heads", C.c_int32), ("n_hidden", C.c_int32),
                ("w_hidden", (_f * 2) * CF_MAX_HEADS), ("b_hidden", (_f * 2) * CF_MAX_HEADS),
                ("w_out", _f * CF_MAX_HEADS), ("b_out", _f * CF_MAX_HEADS),
                ("out", _f * CF_MAX_HEADS), ("out2", _f * CF_MAX_HEADS),
                ("c_base", C.c_int32 * CF_MAX_HEADS), ("n_out", C.c_int32 * CF_MAX_HEADS),
                ("act", C.c_int32 * CF_MAX_HEADS)]


class HeadFusedArgs(C.Structure):
    _fields_ = [("tail", HeadTailArgs), ("src", _f * 2), ("src_c", C.c_int32 * 2), ("n_src", C.c_int32),
                ("slots", _f), ("K_pad", C.c_int32), ("w_first", _f * CF_MAX_HEADS),
                ("b_first", _f * CF_MAX_HEADS), ("layout3x3", C.c_int32), ("w_out_perm", _f * CF_MAX_HEADS),
                ("mfma16", C.c_int32), ("mx", C.c_int32), ("first_scale", C.c_float * CF_MAX_HEADS)]


class StemArgs(C.Structure):
    _fields_ = [("x", _f), ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("w_base", _f), ("b_base", _f), ("scale_base", C.c_float),
                ("w_level0", _f), ("b_level0", _f), ("scale_level0", C.c_float),
                ("w_level1", _f), ("b_level1", _f), ("scale_level1", C.c_float),
                ("out", _f), ("out_pool", _f), ("in_scale", C.c_float * 3)]


class PackSrc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("stride", C.c_int32), ("c_base", C.c_int32)]


class PackBn(C.Structure):
    _fields_ = [("gamma", _f), ("beta", _f), ("mean", _f), ("var", _f), ("eps", C.c_float)]


class PackConvDesc(C.Structure):
    _fields_ = [("weight", _f), ("bias", _f), ("bn", PackBn),
                ("cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("dilation", C.c_int32),
                ("src", C.POINTER(PackSrc)), ("n_src", C.c_int32),
                ("proj_weight", _f), ("proj_bias", _f), ("proj_bn", PackBn), ("proj", PackSrc)]


class PackInfo(C.Structure):
    _fields_ = [("n_pad", C.c_int32), ("k_pad", C.c_int32), ("n_slots", C.c_int32), ("patch", C.c_int32),
                ("out_scale", C.c_float), ("weight_bytes", C.c_size_t)]


class DecodeArgs(C.Structure):
    _fields_ = [("scores", _f), ("inds", _f), ("classes", _f), ("reg", _f), ("wh", _f),
                ("depth", _f), ("rot", _f), ("dim", _f), ("amodal", _f), ("att", _f), ("vel", _f),
                ("B", C.c_int32), ("K", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("out_h", C.c_int32), ("out_w", C.c_int32), ("norm2d", C.c_int32), ("det", _f)]


class SerializeArgs(C.Structure):
    _fields_ = [("post", _f), ("B", C.c_int32), ("K", C.c_int32), ("trans_matrix", _f),
                ("velocity_matrix", _f), ("cs_rot", _f), ("pose_rot", _f), ("rows", _f), ("rotation", _f),
                ("n_samples", C.c_int32), ("sample_ptr", _f), ("sample_frames", _f),
                ("max_per_sample", C.c_int32), ("order", _f), ("counts", _f)]


# every symbol include/cf_hip.h declares: (restype, argtypes)
_i, _d = C.c_int, C.c_double
SYMBOLS = {
    "cf_conv2d_fused": (_i, [C.POINTER(ConvArgs), _f]),
    "cf_conv2d_bf16x3": (_i, [C.POINTER(ConvArgs), _f]),
    "cf_conv2d_f16x3": (_i, [C.POINTER(ConvArgs), _f]),
    "cf_conv3x3_f16x3": (_i, [C.POINTER(ConvArgs), _f]),
    "cf_conv3x3_root_f16x3": (_i, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), C.POINTER(C.c_int32), _f]),
    "cf_conv3x3_proj_f16x3": (_i, [C.POINTER(ConvArgs), C.POINTER(C.c_int32), _f]),
    "cf_stem_fused": (_i, [C.POINTER(StemArgs), _f]),
    "cf_split_bf16": (_i, [_f, _f, C.c_long, _i, _i, _i, _f]),
    "cf_head_tail": (_i, [C.POINTER(HeadTailArgs), _f]),
    "cf_head_fused": (_i, [C.POINTER(HeadFusedArgs), _f]),
    "cf_pack_feat_mx": (_i, [_f, _i, _f, C.c_long, _f]),
    "cf_pack_feat_mx_scaled": (_i, [_f, _i, _f, C.c_long, C.c_float, _f]),
    "cf_absmax_f32": (_i, [_f, C.c_long, _i, _i, _f, _f]),
    "cf_dcn_v2_fused": (_i, [C.POINTER(DcnArgs), _f]),
    "cf_dcn_v2_f16x3": (_i, [C.POINTER(DcnArgs), _f]),
    "cf_dcn_v2_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "cf_upsample_dw": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _f]),
    "cf_maxpool2x2": (_i, [_f, _f, _i, _i, _i, _i, _f]),
    "cf_nchw_to_nhwc4": (_i, [_f, _f, _i, _i, _i, _i, _f]),
    "cf_nhwc_to_nchw": (_i, [_f, _f, _i, _i, _i, _i, _i, _f]),
    "cf_nchw_to_nhwc": (_i, [_f, _f, _i, _i, _i, _i, _i, _i, _f]),
    "cf_radar_ingest": (_i, [_f, _f, _i, _i, _i, _f, _i, _i, _d, _d, _i, _f, _f, _f, _f]),
    "cf_preprocess_images": (_i, [_f, _i, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_float),
                                 C.POINTER(C.c_float), _i, _i, _f, _f]),
    "cf_topk_workspace_bytes": (C.c_size_t, [_i, _i]),
    "cf_topk_workspace_bytes_nms": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "cf_topk_peaks": (_i, [_f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f]),
    "cf_topk_peaks_if_changed": (_i, [_f, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f]),
    "cf_checksum64": (_i, [_f, C.c_long, _f, _f]),
    "cf_frustum_assoc": (_i, [_f, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, C.c_float, _f, _f, _f, _f]),
    "cf_topk_frustum": (_i, [_f, _i, _i, _f, _f, _f, _f, _f, _f, _i, _i, _i, C.c_float, _f, _f, _f, _f, _f, _f, _f, _f]),
    "cf_pillar_expand": (_i, [_f, _f, _f, _i, _i, _i, _f, _f, _i, _i, _d, _d, _d, _f, _f, _f, _f]),
    "cf_decode_gather": (_i, [C.POINTER(DecodeArgs), _f]),
    "cf_post_process": (_i, [_f, _f, _f, _i, _i, _i, _i, _f, _f]),
    "cf_decode_post": (_i, [C.POINTER(DecodeArgs), _f, _f, _f, _f]),
    "cf_serialize_nuscenes": (_i, [C.POINTER(SerializeArgs), _f]),
    "cf_serialize_max_candidates": (_i, []),
    "cf_spin_us": (_i, [_i, _f]),
    "cf_last_error": (C.c_char_p, []),
    "cf_pack_conv_f16x3_info": (_i, [C.POINTER(PackConvDesc), C.POINTER(PackInfo)]),
    "cf_pack_conv_f16x3": (_i, [C.POINTER(PackConvDesc), _f, _f, _f, C.POINTER(PackInfo)]),
    "cf_pack_dcn_f16_info": (_i, [_i, _i, C.POINTER(PackInfo)]),
    "cf_pack_dcn_f16": (_i, [_f, _f, C.POINTER(PackBn), _i, _i, _f, _f, C.POINTER(PackInfo)]),
    "cf_abi_version": (_i, []),
}

_lib = None


class CfHipError(RuntimeError):
    pass


def load():
    """Load libcfhip.so (once).  Raises if it has not been built - there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CfHipError(
                f"{LIB_PATH} is missing: build it with `python -m centerfusiondetect3d_amd.build` "
                "(hipcc, gfx950). The CenterFusion forward has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.cf_abi_version() != ABI_VERSION:
            raise CfHipError(f"{LIB_PATH} has ABI version {lib.cf_abi_version()}, this package needs {ABI_VERSION}: "
                             "rebuild it with `python -m centerfusiondetect3d_amd.build`")
        _lib = lib
    return _lib


def check(status, what=""):
    if status != 0:
        msg = load().cf_last_error().decode()
        raise CfHipError(f"{what} failed with status {status}: {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
