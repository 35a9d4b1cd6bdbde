"""ctypes binding of libtubedetr_hip.so (include/tubedetr_hip.h).

The product path has NO CPU fallback: if the library is missing or a kernel reports an error this
module raises.  Tensors cross the boundary as raw device pointers + the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

TD_F32, TD_BF16 = 0, 1
EXPECTED_ABI = 9  # td_abi_version() of the library these signatures were written against
_LIB_PATH = os.environ.get("TD_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtubedetr_hip.so")  # (TD_HIP_LIB: an A/B build of the same ABI, tools/build_variant.sh)
_lib = None


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "Hs", "Ws", "C", "Ho", "Wo", "R", "S", "stride", "pad", "mode", "Nc", "ldc", "out_sp", "out_H", "out_W",
                                       "aniso", "stride_w", "pad_w")]


class PrepItem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W", "bn_w", "bn_b", "bn_rm", "bn_rv", "bias", "w_fwd", "w_dgrad", "bias_out", "scale_out")] + \
               [(n, C.c_int) for n in ("Co", "Ci", "RS", "Cpad", "Co_alloc", "blk0")]


class WgradJob(C.Structure):
    """td_wgrad_job (include/tubedetr_hip.h)."""
    _fields_ = [("g", C.c_void_p), ("src", C.c_void_p), ("dW", C.c_void_p), ("scale", C.c_void_p), ("d", ConvDesc), ("ldg", C.c_int), ("ci_real", C.c_int), ("dbias", C.c_void_p), ("accumulate", C.c_int), ("prezeroed", C.c_int)]


TD_U8 = 2


class FrameSource(C.Structure):
    """td_frame_source."""
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int), ("n", C.c_int), ("index", C.c_void_p), ("valid_hw", C.c_void_p)]


class LinearExDesc(C.Structure):
    """td_linear_ex_desc."""
    _fields_ = [(n, C.c_int) for n in ("M", "N", "K1", "K2", "lda1", "lda2", "ldc", "ldr", "w_shared")] + \
               [("rows1", C.c_longlong), ("rows2", C.c_longlong)] + [(n, C.c_void_p) for n in ("a1_map", "a2_map", "out_map", "res_map")]


class OptimSegment(C.Structure):
    """td_optim_segment."""
    _fields_ = [("begin", C.c_longlong), ("end", C.c_longlong), ("group", C.c_int), ("active", C.c_int)]


class Epilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p),
        ("residual", C.c_void_p),
        ("mask_src", C.c_void_p),
        ("relu", C.c_int),
        ("sigmoid", C.c_int),
        ("dropout_p", C.c_float),
        ("dropout_seed", C.c_uint32),
        ("alpha", C.c_float),
        ("dropout_counter", C.c_void_p),
    ]


_P, _I, _F, _U32, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_size_t
_SIGS = {
    "td_prof_enable": [_I],
    "td_prof_dump": [C.c_char_p],
    "td_debug_set_stamp_buffer": [_P],
    "td_prof_collect": [_I, _I, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "td_prof_collect_bytes": [_I, _I, C.POINTER(C.c_double)],
    "td_conv_gemm": [_P, _P, _P, C.POINTER(ConvDesc), C.POINTER(Epilogue), _I, _P],
    "td_conv_wgrad": [_P, _P, _P, C.POINTER(ConvDesc), _I, _I, _I, _P],
    "td_conv_wgrad_bias": [_P, _P, _P, _P, C.POINTER(ConvDesc), _I, _I, _I, _P],
    "td_conv_wgrad_batch": [C.POINTER(WgradJob), _I, _I, _P, _P, _SZ, _P],
    "td_resnet_num_convs": [C.POINTER(C.c_int)],
    "td_resnet_fwd": [C.POINTER(FrameSource), _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _I, _I, _I, C.POINTER(C.c_int), C.POINTER(_P), C.POINTER(_P), _I, _P, _SZ,
                      C.POINTER(_P), C.POINTER(C.c_int), _I, _I, _I, _P],
    "td_bottleneck_fused": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "td_stem_pair_weights": [_P, _P, _I, _I, _P],
    "td_stem_pool": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "td_pw_chain2": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "td_set_deterministic": [_I],
    "td_get_deterministic": [],
    "td_frames_to_nhwc": [C.POINTER(FrameSource), _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _I, _P],
    "td_resnet_bwd": [_P, _I, _I, _I, _I, C.POINTER(C.c_int), _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P, _P, _SZ, _P, _P, _SZ, _I, _I, _I, _P],
    "td_weight_prep_batch": [_P, _I, _I, _I, _P],
    "td_weight_prep": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "td_wgrad_finalize": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "td_nchw_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "td_nhwc_to_nchw": [_P, _P, _I, _I, _I, _I, _I, _P],
    "td_cast": [_P, _P, _SZ, _I, _I, _P],
    "td_maxpool3x3s2": [_P, _P, _I, _I, _I, _I, _I, _P],
    "td_add_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "td_add_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "td_colsum": [_P, _P, _I, _I, _I, _I, _P],
    "td_add": [_P, _P, _P, _SZ, _I, _P],
    "td_relu_bwd": [_P, _P, _P, _SZ, _F, _I, _P],
    "td_gelu_fwd": [_P, _P, _SZ, _I, _P],
    "td_gelu_bwd": [_P, _P, _P, _SZ, _I, _P],
    "td_dropout": [_P, _P, _SZ, _F, _U32, _P, _I, _P],
    "td_pos_sine": [_P, _P, _I, _I, _I, _I, _F, _I, _I, _P],
    "td_linear_ex": [_P, _P, _P, _P, C.POINTER(LinearExDesc), C.POINTER(Epilogue), _I, _P],
    "td_rows_copy": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "td_rows_segment_sum": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "td_criterion_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "td_criterion_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "td_sted_decode": [_P, _P, _I, _I, _P],
    "td_grad_norm_clip": [_P, _SZ, C.POINTER(OptimSegment), _I, _F, _P, _SZ, _P, _P, _P],
    "td_adamw_ema_step": [_P, _P, _P, _P, _P, _SZ, C.POINTER(OptimSegment), _I, _P, _P, _P, _F, _F, _F, _F, _F, _P],
    "td_mha_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _U32, _P, _I, _P],
    "td_mha_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _U32, _P, _I, _P],
    "td_cross_q1_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _U32, _P, _I, _P],
    "td_cross_q1_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _U32, _P, _I, _P],
    "td_cross_q1_bwd_coef": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _U32, _P, _I, _P],
    "td_cross_q1_dmem": [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "td_head_blocks_expand": [_P, _P, _F, _P, _P, _I, _I, _I, _P],
    "td_head_blocks_extract": [_P, _F, _P, _P, _I, _I, _P],
    "td_mha_lean_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _U32, _P, _I, _P],
    "td_mha_lean_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _U32, _P, _I, _P],
}
_SIZE_SIGS = {
    "td_grad_norm_ws_bytes": [],
    "td_mha_lean_stats_bytes": [_I, _I, _I],
    "td_conv_wgrad_batch_table_bytes": [_I],
    "td_resnet_bwd_table_bytes": [C.POINTER(C.c_int), _I],
    "td_resnet_fwd_ws_bytes": [_I, _I, _I, C.POINTER(C.c_int), _I, _I],
    "td_resnet_bwd_ws_bytes": [_I, _I, _I, C.POINTER(C.c_int), _I, _I],
}
EXPORTS = ["td_last_error", "td_abi_version"] + list(_SIGS) + list(_SIZE_SIGS)


def lib_path() -> str:
    return _LIB_PATH


def lib() -> C.CDLL:
    """Load the shared library (raises if it was not built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -m tubedetr_amd.build` (or __graft_entry__.build()); "
                "tubedetr_amd has no CPU / eager fallback"
            )
        L = C.CDLL(_LIB_PATH)
        L.td_last_error.restype = C.c_char_p
        L.td_last_error.argtypes = []
        L.td_abi_version.restype = C.c_int
        L.td_abi_version.argtypes = []
        got = L.td_abi_version()
        if got != EXPECTED_ABI:  # a stale lib/ from an older checkout would be called with a different argument list
            raise RuntimeError(f"{_LIB_PATH} has ABI version {got}, this binding needs {EXPECTED_ABI}: rebuild it with "
                               "`python -m tubedetr_amd.build --force`")
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = sig
        for name, sig in _SIZE_SIGS.items():
            fn = getattr(L, name)
            fn.restype = C.c_size_t
            fn.argtypes = sig
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RuntimeError(f"libtubedetr_hip {what} failed ({rc}): {lib().td_last_error().decode()}")


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream (the C getter is ~20x cheaper than torch.cuda.current_stream())."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return TD_F32
    if dt == torch.bfloat16:
        return TD_BF16
    raise TypeError(f"unsupported compute dtype {dt}")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous & on the GPU."""
    if t is None:
        return None
    assert t.is_cuda, "tubedetr_amd kernels only run on the GPU"
    return t.data_ptr()
