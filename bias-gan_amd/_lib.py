"""ctypes binding of libbgamd.so (the C ABI declared in include/bgamd.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, this raises.  The product path never routes through PyTorch ops or the
CPU oracle for its arithmetic.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BGAMD_LIB") or os.path.join(_HERE, "libbgamd.so")   # BGAMD_LIB: experiment builds

BF16, F32, FP8 = 0, 1, 2
FP8_E4M3, FP8_E5M2 = 0, 1
ABI_VERSION = 4

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [(n, c_i32) for n in ("dtype", "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad",
                                     "dil", "ldx", "ldy")]


class DwDesc(C.Structure):
    _fields_ = [(n, c_i32) for n in ("dtype", "N", "H", "W", "C", "Ho", "Wo", "stride", "dil", "ldx", "ldy")]


class Dw3Desc(C.Structure):
    _fields_ = [(n, c_i32) for n in ("dtype", "N", "D", "H", "W", "C", "Do", "Ho", "Wo", "stride", "dil", "ldx", "ldy")]


class NpyInfo(C.Structure):
    _fields_ = [("dtype_code", c_i32), ("typesize", c_i32), ("fortran_order", c_i32), ("ndim", c_i32),
                ("shape", c_i64 * 8), ("data_offset", c_i64), ("file_size", c_i64)]


# host-side entry points (no stream argument): name -> argtypes
_HOST_SIGS = {
    "bg_npy_parse": [C.c_char_p, C.POINTER(NpyInfo)],
    "bg_ring_create": [c_i32, c_i32, c_i64, c_i32, C.POINTER(c_vp)],
    "bg_ring_destroy": [c_vp],
    "bg_ring_submit": [c_vp, C.c_char_p, c_i64, c_i64, c_i32, C.POINTER(c_i64)],
    "bg_ring_acquire": [c_vp, c_i64, c_vp, C.POINTER(c_vp)],
    "bg_ring_copy_out": [c_vp, c_i64, c_vp, c_i64, c_vp],
    "bg_ring_release": [c_vp, c_i64, c_vp],
    "bg_stream_create": [C.POINTER(c_vp)],
    "bg_stream_destroy": [c_vp],
}

# name -> argtypes (restype is int for all but bg_last_error); mirrors include/bgamd.h
_SIGS = {
    "bg_abi_version": [],
    "bg_conv2d_fwd": [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp],
    "bg_conv2d_fwd_stats": [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp],
    "bg_conv2d_bwd_data": [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp],
    "bg_conv2d_bwd_weight": [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp],
    "bg_conv2d_bwd_weight_ws": [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "bg_conv2d_bwd_weight_ws_bytes": [C.POINTER(ConvDesc), c_vp],
    "bg_conv2d_bwd_weight_grouped": [c_i32, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_conv2d_bwd_weight_grouped_taps": [C.POINTER(ConvDesc), c_vp, c_i32, c_vp],
    "bg_pack_conv_weights": [c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp],
    "bg_quant_fp8": [c_i32, c_vp, c_i32, c_i64, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp],
    "bg_fp8_roll": [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp],
    "bg_pack_conv_weights_fp8": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp],
    "bg_conv2d_fwd_fp8": [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp],
    "bg_conv2d_bwd_data_fp8": [C.POINTER(ConvDesc), c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp],
    "bg_dwconv3x3_fwd": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_vp],
    "bg_dwconv3x3_bwd_data": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_vp],
    "bg_dwconv3x3_bwd_fused": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp,
                               c_vp, c_vp],
    "bg_dwconv3x3_bwd_fork": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32,
                              c_vp, c_vp, c_vp, c_vp],
    "bg_dwconv3x3_bwd_weight": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_vp],
    "bg_dwconv3x3_bwd_data_add": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_i32, c_vp, c_vp],
    "bg_dwconv3x3_fwd_pre": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp],
    "bg_dwconv3x3_bwd_weight_pre": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp],
    "bg_dwconv3x3_fwd_pre_stats": [C.POINTER(DwDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                   c_vp, c_i32, c_i32, c_vp, c_vp, c_vp],
    "bg_norm_stats": [c_i32, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp],
    "bg_norm_finalize": [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                         c_vp],
    "bg_norm_finalize_affine": [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                c_vp],
    "bg_norm_eval_affine": [c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp],
    "bg_norm_act_fwd": [c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp],
    "bg_norm_act_fwd_stats": [c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                              c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp],
    "bg_norm_act_bwd_apply_stats": [c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                                    c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp],
    "bg_norm_act_bwd_apply_stats_q8": [c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32,
                                    c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp],
    "bg_norm_act_bwd_reduce": [c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32,
                               c_i32, c_vp, c_vp, c_vp],
    "bg_norm_bwd_finalize": [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                             c_vp],
    "bg_norm_act_bwd_apply": [c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32,
                              c_i64, c_i32, c_i32, c_i32, c_vp],
    "bg_depth_unfold": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_depth_fold": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_dwconv3x3x3_fwd": [C.POINTER(Dw3Desc), c_vp, c_vp, c_vp, c_vp],
    "bg_dwconv3x3x3_bwd_data": [C.POINTER(Dw3Desc), c_vp, c_vp, c_vp, c_vp],
    "bg_dwconv3x3x3_bwd_weight": [C.POINTER(Dw3Desc), c_vp, c_vp, c_vp, c_vp],
    "bg_depth_resize_fwd": [c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_depth_resize_bwd": [c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_conv2d_fwd_splitk": [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_i32, c_vp],
    "bg_conv2d_bwd_data_splitk": [C.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_i32, c_vp],
    "bg_splitk_reduce": [c_i32, c_vp, c_i32, c_i64, c_i32, c_vp, c_i32, c_vp],
    "bg_depth_avg2": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_mask_window": [c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                       c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp],
    "bg_resize_nearest3d_rows": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_mul_rows": [c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp],
    "bg_scale_rows": [c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp],
    "bg_resize_nearest3d_fwd": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_resize_nearest3d_bwd": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_resize_trilinear3d_fwd": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_resize_trilinear3d_bwd": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_pc_dropout": [c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i64, c_i32, c_f32, c_vp],
    "bg_blend_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
    "bg_tv_loss_fwd": [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp],
    "bg_tv_loss_bwd": [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp],
    "bg_avgpool2x2": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_resize_bilinear_fwd": [c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_resize_bilinear_bwd": [c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_colsum": [c_i32, c_vp, c_i32, c_i64, c_i32, c_i32, c_f32, c_vp, c_vp],
    "bg_broadcast_rows": [c_i32, c_vp, c_f32, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp],
    "bg_cast_rows": [c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp],
    "bg_nchw_to_nhwc": [c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "bg_nhwc_to_nchw": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp],
    "bg_fill_f32": [c_vp, c_f32, c_i64, c_vp],
    "bg_axpy_rows": [c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp],
    "bg_linear_head_fwd": [c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "bg_linear_head_bwd": [c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "bg_bce_logits": [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp],
    "bg_l1_loss_fwd": [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp],
    "bg_l1_loss_bwd": [c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp],
    "bg_pixel_loss_fwd": [c_i32, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp],
    "bg_pixel_loss_bwd": [c_i32, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp],
    "bg_gp_penalty": [c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp],
    "bg_adam_step": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, C.c_double, C.c_double, c_f32, c_f32, c_i32, c_f32, c_f32,
                     c_f32, c_vp],
    "bg_set_floats": [c_vp, c_i32, c_f32, c_f32, c_f32, c_f32, c_vp],
    "bg_adam_step_dev": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, C.c_double, C.c_double, c_f32, c_f32, c_i32, c_vp],
    "bg_sumsq_f32": [c_vp, c_i64, c_f32, c_vp, c_vp],
    "bg_lamb_stage1": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_f32, C.c_double, C.c_double, c_i32, c_f32, c_f32, c_i32, c_f32, c_f32, c_f32,
                       c_vp, c_vp, c_vp],
    "bg_lamb_stage2": [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_f32, c_f32, c_i32, c_vp],
    "bg_cast_f32_to_bf16": [c_vp, c_vp, c_i64, c_vp],
}
EXPORTS = sorted(list(_SIGS) + list(_HOST_SIGS) + ["bg_last_error", "bg_conv_weight_kpad", "bg_conv_set_variant"])

_lib = None


def load():
    """Load libbgamd.so, failing loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"bias_gan_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no PyTorch/CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_i32
    for name, args in _HOST_SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_i32
    lib.bg_last_error.argtypes = []
    lib.bg_last_error.restype = C.c_char_p
    lib.bg_conv_weight_kpad.argtypes = [c_i32]
    lib.bg_conv_weight_kpad.restype = c_i32
    lib.bg_conv_set_variant.argtypes = [c_i32]
    lib.bg_conv_set_variant.restype = c_i32
    if lib.bg_abi_version() != ABI_VERSION:
        raise RuntimeError("bias_gan_amd: libbgamd.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def kpad(dtype: torch.dtype) -> int:
    """Reduction-dimension padding granule of the packed conv weight copies."""
    return load().bg_conv_weight_kpad(dt(dtype))


def conv_variant(v: int) -> None:
    """GEMM tile family of the conv launches: -1 heuristics, 0 classic tiles only, 2 fat tiles wherever legal."""
    if load().bg_conv_set_variant(int(v)) != 0:
        raise RuntimeError(f"bg_conv_set_variant({v}): {_lib.bg_last_error().decode()}")


def dt(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {dtype}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw handle of torch's current HIP stream on the current device (the launch path: ~0.3 us through the
    C accessor instead of ~9 us through the torch.cuda.Stream object)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# When set to a list, every call is bracketed by HIP events recorded on the stream the
# kernel is launched on (torch's current stream IS that stream): bench.py uses it for
# the per-kernel durations of its roofline object.  None (default) = no overhead.
PROFILE = None
PROFILE_SHAPES = None


def _conv_flops(desc) -> float:
    return 2.0 * desc.N * desc.Ho * desc.Wo * desc.Cout * desc.Cin * desc.KH * desc.KW


def _es(dtype_code) -> int:
    return 2 if dtype_code == BF16 else 4


def _alg_bytes(name, a) -> float:
    """Algorithmic HBM bytes of one launch of an HBM-bound entry point: every operand
    element read or written exactly once."""
    nn = lambda *idx: sum(1 for i in idx if a[i] is not None)  # noqa: E731
    if name == "bg_conv2d_bwd_weight_grouped":
        return float(a[2]) * (float(a[3]) * (a[4] + a[5]) * 2 + 4.0 * a[4] * a[5])
    if name == "bg_conv2d_bwd_weight_grouped_taps":   # x and dy once, fp32 dW once, per layer
        d = a[0]
        return float(a[2]) * (float(d.N) * d.H * d.W * (d.Cin + d.Cout) * 2 + 4.0 * d.Cout * d.Cin * d.KH * d.KW)
    if name == "bg_conv2d_fwd_fp8":       # fp8 bytes in, fp8 weights, bf16 out
        d = a[0]
        return float(d.N) * (d.H * d.W * d.Cin + 2 * d.Ho * d.Wo * d.Cout) + d.Cout * d.Cin * d.KH * d.KW
    if name == "bg_conv2d_bwd_data_fp8":  # fp8 dy in, fp8 weights, bf16 dx out
        d = a[0]
        return float(d.N) * (2 * d.H * d.W * d.Cin + d.Ho * d.Wo * d.Cout) + d.Cout * d.Cin * d.KH * d.KW
    if name == "bg_quant_fp8":            # source once, fp8 copy once
        return float(a[3]) * (a[4] * _es(a[0]) + a[7])
    if name.startswith("bg_conv2d"):     # every operand once: activation in, activation out, weights
        d = a[0]
        es = _es(d.dtype)
        wbytes = d.Cout * d.Cin * d.KH * d.KW * (4 if name in ("bg_conv2d_bwd_weight", "bg_conv2d_bwd_weight_ws") else es)
        return float(d.N) * (d.H * d.W * d.Cin + d.Ho * d.Wo * d.Cout) * es + wbytes
    if name == "bg_dwconv3x3_bwd_fused":      # dy in, x in, da out
        d = a[0]
        return float(d.N) * 3 * d.H * d.W * d.C * _es(d.dtype)
    if name == "bg_dwconv3x3_bwd_fork":       # dy in, a0 in, skip in, z in, gout out
        d = a[0]
        return float(d.N) * 5 * d.H * d.W * d.C * _es(d.dtype)
    if name == "bg_dwconv3x3_bwd_data_add":   # dy in, addend in, dx out
        d = a[0]
        return float(d.N) * 3 * d.H * d.W * d.C * _es(d.dtype)
    if name.startswith("bg_dwconv3x3"):
        d = a[0]
        return float(d.N) * (d.H * d.W + d.Ho * d.Wo) * d.C * _es(d.dtype)
    if name == "bg_norm_stats":
        return float(a[2]) * a[3] * _es(a[0])
    if name == "bg_norm_act_fwd":
        return float(a[9]) * a[10] * _es(a[0]) * (2 + nn(5))
    if name == "bg_norm_act_fwd_stats":
        return float(a[17]) * a[18] * _es(a[0]) * (2 + nn(13))
    if name == "bg_norm_act_bwd_reduce":
        return float(a[11]) * a[12] * _es(a[0]) * nn(1, 3, 5)
    if name == "bg_norm_act_bwd_apply":
        return float(a[14]) * a[15] * _es(a[0]) * nn(1, 3, 5, 10, 12)
    if name == "bg_norm_act_bwd_apply_stats_q8":     # + one byte per element for the e5m2 copy
        return float(a[20]) * a[21] * (_es(a[0]) * nn(1, 3, 5, 16, 18) + 1)
    if name == "bg_norm_act_bwd_apply_stats":
        return float(a[20]) * a[21] * _es(a[0]) * nn(1, 3, 5, 16, 18)
    if name == "bg_nchw_to_nhwc":
        return float(a[3]) * a[5] * (4 * a[4] + _es(a[0]) * a[6])
    if name == "bg_nhwc_to_nchw":
        return float(a[4]) * a[6] * a[5] * (4 + _es(a[0]))
    if name in ("bg_adam_step", "bg_adam_step_dev"):
        return float(a[5]) * (7 * 4 + (2 if a[4] is not None else 0))
    if name == "bg_resize_bilinear_fwd":
        return float(a[6]) * a[11] * (a[7] * a[8] * _es(a[0]) + a[9] * a[10] * _es(a[1]))
    if name == "bg_resize_bilinear_bwd":
        return float(a[6]) * a[11] * (a[7] * a[8] * _es(a[1]) + a[9] * a[10] * _es(a[0]))
    if name == "bg_axpy_rows":
        return float(a[5]) * a[6] * _es(a[0]) * 3
    return 0.0


def host_call(name, *args):
    """Call a host-side entry point (no stream appended); raise on a non-zero status.
    BG_E_IO (-3) raises IndexError when the message says 'file corruption' (the reference's
    test contract, tests/reader_test.py:192-209), RuntimeError otherwise."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.bg_last_error().decode()
        if "file corruption" in msg:
            raise IndexError(f"{name}: {msg}")
        raise RuntimeError(f"{name} failed ({rc}): {msg}")


_WS_BYTES = {}


def wgrad_ws_bytes(desc) -> int:
    """Workspace of bg_conv2d_bwd_weight_ws for this descriptor (cached per shape)."""
    key = (desc.dtype, desc.N, desc.H, desc.W, desc.Cin, desc.Ho, desc.Wo, desc.Cout, desc.KH, desc.KW, desc.stride, desc.pad, desc.dil,
           desc.ldx, desc.ldy)
    n = _WS_BYTES.get(key)
    if n is None:
        out = C.c_int64(0)
        host_call("bg_conv2d_bwd_weight_ws_bytes", C.byref(desc), C.cast(C.byref(out), c_vp))
        n = _WS_BYTES[key] = int(out.value)
    return n


_FN = {}


def call_on(raw_stream, name, *args):
    """Call an entry point on an explicit raw HIP stream handle (no change of torch's current stream)."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*args, raw_stream)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {_lib.bg_last_error().decode()}")


def call(name, *args):
    """Call an entry point on the current HIP stream; raise on a non-zero status."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    lib = _lib
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args, stream())
        e1.record()
        if name == "bg_conv2d_bwd_weight_grouped":
            flops = 2.0 * args[2] * args[3] * args[4] * args[5]
        elif name == "bg_conv2d_bwd_weight_grouped_taps":
            flops = _conv_flops(args[0]) * args[2]
        else:
            flops = _conv_flops(args[0]) if name.startswith("bg_conv2d") else 0.0
        PROFILE.append((name, flops, e0, e1, _alg_bytes(name, args)))
        if PROFILE_SHAPES is not None:      # bench.py --dump-launches: which layer a convolution launch was
            d = args[0]
            PROFILE_SHAPES.append(f"{d.N}x{d.H}x{d.W},{d.Cin}->{d.Cout},k{d.KH}s{d.stride}d{d.dil}" if isinstance(d, ConvDesc) else "-")
    else:
        rc = fn(*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.bg_last_error().decode()}")


_SIDE_STREAMS = {}
_OWNED_STREAMS = []     # raw handles from bg_stream_create, destroyed at interpreter exit


def _destroy_side_streams():
    while _OWNED_STREAMS:
        h = _OWNED_STREAMS.pop()
        try:
            host_call("bg_stream_destroy", c_vp(h))
        except Exception:   # noqa: BLE001 -- the runtime may already be tearing down
            pass
    _SIDE_STREAMS.clear()


import atexit as _atexit
_atexit.register(_destroy_side_streams)


def side_stream(device, tag: str):
    """The package's dedicated side stream `tag` on `device` (created once, never shared with PyTorch's stream pool -- see
    bg_stream_create): a torch.cuda.ExternalStream over a HIP stream the library owns."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, tag)
    s = _SIDE_STREAMS.get(key)
    if s is None and os.environ.get("BGAMD_POOL_STREAMS"):   # A/B switch only: PyTorch's pool streams (may alias each other)
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", idx))
    if s is None:
        out = c_vp()
        with torch.cuda.device(idx):
            host_call("bg_stream_create", C.byref(out))
        _OWNED_STREAMS.append(out.value)
        s = _SIDE_STREAMS[key] = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", idx))
    return s
