"""ctypes binding of libtricolo_hip.so (C ABI declared in include/tricolo_hip.h).

The product path has NO CPU fallback: if the library is missing, or a tensor is not on a GPU, the call raises.
Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C tricolo_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtricolo_hip.so")


class TriConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "ID", "IH", "IW", "Cin", "OD", "OH", "OW", "Cout",
                                       "KD", "KH", "KW", "stride", "pad_d", "pad_h", "pad_w")]


class TriPrepDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("hi", C.c_void_p), ("lo", C.c_void_p), ("s_row", C.c_long), ("s_tap", C.c_long),
                ("s_inner", C.c_long), ("rows", C.c_int), ("ntaps", C.c_int), ("inner", C.c_int), ("inner_pad", C.c_int),
                ("kpad", C.c_int), ("fmt", C.c_int), ("frag", C.c_int)]


class TriWgradReduce(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("dw", C.c_void_p), ("s_co", C.c_long), ("s_tap", C.c_long), ("s_ci", C.c_long)] + \
               [(n, C.c_int) for n in ("splits", "Cout", "Kpad", "ntaps", "cin_stored", "cin_real", "zlanes", "nblocks")] + \
               [("out_scale", C.c_float), ("kw_real", C.c_int), ("kw_shift", C.c_int)]


class TriWgradJob(C.Structure):
    _fields_ = [("d", C.POINTER(TriConvDesc)), ("inp", C.c_void_p), ("dout", C.c_void_p), ("plan", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("dw", C.c_void_p), ("s_co", C.c_long), ("s_tap", C.c_long), ("s_ci", C.c_long),
                ("cin_real", C.c_int), ("out_scale", C.c_float), ("row_pos", C.c_void_p), ("row_count", C.c_void_p)]


class TriConvBnSums(C.Structure):
    _fields_ = [("y", C.c_void_p), ("relu_scale", C.c_void_p), ("relu_shift", C.c_void_p), ("relu_out", C.c_void_p), ("partial", C.c_void_p)]


TRI_WGRAD_JOBS_MAX = 12
TRI_ERR_ARG, TRI_ERR_UNSUPPORTED = -1, -2                   # common.h

P, I, L, F, Z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t
DP = C.POINTER(TriConvDesc)

# name -> (restype, argtypes); must list every symbol of include/tricolo_hip.h (checked by tests/test_abi.py)
SIGNATURES = {
    "tri_version": (I, []),
    "tri_last_error": (C.c_char_p, []),
    "tri_conv_kpad": (I, [I, I]),
    "tri_weight_prep": (I, [P, L, L, L, I, I, I, I, P, P, I, I, P]),
    "tri_weight_prep_multi": (I, [P, I, P]),
    "tri_embedding_fwd": (I, [P, P, I, I, I, P, P]),
    "tri_embedding_bwd": (I, [P, P, I, I, I, I, I, P, P]),
    "tri_retrieval_topk": (I, [P, P, P, I, I, I, I, P, P, P, P]),
    "tri_linear_small_supported": (I, [I, I, I]),
    "tri_linear_small_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "tri_linear_small_dgrad": (I, [P, P, P, P, I, I, I, I, I, P]),
    "tri_linear_small_wgrad": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "tri_linear_small_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "tri_conv_num_mtiles": (I, [DP, I]),
    "tri_conv_num_records": (I, [DP, I, I]),
    "tri_conv_workspace": (Z, [DP, I]),
    "tri_conv_kernel_family": (I, [DP, I, I]),
    "tri_conv_wgrad_kernel_family": (I, [DP, I]),
    "tri_conv_fwd": (I, [DP, P, P, P, P, P, P, I, I, P, I, P, Z, P, P, P]),
    "tri_conv_dgrad": (I, [DP, P, P, P, P, P, I, I, P, Z, P, P, P]),
    "tri_conv_dgrad_bn_records": (I, [DP, I, I]),
    "tri_conv_dgrad_bn": (I, [DP, P, P, P, P, I, I, P, Z, P, C.POINTER(TriConvBnSums), P]),
    "tri_conv_wgrad_workspace": (Z, [DP]),
    "tri_conv_plan_bytes": (Z, [DP]),
    "tri_conv_plan_build": (I, [DP, P, P]),
    "tri_conv_wgrad": (I, [DP, P, P, P, P, P, Z, P, L, L, L, I, I, I, F, P, P, P]),
    "tri_conv_wgrad_partial": (I, [DP, P, P, P, P, P, Z, P, L, L, L, I, I, I, F, P, P, P, P]),
    "tri_wgrad_reduce_grouped": (I, [P, I, P]),
    "tri_wgrad_reduce_grouped_noted": (I, [P, I, P, P]),
    "tri_conv_stem_wgrad_bn": (I, [DP, P, P, P, P, P, P, P, P, P, P, Z, P, L, L, L, I, I, F, P, P]),
    "tri_conv_wgrad_group_info": (I, [DP, I, P, P, P]),
    "tri_conv_wgrad_partial_group": (I, [P, I, I, P, P]),
    "tri_bn_finalize": (I, [P, I, I, P, I, P, P, P, P, P, F, F, P, P, P, P, P]),
    "tri_bn_eval_coeffs": (I, [I, P, P, P, P, F, P, P, P, P, P]),
    "tri_bn_act": (I, [P, P, P, P, P, P, P, L, I, I, I, P]),
    "tri_relu_bwd": (I, [P, P, P, L, I, P]),
    "tri_bn_bwd_num_blocks": (I, [L]),
    "tri_bn_bwd_reduce": (I, [P, P, L, I, P, P, P, P, P, I, P]),
    "tri_bn_bwd_finalize": (I, [P, I, I, P, I, P, P, P, P, P, P, P, P, F, P]),
    "tri_bn_bwd_small": (I, [P, P, L, I, P, I, P, P, P, P, P, P, P, P, I, P, P, P, F, I, P]),
    "tri_bn_bwd_apply": (I, [P, P, P, P, P, P, P, L, I, P, P, P, P, I, I, P]),
    "tri_bn_bwd_pair_reduce": (I, [P, P, P, P, L, I, P, P, I, P]),
    "tri_bn_bwd_pair_finalize": (I, [P, P, I, I, I, P, P, P, P, P, P, P, P, F, P]),
    "tri_bn_bwd_pair_apply": (I, [P, P, P, P, P, P, P, P, P, L, I, I, P]),
    "tri_bn_relu_pool3d_fwd": (I, [P, P, P, P, I, I, I, P, P, I, P]),
    "tri_pool3d_bwd_route": (I, [P, P, P, P, P, P, I, I, I, P, I, P]),
    "tri_pool3d_bwd_route_rows": (I, [P, P, P, P, P, P, I, I, I, P, P, P, I, P]),
    "tri_pool3d_bwd_route_rows_num_blocks": (I, [I, I, I]),
    "tri_pool3d_bwd_route_rows_reduce": (I, [P, P, P, P, P, P, I, I, I, P, P, P, P, I, P]),
    "tri_bn_bwd_apply_rows": (I, [P, P, P, P, P, P, I, P, P, L, I, P]),
    "tri_bn_bwd_rows_scratch": (C.c_size_t, [I]),
    "tri_bn_bwd_rows": (I, [P, P, I, P, P, L, P, P, P, P, P, P, F, P, I, P]),
    "tri_pool3d_bwd_route_reduce_num_blocks": (I, [I, I, I]),
    "tri_pool3d_bwd_route_reduce": (I, [P, P, P, P, P, P, I, I, I, P, P, I, P]),
    "tri_maxpool2d_fwd": (I, [P, I, I, I, I, P, P, P, P, I, P]),
    "tri_maxpool2d_bwd": (I, [P, P, I, I, I, I, P, I, P]),
    "tri_maxpool_bn_bwd_num_blocks": (I, [I, I, I]),
    "tri_maxpool_bn_bwd_reduce": (I, [P, P, P, I, I, I, I, P, P, P, I, P]),
    "tri_maxpool_bn_bwd_pooled_num_blocks": (I, [I, I, I]),
    "tri_maxpool_bn_bwd_reduce_pooled": (I, [P, P, P, P, I, I, I, I, P, P, P, P, I, P]),
    "tri_maxpool_bn_bwd_apply": (I, [P, P, P, I, I, I, I, P, P, P, P, P, P, I, P]),
    "tri_avgpool_viewmax_fwd": (I, [P, I, I, I, I, P, P, I, P]),
    "tri_avgpool_viewmax_bwd": (I, [P, P, I, I, I, I, P, I, F, P]),
    "tri_voxel_scatter": (I, [P, P, I, I, I, P, P, I, P]),
    "tri_voxel_from_rgba_u8": (I, [P, I, I, P, P, I, P]),
    "tri_nchw3_u8_to_nhwc4": (I, [P, I, I, I, P, P, P, I, P]),
    "tri_mask_count": (I, [P, L, P, P]),
    "tri_debug_stamp": (I, [P, P]),
    "tri_debug_buffer_b128_probe": (I, [P, P, L, I, P]),
    "tri_mask_compact_scratch": (Z, [L]),
    "tri_mask_compact": (I, [P, L, P, P, P, P]),
    "tri_mask_pyramid": (I, [P, I, I, P, P]),
    "tri_mask_compact_multi_scratch": (Z, [P, I]),
    "tri_mask_compact_multi": (I, [P, P, I, P, P, P, P]),
    "tri_nchw3_to_nhwc4": (I, [P, I, I, I, P, I, P]),
    "tri_l2norm_fwd": (I, [P, I, I, F, P, P, P]),
    "tri_l2norm_bwd": (I, [P, P, P, I, I, F, P, P]),
    "tri_colsum": (I, [P, L, I, P, P]),
    "tri_axpy": (I, [P, F, P, L, P]),
    "tri_act_bwd": (I, [P, P, P, L, I, P]),
    "tri_cast_from_f32": (I, [P, P, L, F, I, P]),
    "tri_cast_to_f32": (I, [P, P, L, I, P]),
    "tri_gru_fwd": (I, [P, P, P, I, I, P, P, P, I, P]),
    "tri_gru_bwd": (I, [P, P, P, P, I, I, P, P, P, P, I, P]),
    "tri_ntxent_workspace": (Z, [I, I]),
    "tri_ntxent_fwd_bwd": (I, [P, P, I, I, F, F, I, P, P, P, P, Z, P]),
    "tri_ntxent_bwd": (I, [P, P, I, I, F, F, I, P, P, P, P, Z, P]),
    "tri_ntxent_multi_workspace": (Z, [I, I, I]),
    "tri_ntxent_multi_fwd": (I, [P, I, I, I, F, F, I, P, P, Z, P]),
    "tri_ntxent_multi_bwd": (I, [P, I, I, I, F, F, I, P, P, P, P, Z, P]),
    "tri_copy_segments": (I, [P, P, P, I, P]),
    "tri_gru_bias_grads": (I, [P, I, P, P, P, P, P]),
    "tri_adam_tick": (I, [P, P]),
    "tri_adam_guard": (I, [P, L, P, P]),
    "tri_adam_guard_segments": (I, [P, P, I, L, P, P]),
    "tri_adam_step": (I, [P, P, P, P, L, P, F, P, F, F, F, F, F, P]),
    "tri_adam_step_segments": (I, [P, P, P, I, P, P, L, P, F, P, F, F, F, F, F, P]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built.  Run `make -C tricolo_amd/csrc` "
                "(or __graft_entry__.build()).  tricolo_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class _TimedLib:
    """Diagnostic proxy (tools/kernel_times.py): every tri_* call that takes a stream is bracketed by two HIP events on the current
    stream; `records` collects (entry point, start, end).  Never installed by the product path."""

    def __init__(self, real):
        self._real, self.records = real, []

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        sig = SIGNATURES.get(name)
        if sig is None or not name.startswith("tri_") or not sig[1] or sig[0] is not I or name in ("tri_last_error",):
            return fn
        if "workspace" in name or "num_" in name or "_info" in name or "kpad" in name or "family" in name or "bytes" in name or "supported" in name or "scratch" in name or "records" in name:
            return fn

        shape_args = {"tri_bn_bwd_reduce": (2, 3), "tri_bn_bwd_apply": (7, 8), "tri_bn_act": (7, 8), "tri_bn_finalize": (1, 2),
                      "tri_bn_bwd_finalize": (1, 2), "tri_bn_bwd_small": (2, 3)}.get(name)

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            key = name if shape_args is None else f"{name}[{int(a[shape_args[0]])}x{int(a[shape_args[1]])}]"
            self.records.append((key, e0, e1))
            return rc
        return timed


def install_timing_proxy():
    """Replace the library handle by a _TimedLib (diagnostics only); returns it."""
    global _lib
    real = lib()
    if isinstance(real, _TimedLib):
        return real
    _lib = _TimedLib(real)
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().tri_last_error().decode(errors="replace")
        raise RuntimeError(f"libtricolo_hip {what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: there is no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("tricolo_amd: tensor is not on a GPU; the HIP path has no CPU fallback")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def make_desc(B, ID, IH, IW, Cin, OD, OH, OW, Cout, KD, KH, KW, stride, pad_d, pad_h, pad_w) -> TriConvDesc:
    return TriConvDesc(B, ID, IH, IW, Cin, OD, OH, OW, Cout, KD, KH, KW, stride, pad_d, pad_h, pad_w)
