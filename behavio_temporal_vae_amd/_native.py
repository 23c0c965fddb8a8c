"""ctypes binding of ``libbtvae_hip.so`` (the C-ABI declared in ``include/btvae_hip.h``).

There is no CPU fallback: if the shared library is missing or a call fails, a ``RuntimeError`` is
raised.  torch is used only to own device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_longlong, c_ulonglong, c_void_p

import torch

# BTVAE_HIP_LIB points at another build of the same library (kernel experiments); default is the in-tree build
_LIB_PATH = os.environ.get("BTVAE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbtvae_hip.so")
_lib = None
_inited = set()

P, I, F, LL, ULL = c_void_p, c_int, c_float, c_longlong, c_ulonglong

# name -> argument types (every function returns int status unless listed in _RET)
_SIGS = {
    "btvae_init": [I],
    "btvae_prof_enable": [I],
    "btvae_prof_filter": [ctypes.c_uint, ctypes.c_double],
    "btvae_prof_collect": [P, P, P, P, I],
    "btvae_prof_dominant": [I, P, P, P],
    "btvae_set_gemm_precision": [I],
    "btvae_linear_relu_stats_fwd": [P, I, P, P, P, P, I, I, I, P],
    "btvae_linear_relu_stats_fwd_pw": [P, I, P, P, P, P, P, I, I, I, P],
    "btvae_linear_relu_affine_fwd": [P, I, P, P, P, P, P, P, I, I, I, P],
    "btvae_linear_relu_affine_fwd_pw": [P, I, P, P, P, P, P, P, P, I, I, I, P],
    "btvae_linear_relu_affine_fwd_pw_ld": [P, I, P, P, P, P, P, P, I, P, I, I, I, I, P],
    "btvae_weight_planes": [P, I, I, I, P, I, P],
    "btvae_weight_planes_multi": [P, P, P, P, P, I, P],
    "btvae_weight_planes_f16": [P, I, I, I, P, I, P, P],
    "btvae_weight_planes_f16_multi": [P, P, P, P, P, P, I, P],
    "btvae_absmax": [P, LL, P, P],
    "btvae_slots_zero": [P, I, P],
    "btvae_guard_check": [P, P, I, P],
    "btvae_weight_image_bf16": [P, I, I, I, I, P, P],
    "btvae_pairs_encode": [P, I, P, I, P, I, I, P],
    "btvae_linear_relu_stats_fwd_pairs": [P, I, P, P, P, P, P, I, P, I, I, I, P],
    "btvae_linear_wgrad_pairs": [P, P, I, P, P, P, LL, I, I, I, I, P],
    "btvae_bn_apply_fwd_pairs": [P, I, P, P, P, P, P, I, P, P, I, P, I, I, P],
    "btvae_pairs_decode": [P, I, P, P, I, I, I, P],
    "btvae_weight_image_pq": [P, I, I, I, I, P, P, P],
    "btvae_linear_relu_affine_fwd_pairs": [P, I, P, P, P, P, P, P, P, I, P, I, P, I, I, I, P],
    "btvae_weight_image_bf16_multi": [P, P, P, P, P, I, P],
    "btvae_linear_head_fwd": [P, I, P, P, I, F, F, P, P, P, I, I, I, P],
    "btvae_linear_head_fwd_ws": [P, I, P, P, I, F, F, P, P, P, P, LL, I, I, I, P],
    "btvae_linear_dgrad": [P, P, P, P, I, I, I, I, P],
    "btvae_linear_dgrad_pw": [P, P, P, P, P, I, I, I, I, P],
    "btvae_linear_wgrad": [P, P, I, P, P, LL, I, I, I, I, P],
    "btvae_matmul_tn": [P, I, P, I, P, P, LL, I, I, I, I, I, P],
    "btvae_matmul_nt_bias": [P, I, P, I, P, P, I, I, I, I, P],
    "btvae_matmul_nt_add": [P, I, P, I, P, I, P, I, I, I, I, P],
    "btvae_matmul_nn_mul": [P, I, P, I, P, P, I, P, I, I, I, I, P],
    "btvae_bn_stats_finalize": [P, I, I, I, P, P, P, P, F, F, P],
    "btvae_bn_fold_eval": [P, P, P, P, F, P, P, I, P],
    "btvae_bn_apply_fwd": [P, P, P, P, P, P, P, F, P, I, I, P],
    "btvae_bn_relu_bwd": [P, P, P, P, P, P, F, P, P, P, P, P, P, I, I, I, P],
    "btvae_relu_bwd": [P, P, P, F, P, P, P, P, I, I, I, P],
    "btvae_bn_fold_linear": [P, P, P, P, P, P, P, P, P, P, I, I, P],
    "btvae_linear_wgrad_affine": [P, P, I, P, P, P, P, P, P, I, P, P, LL, I, I, I, I, P],
    "btvae_bn_bwd_sums_from_products": [P, P, P, P, P, P, P, I, I, P],
    "btvae_linear_dgrad_bnbwd": [P, P, P, P, P, P, P, P, LL, P, P, I, I, I, P],
    "btvae_linear_dgrad_bnbwd_pw": [P, P, P, P, P, P, P, P, P, LL, P, P, I, I, I, P],
    "btvae_colsum_partials": [P, I, I, P, P, I, P],
    "btvae_linear_dgrad_gstats": [P, P, P, P, P, P, P, P, I, I, I, P],
    "btvae_linear_dgrad_gstats_pw": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "btvae_bn_bwd_sums_from_partials": [P, I, I, P, P, P, I, P],
    "btvae_bn_stats_local": [P, I, I, I, P, P],
    "btvae_bn_stats_merge": [P, I, I, I, P, P, P, P, F, F, P],
    "btvae_bn_relu_bwd_sums": [P, P, P, P, P, F, P, P, P, P, I, I, I, P],
    "btvae_bn_relu_bwd_apply": [P, P, P, P, P, P, F, P, LL, P, P, P, P, I, I, I, P],
    "btvae_dropout_keep_mask": [P, LL, F, ULL, ULL, P],
    "btvae_sample_error_update": [P, P, P, P, I, I, I, I, P],
    "btvae_dirichlet_sample": [P, P, P, I, I, I, I, I, ULL, ULL, P],
    "btvae_colsum": [P, P, P, P, I, I, I, P],
    "btvae_colsum_noalloc": [P, P, P, I, I, I, P],
    "btvae_reparam_kl_fwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, ULL, ULL, P],
    "btvae_reparam_kl_bwd": [P, P, P, P, P, P, P, P, I, I, I, I, F, P],
    "btvae_diag_gauss_loglik": [P, P, P, P, P, P, P, P, P, F, I, I, I, P],
    "btvae_dict_prepare": [P, P, P, P, P, I, I, F, P],
    "btvae_dict_grad": [P, P, P, P, I, I, I, P],
    "btvae_mvn_rows_fwd": [P, P, P, P, I, I, I, P],
    "btvae_mvn_rows_fwd_bwd": [P, P, P, P, P, F, P, P, P, I, I, I, P],
    "btvae_elbo_scalars": [P, I, P, I, F, P, P],
    "btvae_iw_lse_update": [P, P, P, P, P, P, I, I, I, I, P],
    "btvae_iw_lse_finalize": [P, P, P, I, I, P],
    "btvae_grad_sumsq": [P, LL, P, P],
    "btvae_clip_adam_step": [P, P, P, P, LL, P, F, F, F, F, F, F, LL, P, P],
    "btvae_scale": [P, LL, F, P],
    "btvae_adam_tick": [P, P, I, P],
    "btvae_clip_adam_step_dev": [P, P, P, P, LL, P, P, P, P],
    "btvae_cond_mvn_sample": [P, P, P, P, ULL, P, P, P, P, I, I, I, P],
    "btvae_cond_mvn_sample_wide": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "btvae_masked_scatter": [P, P, P, LL, P],
    "btvae_lse_over_k": [P, P, I, I, P],
    "btvae_pack_lower": [P, P, LL, I, P],
}
_RET = {
    "btvae_get_gemm_precision": (c_int, []),
    "btvae_last_error": (ctypes.c_char_p, []),
    "btvae_version": (c_int, []),
    "btvae_strip_chunks": (c_int, [I]),
    "btvae_bn_relu_bwd_scratch": (c_longlong, [I, I]),
    "btvae_weight_planes_elems": (c_longlong, [I, I]),
    "btvae_weight_image_bf16_bytes": (c_longlong, [I, I, I]),
    "btvae_weight_image_pq_bytes": (c_longlong, [I, I, I]),
    "btvae_next_bf16": (c_int, [I]),
    "btvae_colsum_noalloc_max_floats": (c_longlong, []),
    "btvae_prof_enabled": (c_int, []),
    "btvae_gemm_operand_maxima": (c_int, [P, P]),
    "btvae_output_maximum": (c_int, [P]),
    "btvae_gemm_guard_site": (c_int, [I]),
    "btvae_guard_sites": (c_int, []),
    "btvae_guard_range_log2": (c_int, [I]),
    "btvae_guard_flags": (ctypes.POINTER(c_int), []),
    "btvae_next_row_pitches": (c_int, [I, I]),
}
EXPORTS = sorted(list(_SIGS) + list(_RET))
PROF_KINDS = ["gemm_fwd", "gemm_dgrad", "gemm_wgrad", "gemm_mvn", "bn", "mvn_rows", "reduce", "optim"]


def lib():
    """The loaded shared library; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"HIP extension not built: {_LIB_PATH} is missing (run __graft_entry__.build())")
        L = ctypes.CDLL(_LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_int
        for name, (ret, args) in _RET.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ret
        _lib = L
    return _lib


_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def last_error() -> str:
    return lib().btvae_last_error().decode("utf-8", "replace")


def ensure_device(device: torch.device):
    """Set the library up on ``device`` (checks that it is a gfx950 part) once per device index, and require that it
    is torch's current device: every library call launches on the current device's current stream, so tensors of
    another device would be read through the wrong device's queue."""
    if device.type != "cuda":
        raise RuntimeError(f"behavio_temporal_vae_amd computes on MI355X only; got tensors on '{device}'")
    cur = _cur_device() if _cur_device is not None else torch.cuda.current_device()
    idx = device.index if device.index is not None else cur
    if idx != cur:
        raise RuntimeError(f"tensors are on cuda:{idx} but the current device is cuda:{cur}; wrap the call in "
                           f"`with torch.cuda.device({idx}):` or call torch.cuda.set_device({idx})")
    if idx not in _inited:
        rc = lib().btvae_init(idx)
        if rc != 0:
            raise RuntimeError(last_error())
        _inited.add(idx)
    return idx


def ptr(t):
    """Device pointer of a contiguous float32/int tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise RuntimeError("non-contiguous tensor passed to the HIP library")
    return c_void_p(t.data_ptr())


def ptr_rows(t):
    """Device pointer of a 2-D tensor whose rows are contiguous but may be a pitch apart (``t.stride(0)`` floats); the
    callee is given that pitch as the leading dimension."""
    if t is None:
        return None
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        raise RuntimeError("tensor passed to the HIP library is not a matrix of contiguous rows")
    return c_void_p(t.data_ptr())


# torch.cuda.current_stream() builds a Stream object per call (several microseconds); a training step makes ~120
# library calls, so the raw handle is read through the same C entry points torch itself uses, when they exist.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    if _raw_stream is not None and _cur_device is not None:
        return c_void_p(_raw_stream(_cur_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name: str, *args):
    """Call a status-returning entry point on the current torch stream; raise on failure."""
    rc = getattr(lib(), name)(*args, stream_ptr())
    if rc != 0:
        msg = last_error()
        # a request meant for the launch that did not happen must not reach a later one
        lib().btvae_gemm_operand_maxima(None, None)
        lib().btvae_output_maximum(None)
        lib().btvae_gemm_guard_site(-1)
        lib().btvae_next_bf16(0)
        lib().btvae_next_row_pitches(0, 0)
        raise RuntimeError(f"{name}: {msg}")


def f32(t, name="tensor"):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


GEMM_PRECISIONS = {"fp32": 0, "bf16": 1, "fp32_split": 2, "fp32_split2": 3}


def set_gemm_precision(name: str):
    """Operand precision of the Linear-layer products, process-wide: ``"fp32_split2"`` (what models ask for unless
    configured otherwise: the Linear products whose callers supply their operands' absolute maxima run on two fp16
    pieces per value, scaled per tensor -- three matrix instructions per product, element errors ~2^-22 relative to the
    tensor's largest magnitude; the others as ``"fp32_split"``), ``"fp32_split"`` (fp32-accurate products as exact
    three-way bf16 splits on the bf16 matrix pipe, six matrix instructions), ``"fp32"`` (the fp32 MFMA instruction; the
    library's own start-up mode) or ``"bf16"`` (bf16-rounded operands, fp32 accumulation)."""
    if name not in GEMM_PRECISIONS:
        raise ValueError(f"unknown GEMM precision {name!r}; choose from {sorted(GEMM_PRECISIONS)}")
    if lib().btvae_set_gemm_precision(GEMM_PRECISIONS[name]) != 0:
        raise RuntimeError(last_error())


def get_gemm_precision() -> str:
    mode = lib().btvae_get_gemm_precision()
    return next(k for k, v in GEMM_PRECISIONS.items() if v == mode)


def prof_enable(on: bool):
    lib().btvae_prof_enable(1 if on else 0)


def prof_filter(kinds=None, min_flops: float = 0.0):
    """Time only the named families (None = all) and launches of at least ``min_flops`` FLOPs: every timed launch
    costs two event records on its stream."""
    mask = 0xFFFFFFFF if kinds is None else sum(1 << PROF_KINDS.index(k) for k in kinds)
    lib().btvae_prof_filter(mask, float(min_flops))


def prof_dominant(kind: str, operands: str = "any"):
    """(ms summed, flops of one launch, launches) of the launches of ``kind`` with the largest FLOP count;
    call after a device sync and before ``prof_collect``.  ``operands``: "any", or only the launches on "plain" (fp32 / bf16)
    operands, or only those with an operand in the "pairs" format (one shape can run on two kernels)."""
    ms, fl, cnt = c_double(), c_double(), c_longlong()
    sel = ("any", "plain", "pairs").index(operands)
    rc = lib().btvae_prof_dominant(PROF_KINDS.index(kind) + 256 * sel, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt))
    if rc != 0:
        raise RuntimeError(last_error())
    return ms.value, fl.value, cnt.value


def prof_collect():
    """{family: (ms, flops, bytes, launches)} since ``prof_enable(True)``; call after a device sync."""
    n = len(PROF_KINDS)
    ms, fl, by = (c_double * n)(), (c_double * n)(), (c_double * n)()
    cnt = (c_longlong * n)()
    rc = lib().btvae_prof_collect(ms, fl, by, cnt, n)
    if rc != 0:
        raise RuntimeError(last_error())
    return {k: (ms[i], fl[i], by[i], cnt[i]) for i, k in enumerate(PROF_KINDS)}
