"""ctypes loader for libsrad.so (the HIP engine).  There is NO CPU fallback: if the library is
missing or a call fails, the product path raises."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRAD_LIB_PATH: a differently built libsrad (an A/B build of tools/), else the in-tree library
LIB_PATH = os.environ.get("SRAD_LIB_PATH") or os.path.join(_HERE, "libsrad.so")

PREC_F32 = 0
PREC_BF16 = 1
PREC_BF16X3 = 2      # split-bf16: operands as hi + lo bf16 terms, three bf16 MFMAs per product, fp32-grade outputs (inference)
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "float32": PREC_F32, "bf16": PREC_BF16, "bfloat16": PREC_BF16,
              "bf16x3": PREC_BF16X3, "split-bf16": PREC_BF16X3}

ACT_NONE, ACT_GELU, ACT_LRELU, ACT_RELU = 0, 1, 2, 3


class DrctConfig(C.Structure):
    _fields_ = [("in_chans", C.c_int32), ("img_size", C.c_int32), ("window_size", C.c_int32),
                ("upscale", C.c_int32), ("embed_dim", C.c_int32), ("n_rdg", C.c_int32),
                ("num_heads", C.c_int32), ("gc", C.c_int32), ("num_feat", C.c_int32),
                ("mlp_ratio", C.c_float), ("img_range", C.c_float), ("precision", C.c_int32),
                ("use_graph", C.c_int32)]


class DrnConfig(C.Structure):
    _fields_ = [("n_colors", C.c_int32), ("scale", C.c_int32), ("n_blocks", C.c_int32),
                ("n_feats", C.c_int32), ("negval", C.c_float), ("rgb_range", C.c_float),
                ("precision", C.c_int32), ("use_graph", C.c_int32)]


_lock = threading.Lock()
_lib = None

_P = C.c_void_p
_SIG = {
    "srad_last_error": (C.c_char_p, []),
    "srad_version": (C.c_int, []),
    # DRCT
    "srad_drct_create": (C.c_int, [C.POINTER(DrctConfig), C.POINTER(_P)]),
    "srad_drct_destroy": (None, [_P]),
    "srad_drct_arena_bytes": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "srad_drct_bind_arena": (C.c_int, [_P, _P, C.c_size_t]),
    "srad_drct_num_params": (C.c_int, [_P]),
    "srad_drct_param_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    "srad_drct_set_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "srad_drct_workspace_bytes": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "srad_drct_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "srad_drct_flops": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    # DRCT training
    "srad_drct_train_param_floats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "srad_drct_train_param_offset": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "srad_drct_train_arena_bytes": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "srad_drct_train_bind": (C.c_int, [_P, _P, C.c_size_t]),
    "srad_drct_sync_params": (C.c_int, [_P, _P, _P]),
    "srad_drct_train_workspace_bytes": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "srad_drct_forward_train": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "srad_drct_num_buckets": (C.c_int, [_P]),
    "srad_drct_bucket_range": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "srad_drct_backward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_size_t, _P, _P, _P]),
    "srad_l1_grad": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, _P]),
    "srad_adam_step": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_int, C.c_float, _P]),
    "srad_adam_step_dev": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _P, _P]),
    "srad_set4": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    # DRN
    "srad_drn_create": (C.c_int, [C.POINTER(DrnConfig), C.POINTER(_P)]),
    "srad_drn_destroy": (None, [_P]),
    "srad_drn_arena_bytes": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "srad_drn_bind_arena": (C.c_int, [_P, _P, C.c_size_t]),
    "srad_drn_num_params": (C.c_int, [_P]),
    "srad_drn_param_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    "srad_drn_set_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "srad_drn_workspace_bytes": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "srad_drn_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(_P), C.c_int, _P, C.c_size_t, _P]),
    "srad_drn_flops": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "srad_drn_train_param_floats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "srad_drn_train_param_offset": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "srad_drn_train_arena_bytes": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "srad_drn_train_bind": (C.c_int, [_P, _P, C.c_size_t]),
    "srad_drn_sync_params": (C.c_int, [_P, _P, _P]),
    "srad_drn_train_workspace_bytes": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "srad_drn_forward_train": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(_P), C.c_int, _P, C.c_size_t, _P]),
    "srad_drn_num_buckets": (C.c_int, [_P]),
    "srad_drn_bucket_range": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "srad_drn_backward": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P, _P, _P]),
    "srad_dual_backward_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "srad_dual_backward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P,
                                     C.c_size_t, C.c_int, _P]),
    "srad_dual_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "srad_dual_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, _P, C.c_int, C.c_int, C.c_int, _P, _P,
                                    C.c_size_t, C.c_int, _P]),
    # scorer
    "srad_to_u8_hwc": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P]),
    "srad_quantize": (C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    "srad_score_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "srad_score_pairs": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int,
                                   _P, _P, _P, _P, C.c_size_t, _P]),
    "srad_val_metrics": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, C.c_size_t, _P]),
    "srad_roc_auc": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double)]),
    "srad_l1_workspace_bytes": (C.c_int, [C.POINTER(C.c_size_t)]),
    "srad_l1_loss": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P]),
    "srad_loss_workspace_bytes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "srad_loss_forward": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P,
                                    _P, C.c_size_t, _P]),
    "srad_loss_backward": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P,
                                     _P, C.c_float, _P, C.c_int, _P, C.c_size_t, _P]),
    # event profiler
    "srad_prof_enable": (C.c_int, [C.c_int]),
    "srad_prof_num_classes": (C.c_int, []),
    "srad_prof_class_name": (C.c_char_p, [C.c_int]),
    "srad_prof_collect": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]),
    # single operators
    "srad_op_gemm": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                               _P, _P, _P, C.c_int, C.c_float, C.c_float, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                               _P, C.c_size_t, _P]),
    "srad_bench_gemm": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                  _P, _P, _P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, C.c_size_t,
                                  C.c_int, C.POINTER(C.c_float), _P]),
    "srad_bench_window_attn": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    "srad_bench_mlp_block": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, C.c_int, C.c_int,
                                       C.POINTER(C.c_float), _P]),
    "srad_bench_qkv_attn": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_size_t,
                                      C.c_int, C.POINTER(C.c_float), _P]),
    "srad_op_swin_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "srad_op_qkv_attn": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P,
                                   C.c_size_t, _P]),
    "srad_op_mlp_block": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P,
                                    _P, _P, C.c_int, C.c_float, C.c_float, _P, C.c_int, _P, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "srad_op_gemm_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "srad_op_window_attn": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, _P]),
    "srad_op_layernorm": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "srad_op_window_attn_qscale": (C.c_float, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "srad_op_ln_qkv_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "srad_op_ln_qkv": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_float, _P, C.c_size_t, _P]),
    "srad_op_window_attn_bf16_in": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    # backward operators
    "srad_op_wgrad": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, _P, C.c_float, _P, _P, _P, _P]),
    "srad_op_wgrad_workspace_bytes": (C.c_size_t, []),
    "srad_op_dgrad": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P,
                                C.c_int, C.c_int, C.c_float, C.c_float, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "srad_op_mlp_bwd_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "srad_op_mlp_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P,
                                  C.c_int, _P, C.c_int, _P, C.c_int, C.c_float, C.c_float, _P, _P, _P, _P, _P,
                                  _P, C.c_size_t, _P, _P]),
    "srad_op_lin_ln_bwd_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "srad_op_lin_ln_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, _P,
                                     _P, C.c_size_t, _P, _P]),
    "srad_op_layernorm_bwd": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_int, _P, _P]),
    "srad_op_window_attn_bwd": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, _P, _P]),
    "srad_op_conv80_h": (C.c_int, [_P, _P, _P, C.c_int, C.c_float, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_size_t, _P]),
    "srad_op_wgrad_conv9_h": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "srad_op_wgrad_deferred": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int,
                                         C.c_float, _P, _P, _P, _P]),
    "srad_bench_wgrad_block": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, _P]),
    "srad_op_window_attn_bwd_h": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, _P, _P]),
}

BUCKET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


def exported_symbols():
    """Names every entry point include/srad.h declares (checked by the CPU test-suite)."""
    return sorted(_SIG)


def lib() -> C.CDLL:
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"libsrad.so not found at {LIB_PATH}: build it with "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                    "There is no CPU fallback for the product path.")
            # PyTorch ships its own libamdhip64 (torch/lib) and libsrad.so names /opt/rocm's by SONAME: whichever is loaded
            # first serves both.  Load torch's first - with two HIP runtimes in one process the engine's launches fail
            # with "no ROCm-capable device is detected" (seen when build() dlopen'ed libsrad before torch was imported).
            import torch  # noqa: F401
            l = C.CDLL(LIB_PATH)
            missing = [n for n in _SIG if not hasattr(l, n)]
            if missing:
                raise RuntimeError(f"{LIB_PATH} does not export {missing}; rebuild it")
            for name, (res, args) in _SIG.items():
                fn = getattr(l, name)
                fn.restype = res
                fn.argtypes = args
            _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().srad_last_error()
        raise RuntimeError(f"libsrad {what} failed ({rc}): {msg.decode() if msg else '?'}")


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t) -> C.c_void_p:
    """Device pointer of a torch tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def prof_enable(on: bool) -> None:
    check(lib().srad_prof_enable(1 if on else 0), "prof_enable")


def prof_collect() -> dict:
    """{class name: dict(launches, ms, flops, bytes)} of everything launched since the last call
    while profiling was enabled (HIP events on the launch stream)."""
    n = lib().srad_prof_num_classes()
    la, ms, fl, by = (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    check(lib().srad_prof_collect(la, ms, fl, by), "prof_collect")
    out = {}
    for i in range(n):
        if la[i]:
            out[lib().srad_prof_class_name(i).decode()] = dict(launches=int(la[i]), ms=float(ms[i]),
                                                               flops=float(fl[i]), bytes=float(by[i]))
    return out
