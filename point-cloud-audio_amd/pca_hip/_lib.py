"""ctypes binding of libpca_hip.so (C ABI declared in include/pca_hip.h).

The shared object is the product: if it is missing or fails to load, everything that
needs it raises -- there is no CPU or PyTorch fallback behind this module.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpca_hip.so")

PCA_F32, PCA_BF16 = 0, 1
K_GEMM_F32, K_MAB1_FWD, K_MAB1_BWD, K_MAB0_FWD, K_MAB0_BWD, K_WGRAD = 1, 2, 3, 4, 5, 6
K_SET_FWD, K_SET_BWD = 7, 8
MODE_F32, MODE_BF16, MODE_FP8 = 0, 1, 2

c_fp = C.c_void_p       # float* (device)
c_vp = C.c_void_p
c_i64p = C.c_void_p


class MabShape(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("B", "nq", "nk", "dq", "dk", "d", "h", "q_shared", "mode",
                 "q_dtype", "k_dtype", "y_dtype")] + [("k_lengths", C.c_void_p), ("ln", C.c_int32)]


class MabParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo",
                                          "ln0_w", "ln0_b", "ln1_w", "ln1_b")]


class MabGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo",
                                          "ln0_w", "ln0_b", "ln1_w", "ln1_b")]


class StConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "N", "din", "d", "h", "m", "k", "C", "mode")]


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("sa_m", C.c_int64), ("sa_k", C.c_int64), ("sb_k", C.c_int64),
                ("sb_n", C.c_int64), ("sc_m", C.c_int64),
                ("nb1", C.c_int32), ("nb2", C.c_int32),
                ("sa_b1", C.c_int64), ("sa_b2", C.c_int64), ("sb_b1", C.c_int64),
                ("sb_b2", C.c_int64), ("sc_b1", C.c_int64), ("sc_b2", C.c_int64),
                ("accumulate", C.c_int32), ("split_k", C.c_int32), ("alpha", C.c_float)]


# name -> (restype, argtypes); every symbol include/pca_hip.h declares
SIGNATURES = {
    "pca_abi_version": (C.c_int, []),
    "pca_last_error": (C.c_char_p, []),
    "pca_stft_num_frames": (C.c_int64, [C.c_int64, C.c_int]),
    "pca_stft_logmag": (C.c_int, [c_fp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, c_fp,
                                  C.c_int64, C.c_int64, c_vp]),
    "pca_stft_logmag_batch": (C.c_int, [c_fp, c_i64p, c_i64p, C.c_int, C.c_int64, C.c_int64,
                                        C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_int64,
                                        C.c_int64, c_vp]),
    "pca_resample": (C.c_int, [c_fp, C.c_int64, C.c_double, c_vp, c_vp, C.c_int, C.c_int, C.c_float, c_fp,
                               C.c_int64, c_vp]),
    "pca_pack_points_2d": (C.c_int, [c_fp, C.c_int64, C.c_int64, c_fp, c_i64p, C.c_int,
                                     C.c_int, c_fp, c_i64p, c_i64p, c_vp]),
    "pca_pack_points_3d": (C.c_int, [c_fp, C.c_int64, C.c_int64, C.c_int64, c_fp, c_fp,
                                     c_i64p, C.c_int, C.c_int, C.c_int, c_fp, c_i64p, c_i64p,
                                     c_vp]),
    "pca_pack_points_2d_seq": (C.c_int, [c_fp, C.c_int64, C.c_int64, c_fp, c_i64p, c_vp, c_vp,
                                         C.c_int, C.c_int, c_fp, c_i64p, c_i64p, c_vp]),
    "pca_pack_defer": (C.c_int, [C.c_int]),
    "pca_pack_points_3d_seq": (C.c_int, [c_fp, C.c_int64, C.c_int64, C.c_int64, c_fp, c_fp,
                                         c_vp, c_i64p, c_vp, c_vp, C.c_int, C.c_int, C.c_int,
                                         c_fp, c_vp, c_i64p, c_i64p, c_vp]),
    "pca_pack_points_3d_var": (C.c_int, [c_fp, C.c_int64, C.c_int64, C.c_int64, c_fp, c_fp,
                                         c_vp, c_i64p, C.c_int, C.c_int, C.c_int, c_fp, c_vp,
                                         c_i64p, c_i64p, c_vp]),
    "pca_mab_saved_bytes": (C.c_size_t, [C.POINTER(MabShape)]),
    "pca_mab_fwd_ws_bytes": (C.c_size_t, [C.POINTER(MabShape)]),
    "pca_mab_bwd_ws_bytes": (C.c_size_t, [C.POINTER(MabShape)]),
    "pca_mab_fwd": (C.c_int, [C.POINTER(MabShape), c_vp, c_vp, C.POINTER(MabParams), c_vp,
                              c_vp, c_vp, c_vp]),
    "pca_mab_bwd": (C.c_int, [C.POINTER(MabShape), c_vp, c_vp, C.POINTER(MabParams), c_vp,
                              c_vp, c_vp, c_vp, C.c_int, C.POINTER(MabGrads), c_vp, c_vp]),
    "pca_linear_fwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_int, C.c_int, c_vp]),
    "pca_linear_bwd": (C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_int,
                                 C.c_int, c_vp, c_vp]),
    "pca_linear_bwd_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "pca_cross_entropy": (C.c_int, [c_fp, c_i64p, C.c_int, C.c_int, C.c_float, c_fp, c_fp,
                                    c_fp, c_vp]),
    "pca_debug_poison_lds": (C.c_int, [c_vp]),
    "pca_subsample_points": (C.c_int, [c_fp, C.c_int64, C.c_int64, C.c_int64, c_fp, c_fp, c_vp,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_uint64, C.c_uint64, c_vp, c_fp, c_vp, c_vp, c_vp,
                                       c_vp]),
    "pca_importance_points": (C.c_int, [c_fp, C.c_int64, C.c_int64, C.c_int64, c_fp, c_fp, c_vp,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_fp,
                                        C.c_int, C.c_uint64, C.c_uint64, c_vp, c_fp, c_vp, c_fp,
                                        c_vp, c_vp, c_vp]),
    "pca_pack_points_2d_ss": (C.c_int, [c_fp, c_fp, c_vp, C.c_int, C.c_int, c_fp, c_vp, c_vp,
                                        c_vp]),
    "pca_adam_step": (C.c_int, [c_fp, c_fp, c_fp, c_fp, C.c_int64, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_float, c_vp, C.c_int, c_vp]),
    "pca_st_param_count": (C.c_int64, [C.POINTER(StConfig)]),
    "pca_st_bucket_split": (C.c_int64, [C.POINTER(StConfig)]),
    "pca_st_ws_bytes": (C.c_size_t, [C.POINTER(StConfig), C.c_int]),
    "pca_st_ws_layout": (C.c_int, [C.POINTER(StConfig), C.c_void_p]),
    "pca_st_handoff_counter": (C.c_int, [C.POINTER(StConfig), C.c_void_p, C.POINTER(C.c_void_p)]),
    "pca_st_forward": (C.c_int, [C.POINTER(StConfig), c_fp, c_fp, c_vp, c_fp, c_vp, c_vp]),
    "pca_st_train_fwd_bwd": (C.c_int, [C.POINTER(StConfig), c_fp, c_fp, c_vp, c_i64p, c_fp,
                                       c_fp, c_fp, c_fp, C.c_float, C.c_int, c_vp, c_vp]),
    "pca_prof_start": (C.c_int, [C.c_int, C.c_int]),
    "pca_prof_stop": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pca_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), c_fp, c_fp, c_fp, c_fp, c_vp]),
    "pca_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), c_fp, c_fp, c_fp, c_fp, c_vp]),
    "pca_softmax_rows": (C.c_int, [c_fp, C.c_int64, C.c_int, C.c_float, c_vp]),
    "pca_softmax_bwd_rows": (C.c_int, [c_fp, c_fp, C.c_int64, C.c_int, C.c_float, c_vp]),
    "pca_colsum": (C.c_int, [c_fp, C.c_int64, C.c_int, c_fp, C.c_int, c_vp]),
}

_lib = None
_lock = threading.Lock()


class PcaHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libpca_hip.so once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise PcaHipError(
                f"{LIB_PATH} not found: build it with point-cloud-audio_amd/csrc/build.sh "
                "(or __graft_entry__.build()); there is no fallback path")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().pca_last_error()
        raise PcaHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
