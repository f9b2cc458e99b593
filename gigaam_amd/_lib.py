"""ctypes binding of libgigaam_hip.so (the C ABI in include/gigaam_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, the product path raises.  (A CPU restatement exists only as the test
oracle under /oracle and is never imported from this package.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libgigaam_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

# enums of include/gigaam_hip.h
SUBS_CONV2D, SUBS_CONV1D = 0, 1
ATT_ROTARY, ATT_REL_POS = 0, 1
NORM_BATCH, NORM_LAYER = 0, 1
HEAD_NONE, HEAD_CTC, HEAD_RNNT, HEAD_EMO = 0, 1, 2, 3
DTYPE_F32, DTYPE_F16, DTYPE_BF16, DTYPE_F64, DTYPE_I64 = 0, 1, 2, 3, 4
GEMM_F32, GEMM_F16X3, GEMM_F16 = 0, 1, 2
PF_CLASSES = ["gemm", "conv2", "attn", "norm", "convmod", "stem", "frontend", "decode", "misc"]


class GamConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "sample_rate", "n_mels", "hop_length", "win_length", "n_fft", "center",
        "feat_in", "n_layers", "d_model", "subsampling", "subs_kernel_size", "subsampling_factor",
        "ff_expansion_factor", "self_attention_model", "n_heads", "pos_emb_max_len",
        "conv_norm_type", "conv_kernel_size",
        "head_type", "num_classes", "pred_hidden", "pred_rnn_layers", "joint_hidden",
    )]


# name -> (restype, argtypes); every symbol include/gigaam_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "gam_abi_version": (C.c_int, []),
    "gam_create": (C.c_int, [C.POINTER(GamConfig), C.c_int, C.POINTER(_P)]),
    "gam_destroy": (None, [_P]),
    "gam_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "gam_finalize": (C.c_int, [_P]),
    "gam_feat_frames": (C.c_int64, [_P, C.c_int64]),
    "gam_enc_frames": (C.c_int64, [_P, C.c_int64]),
    "gam_frontend": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P]),
    "gam_encode": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P]),
    "gam_encode_ex": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, C.c_int, _P, _P]),
    "gam_encode_varlen": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int64), C.c_int, C.c_int64, _P, _P, C.c_int, _P, _P]),
    "gam_last_encode_rows": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "gam_ctc_head": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P]),
    "gam_ctc_greedy": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P, _P]),
    "gam_rnnt_greedy": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P]),
    "gam_emo_probs": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P]),
    "gam_rnnt_predict": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P, _P, _P]),
    "gam_rnnt_joint": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "gam_set_gemm_mode": (C.c_int, [_P, C.c_int]),
    "gam_get_gemm_mode": (C.c_int, [_P]),
    "gam_set_rnnt_cluster": (C.c_int, [_P, C.c_int]),
    "gam_get_rnnt_cluster": (C.c_int, [_P]),
    "gam_debug_buffer_hash": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]),
    "gam_range_flag": (C.c_int, [_P, C.POINTER(C.c_int), _P]),
    "gam_range_flag_fetch": (C.c_int, [_P, _P, _P]),
    "gam_op_gemm": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "gam_op_attention": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "gam_tune_sp": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "gam_plan_sp": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gam_plan_sp_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gam_tune_sp_stages": (C.c_int, [C.c_int]),
    "gam_profile_enable": (C.c_int, [_P, C.c_int]),
    "gam_profile_pause": (C.c_int, [_P, C.c_int]),
    "gam_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "gam_profile_read_bytes": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "gam_last_error": (C.c_char_p, [_P]),
    # multi-GPU exchange (RCCL behind the boundary)
    "gam_comm_unique_id": (C.c_int, [C.c_char_p]),
    "gam_comm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "gam_comm_world": (C.c_int, [_P]),
    "gam_gather_ids": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "gam_comm_last_error": (C.c_char_p, [_P]),
    "gam_comm_destroy": (None, [_P]),
}

_lib: Optional[C.CDLL] = None


class GigaAMHipError(RuntimeError):
    pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the HIP library and bind every symbol of the header; raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GIGAAM_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise GigaAMHipError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.gam_abi_version() != 1:
        raise GigaAMHipError("libgigaam_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib
