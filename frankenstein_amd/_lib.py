"""ctypes binding of libfranken_hip.so (include/franken_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (with fk_last_error()).  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

# FRANKEN_HIP_LIB selects another build of the same ABI (A/B timing of kernel variants on one GPU box)
LIB_PATH = Path(os.environ.get("FRANKEN_HIP_LIB") or Path(__file__).resolve().parent / "libfranken_hip.so")

FK_F32, FK_BF16 = 0, 1
MASK_NONE, MASK_CAUSAL, MASK_BLOCK_CAUSAL, MASK_PREFIX, MASK_KEYPAD, MASK_DENSE = 0, 1, 2, 3, 4, 5
NORM_LAYER, NORM_RMS = 0, 1
ATTN_Q_PRESCALED = 1

_p, _i64, _int, _f32, _f64, _sz, _u32 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint32

# name -> (restype, argtypes); mirrors include/franken_hip.h one to one
SIGNATURES = {
    "fk_version": (_int, []),
    "fk_last_error": (C.c_char_p, []),
    "fk_gemm_nt": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _p, _i64, _i64, _int, _int, _p]),
    "fk_gemm_nt_rope": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_gemm_nt_swiglu": (_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_gemm_nt_dswiglu": (_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_mlp_bwd_fused": (_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_gemm_tn_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "fk_gemm_tn": (_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _int, _int, _p, _sz, _p]),
    "fk_colsum_workspace_bytes": (_sz, [_i64, _i64]),
    "fk_colsum": (_int, [_p, _i64, _p, _i64, _i64, _int, _int, _p, _sz, _p]),
    "fk_attn_fwd": (_int, [_p, _p, _p, _p, _p] + [_i64] * 13 + [_int, _i64, _i64, _i64, _p, _p, _f32, _int, _int, _p]),
    "fk_attn_bwd": (_int, [_p] * 10 + [_i64] * 13 + [_int, _i64, _i64, _i64, _p, _p, _f32, _p, _i64, _i64, _int, _int, _p]),
    "fk_attn_fwd_dropout": (_int, [_p, _p, _p, _p, _p] + [_i64] * 13 + [_int, _i64, _i64, _i64, _p, _p, _f32, _int, _f32, _p, _u32, _int, _p]),
    "fk_attn_bwd_dropout": (_int, [_p] * 10 + [_i64] * 13 + [_int, _i64, _i64, _i64, _p, _p, _f32, _p, _i64, _i64, _int, _f32, _p, _u32, _int, _p]),
    "fk_dropout": (_int, [_p, _p, _p, _i64, _f32, _p, _u32, _int, _p]),
    "fk_norm_fwd": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _f32, _int, _int, _p]),
    "fk_norm_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "fk_norm_bwd": (_int, [_p] * 9 + [_i64, _i64, _int, _int, _int, _p, _sz, _p]),
    "fk_rope": (_int, [_p, _i64, _i64, _i64, _i64, _i64, _p, _i64, _i64, _int, _int, _p]),
    "fk_patchify": (_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_swiglu_fwd": (_int, [_p, _p, _i64, _i64, _int, _p]),
    "fk_swiglu_bwd": (_int, [_p, _p, _p, _i64, _i64, _int, _p]),
    "fk_gelu_fwd": (_int, [_p, _p, _i64, _int, _p]),
    "fk_gelu_bwd": (_int, [_p, _p, _p, _i64, _int, _p]),
    "fk_cast_pack": (_int, [_p, _i64, _p, _i64, _i64, _i64, _int, _int, _p]),
    "fk_cast_pack_rows": (_int, [_p, _i64, _p, _i64, _i64, _i64, _int, _i64, _i64, _i64, _int, _p]),
    "fk_cast_pack_multi": (_int, [_p, _i64, _i64, _int, _p]),
    "fk_cast": (_int, [_p, _int, _p, _int, _i64, _p]),
    "fk_add": (_int, [_p, _p, _p, _i64, _int, _p]),
    "fk_gather_rows": (_int, [_p, _i64, _int, _p, _i64, _p, _i64, _int, _i64, _i64, _i64, _int, _p]),
    "fk_scatter_add_rows": (_int, [_p, _int, _p, _i64, _p, _i64, _i64, _p]),
    "fk_prefix_mask": (_int, [_p, _p, _i64, _p, _p, _i64, _i64, _i64, _p]),
    "fk_copy2d": (_int, [_p, _i64, _p, _i64, _i64, _i64, _int, _p]),
    "fk_attn_combine": (_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_add2d": (_int, [_p, _i64, _p, _i64, _i64, _i64, _p]),
    "fk_gpt_embed_step": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _int, _p]),
    "fk_kv_append": (_int, [_p, _p, _p, _i64, _i64, _i64, _int, _p]),
    "fk_attn_decode": (_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _i64, _f32, _int, _p]),
    "fk_sample_topk": (_int, [_p, _i64, _i64, _i64, _f32, _i64, _p, _p, _p, _p, _p, _i64, _i64, _p, _p]),
    "fk_im2col1d": (_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_col2im1d": (_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_elu_fwd": (_int, [_p, _p, _i64, _int, _p]),
    "fk_elu_bwd": (_int, [_p, _p, _p, _i64, _int, _p]),
    "fk_argmax_rows": (_int, [_p, _i64, _p, _i64, _i64, _int, _p]),
    "fk_block_stats_workspace_bytes": (_sz, [_i64, _i64]),
    "fk_block_stats": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _sz, _p]),
    "fk_zscore_smooth_pad": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f64, _p]),
    "fk_gpt_embed_fwd": (_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_gpt_embed_bwd_wte": (_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_loss_workspace_bytes": (_sz, [_i64]),
    "fk_l1_loss_fwd": (_int, [_p, _p, _p, _i64, _int, _p, _i64, _int, _p, _sz, _p]),
    "fk_l1_loss_bwd": (_int, [_p, _p, _p, _p, _i64, _int, _p, _i64, _p, _int, _p]),
    "fk_head_ce_workspace_bytes": (_sz, [_i64, _i64]),
    "fk_head_ce_fwd": (_int, [_p, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i64, _int, _p, _sz, _p]),
    "fk_head_ce_bwd": (_int, [_p, _i64, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_transpose2d": (_int, [_p, _i64, _p, _i64, _i64, _i64, _int, _p]),
    "fk_ce_workspace_bytes": (_sz, [_i64]),
    "fk_ce_loss_fwd": (_int, [_p, _i64, _p, _p, _p, _i64, _i64, _i64, _int, _p, _sz, _p]),
    "fk_ce_loss_bwd": (_int, [_p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_ce_chunk_fwd": (_int, [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _int, _p]),
    "fk_ce_chunk_finish": (_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _sz, _p]),
    "fk_ce_chunk_bwd": (_int, [_p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p]),
    "fk_adamw_step": (_int, [_p, _p, _p, _p, _i64, _f64, _f64, _f64, _f64, _f64, _i64, _f64, _f64, _int, _p]),
}

_lib = None


class FrankenHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the C-ABI library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FrankenHipError(
                f"{LIB_PATH} is missing: build it with `python -m frankenstein_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback.")
        h = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)      # AttributeError if the header and the library disagree
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().fk_last_error().decode(errors="replace")
        raise FrankenHipError(f"{what or 'libfranken_hip'} failed (code {rc}): {msg}")


def call(name: str, *args):
    """Call an int-returning entry point and raise on a non-zero code."""
    check(getattr(lib(), name)(*args), name)
