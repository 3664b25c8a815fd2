"""ctypes binding of include/omnitok.h (libomnitok.so).

The HIP library is the product: if it is missing or fails to load this module raises -- there is
no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char, c_char_p, c_float, c_int, c_int64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# OMNITOK_LIB: another build of the same library (measurement builds of tools/: ablation arms compiled with extra -D flags)
LIB_PATH = os.environ.get("OMNITOK_LIB") or os.path.join(HERE, "lib", "libomnitok.so")


class OmnitokConfig(Structure):
    _fields_ = [
        ("resolution", c_int), ("image_channels", c_int), ("patch_size", c_int),
        ("temporal_patch_size", c_int), ("dim", c_int), ("heads", c_int), ("dim_head", c_int),
        ("ff_inner", c_int), ("window_size", c_int), ("n_codes", c_int), ("codebook_dim", c_int),
        ("l2_code", c_int), ("spatial_rope", c_int), ("legacy_attention", c_int),
        ("causal_temporal", c_int), ("causal_peg", c_int), ("temporal_depth", c_int),
        ("enc_block", c_char * 16), ("dec_block", c_char * 16), ("use_vae", c_int),
        ("patch_embed_cnn", c_int), ("defer_temporal_pool", c_int), ("defer_spatial_pool", c_int),
        ("gen_upscale", c_int), ("external_codebook", c_int),
    ]


class OmnitokLmConfig(Structure):
    _fields_ = [("vocab_size", c_int), ("block_size", c_int), ("n_layer", c_int), ("n_head", c_int),
                ("n_embd", c_int)]


class OmnitokPlGemm(Structure):
    """omnitok_pl_gemm (include/omnitok.h): arguments of the plane x plane GEMM."""
    _fields_ = [
        ("a", c_void_p), ("a_scale", c_void_p), ("a_scale_const", c_float),
        ("a2", c_void_p), ("a2_scale", c_void_p), ("a2_scale_const", c_float), ("a_split_n", c_int),
        ("w", c_void_p), ("w_scale", c_void_p), ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_int64),
        ("c", c_void_p), ("ldc", c_int64), ("c2", c_void_p), ("ldc2", c_int64), ("c_split_n", c_int),
        ("out_planes", c_void_p), ("out_planes_k", c_int), ("out_bound", c_float),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float), ("epilogue", c_int),
        ("fold_stats", c_void_p), ("fold_b", c_void_p), ("fold_u", c_void_p), ("fold_cols", c_int),
        ("qp", c_void_p), ("kp", c_void_p), ("vp", c_void_p), ("qk_k0", c_int), ("n_tokens", c_int), ("heads", c_int),
        ("rope_cos", c_void_p), ("rope_sin", c_void_p), ("q_scale", c_void_p), ("k_scale", c_void_p),
        ("q_mul", c_float), ("q_bound", c_float), ("k_bound", c_float), ("v_bound", c_float),
        ("v_bound_dev", c_void_p), ("v_bound_stride", c_int), ("rows_per_clip", c_int64),
        ("M", c_int64), ("N", c_int), ("K", c_int), ("cfg", c_int), ("debug_cycles", c_void_p),
        ("a_rpg", c_int64), ("a_gstride", c_int64), ("a_goff", c_int64),
        ("up_C", c_int), ("up_F", c_int), ("up_H", c_int), ("up_W", c_int), ("up_f0", c_int), ("up_t", c_int),
        ("up_pt", c_int), ("up_p", c_int), ("k_valid", c_int),
        ("tp", c_void_p), ("t_nseq", c_int), ("t_heads", c_int), ("t_seqs_per_clip", c_int), ("t_alibi", c_void_p),
        ("t_out_scale", c_void_p),
    ]


class OmnitokError(RuntimeError):
    pass


P = c_void_p
I64 = c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/omnitok.h
_PROTOS = {
    "omnitok_layernorm": [P, P, P, P, I64, c_int, c_float, I64, I64, I64, P],
    "omnitok_gemm": [P, I64, P, I64, P, P, I64, P, I64, I64, c_int, c_int, c_int, I64, I64, I64, P],
    "omnitok_gemm_x3": [P, I64, P, I64, P, P, I64, P, I64, I64, c_int, c_int, c_int, I64, I64, I64, P, P, P, c_int, P, I64,
                        c_int, P],
    "omnitok_h2_pack_weight": [P, I64, c_int, c_int, P, P, P],
    "omnitok_gemm_h2": [P, I64, P, P, P, P, I64, P, I64, I64, c_int, c_int, c_int, I64, I64, I64, c_float, P, c_int,
                        I64, P, P, P, c_int, c_float, P, I64, c_int, P],
    "omnitok_gemm_h2_vpack": [P, I64, P, P, P, P, I64, P, I64, I64, c_int, c_int, c_int, I64, I64, I64, c_float, P, c_int,
                              I64, P, P, P, c_int, c_float, P, I64, c_int, P, c_int, c_int, c_int, c_float, P, c_int, P],
    "omnitok_row_stats": [P, I64, c_int, c_float, P, P, I64, P],
    "omnitok_weight_range": [P, I64, c_int, c_int, P, P],
    "omnitok_pack_geglu_weight": [P, c_int, c_int, c_int, P, P],
    "omnitok_patchify_ln": [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_float, P, I64, P],
    "omnitok_unpatchify": [P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "omnitok_peg3d": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "omnitok_pack_peg_weight": [P, c_int, P, P],
    "omnitok_transpose_tokens": [P, P, I64, I64, I64, c_int, P],
    "omnitok_rope_table": [c_int, c_int, c_float, P, P],
    "omnitok_qk_prep": [P, I64, P, I64, I64, c_int, c_int, P, P, P, P, c_float, P],
    "omnitok_attn_spatial": [P, I64, P, P, I64, P, I64, c_int, c_int, c_int, P, c_int, c_int, P],
    "omnitok_attn_pack": [P, I64, P, P, I64, I64, c_int, c_int, P, P, P, P, c_float, c_float, c_float, c_float, P, c_int,
                          I64, P, P, P, P],
    "omnitok_attn_spatial_h2": [P, P, P, P, I64, c_int, c_int, c_int, c_float, c_float, c_float, P, c_int, c_int, P,
                                c_int, c_int, P],
    "omnitok_attn_window": [P, I64, P, P, I64, c_int, c_int, c_int, c_int, P],
    "omnitok_attn_temporal": [P, I64, P, P, I64, P, I64, I64, c_int, c_int, P, P, c_float, c_int, P, P],
    "omnitok_pre_vq": [P, P, P, P, I64, c_int, c_int, c_int, P],
    "omnitok_vq_prepare": [P, c_int, c_int, P, P, P],
    "omnitok_vq_argmin": [P, P, P, I64, c_int, P, P],
    "omnitok_vq_screen_prepare": [P, P, c_int, c_int, P, P],
    "omnitok_vq_argmin_screened": [P, P, P, P, I64, c_int, P, P],
    "omnitok_vq_argmax_cos": [P, P, I64, c_int, P, P],
    "omnitok_vq_argmin_cdist": [P, P, P, I64, c_int, P, P],
    "omnitok_dequant_post_vq": [P, P, c_int, c_int, P, P, P, I64, c_int, P, P],
    "omnitok_dequant_table": [P, c_int, c_int, P, P, P, c_int, P, P],
    "omnitok_gather_rows": [P, P, c_int, P, I64, c_int, P, P],
    "omnitok_gather_rows_transposed": [P, P, c_int, P, I64, c_int, c_int, c_int, P, P],
    "omnitok_layernorm_transposed": [P, P, P, P, I64, c_int, c_int, c_int, c_float, P],
    "omnitok_layernorm_prevq": [P, P, P, P, P, P, I64, c_int, c_int, c_int, c_float, c_int, c_int, P],
    "omnitok_token_resample": [P, P, c_int, I64, c_int, c_int, c_int, c_int, P],
    "omnitok_vae_sample": [P, P, P, P, P, P, I64, I64, c_int, c_int, P],
    "omnitok_post_vq": [P, c_int, I64, I64, c_int, P, P, P, c_int, P],
    "omnitok_vq_embed_st": [P, P, P, c_int, I64, I64, P, P],
    "omnitok_vq_stats": [P, I64, c_int, P, P, P, c_int, c_float, P, P],
    "omnitok_engine_create": [POINTER(OmnitokConfig), POINTER(P)],
    "omnitok_engine_destroy": [P],
    "omnitok_engine_set_weight": [P, c_char_p, P, POINTER(I64), c_int, c_int, P],
    "omnitok_engine_finalize": [P, P],
    "omnitok_engine_missing": [P, c_char_p, c_int],
    "omnitok_encode": [P, P, c_int, c_int, c_int, c_int, P, P, P, P],
    "omnitok_decode": [P, P, c_int, c_int, c_int, c_int, P, P],
    "omnitok_encode_vae": [P, P, c_int, c_int, c_int, c_int, P, P, P, P],
    "omnitok_decode_vae": [P, P, c_int, c_int, c_int, c_int, c_int, P, P],
    "omnitok_engine_encode_shape": [P, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)],
    "omnitok_engine_decode_shape": [P, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)],
    "omnitok_engine_check_ids": [P, P],
    "omnitok_engine_workspace_bytes": [P],
    "omnitok_engine_workspace_need_encode": [P, c_int, c_int, c_int, c_int],
    "omnitok_engine_workspace_need_decode": [P, c_int, c_int, c_int, c_int],
    "omnitok_engine_set_workspace": [P, P, I64],
    "omnitok_engine_set_timing": [P, c_int],
    "omnitok_engine_set_option": [P, c_char_p, c_int],
    "omnitok_engine_timing_report": [P, c_char_p, c_int],
    # include/omnitok_lm.h
    "omnitok_lm_create": [POINTER(OmnitokLmConfig), POINTER(P)],
    "omnitok_lm_destroy": [P],
    "omnitok_lm_set_weight": [P, c_char_p, P, POINTER(I64), c_int, P],
    "omnitok_lm_finalize": [P, P],
    "omnitok_lm_alloc_cache": [P, c_int, c_int],
    "omnitok_lm_cache_bytes": [P],
    "omnitok_lm_overflowed": [P, P],
    "omnitok_lm_step": [P, P, P, P, c_int, P, c_int, P],
    "omnitok_lm_prefill": [P, P, P, P, c_int, c_int, P, P],
    "omnitok_lm_step_ex": [P, P, P, P, P, P, c_int, P, c_int, P],
    "omnitok_lm_prefill_ex": [P, P, c_int, P, c_int, P, P, P, c_int, P, P],
    "omnitok_lm_select": [P, P, c_int, c_int, c_float, c_float, c_float, c_int, c_float, c_int, P, P, P, P, P],
    "omnitok_lm_gemv": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "omnitok_lm_attn_decode": [P, P, P, P, c_int, c_int, c_int, c_int, P, P, P],
    "omnitok_pl_planes_bytes": [I64, c_int, c_int],
    "omnitok_pl_unscale": [c_float],
    "omnitok_pl_pack_weight": [P, I64, c_int, c_int, c_int, P, P, P],
    "omnitok_pl_pack_rows": [P, I64, I64, c_int, I64, P, P, c_float, P],
    "omnitok_gemm_pl": [POINTER(OmnitokPlGemm), P],
    "omnitok_stats_pack": [P, I64, c_int, c_float, c_int, P, I64, P, P, P, I64, P],
    "omnitok_attn_spatial_h2_planes": [P, P, P, P, I64, P, P, c_int, c_int, c_int, c_float, c_float, c_float, P, c_int,
                                       c_int, P, c_int, c_int, P],
    "omnitok_attn_window_planes": [P, I64, P, P, I64, P, c_float, c_int, c_int, c_int, c_int, P],
    "omnitok_layernorm_planes": [P, I64, c_int, c_float, P, P, c_float, P, I64, P],
    "omnitok_attn_window_h2": [P, P, P, P, P, I64, P, c_float, c_float, c_float, c_int, c_int, c_int, c_int, P],
    "omnitok_stats_pack_windows": [P, I64, c_int, c_float, c_int, P, I64, P, P, c_int, c_int, c_int, P],
    "omnitok_stats_pack_temporal": [P, I64, c_int, c_float, P, P, P, P, I64, P],
    "omnitok_attn_temporal_planes": [P, I64, P, P, I64, P, I64, P, P, c_float, P, c_int, I64, I64, c_int, c_int, P, P,
                                     c_float, c_int, P, P],
    # include/omnitok_comm.h
    "omnitok_comm_available": [c_char_p, c_int],
    "omnitok_comm_unique_id": [P],
    "omnitok_comm_create": [P, c_int, c_int, POINTER(P)],
    "omnitok_comm_destroy": [P],
    "omnitok_comm_world": [P],
    "omnitok_comm_rank": [P],
    "omnitok_comm_allgather_i32": [P, P, P, I64, P],
    "omnitok_comm_allgather_ids": [P, P, P, I64, P],
    "omnitok_set_option": [c_char_p, c_int],
    "omnitok_get_option": [c_char_p, POINTER(c_int)],
    "omnitok_debug_set_gemm_trace": [P],
    "omnitok_debug_mfma_peak": [P, P, c_int, c_int, c_int, P, P],
    "omnitok_last_error": [],
    "omnitok_version": [],
}
_RESTYPES = {"omnitok_last_error": c_char_p, "omnitok_version": c_char_p,
             "omnitok_engine_destroy": None, "omnitok_engine_workspace_bytes": c_int64,
             "omnitok_engine_workspace_need_encode": c_int64, "omnitok_engine_workspace_need_decode": c_int64,
             "omnitok_lm_destroy": None, "omnitok_comm_destroy": None, "omnitok_lm_cache_bytes": c_int64,
             "omnitok_pl_planes_bytes": c_int64, "omnitok_pl_unscale": c_float}

EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


def load():
    """Loads libomnitok.so (building nothing: run `python omnitokenizer_amd/build.py` or
    __graft_entry__.build() first)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OmnitokError(
            f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
            "Build it with `python omnitokenizer_amd/build.py`.")
    # PyTorch-ROCm ships its own HIP runtime; it must be the one in the process before libomnitok.so is mapped,
    # otherwise the library binds a second runtime (/opt/rocm) that sees no device once torch has initialised its
    # own ("no ROCm-capable device is detected" at the first hipMalloc).  Importing torch first pins the order no
    # matter what the caller imported before.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    _lib = lib
    # tuning knobs for A/B runs: OMNITOK_OPTS="gemm_variant=1,gemm_lds_pad_kb=40"
    for kv in filter(None, os.environ.get("OMNITOK_OPTS", "").split(",")):
        k, v = kv.split("=")
        if lib.omnitok_set_option(k.strip().encode(), int(v)) != 0:
            raise OmnitokError(lib.omnitok_last_error().decode())
    return lib


def get_option(name: str) -> int:
    v = c_int()
    check(load().omnitok_get_option(name.encode(), ctypes.byref(v)), "get_option")
    return v.value


def set_option(name: str, value: int):
    check(load().omnitok_set_option(name.encode(), int(value)), "set_option")


def check(rc: int, what: str = ""):
    """Raises on a negative status: ValueError for invalid arguments (the reference's asserts),
    RuntimeError otherwise."""
    if rc >= 0:
        return rc
    msg = load().omnitok_last_error().decode(errors="replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -4:
        raise NotImplementedError(msg)
    raise OmnitokError(f"{what}: {msg} (status {rc})")
