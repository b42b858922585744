"""ctypes binding of libsnag_hip.so (the C ABI in include/snag_hip.h).

There is NO fallback: if the shared library is missing or a call is refused, this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNAG_HIP_LIB") or os.path.join(_HERE, "libsnag_hip.so")   # override: A/B builds of the same ABI

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float
_U64 = _c.c_uint64
SCORE_TICKET_WORDS = 1056      # SNAG_SCORE_TICKET_WORDS (include/snag_hip.h)
MAX_MASK_REGIONS = 8           # SNAG_MAX_MASK_REGIONS
ABI_VERSION = 3                # SNAG_ABI_VERSION: the struct layouts and argument lists below are those of this version



class GemmDesc(_c.Structure):
    """struct snag_gemm_desc (include/snag_hip.h)."""
    _fields_ = [("M", _c.c_int32), ("N", _c.c_int32), ("K", _c.c_int32),
                ("a_kcontig", _c.c_int32), ("b_kcontig", _c.c_int32),
                ("A", _P), ("lda", _L), ("B", _P), ("ldb", _L), ("C", _P), ("ldc", _L),
                ("bias", _P), ("alpha", _F), ("act", _c.c_int32), ("pre_act", _P),
                ("keep_mask", _P), ("keep_scale", _F),
                ("add1", _P), ("ld1", _L), ("add2", _P), ("ld2", _L),
                ("accumulate", _c.c_int32), ("workspace", _P), ("workspace_floats", _L),
                ("aux", _P), ("ldaux", _L), ("ldmask", _L), ("defer_reduce", _c.c_int32),
                ("mirror", _P), ("ldmirror", _L), ("mirror_rowmask", _P), ("mirror_scale", _F),
                ("c_transposed", _c.c_int32), ("splits", _c.c_int32)]


class MaskRegion(_c.Structure):
    """struct snag_mask_region."""
    _fields_ = [("seed", _U64), ("offset", _L), ("n", _L), ("p", _F), ("invert", _c.c_int32), ("out", _P), ("period", _L), ("skip", _L), ("take", _L)]


class FusionLayer(_c.Structure):
    """struct snag_fusion_layer."""
    _fields_ = [(n, _P) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b",
                                  "keep_probs", "keep_attn_out", "keep_ffn_out")]


class FusionLayerGrad(_c.Structure):
    """struct snag_fusion_layer_grad."""
    _fields_ = [(n, _P) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")]


class FusionDesc(_c.Structure):
    """struct snag_fusion_desc."""
    _fields_ = [("N", _c.c_int32), ("M", _c.c_int32), ("D", _c.c_int32), ("H", _c.c_int32), ("I", _c.c_int32),
                ("L", _c.c_int32), ("use_intermediate", _c.c_int32), ("drop_p", _F),
                ("layers", _c.POINTER(FusionLayer)), ("weight_raw", _P), ("x0", _P),
                ("joint", _P), ("joint_fz", _P), ("hidden", _P), ("weight_norm", _P),
                ("save", _P), ("save_floats", _L), ("work", _P), ("work_floats", _L), ("readout_slot_p1", _c.c_int32),
                ("fold_others", _c.c_int32), ("co_fwd", _c.POINTER(GemmDesc) * 2), ("n_co_fwd", _c.c_int32), ("weight_images_ready", _c.c_int32), ("force_one_launch", _c.c_int32)]


class SmallMM(_c.Structure):
    """struct snag_small_mm."""
    _fields_ = [("M", _c.c_int32), ("N", _c.c_int32), ("K", _c.c_int32), ("accumulate", _c.c_int32), ("A", _P), ("sam", _L), ("sak", _L),
                ("B", _P), ("sbk", _L), ("sbn", _L), ("C", _P), ("ldc", _L), ("A2", _P), ("B2", _P)]


class FusionBwdDesc(_c.Structure):
    """struct snag_fusion_bwd_desc."""
    _fields_ = [("f", FusionDesc), ("d_joint", _P), ("d_joint_fz", _P), ("d_hidden", _P), ("d_weight_norm", _P),
                ("d_x0", _P), ("grads", _c.POINTER(FusionLayerGrad)), ("d_weight_raw", _P), ("accumulate", _c.c_int32),
                ("zero_d_hidden", _c.c_int32), ("d_slot_out", _P), ("d_slot_ld", _L), ("d_slot_rowmask", _P), ("d_slot_scale", _F),
                ("co_dx", _c.POINTER(GemmDesc) * 2), ("n_co_dx", _c.c_int32), ("co_dw", _c.POINTER(GemmDesc) * 2), ("n_co_dw", _c.c_int32),
                ("ln_bwd_done", _c.c_int32), ("wo_images_ready", _c.c_int32), ("separate_attention_backward", _c.c_int32),
                ("next_group", _P), ("n_next_group", _c.c_int32)]


class LnBwdTail(_c.Structure):
    """struct snag_ln_bwd_tail."""
    _fields_ = [("t", _P), ("stats", _P), ("gamma", _P), ("keep", _P), ("keep_ld", _L), ("keep_scale", _F), ("dt", _P), ("dtm", _P), ("part", _P)]


# name -> argtypes; every entry returns int (0 = enqueued) except the two listed in _SPECIAL
SIGNATURES = {
    "snag_score_margin_fwd": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _P, _P, _P, _P],
    "snag_score_margin_fwd_bf16": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _P, _P, _P, _P],
    "snag_score_margin_bwd": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _L, _P, _P],
    "snag_score_margin_fwd_bwd": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "snag_score_margin_fwd_bwd_seg": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _L, _P, _I, _P, _L, _P],
    "snag_score_margin_fwd_bwd_touched": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _L, _P, _P, _L, _P],
    "snag_score_margin_fwd_bwd_touched_bf16": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _L, _P, _P, _L, _P],
    "snag_score_margin_fwd_bwd_bf16": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "snag_score_margin_bwd_bf16": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _L, _P, _P],
    "snag_score_margin_bwd_fixed_bf16": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P],
    "snag_adamw_step_bf16": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _P, _F, _P, _P],
    "snag_f32_to_bf16": [_P, _L, _P, _P],
    "snag_zero_touched_rows": [_P, _L, _I, _P, _P, _P, _I, _L, _I, _P],
    "snag_reduce_sum": [_P, _L, _P, _P],
    "snag_rank_full_table": [_P, _L, _L, _P, _I, _P, _P, _P],
    "snag_gemm_f32": [_c.POINTER(GemmDesc), _P],
    "snag_gemm_reduce_pair": [_P, _P, _P],
    "snag_gemm_reduce_group": [_P, _c.c_int, _P],
    "snag_gemm_reduce_group_mm": [_P, _c.c_int, _P, _c.c_int, _P],
    "snag_gemm_f32_group": [_P, _c.c_int, _P],
    "snag_gemm_f32_pair": [_c.POINTER(GemmDesc), _c.POINTER(GemmDesc), _P],
    "snag_fusion_fwd": [_c.POINTER(FusionDesc), _P],
    "snag_fusion_fwd_prepare_reduce": [_c.POINTER(FusionDesc), _P, _c.c_int, _c.POINTER(_c.c_int32), _P],
    "snag_fusion_bwd": [_c.POINTER(FusionBwdDesc), _P],
    "snag_fusion_readout_offsets": [_c.POINTER(FusionDesc), _P, _P],
    "snag_fusion_readout_ln_offsets": [_c.POINTER(FusionDesc), _P, _P, _P, _P, _P],
    "snag_score_margin_fwd_bwd_seg_ln": [_P, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _L, _c.POINTER(LnBwdTail), _P],
    "snag_small_matmul": [_P, _I, _P],
    "snag_pack_tokens": [_c.POINTER(_P), _I, _L, _I, _I, _P, _P, _F, _P, _P],
    "snag_token_mean": [_P, _L, _I, _I, _P, _P],
    "snag_atten_weight_fwd": [_P, _L, _I, _I, _P, _P, _L, _P, _P],
    "snag_atten_weight_bwd": [_P, _L, _I, _I, _P, _P, _P, _L, _P, _P, _I, _P, _P],
    "snag_adaptive_avg_pool_rows": [_P, _L, _L, _I, _I, _I, _I, _P, _L, _P],
    "snag_colstats": [_P, _L, _I, _P, _L, _P, _P, _P, _P],
    "snag_noise_apply": [_P, _L, _I, _P, _P, _P, _P, _P, _U64, _F, _P, _P],
    "snag_dropout_keep_mask": [_U64, _P, _L, _F, _P, _P],
    "snag_gaussian_fill": [_U64, _L, _I, _P, _P, _P, _P],
    "snag_colsum": [_P, _L, _I, _L, _P, _I, _P, _P],
    "snag_colsum_pair": [_P, _P, _L, _I, _L, _P, _P, _I, _P, _P],
    "snag_token_mean_bwd": [_P, _L, _I, _I, _P, _I, _P],
    "snag_entity_token_bwd": [_P, _L, _L, _I, _P, _F, _P, _I, _P],
    "snag_adamw_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _P, _F, _P],
    "snag_adamw_step_groups": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P, _P, _I, _L, _P, _F, _P, _I, _P],
    "snag_clip_adamw_step_groups": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _P, _P, _I, _P, _F, _F, _P, _P, _P],
    "snag_clip_grad_norm": [_P, _L, _F, _F, _P, _P, _P, _P],
    "snag_tick": [_P, _P],
    "snag_gat_layer_fwd": [_P, _L, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _F, _P],
    "snag_gat_layer_bwd": [_P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _F, _P],
    "snag_keep_scale": [_P, _L, _P, _F, _P, _P],
    "snag_build_filter_mask": [_P, _P, _P, _L, _I, _I, _P, _L, _P, _P],
    "snag_sample_negatives": [_U64, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _L, _P, _P, _P],
    "snag_step_prologue": [_U64, _P, _P, _P, _P, _L, _I, _I, _I, _P, _L, _P, _P, _L, _I, _P, _P, _F, _P, _L, _P, _I, _P, _I, _P],
    "snag_bernoulli_regions": [_P, _I, _P, _P],
    "snag_noise_refresh": [_P, _L, _I, _P, _P, _P, _P, _U64, _L, _F, _P, _P, _L, _P],
    "snag_sample_negatives_distinct": [_U64, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _L, _P, _I, _P, _P, _P],
    "snag_gather_normalize_fwd": [_P, _L, _L, _I, _P, _I, _L, _I, _P, _P, _P],
    "snag_gather_normalize_bwd": [_P, _P, _P, _I, _P, _I, _L, _I, _P, _L, _P],
    "snag_spmm_csr": [_P, _P, _P, _P, _L, _I, _P, _P, _P],
    "snag_cityblock_dist": [_P, _L, _I, _P, _L, _I, _I, _P, _L, _P],
    "snag_transpose_f32": [_P, _L, _L, _L, _P, _L, _P],
    "snag_score_margin_bwd_fixed": [_P, _L, _L, _P, _L, _I, _P, _P, _P, _P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P],
    "snag_fixed_to_float": [_P, _L, _I, _P, _L, _I, _P],
    "snag_selfadv_loss": [_P, _P, _L, _I, _F, _F, _P, _P, _P, _P],
    "snag_gram_f32": [_P, _L, _I, _I, _F, _P, _L, _P],
    "snag_infonce_rows": [_P, _L, _I, _P, _P, _P],
    "snag_infonce_grad": [_P, _L, _I, _P, _P, _F, _F, _P, _P],
    "snag_row_sqnorm": [_P, _L, _I, _L, _P, _P],
    "snag_pairwise_dist": [_P, _L, _I, _P, _L, _I, _I, _P, _P, _P, _L, _I, _P],
    "snag_csls_neighbourhood": [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P],
    "snag_csls_apply": [_P, _L, _I, _I, _P, _P, _P, _P],
    "snag_rank_diagonal": [_P, _L, _I, _I, _P, _P, _P, _P, _P],
    "snag_col_topk": [_P, _L, _I, _I, _I, _I, _P, _P, _P],
    "snag_col_topk_mean": [_P, _I, _I, _I, _P, _P],
    "snag_rank_rows_block": [_P, _L, _I, _I, _I, _P, _P, _P, _P, _P],
    "snag_rank_cols_block": [_P, _L, _I, _I, _P, _P, _P, _P, _P],
    "snag_rank_summary": [_P, _I, _P, _I, _P, _P],
    "snag_rank_summary_pair": [_P, _P, _I, _P, _I, _P, _P],
    "snag_argmin_rows_cols": [_P, _L, _I, _I, _P, _P, _P, _P],
    "snag_entity_token_fwd": [_P, _L, _I, _P, _P, _F, _P, _L, _P],
}
_SPECIAL = {"snag_version": ([], _I), "snag_last_error": ([], _c.c_char_p),
            "snag_gemm_workspace_floats": ([_I, _I, _I], _L),
            "snag_fusion_save_floats": ([_I] * 7, _L), "snag_fusion_work_floats": ([_I] * 7, _L),
            "snag_colstats_work_floats": ([_I], _L), "snag_atten_weight_work_floats": ([_I], _L), "snag_eval_work_floats": ([_I, _I], _L),
            "snag_gat_work_floats": ([_L, _I, _I], _L), "snag_score_seg_work_bytes": ([_L, _L, _L, _I, _I], _L),
            "snag_score_touched_work_bytes": ([_L, _L, _L, _I, _I], _L)}

_lib = None


class SnagHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise with the build recipe if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SnagHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m snag_amd.build` "
            "(hipcc --offload-arch=gfx950). snag_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argt
        fn.restype = _I
    for name, (argt, rest) in _SPECIAL.items():
        fn = getattr(lib, name)
        fn.argtypes = argt
        fn.restype = rest
    got = lib.snag_version()
    if got != ABI_VERSION:
        raise SnagHipError(f"{LIB_PATH} reports ABI version {got}, this binding is written against {ABI_VERSION}: a stale or variant library "
                           "would be read with the wrong struct offsets -- rebuild it (`python -m snag_amd.build --force`)")
    _lib = lib
    return lib


def last_error() -> str:
    return load().snag_last_error().decode("utf-8", "replace")


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise SnagHipError(f"{name} failed ({rc}): {last_error()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    if not torch.cuda.is_available():
        raise SnagHipError("snag_amd needs a visible MI355X (torch.cuda.is_available() is False); there is no CPU path")
