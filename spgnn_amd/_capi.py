"""ctypes binding of libspgnn_hip.so (include/spgnn_hip.h).

There is no CPU fallback: if the library is missing the import of any op fails loudly with
build instructions.  Symbols are bound with full argtypes so a header/library mismatch shows
up at load time.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspgnn_hip.so")
ABI_VERSION = 63

_i32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
_f32p = C.c_void_p
_i64, _i32, _f32, _u64, _vp = C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_void_p



class GemmNtProblem(C.Structure):         # spgnn_gemm_nt_problem
    _fields_ = [("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("C", _vp), ("ldc", _i64), ("M", _i64), ("N", _i64), ("K", _i64),
                ("scale_a", _vp), ("scale_b", _vp), ("upd_u", _vp), ("upd_u_stride", _i64), ("upd_v", _vp), ("upd_v_stride", _i64),
                ("bias", _vp), ("score_l", _vp), ("score_r", _vp), ("score_out", _vp), ("addend", _vp), ("addend_stride", _i64),
                ("absmax_out", _vp), ("upd_j", _i32), ("activation", _i32), ("score_cols", _i32), ("reserved", _i32),
                ("drop_seed_offset", _vp), ("drop_seed", _u64), ("drop_p", _f32), ("reserved2", _f32)]


class GemmTnProblem(C.Structure):         # spgnn_gemm_tn_problem
    _fields_ = [("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("C", _vp), ("ldc", _i64), ("split_stride", _i64), ("R", _i64),
                ("M", _i64), ("N", _i64), ("scale_a", _vp), ("scale_b", _vp), ("colsum_a", _vp), ("colsum_stride", _i64),
                ("colsum_split_stride", _i64), ("splits", _i32), ("flags", _i32)]


class ScoresBwdWJob(C.Structure):         # spgnn_scores_bwd_w_job
    _fields_ = [("g_s", _vp), ("g_s_stride", _i64), ("x", _vp), ("x_stride", _i64), ("partials", _vp), ("splits", _i32), ("Kp", _i32),
                ("K", _i32), ("J", _i32)]


class LspeFwdGroup(C.Structure):          # spgnn_lspe_fwd_group
    _fields_ = [("ft", _vp), ("ft_stride", _i64), ("res", _vp), ("res_stride", _i64), ("bias", _vp), ("el", _vp), ("er", _vp),
                ("s_stride", _i64), ("attn", _vp), ("score_parts", _vp), ("H", _i32), ("act", _i32), ("slope", _f32), ("p_drop", _f32),
                ("seed", _u64)]


class LspeBwdDstGroup(C.Structure):       # spgnn_lspe_bwd_dst_group
    _fields_ = [("ft", _vp), ("ft_stride", _i64), ("el", _vp), ("er", _vp), ("s_stride", _i64), ("attn", _vp), ("g_pre", _vp),
                ("g_pre_stride", _i64), ("g_e", _vp), ("g_er", _vp), ("gs_stride", _i64), ("absmax", _vp), ("H", _i32), ("act", _i32),
                ("slope", _f32), ("p_drop", _f32), ("seed", _u64)]


class LspeBwdSrcGroup(C.Structure):       # spgnn_lspe_bwd_src_group
    _fields_ = [("attn", _vp), ("g_e", _vp), ("g_pre", _vp), ("g_pre_stride", _i64), ("g_ft", _vp), ("g_ft_stride", _i64),
                ("g_el", _vp), ("g_er", _vp), ("gs_stride", _i64), ("score_l", _vp), ("score_r", _vp), ("absmax", _vp), ("H", _i32),
                ("p_drop", _f32), ("seed", _u64)]


class CopyPadJob(C.Structure):           # spgnn_copy_pad_job
    _fields_ = [("dst", _vp), ("src", _vp), ("pad", _vp), ("n", _i32), ("n_pad", _i32), ("pad_add", _i32), ("reserved", _i32)]


class CopyPadJobs(C.Structure):          # spgnn_copy_pad_jobs
    _fields_ = [("job", CopyPadJob * 8), ("n_jobs", _i32)]


class RowCopyJob(C.Structure):           # spgnn_row_copy_job
    _fields_ = [("dst", _vp), ("src", _vp), ("dst_stride", _i64), ("src_stride", _i64), ("rows_copy", _i64), ("rows_total", _i64),
                ("dst_col", _i32), ("width", _i32)]


class RowCopyJobs(C.Structure):          # spgnn_row_copy_jobs
    _fields_ = [("job", RowCopyJob * 16), ("n_jobs", _i32)]


class SumJob(C.Structure):               # spgnn_sum_job
    _fields_ = [("kind", _i32), ("splits", _i32), ("partials", _vp), ("split_stride", _i64), ("out", _vp), ("out_stride", _i64),
                ("n", _i64), ("H", _i32), ("D", _i32), ("ld", _i32), ("M", _i32), ("N", _i32), ("split_col", _i32), ("ld_in", _i64),
                ("out2", _vp), ("out2_stride", _i64), ("extra", _vp), ("extra_col", _i32), ("reserved", _i32)]


class WeightCatBf16Job(C.Structure):     # spgnn_weight_cat_bf16_job
    _fields_ = [("a", _vp), ("a_stride", _i64), ("b", _vp), ("b_stride", _i64), ("w", _vp), ("w_stride", _i64), ("w_t", _vp),
                ("w_t_stride", _i64), ("rows_a", _i32), ("rows_b", _i32), ("K", _i32), ("first_block", _i32), ("tiles_x", _i32),
                ("reserved", _i32)]


class WeightPrepLayer(C.Structure):      # spgnn_weight_prep_layer
    _fields_ = [("a", _vp), ("a_stride", _i64), ("b", _vp), ("b_stride", _i64), ("dst", _vp), ("ps", _vp), ("dst_stride", _i64),
                ("dst_t", _vp), ("ps_t", _vp), ("dst_t_stride", _i64), ("scale", _vp), ("first_block", _i64), ("rows_a", _i32),
                ("rows_b", _i32), ("K", _i32), ("mode", _i32)]


# name -> argtypes; must list every function include/spgnn_hip.h declares (tests check this)
_TILE_FWD = [_i32p, _i64, _i32, _i32p, _i32p, _vp, _i64, _f32p, _f32p, _i64, _vp, _i64, _f32p, _vp, _i64, _f32p, _f32p, _i64, _i32, _i32,
             _f32, _i32, _f32, _u64, _vp, _f32, _u64, _i32, _i32, _vp]
_TILE_DST = [_i32p, _i64, _i32, _i32p, _i32p, _vp, _i64, _f32p, _f32p, _i64, _f32p, _vp, _i64, _vp, _i64, _vp, _i64, _f32p, _f32p, _i64,
             _f32p, _i64, _i32, _i32, _f32, _i32, _f32, _u64, _vp, _f32, _u64, _i32, _i32, _vp]
_TILE_SRC = [_i32p, _i64, _i32, _i32p, _i32p, _i32p, _i32p, _f32p, _f32p, _vp, _i64, _vp, _i64, _f32p, _i64, _f32p, _f32p, _f32p, _f32p,
             _i64, _i32, _i32, _f32, _u64, _vp, _vp]

SIGNATURES = {
    "spgnn_gemm_nt_skinny": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _i64, _i64, _i64, _f32p, _i32, _f32p, _f32p, _f32p, _i32, _vp],
    "spgnn_act_bwd_proj_wgrad_blocks": [_i64],
    "spgnn_act_bwd_proj_wgrad": [_f32p, _i64, _i32, _f32p, _i64, _f32p, _i64, _f32p, _i64, _f32p, _f32p, _i64, _i32, _i32, _i32, _vp],
    "spgnn_gat_tile_supported": [_i32, _i32, _i32, _i32],
    "spgnn_gat_fwd_tile": _TILE_FWD, "spgnn_gat_fwd_tile_bf16": _TILE_FWD,
    "spgnn_gat_bwd_dst_tile": _TILE_DST, "spgnn_gat_bwd_dst_tile_bf16": _TILE_DST,
    "spgnn_gat_bwd_src_tile": _TILE_SRC, "spgnn_gat_bwd_src_tile_bf16": _TILE_SRC,
    "spgnn_sum_partials_multi": [C.POINTER(SumJob), _i32, _vp],
    "spgnn_weight_prep_blocks": [_i32, _i64, _i64],
    "spgnn_weight_prep": [_vp, _i32, _i64, _vp, _vp],
    "spgnn_build_csc_count": [_vp, _vp, _vp, _i64, _i32p, _i32p, _vp],
    "spgnn_ell_rows": [_i32p, _i32p, _vp, _i64, _i64, _i32p, _vp, _vp],
    "spgnn_copy_pad_i32": [_vp, _vp],
    "spgnn_build_csc": [_vp, _vp, _vp, _i64, _vp, _vp, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i64, _i64, _vp],
    "spgnn_lspe_supported": [_i32],
    "spgnn_lspe_fwd": [_i32p, _i32p, C.POINTER(LspeFwdGroup), _f32p, _i64, _f32, _u64, _f32p, _i64, _f32, _u64, _f32p, _f32p, _i64, _i64,
                       _i32, _vp, _vp],
    "spgnn_lspe_bwd_dst": [_i32p, _i32p, C.POINTER(LspeBwdDstGroup), _f32p, _i64, _f32p, _i64, _f32p, _i64, _f32, _u64, _f32p, _i64, _f32,
                           _u64, _i64, _i64, _i32, _vp, _vp],
    "spgnn_lspe_bwd_src": [_i32p, _i32p, _i32p, C.POINTER(LspeBwdSrcGroup), _i64, _i64, _i32, _vp, _vp],
    "spgnn_abi_version": [],
    "spgnn_last_error": [],
    "spgnn_gat_fwd": [_i32p, _i32p, _i32p, _f32p, _i64, _f32p, _f32p, _i64, _f32p, _i64, _f32p, _f32p, _i64, _f32p, _i64,
                      _f32p, _i64, _i64, _i32, _i32, _f32, _i32, _f32, _u64, _vp, _f32, _u64, _i32, _i32, _f32p, _vp],
    "spgnn_gat_can_fuse_mean": [_i32, _i32],
    "spgnn_gat_bwd_dst": [_i32p, _i32p, _i32p, _f32p, _i64, _f32p, _f32p, _i64, _f32p, _f32p, _i64, _i32, _f32p, _i64,
                          _f32p, _i64, _f32p, _f32p, _i64, _f32p, _i64, _i64, _i32, _i32, _f32, _i32, _f32, _u64, _vp, _f32, _u64, _i32, _i32, _vp],
    "spgnn_gat_bwd_src": [_i32p, _i32p, _i32p, _i32p, _i32p, _f32p, _f32p, _f32p, _i64, _f32p, _i64, _f32p, _i64, _f32p, _f32p, _f32p, _f32p,
                          _i64, _i64, _i32, _i32, _f32, _u64, _vp, _vp],
    "spgnn_scores_fwd": [_f32p, _i64, _f32p, _i32, _f32p, _i64, _f32p, _f32p, _i64, _i32, _i32, _vp],
    "spgnn_scale_from_partials": [_f32p, _i64, _f32, _f32p, _vp, _vp],
    "spgnn_scores_bwd_w": [_f32p, _i64, _f32p, _i64, _f32p, _i32, _i32, _i64, _i32, _i32, _vp],
    "spgnn_scores_bwd_w_pair": [_f32p, _i64, _f32p, _i64, _f32p, _i32, _i32, _i32, _i32, _f32p, _i64, _f32p, _i64, _f32p, _i32, _i32, _i32,
                                _i32, _i64, _vp],
    "spgnn_scores_bwd_x": [_f32p, _i64, _f32p, _i32, _f32p, _i64, _i32, _i64, _i32, _i32, _vp],
    "spgnn_spmm_sum": [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _f32p, _i32, _f32p, _i64, _i64, _i64, _i32, _f32p, _vp],
    "spgnn_spmm_sum_dropout": [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _f32p, _f32p, _i32, _f32p, _i64, _i64, _i64, _i32, _f32p, _f32, _u64,
                               _vp, _vp],
    "spgnn_spmm_max_fwd": [_i32p, _i32p, _f32p, _i64, _f32p, _i64, _i32p, _i64, _i64, _i64, _i32, _vp],
    "spgnn_spmm_max_bwd": [_i32p, _i32p, _i32p, _f32p, _i64, _i32p, _i64, _f32p, _i64, _i64, _i64, _i32, _vp],
    "spgnn_spmm_max_u8_supported": [_i32],
    "spgnn_spmm_max_fwd_u8": [_i32p, _i32p, _f32p, _i64, _f32p, _i64, _vp, _i64, _i64, _i64, _i32, _vp],
    "spgnn_spmm_max_bwd_u8": [_i32p, _i32p, _i32p, _i32p, _f32p, _i64, _vp, _i64, _f32p, _i64, _i64, _i64, _i32, _vp],
    "spgnn_spmm_max_bwd_u8_relu": [_i32p, _i32p, _i32p, _i32p, _f32p, _i64, _vp, _i64, _f32p, _i64, _f32p, _i64, _f32p, _i64, _i64, _i32, _vp],
    "spgnn_gemm_nt": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _i64, _i64, _i64, _f32p, _f32p, _f32p, _i64, _f32p, _i64, _i32,
                      _f32p, _i32, _f32p, _f32p, _f32p, _i32, _i32, _vp],
    "spgnn_gemm_nt_tile": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _i64, _i64, _i64, _f32p, _f32p, _f32p, _i64, _f32p, _i64, _i32,
                           _f32p, _i32, _f32p, _f32p, _f32p, _i32, _i32, _i32, _vp],
    "spgnn_gemm_nt_headmean": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _i64, _i64, _i64, _f32p, _f32p, _f32p, _i32, _f32p, _i64,
                               _f32p, _i64, _i32, _vp],
    "spgnn_gemm_nt_add": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _i64, _i64, _i64, _f32p, _f32p, _f32p, _i32, _f32p, _i64, _i32, _vp],
    "spgnn_presplit": [_f32p, _i64, _f32p, _f32p, _f32p, _i64, _i64, _i64, _f32p, _f32p, _i64, _i64, _i64, _f32p, _i32, _vp],
    "spgnn_gat_agg_supported": [_i32, _i32],
    "spgnn_gat_agg_fwd": [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _i64, _f32p, _f32p, _i64, _i32, _i32, _f32p, _i64, _i64,
                          _i32, _i32, _f32, _f32, _u64, _vp, _vp],
    "spgnn_gat_agg_bwd_dst": [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _i64, _f32p, _f32p, _i64, _i32, _f32p, _f32p, _i64,
                              _i64, _i64, _i32, _i32, _f32, _f32, _u64, _vp, _vp],
    "spgnn_gat_agg_bwd_src": [_i32p, _i32p, _i32p, _f32p, _f32p, _f32p, _i64, _i32, _i32, _f32p, _f32p, _i64, _f32p, _i64,
                              _f32p, _i64, _i64, _i64, _i32, _i32, _f32, _u64, _vp, _vp],
    "spgnn_gat_agg_bwd_dst_rows": [_i32p, _i32p, _f32p, _i64, _f32p, _f32p, _i64, _f32p, _f32p, _i64, _i32, _vp, _f32p, _f32p, _i64,
                                   _i64, _i64, _i32, _i32, _f32, _f32, _u64, _vp, _vp],
    "spgnn_gat_agg_bwd_src_rows": [_i32p, _i32p, _i32p, _f32p, _f32p, _f32p, _i64, _i32, _i32, _vp, _f32p, _f32p, _i64, _f32p, _i64,
                                   _f32p, _i64, _i64, _i64, _i32, _i32, _f32, _u64, _vp, _vp],
    "spgnn_fold_scores_fwd": [_f32p, _i64, _f32p, _f32p, _f32p, _i32, _i32, _i32, _i32, _vp],
    "spgnn_fold_scores_bwd": [_f32p, _i64, _f32p, _f32p, _f32p, _i32, _f32p, _i64, _f32p, _f32p, _i32, _i32, _i32, _vp],
    "spgnn_cat_dropout": [_f32p, _i64, _f32p, _i64, _i64, _i32, _i32, _i32, _f32, _u64, _vp, _i32, _f32p, _vp],
    "spgnn_cat_dropout_blocks": [_i64, _i32],
    "spgnn_scores_from_parts": [_f32p, _f32p, _i64, _i64, _i32, _i32, _vp],
    "spgnn_masked_ce": [_f32p, _i64, _vp, _f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i32, _vp],
    "spgnn_masked_ce_step": [_f32p, _i64, _vp, _f32p, _u64, _vp, _f32p, _f32p, _f32p, _f32p, _vp, _f32p, _i64, _f32p, _f32p, _i64, _i32, _vp],
    "spgnn_loss_rows": [_f32p, _u64, _vp, _f32p, _i64, _vp, _i32, _vp, _vp, _vp, _vp],
    "spgnn_gather_rows": [_f32p, _i64, _vp, _vp, _i64, _i32, _f32p, _i64, _vp],
    "spgnn_expand_rows": [_f32p, _i64, _vp, _i64, _i32, _f32p, _i64, _vp],
    "spgnn_masked_ce_rows": [_f32p, _i64, _vp, _vp, _vp, _f32p, _f32p, _f32p, _vp, _f32p, _i64, _f32p, _f32p, _i64, _i32, _vp],
    "spgnn_arena_load": [_vp, _vp, _vp],
    "spgnn_ell_rows_both": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp],
    "spgnn_classifier_ce_rows_per_block": [_i64],
    "spgnn_classifier_ce_partial_slices": [_i64, _i32, _i32],
    "spgnn_classifier_ce": [_f32p, _i64, _f32p, _i32, _f32p, _vp, _f32p, _u64, _vp, _f32p, _f32p, _vp, _f32p, _i64, _f32p, _i64, _f32p, _f32p,
                            _f32p, _vp, _f32p, _f32p, _i64, _i32, _i32, _vp],
    "spgnn_classifier_ce_bf16": [_vp, _i64, _f32p, _i32, _f32p, _vp, _f32p, _u64, _vp, _f32p, _f32p, _vp, _f32p, _i64, _f32p, _i64, _f32p, _f32p,
                                 _f32p, _vp, _f32p, _f32p, _i64, _i32, _i32, _vp],
    "spgnn_sgd_momentum_step_mean": [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i64, _f32, _f32, _f32, _i32, _vp],
    "spgnn_sgd_momentum_step_guarded": [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _vp, _i64, _f32, _f32, _f32, _i32, _vp],
    "spgnn_masked_ce_step_flagged": [_f32p, _i64, _vp, _f32p, _u64, _vp, _f32p, _vp, _f32p, _f32p, _f32p, _vp, _f32p, _i64, _f32p, _f32p,
                                     _i64, _i32, _vp],
    "spgnn_step_begin": [_vp, _f32p, _i32, _vp, _vp],
    "spgnn_linear_mean_fold_fwd": [_f32p, _i64, _f32p, _i64, _f32p, _f32p, _i64, _f32p, _i32, _i32, _i32, _i32, _f32p, _i64, _vp, _i64,
                                   _f32p, _f32p, _i64, _f32p, _f32p, _f32p, _vp, _i32, _vp],
    "spgnn_linear_mean_fold_workspace": [_i32, _i32],
    "spgnn_linear_mean_fold_bwd": [_f32p, _i64, _f32p, _f32p, _i64, _f32p, _i64, _f32p, _i32, _i32, _i32, _i32, _f32p, _i64, _f32p, _i64,
                                   _f32p, _f32p, _i64, _i32, _vp],
    "spgnn_gemm_nt_pair": [_vp, _vp, _i32, _vp],
    "spgnn_gemm_nt_problem_run": [_vp, _i32, _vp],
    "spgnn_gemm_tn_pair": [_vp, _vp, _vp],
    "spgnn_gemm_tn_problem_run": [_vp, _vp],
    "spgnn_gemm_tn_tile_rows": [_i64, _i64, _i64, _i32],
    "spgnn_scores_bwd_w_multi": [_vp, _i32, _i64, _i32, _vp],
    "spgnn_sample_neighbors": [_i32p, _i32p, _i32p, _i64, _vp, _i64, _i32, _i32p, _u64, _i32p, _i32p, _i32p, _i32p, _vp],
    "spgnn_block_relabel": [_i32p, _i32p, _i32p, _i64, _i64, _i32p, _i64, _vp, _i32p, _vp],
    "spgnn_head_mean": [_f32p, _i64, _f32p, _i64, _i64, _i32, _i32, _vp],
    "spgnn_act_bwd_proj": [_f32p, _i64, _i32, _f32p, _i64, _f32p, _i64, _f32p, _i64, _f32p, _i64, _i32, _i32, _i32, _vp],
    "spgnn_act_bwd_proj_rows": [_f32p, _i64, _i32, _f32p, _i64, _f32p, _i64, _vp, _vp, _f32p, _i64, _f32p, _i64, _i32, _i32, _i32, _vp],
    "spgnn_act_bwd_proj_blocks": [_i64],
    "spgnn_act_bwd_dropout": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _f32p, _i64, _i32, _i32, _f32, _u64, _vp, _vp],
    "spgnn_act_bwd_dropped": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _f32p, _i64, _i32, _i32, _f32, _u64, _vp, _vp],
    "spgnn_act_bwd_colsum": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _f32p, _f32p, _i64, _i32, _i32, _f32, _u64, _vp, _f32p, _i64, _vp],
    "spgnn_act_bwd_colsum_blocks": [_i64, _i32],
    "spgnn_act_bwd": [_f32p, _i64, _i32, _f32p, _i64, _f32p, _i64, _f32p, _i64, _i32, _i32, _i32, _vp],
    "spgnn_gemm_tn": [_f32p, _i64, _f32p, _i64, _f32p, _i64, _i64, _i32, _i64, _i64, _i64, _f32p, _f32p, _f32p, _i64, _i64, _vp],
    "spgnn_pow2_scale": [_f32p, _i64, _i64, _i64, _f32p, _f32p, _i32, _vp],
    "spgnn_sum_partials": [_f32p, _i64, _i32, _i64, _f32p, _vp],
    "spgnn_sum_partials_blockdiag": [_f32p, _i64, _i32, _i32, _i32, _i32, _f32p, _vp],
    "spgnn_sum_partials_compact": [_f32p, _i64, _i32, _i32, _i32, _i64, _f32p, _i64, _f32p, _i64, _i32, _f32p, _i32, _vp],
    "spgnn_weight_cat": [_f32p, _i64, _i32, _f32p, _i64, _i32, _i32, _f32p, _i64, _f32p, _i64, _f32p, _vp],
    "spgnn_weight_cat_partials": [_i32, _i32, _i64, _i64],
    "spgnn_tree_distance_encoding": [_i32p, _i32p, _vp, _i32p, _i32, _f32p, _i64, _i32p, _i64, _i64, _vp],
    "spgnn_sgd_momentum_step": [_f32p, _f32p, _f32p, _f32p, _f32p, _i64, _f32, _f32, _f32, _i32, _vp],
    "spgnn_tree_anchors_workspace": [_i64, _i32, _i64],
    "spgnn_tree_anchors": [_f32p, _i64, _i32p, _i32p, _vp, _i64, _i64, _i64, _i32, _i32, _i32p, _vp, _vp],
    # bf16-storage path
    "spgnn_gat_fwd_bf16": [_i32p, _i32p, _i32p, _vp, _i64, _f32p, _f32p, _i64, _vp, _i64, _f32p, _vp, _i64, _f32p, _i64,
                           _f32p, _i64, _i64, _i32, _i32, _f32, _i32, _f32, _u64, _vp, _f32, _u64, _i32, _i32, _vp],
    "spgnn_gat_bwd_dst_bf16": [_i32p, _i32p, _i32p, _vp, _i64, _f32p, _f32p, _i64, _f32p, _vp, _i64, _i32, _vp, _i64,
                               _vp, _i64, _f32p, _f32p, _i64, _i64, _i64, _i32, _i32, _f32, _i32, _f32, _u64, _vp, _f32, _u64, _i32, _i32, _vp],
    "spgnn_gat_bwd_src_bf16": [_i32p, _i32p, _i32p, _i32p, _i32p, _f32p, _f32p, _vp, _i64, _vp, _i64, _f32p, _i64, _f32p, _f32p, _f32p,
                               _i64, _i64, _i32, _i32, _f32, _u64, _vp, _vp],
    "spgnn_gat_agg_fwd_bf16": [_i32p, _i32p, _vp, _i64, _f32p, _f32p, _i64, _f32p, _vp, _i64, _i32, _i32, _i64, _i64,
                               _i32, _i32, _f32, _f32, _u64, _vp, _vp],
    "spgnn_gat_agg_bwd_dst_bf16": [_i32p, _i32p, _vp, _i64, _f32p, _f32p, _i64, _f32p, _vp, _i64, _i32, _f32p, _f32p, _i64,
                                   _i64, _i64, _i32, _i32, _f32, _f32, _u64, _vp, _vp],
    "spgnn_gat_agg_bwd_src_bf16": [_i32p, _i32p, _i32p, _f32p, _f32p, _vp, _i64, _i32, _i32, _f32p, _f32p, _i64, _vp, _i64,
                                   _f32p, _i64, _i64, _i64, _i32, _i32, _f32, _u64, _vp, _vp],
    "spgnn_scores_fwd_bf16": [_vp, _i64, _f32p, _i32, _f32p, _i64, _f32p, _i64, _i32, _i32, _vp],
    "spgnn_scores_bwd_x_bf16": [_f32p, _i64, _f32p, _i32, _vp, _i64, _i32, _i64, _i32, _i32, _vp],
    "spgnn_scores_bwd_w_bf16": [_f32p, _i64, _vp, _i64, _f32p, _i32, _i32, _i64, _i32, _i32, _vp],
    "spgnn_cat_dropout_bf16": [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _f32, _u64, _vp, _i32, _vp],
    "spgnn_gemm_nt_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _i64, _i64, _f32p, _i32, _f32p, _f32p, _f32p, _i32, _vp],
    "spgnn_gemm_nt_bf16_tile": [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _i64, _i64, _f32p, _i32, _f32p, _f32p, _f32p, _i32, _i32, _vp],
    "spgnn_gemm_nt_bf16_scores": [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _i64, _i64, _f32p, _f32p, _f32p, _i32, _i32, _vp],
    "spgnn_gemm_tn_bf16": [_vp, _i64, _vp, _i64, _f32p, _i64, _i64, _i32, _i64, _i64, _i64, _f32p, _i64, _i64, _vp],
    "spgnn_weight_cat_bf16": [_f32p, _i64, _i32, _f32p, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp],
    "spgnn_weight_cat_bf16_blocks": [_i32, _i32, _i64, _i64, _vp],
    "spgnn_weight_cat_bf16_multi": [_vp, _i32, _i32, _vp],
    "spgnn_cast_rows_bf16": [_f32p, _i64, _i64, _i32, _vp, _i64, _vp],
}

_lib: Optional[C.CDLL] = None


class SpgnnLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load and bind the library once.  Raises SpgnnLibraryError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpgnnLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or spgnn_amd/csrc/build.py). "
            f"There is no CPU fallback for the message-passing ops.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SpgnnLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "spgnn_last_error" else C.c_int64 if name in ("spgnn_cat_dropout_blocks", "spgnn_weight_cat_partials", "spgnn_tree_anchors_workspace", "spgnn_weight_prep_blocks", "spgnn_linear_mean_fold_workspace", "spgnn_classifier_ce_partial_slices") else C.c_int
    ver = lib.spgnn_abi_version()
    if ver != ABI_VERSION:
        raise SpgnnLibraryError(f"{LIB_PATH} has ABI version {ver}, python side expects {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().spgnn_last_error()
        raise RuntimeError(f"{what} failed with code {code}: {msg.decode() if msg else ''}")
