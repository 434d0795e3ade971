"""ctypes binding of libgvd_hip.so (C-ABI declared in include/gvd_hip.h).

The product path has NO fallback: if the library is missing or a call returns non-zero, a
`GvdHipError` is raised.  torch is used here only to obtain device pointers and the current HIP stream.
"""
import ctypes as C
import os

import torch

from . import build as _build


class GvdHipError(RuntimeError):
    pass


c_f32p = C.c_void_p
c_u8p = C.c_void_p
c_i64p = C.c_void_p


class GemmSeg(C.Structure):
    _fields_ = [('A', c_f32p), ('lda', C.c_int64), ('a_batch_stride', C.c_int64),
                ('W', c_f32p), ('ldw', C.c_int64), ('w_batch_stride', C.c_int64),
                ('K', C.c_int)]


class GemmArgs(C.Structure):
    _fields_ = [('seg', GemmSeg * 3), ('nseg', C.c_int),
                ('nbias', c_f32p), ('nbias2', c_f32p),
                ('mbias', c_f32p), ('mbias_batch_stride', C.c_int64),
                ('rowbias', c_f32p), ('rowbias_ld', C.c_int64), ('rowbias_batch_stride', C.c_int64),
                ('mask', c_u8p), ('mask_ldm', C.c_int64), ('mask_batch_stride', C.c_int64),
                ('C', c_f32p), ('ldc', C.c_int64), ('c_batch_stride', C.c_int64),
                ('M', C.c_int), ('N', C.c_int), ('batch', C.c_int), ('act', C.c_int), ('m_dev', C.c_void_p),
                ('a_row_map', C.c_void_p), ('a_src_rows', C.c_int64),
                ('a_kstrided', C.c_int), ('w_kstrided', C.c_int),
                ('batch_inner', C.c_int), ('a_inner_stride', C.c_int64), ('w_inner_stride', C.c_int64),
                ('c_inner_stride', C.c_int64)]


class LstmArgs(C.Structure):
    _fields_ = [('seg', GemmSeg * 3), ('nseg', C.c_int),
                ('b_ih', c_f32p), ('b_hh', c_f32p),
                ('rowbias', c_f32p), ('rowbias_ld', C.c_int64),
                ('c_prev', c_f32p), ('ldc_prev', C.c_int64),
                ('h_out', c_f32p), ('ldh', C.c_int64),
                ('c_out', c_f32p), ('ldc_out', C.c_int64),
                ('gates_out', c_f32p), ('ldg', C.c_int64),
                ('B', C.c_int), ('H', C.c_int)]


class AttnSide(C.Structure):
    _fields_ = [('feats', c_f32p), ('p_feats', c_f32p), ('q', c_f32p), ('ldq', C.c_int64),
                ('w', c_f32p), ('alpha_bias', c_f32p),
                ('att_mask', c_u8p), ('ld_att_mask', C.c_int64),
                ('pnt_mask', c_u8p), ('ld_pnt_mask', C.c_int64),
                ('logits_out', c_f32p), ('ld_logits', C.c_int64),
                ('scores_out', c_f32p), ('ld_scores', C.c_int64),
                ('row_map', C.c_void_p),
                ('N', C.c_int), ('group', C.c_int), ('score_mode', C.c_int)]


class GreedyArgs(C.Structure):
    _fields_ = [('fc', c_f32p), ('conv', c_f32p), ('p_conv', c_f32p), ('pool', c_f32p), ('p_pool', c_f32p),
                ('pnt_mask', c_u8p), ('pool_row_map', C.c_void_p), ('embed', c_f32p),
                ('att_w_ih', c_f32p), ('att_w_hh', c_f32p), ('att_b_ih', c_f32p), ('att_b_hh', c_f32p),
                ('lang_w_ih', c_f32p), ('lang_w_hh', c_f32p), ('lang_b_ih', c_f32p), ('lang_b_hh', c_f32p),
                ('att1_h2att_w', c_f32p), ('att1_h2att_b', c_f32p), ('att1_alpha_w', c_f32p), ('att1_alpha_b', c_f32p),
                ('att2_h2att_w', c_f32p), ('att2_h2att_b', c_f32p), ('att2_alpha_w', c_f32p), ('att2_alpha_b', c_f32p),
                ('logit_w', c_f32p), ('logit_b', c_f32p),
                ('B', C.c_int), ('Ft', C.c_int), ('R', C.c_int), ('H', C.c_int), ('A', C.c_int), ('E', C.c_int),
                ('V', C.c_int), ('L', C.c_int), ('unk_idx', C.c_int), ('no_persistent', C.c_int),
                ('seq', c_i64p), ('seq_logprobs', c_f32p), ('att2_weights', c_f32p), ('workspace', C.c_void_p),
                ('prof', C.c_void_p), ('status', C.c_void_p), ('trace', C.c_void_p), ('att_input_mode', C.c_int),
                ('region_attn_mode', C.c_int)]


class BeamStepArgs(C.Structure):
    _fields_ = [('ys', c_f32p), ('ix', c_i64p), ('sums', c_f32p), ('att2_ind', c_i64p),
                ('beam_seq', c_i64p), ('beam_lps', c_f32p), ('beam_att', c_i64p),
                ('best_p', c_f32p), ('best_seq', c_i64p), ('best_lps', c_f32p), ('best_vix', c_i64p),
                ('parent', c_i64p), ('word', c_i64p),
                ('B', C.c_int), ('K', C.c_int), ('L', C.c_int), ('t', C.c_int)]


DX_MAX_GROUPS = 4          # GVD_DX_MAX_GROUPS


class DxGroup(C.Structure):
    """gvd_dx_group: one product out[M, ncols] = A[M, Kred] . W[Kred, ncols] (+ addend) of a gvd_gemm_dx_small_f32 launch."""
    _fields_ = [('A', c_f32p), ('lda', C.c_int64), ('W', c_f32p), ('ldw', C.c_int64), ('Kred', C.c_int), ('ncols', C.c_int),
                ('out', c_f32p), ('ldo', C.c_int64), ('addend', c_f32p), ('ld_add', C.c_int64)]


OPT_MAX_TENSORS = 32       # GVD_OPT_MAX_TENSORS
OPT_CHUNK = 16384          # GVD_OPT_CHUNK


class OptGroup(C.Structure):
    """gvd_opt_group: up to 32 parameter tensors of one optimiser launch, passed by value."""
    _fields_ = [('p', C.c_void_p * OPT_MAX_TENSORS), ('g', C.c_void_p * OPT_MAX_TENSORS),
                ('m', C.c_void_p * OPT_MAX_TENSORS), ('v', C.c_void_p * OPT_MAX_TENSORS),
                ('n', C.c_int64 * OPT_MAX_TENSORS), ('chunk0', C.c_int * (OPT_MAX_TENSORS + 1)),
                ('lr', C.c_float * OPT_MAX_TENSORS), ('bc1', C.c_float * OPT_MAX_TENSORS),
                ('bc2_sqrt', C.c_float * OPT_MAX_TENSORS), ('vec_ok', C.c_uint8 * OPT_MAX_TENSORS),
                ('count', C.c_int), ('part0', C.c_int)]


# every symbol include/gvd_hip.h declares: (restype, argtypes)
_SIG = {
    'gvd_version': (C.c_char_p, []),
    'gvd_abi_version': (C.c_int, []),
    'gvd_prof_create': (C.c_void_p, [C.c_int]),
    'gvd_prof_destroy': (None, [C.c_void_p]),
    'gvd_prof_reset': (None, [C.c_void_p]),
    'gvd_prof_read': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    'gvd_attn_fwd_prof': (C.c_int, [C.POINTER(AttnSide), C.POINTER(AttnSide), C.c_int, C.c_int, C.c_int,
                                    c_f32p, C.c_int64, c_f32p, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'gvd_gemm_nt_f32': (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    'gvd_gemm_prof_set': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'gvd_prof_read_pairs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'gvd_lstm_cell_fwd': (C.c_int, [C.POINTER(LstmArgs), C.c_void_p]),
    'gvd_tanh_fast_f32': (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    'gvd_attn_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'gvd_attn_fwd': (C.c_int, [C.POINTER(AttnSide), C.POINTER(AttnSide), C.c_int, C.c_int, C.c_int,
                               c_f32p, C.c_int64, c_f32p, c_f32p, C.c_void_p, C.c_void_p]),
    'gvd_add_layernorm_unbiased': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_int,
                                             C.c_float, C.c_void_p]),
    'gvd_add_layernorm_unbiased_bwd_parts': (C.c_int, [C.c_int64]),
    'gvd_add_layernorm_unbiased_bwd': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int,
                                                 C.c_float, C.c_void_p]),
    'gvd_add_layernorm_unbiased_drop': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_float,
                                                  C.c_float, C.c_uint64, C.c_void_p]),
    'gvd_add_layernorm_unbiased_drop_bwd': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64,
                                                      C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p]),
    'gvd_flash_attn_train_fwd_f32': (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, C.c_float, C.c_uint64,
                                               C.c_void_p]),
    'gvd_enc_attn_bwd_maps': (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                        c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_uint64, C.c_void_p]),
    'gvd_enc_dropout_mask': (C.c_int, [c_u8p, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_void_p]),
    'gvd_region_feature_rows': (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, C.c_int64, c_u8p, C.c_int64, C.c_int64,
                                          c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_void_p, C.c_int, C.c_float,
                                          C.c_void_p]),
    'gvd_region_feature_rows_bwd': (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p, C.c_int, c_u8p, C.c_int64, C.c_int64,
                                              c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int64,
                                              C.c_int, C.c_float, C.c_void_p]),
    'gvd_flash_attn_padded_f32': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'gvd_flash_attn_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'gvd_compact_index': (C.c_int, [c_u8p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    c_f32p, c_u8p, C.c_void_p]),
    'gvd_gather_rows_f32': (C.c_int, [c_f32p, C.c_int64, C.c_void_p, c_f32p, C.c_int64, C.c_int, C.c_int64, C.c_void_p,
                                      C.c_void_p]),
    'gvd_fc_feature': (C.c_int, [c_f32p, c_i64p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                C.c_void_p]),
    'gvd_loc_features': (C.c_int, [c_f32p, C.c_void_p, C.c_void_p, c_f32p, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    'gvd_affine_relu_rows': (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_void_p]),
    'gvd_zero_rows_outside_window': (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'gvd_check_masked_rows_zero': (C.c_int, [c_f32p, C.c_int, c_u8p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p]),
    'gvd_grid_sync_words': (C.c_int, []),
    'gvd_gru_bidir_layer': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    'gvd_lstm_bidir_layer': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'gvd_gru_bwd_step': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'gvd_lstm_cell_bwd': (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p,
                                    C.c_int64, c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p, C.c_int64,
                                    C.c_void_p]),
    'gvd_attn_bwd_chunks': (C.c_int, [C.c_int, C.c_int]),
    'gvd_attn_bwd_step': (C.c_int, [C.POINTER(AttnSide), C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, c_f32p,
                                    C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p,
                                    c_f32p, c_f32p, C.c_void_p]),
    'gvd_attn_bwd_pfeats': (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, C.c_int64, c_f32p,
                                      C.c_int64, C.c_int64, c_f32p, C.c_int, c_f32p, C.c_int, C.c_void_p]),
    'gvd_logsoftmax_top2_embed': (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, c_i64p, C.c_int64,
                                            c_f32p, C.c_int64, c_f32p, C.c_int, c_f32p, C.c_int64, C.c_void_p]),
    'gvd_embed_relu': (C.c_int, [c_i64p, C.c_int64, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int64, C.c_void_p]),
    'gvd_logsoftmax_rows': (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, c_i64p, c_f32p, C.c_int,
                                      c_f32p, c_i64p, C.c_void_p]),
    'gvd_beam_step': (C.c_int, [C.POINTER(BeamStepArgs), C.c_void_p]),
    'gvd_greedy_workspace_bytes': (C.c_size_t, [C.c_int] * 7),
    'gvd_greedy_decode': (C.c_int, [C.POINTER(GreedyArgs), C.c_void_p]),
    'gvd_pread_rows': (C.c_int64, [C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    'gvd_npy_read_rows_f32': (C.c_int64, [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    'gvd_npy_read_batch_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    'gvd_zero_masked_rows': (C.c_int, [c_f32p, C.c_int64, C.c_int, c_u8p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    'gvd_iou_targets': (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, c_u8p, c_u8p, C.c_int, C.c_int, C.c_int,
                                  c_f32p, c_i64p, C.c_void_p]),
    'gvd_step_targets': (C.c_int, [c_f32p, c_u8p, c_u8p, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   c_f32p, c_u8p, C.c_void_p]),
    'gvd_cls_loss': (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_i64p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p,
                               C.c_void_p]),
    'gvd_masked_lsm_loss': (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, c_f32p,
                                      C.c_void_p]),
    'gvd_dropout_rows': (C.c_int, [c_f32p, c_f32p, C.c_int64, C.c_float, C.c_uint64, C.c_void_p]),
    'gvd_relu_dropout_bwd_parts': (C.c_int, [C.c_int64]),
    'gvd_relu_dropout_bwd_colsum': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    'gvd_sum_chunks_pair': (C.c_int, [c_f32p, C.c_int, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, C.c_void_p]),
    'gvd_grounder_fwd_f32': (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_f32p, C.c_int64, C.c_int64, c_f32p, C.c_int64, c_f32p,
                                       C.c_int64, C.c_int64, c_u8p, C.c_int64, C.c_int64, c_f32p, C.c_int64, C.c_int64,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'gvd_rows_contract_f32': (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_u8p, C.c_int64, C.c_int64, c_f32p, c_f32p, C.c_int64,
                                        C.c_int64, c_f32p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    'gvd_rank_update_f32': (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_u8p, C.c_int64, C.c_int64, c_f32p, C.c_int64,
                                      C.c_int64, c_f32p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    'gvd_gemm_dx_small_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'gvd_gemm_dx_small_f32': (C.c_int, [C.POINTER(DxGroup), C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'gvd_softmax_rows': (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, C.c_int64, C.c_void_p]),
    'gvd_masked_lsm_bwd': (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                                     C.c_int64, C.c_void_p]),
    'gvd_nll_gather_bwd': (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, c_i64p, c_f32p, c_f32p, c_f32p, C.c_int64,
                                     C.c_void_p]),
    'gvd_masked_copy_rowsum': (C.c_int, [c_f32p, C.c_int64, C.c_int64, c_u8p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                         c_f32p, c_f32p, c_f32p, C.c_void_p]),
    'gvd_bn_parts': (C.c_int, [C.c_int64]),
    'gvd_bn_train_fwd': (C.c_int, [c_f32p, C.c_int64, C.c_int, c_f32p, c_f32p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p,
                                   c_f32p, c_f32p, C.c_void_p]),
    'gvd_bn_train_bwd': (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    'gvd_opt_chunk': (C.c_int, []),
    'gvd_sumsq_partials': (C.c_int, [C.c_void_p, c_f32p, C.c_void_p]),
    'gvd_clip_coef': (C.c_int, [c_f32p, C.c_int, C.c_float, c_f32p, C.c_void_p]),
    'gvd_adam_step': (C.c_int, [C.c_void_p, c_f32p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_void_p]),
}

EXPORTS = tuple(_SIG)
ABI_VERSION = 20        # must equal gvd_abi_version() of the loaded library (struct layouts above are part of the ABI)
_lib = None


def lib():
    """Load libgvd_hip.so (built in-tree by build.py).  Raises GvdHipError if it is missing."""
    global _lib
    if _lib is None:
        path = _build.library_path()
        if not _build.is_fresh():
            # not built yet (or sources changed): compile the HIP sources in-tree now (serialised across processes by a
            # file lock, published atomically).  This is still the HIP path — there is no CPU/eager fallback.  A stale
            # library is NEVER loaded silently: its struct layouts may no longer match the ctypes mirrors above.
            try:
                _build.build_library(force=False, verbose=False)
            except Exception as e:
                raise GvdHipError('libgvd_hip.so (%s) is %s and rebuilding it failed (%s): run `python '
                                  '__graft_entry__.py build`; there is no CPU/eager fallback for the hot path'
                                  % (path, 'stale (sources changed since it was built)' if os.path.exists(path)
                                     else 'not built', e))
        try:
            l = C.CDLL(path)
        except OSError as e:   # e.g. libamdhip64 missing
            raise GvdHipError('cannot load %s: %s' % (path, e))
        try:
            l.gvd_abi_version.restype = C.c_int
            got = l.gvd_abi_version()
        except AttributeError:
            got = None
        if got != ABI_VERSION:
            raise GvdHipError('%s reports ABI %r, the Python binding expects %d (stale build?)' % (path, got, ABI_VERSION))
        for name, (res, args) in _SIG.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                raise GvdHipError('libgvd_hip.so does not export %s (stale build?)' % name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise GvdHipError('%s failed with code %d (%s)' % (
            what, rc, 'unsupported shape/alignment' if rc == -1 else 'hipError_t'))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def require_cuda_f32(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GvdHipError('HIP hot path called with a non-GPU tensor; there is no CPU fallback')
        if t.dtype != torch.float32:
            raise GvdHipError('expected float32, got %s' % t.dtype)


class GemmProfile:
    """Arms gvd_gemm_prof_set for a `with` block: event pairs around every pipelined-GEMM launch + a device flop counter.
    .read() -> (total_ms, launches, flops).  Measurement only (bench.py roofline_mfma); process-global."""

    def __init__(self, max_pairs=8192):
        self.max_pairs = max_pairs
        self.timer = KernelTimer(max_pairs)
        self.flops = torch.zeros(1, dtype=torch.float64, device='cuda')
        self.rows = torch.zeros(max_pairs, dtype=torch.int32, device='cuda')

    def __enter__(self):
        self.timer.reset()
        self.flops.zero_()
        check(lib().gvd_gemm_prof_set(self.timer.h, ptr(self.flops), ptr(self.rows), self.max_pairs), 'gvd_gemm_prof_set')
        return self

    def __exit__(self, *exc):
        lib().gvd_gemm_prof_set(None, None, None, 0)
        return False

    def read(self):
        torch.cuda.synchronize()
        ms, n = self.timer.read()
        return ms, n, float(self.flops.item())

    def table(self, peak_tflops):
        """Per shape (live rows M, N, K, batch, operand forms): launches, mean microseconds, TFLOP/s and the fraction of
        `peak_tflops` - which products pull the aggregate down.  Call after read()."""
        n = self.max_pairs
        ms = (C.c_float * n)()
        tags = (C.c_int64 * (8 * n))()
        got = lib().gvd_prof_read_pairs(self.timer.h, ms, tags, n)
        if got < 0:
            raise GvdHipError('gvd_prof_read_pairs -> %d' % got)
        rows = self.rows[:got].tolist()
        agg = {}
        for i in range(got):
            M, N, K, batch, a_t, w_t, has_mdev, has_add = tags[8 * i:8 * i + 8]
            key = (rows[i] if has_mdev else M, N, K, batch, 'dW' if a_t else ('dX' if w_t else 'fwd'), bool(has_add))
            r = agg.setdefault(key, [0, 0.0])
            r[0] += 1
            r[1] += ms[i]
        out = []
        for (M, N, K, batch, form, add), (cnt, tot) in agg.items():
            fl = 2.0 * M * N * K * batch
            tf = fl * cnt / (tot * 1e-3) / 1e12 if tot > 0 else 0.0
            out.append({'M': M, 'N': N, 'K': K, 'batch': batch, 'form': form + ('+addend' if add else ''), 'launches': cnt,
                        'us': round(1e3 * tot / cnt, 1), 'tflops': round(tf, 1), 'frac': round(tf / peak_tflops, 3)})
        out.sort(key=lambda r: -r['us'] * r['launches'])
        return out


class KernelTimer:
    """Event pairs around the attention streaming kernel (gvd_prof_*): live kernel durations on its stream."""

    def __init__(self, max_pairs=4096):
        self.h = C.c_void_p(lib().gvd_prof_create(max_pairs))
        if not self.h:
            raise GvdHipError('gvd_prof_create failed')

    def reset(self):
        lib().gvd_prof_reset(self.h)

    def read(self):
        ms, n = C.c_float(0), C.c_int(0)
        check(lib().gvd_prof_read(self.h, C.byref(ms), C.byref(n)), 'gvd_prof_read')
        return ms.value, n.value

    def __del__(self):
        try:
            lib().gvd_prof_destroy(self.h)
        except Exception:
            pass
