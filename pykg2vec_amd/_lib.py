"""ctypes binding of libkge_hip.so (C ABI in include/kge_hip.h).

The HIP library IS the product path: if it is missing or a call fails this module raises -- there is no
CPU / PyTorch fallback anywhere in `pykg2vec_amd`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KGE_HIP_LIB", os.path.join(_HERE, "libkge_hip.so"))  # env override: A/B builds in dev tools

KGE_MAX_TABLES = 12
ABI_VERSION = 3

# enum kge_model
(TRANSE, TRANSH, TRANSD, ROTATE, RESCAL, NTN, DISTMULT, COMPLEX, ANALOGY, TRANSM, CP, SIMPLE, SIMPLE_IGNR,
 QUATE, TRANSR) = range(15)
FLAG_L1 = 1
OPT_SGD, OPT_ADAM, OPT_ADAGRAD, OPT_RMSPROP = range(4)
REG_NONE, REG_F2, REG_N3, REG_N3_ABS, REG_ID_F2, REG_ID_N3 = range(6)
LOSS_SLOTS, LOSS_STRIDE = 32, 32  # loss accumulators: float[32*32], total = sum of [k*32]

c_i64p = ctypes.c_void_p  # all device pointers travel as void*


class ModelDesc(ctypes.Structure):
    """struct kge_model_desc"""
    _fields_ = [
        ("model", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("tot_entity", ctypes.c_int64),
        ("tot_relation", ctypes.c_int64),
        ("dim", ctypes.c_int32),
        ("rel_dim", ctypes.c_int32),
        ("margin", ctypes.c_float),
        ("phase_scale", ctypes.c_float),
        ("tables", ctypes.c_void_p * KGE_MAX_TABLES),
        ("grads", ctypes.c_void_p * KGE_MAX_TABLES),
    ]


class PullLists(ctypes.Structure):
    """struct kge_pull_lists"""
    _fields_ = [("pc", ctypes.c_void_p), ("count", ctypes.c_void_p), ("bucket", ctypes.c_void_p), ("head", ctypes.c_void_p),
                ("next", ctypes.c_void_p), ("sdesc", ctypes.c_void_p), ("dbucket", ctypes.c_void_p)]


PULL_BUCKET = 16


class PullBatch(ctypes.Structure):
    """struct kge_pull_batch"""
    _fields_ = [("pairs", ctypes.c_void_p), ("items", ctypes.c_void_p), ("n_items", ctypes.c_int64), ("dense_skip", ctypes.c_void_p),
                ("inc", ctypes.c_void_p), ("inv", ctypes.c_void_p),
                ("multi", ctypes.c_void_p), ("n_multi", ctypes.c_int64), ("n_pairs", ctypes.c_int64)]


class PullDirection(ctypes.Structure):
    """struct kge_pull_direction"""
    _fields_ = [("codes", ctypes.c_void_p), ("recs", ctypes.c_void_p), ("n_pairs", ctypes.c_int64),
                ("lists_without_descriptors", ctypes.c_int32)]


class PullPlanC(ctypes.Structure):
    """struct kge_pull_plan"""
    _fields_ = [("model", ModelDesc * 2), ("hat", (ctypes.c_void_p * 2) * 2), ("norm", ctypes.c_void_p * 2),
                ("state1", ctypes.c_void_p * 2), ("state2", ctypes.c_void_p * 2), ("lists", PullLists * 2),
                ("batches", ctypes.POINTER(PullBatch)), ("n_batches", ctypes.c_int64), ("partials", ctypes.c_void_p),
                ("margin", ctypes.c_float), ("optimizer", ctypes.c_int32), ("lr", ctypes.c_float),
                ("bern_prob", ctypes.c_void_p), ("slots", ctypes.c_void_p), ("n_slots", ctypes.c_int64),
                ("seed", ctypes.c_uint64), ("draws_per_batch", ctypes.c_int64), ("loss", ctypes.c_void_p),
                ("direction", PullDirection)]

class TransXPlanC(ctypes.Structure):
    """struct kge_transx_plan"""
    _fields_ = [("model", ModelDesc), ("lists", PullLists * 2), ("batches", ctypes.POINTER(PullBatch)), ("n_batches", ctypes.c_int64),
                ("partials", ctypes.c_void_p), ("stage", ctypes.c_void_p), ("recs", ctypes.c_void_p), ("margin", ctypes.c_float),
                ("flat_param", ctypes.c_void_p), ("flat_grad", ctypes.c_void_p), ("flat_state1", ctypes.c_void_p),
                ("flat_state2", ctypes.c_void_p), ("flat_numel", ctypes.c_int64), ("optimizer", ctypes.c_int32), ("lr", ctypes.c_float),
                ("bern_prob", ctypes.c_void_p), ("slots", ctypes.c_void_p), ("n_slots", ctypes.c_int64), ("seed", ctypes.c_uint64),
                ("draws_per_batch", ctypes.c_int64), ("loss", ctypes.c_void_p)]


class OwnPlanC(ctypes.Structure):
    """struct kge_own_plan"""
    _fields_ = [("model", ModelDesc), ("state1", ctypes.c_void_p * KGE_MAX_TABLES), ("state2", ctypes.c_void_p * KGE_MAX_TABLES),
                ("lists", PullLists * 2), ("batches", ctypes.POINTER(PullBatch)), ("n_batches", ctypes.c_int64),
                ("partials", ctypes.c_void_p), ("optimizer", ctypes.c_int32), ("lr", ctypes.c_float), ("lmbda", ctypes.c_float),
                ("reg_type", ctypes.c_int32), ("bern_prob", ctypes.c_void_p), ("slots", ctypes.c_void_p), ("n_slots", ctypes.c_int64),
                ("seed", ctypes.c_uint64), ("draws_per_batch", ctypes.c_int64), ("loss", ctypes.c_void_p), ("stage", ctypes.c_void_p)]


class RescalStage(ctypes.Structure):
    """struct kge_rescal_stage"""
    _fields_ = [("gstage", ctypes.c_void_p), ("dsv", ctypes.c_void_p), ("count", ctypes.c_void_p), ("bucket", ctypes.c_void_p),
                ("head", ctypes.c_void_p), ("next", ctypes.c_void_p), ("cap", ctypes.c_int32)]


class StagedTable(ctypes.Structure):
    """struct kge_staged_table"""
    _fields_ = [("cls", ctypes.c_int32), ("site_a", ctypes.c_int32), ("site_b", ctypes.c_int32), ("dsite", ctypes.c_int32),
                ("flat_off", ctypes.c_int64), ("rows", ctypes.c_int64)]


class StagedStep(ctypes.Structure):
    """struct kge_staged_step"""
    _fields_ = [("param", ctypes.c_void_p), ("state1", ctypes.c_void_p), ("state2", ctypes.c_void_p),
                ("tables", StagedTable * 8), ("n_tables", ctypes.c_int32), ("dim", ctypes.c_int32),
                ("ent_off", ctypes.c_void_p), ("ent_inc", ctypes.c_void_p), ("rel_off", ctypes.c_void_p), ("rel_inc", ctypes.c_void_p),
                ("rel_chunk_off", ctypes.c_void_p), ("chunk_rel", ctypes.c_void_p), ("rel_partials", ctypes.c_void_p),
                ("n_chunks", ctypes.c_int32),
                ("dyn_count", ctypes.c_void_p), ("dyn_bucket", ctypes.c_void_p), ("dyn_head", ctypes.c_void_p),
                ("dyn_next", ctypes.c_void_p), ("dyn_cap", ctypes.c_int32),
                ("dyn_count_next", ctypes.c_void_p), ("dyn_head_next", ctypes.c_void_p),
                ("touched_ent", ctypes.c_void_p), ("n_touched_ent", ctypes.c_int32), ("touched_rel", ctypes.c_void_p),
                ("n_touched_rel", ctypes.c_int32), ("dyn_list", ctypes.c_void_p),
                ("dyn_scale", ctypes.c_void_p),
                ("stage", ctypes.c_void_p), ("stage_stride", ctypes.c_int64),
                ("static_slots", ctypes.c_int32), ("dynamic_slots", ctypes.c_int32),
                ("n_pos", ctypes.c_int64), ("n_neg", ctypes.c_int64),
                ("tot_entity", ctypes.c_int64), ("tot_relation", ctypes.c_int64), ("stage_spare", ctypes.c_int32)]


_SIGNATURES = {
    "kge_staged_step_bytes": (ctypes.c_size_t, []),
    "kge_train_pairwise_selfadv_sampled_staged": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                                  ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p,
                                                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64,
                                                                  ctypes.POINTER(StagedStep), ctypes.c_void_p, ctypes.c_void_p]),
    "kge_train_pointwise_logistic_sampled_staged": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                                    ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                                    ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_float,
                                                                    ctypes.c_int32, ctypes.POINTER(StagedStep), ctypes.c_void_p,
                                                                    ctypes.c_void_p]),
    "kge_optimizer_step_staged": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(StagedStep), ctypes.c_float, ctypes.c_int64, ctypes.c_void_p]),
    "kge_abi_version": (ctypes.c_int, []),
    "kge_set_debug": (ctypes.c_int, [ctypes.c_int32]),
    "kge_set_switch": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int32]),
    "kge_get_debug": (ctypes.c_int, []),
    "kge_check_ids": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "kge_debug_marker": (ctypes.c_int, [ctypes.c_int32, ctypes.c_void_p]),
    "kge_filter_csr_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64]),
    "kge_filter_csr_count": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_filter_csr_fill": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_last_error": (ctypes.c_char_p, []),
    "kge_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ModelDesc), ctypes.c_int64]),
    "kge_score_forward": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "kge_score_backward": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "kge_rescal_normalize": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]),
    "kge_rescal_normalize_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int32]),
    "kge_rescal_normalize_ws": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "kge_train_pairwise_hinge": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_train_pairwise_hinge_sampled": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_train_pairwise_selfadv": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_train_pairwise_selfadv_sampled": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_train_pointwise_logistic": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_train_pointwise_logistic_sampled": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                            ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                            ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_float, ctypes.c_int32,
                                                            ctypes.c_void_p, ctypes.c_void_p]),
    "kge_l2norm_reg": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_head_1n_rank_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32]),
    "kge_head_1n_rank": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_head_1n_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_head_1n_forward_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_head_1n_backward_workspace_bytes": (ctypes.c_size_t, []),
    "kge_head_1n_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64]
                             + [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]),
    "kge_head_1n_bce_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "kge_head_1n_bce": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float,
                                       ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_void_p] * 5),
    "kge_optimizer_step": (ctypes.c_int, [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_float, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_optimizer_step_rows": (ctypes.c_int, [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                                               ctypes.c_int64, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 4),
    "kge_optimizer_step_rownorm_ok": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int64]),
    "kge_optimizer_step_rownorm": (ctypes.c_int, [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int64,
                                                  ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "kge_optimizer_step_rows_rownorm": (ctypes.c_int, [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 4
                                        + [ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
                                        + [ctypes.c_void_p] * 3 + [ctypes.POINTER(RescalStage), ctypes.c_void_p, ctypes.c_size_t]
                                        + [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]),
    "kge_rescal_pair_step": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_float, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_rescal_pair_step_ok": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_int64]),
    "kge_rescal_stage_ok": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_int64]),
    "kge_rescal_pair_step_staged": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_float, ctypes.c_void_p,
                                                   ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(RescalStage), ctypes.c_void_p]),
    "kge_optimizer_step_rows_staged": (ctypes.c_int, [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                                                      ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 3 + [ctypes.POINTER(RescalStage), ctypes.c_void_p]),
    "kge_step_advance": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]),
    "kge_optimizer_step_advance": (ctypes.c_int, [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_float, ctypes.c_int32]
                                   + [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]),
    "kge_eval_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ModelDesc), ctypes.c_int64]),
    "kge_eval_ranks": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_eval_ranks_ties": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_eval_ranks_grouped_ties": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
                                    + [ctypes.c_void_p] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_eval_grouped_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(ModelDesc), ctypes.c_int64, ctypes.c_int64]),
    "kge_eval_ranks_grouped": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
                               + [ctypes.c_void_p] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_eval_sweep_scores": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_eval_sweep_scores_side": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_rank_from_scores": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_void_p]),
    "kge_triple_set_build": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "kge_corrupt": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64] + [ctypes.c_void_p] * 3 + [ctypes.c_void_p]),
    "kge_sample_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32] + [ctypes.c_void_p] * 6 + [ctypes.c_void_p, ctypes.c_void_p]),
    "kge_pull_partial_stride": (ctypes.c_int, [ctypes.c_int32]),
    "kge_pull_hat_stride": (ctypes.c_int, [ctypes.c_int32]),
    "kge_pull_groups_per_block": (ctypes.c_int, [ctypes.c_int32]),
    "kge_row_norms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_pull_sample": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.POINTER(PullLists), ctypes.c_void_p]),
    "kge_pull_lists_explicit": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.POINTER(PullLists), ctypes.c_void_p]),
    "kge_pull_step": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 8 + [ctypes.POINTER(PullLists), ctypes.c_void_p,
                                     ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float,
                                     ctypes.c_int32, ctypes.c_float, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(PullLists), ctypes.c_void_p,
                                     ctypes.POINTER(PullDirection), ctypes.c_void_p]),
    "kge_pull_direction_bytes": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t),
                                                ctypes.POINTER(ctypes.c_size_t)]),
    "kge_transx_groups_per_block": (ctypes.c_int, [ctypes.c_int32]),
    "kge_transx_partial_stride": (ctypes.c_int, [ctypes.c_int32]),
    "kge_transx_scratch_bytes": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_size_t),
                                                ctypes.POINTER(ctypes.c_size_t)]),
    "kge_transx_grad_step": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(PullLists), ctypes.c_void_p,
                                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                            ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(PullLists), ctypes.c_void_p, ctypes.c_void_p]),
    "kge_transx_plan_bytes": (ctypes.c_size_t, []),
    "kge_transx_run": (ctypes.c_int, [ctypes.POINTER(TransXPlanC), ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p]),
    "kge_own_groups_per_block": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32]),
    "kge_own_partial_stride": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32]),
    "kge_own_stage_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]),
    "kge_own_step": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(PullLists), ctypes.c_void_p,
                                    ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_float,
                                    ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(PullLists),
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "kge_own_apply": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                     ctypes.POINTER(PullLists), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_int64,
                                     ctypes.c_void_p]),
    "kge_own_plan_bytes": (ctypes.c_size_t, []),
    "kge_own_run": (ctypes.c_int, [ctypes.POINTER(OwnPlanC), ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                   ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p]),
    "kge_pull_index_geometry": (ctypes.c_int, [ctypes.c_int64] * 4 + [ctypes.c_int32] * 3 + [ctypes.POINTER(ctypes.c_int64)] * 3
                                + [ctypes.POINTER(ctypes.c_size_t)]),
    "kge_pull_index_build": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int64] * 6 + [ctypes.c_int32] * 3
                             + [ctypes.c_void_p] * 8 + [ctypes.c_size_t, ctypes.c_void_p]),
    "kge_pull_plan_bytes": (ctypes.c_size_t, []),
    "kge_pull_run": (ctypes.c_int, [ctypes.POINTER(PullPlanC), ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                    ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class KgeHipError(RuntimeError):
    pass


def load():
    """Load libkge_hip.so (once).  Raises if it has not been built: build with
    `python -c 'import __graft_entry__ as g; g.build()'` or `make -C pykg2vec_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KgeHipError("libkge_hip.so not found at %s -- the HIP extension is mandatory (no CPU fallback); "
                          "run __graft_entry__.build()" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.kge_abi_version() != ABI_VERSION:
        raise KgeHipError("libkge_hip.so ABI %d != expected %d" % (lib.kge_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().kge_last_error()
        raise KgeHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
