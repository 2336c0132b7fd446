"""ctypes binding of include/cdcmdr.h (the C-ABI of the HIP hot path).

The structures below mirror the header field for field; tests/test_abi.py parses the header and
checks field order and sizes against them.

There is NO fallback: if libcdcmdr.so cannot be loaded (and cannot be built), every op raises.
"""
import ctypes as C
import os

from . import build as _build

MAX_GROUPS = 32
MAX_TENSORS = 48
MAX_GATES = 16
MAX_SEL = 16
MAX_BN_SEGS = 24
SORT_MAX_B = 16384
SORT_MAX_ROWS = 32768
BN_ROWS_PER_BLOCK = 64
ROWDOT_PARTS = 256
PREC_BF16 = 0
PREC_F32 = 1

c_f = C.c_float
c_i32 = C.c_int32
c_i64 = C.c_int64
c_p = C.c_void_p


class AdamHP(C.Structure):
    _fields_ = [("lerp_w", c_f), ("beta2", c_f), ("one_minus_beta2", c_f), ("eps", c_f),
                ("weight_decay", c_f), ("l2_twice", c_f), ("step_scalars", c_p), ("n_scalars", c_i32), ("fast_replay", c_i32),
                ("inv_bc2", c_p), ("replay_tab", c_p), ("k1", c_f), ("k2", c_f), ("ik1", c_f), ("ik2", c_f), ("k1_lo", c_f), ("k2_lo", c_f)]


class LinGroup(C.Structure):
    _fields_ = [("x", c_p), ("ldx", c_i64), ("w", c_p), ("ldw", c_i64), ("bias", c_p),
                ("y", c_p), ("ldy", c_i64), ("M", c_i32), ("N", c_i32), ("K", c_i32), ("act_cols", c_i32),
                ("bn_partial", c_p), ("bn_col0", c_i32), ("bn_total_c", c_i32)]


class LinFwdArgs(C.Structure):
    _fields_ = [("n_groups", c_i32), ("relu", c_i32), ("drop_p", c_f), ("seed", C.c_uint64),
                ("seed_offset_dev", c_p), ("row_offsets", c_p), ("g", LinGroup * MAX_GROUPS)]


class BwdxSeg(C.Structure):
    _fields_ = [("dz", c_p), ("lddz", c_i64), ("w", c_p), ("ldw", c_i64), ("wt", c_p), ("ldwt", c_i64), ("N", c_i32), ("out", c_i32),
                ("wt_bf16", c_i32)]


class BwdxOut(C.Structure):
    _fields_ = [("dx", c_p), ("lddx", c_i64), ("mask_y", c_p), ("ldmask", c_i64), ("M", c_i32), ("K", c_i32),
                ("mask_cols", c_i32), ("accumulate", c_i32)]


class LinBwdxArgs(C.Structure):
    _fields_ = [("n_out", c_i32), ("n_seg", c_i32), ("mask_scale", c_f), ("row_offsets", c_p),
                ("o", BwdxOut * MAX_GROUPS), ("s", BwdxSeg * MAX_GROUPS)]


class TransposeItem(C.Structure):
    _fields_ = [("src", c_p), ("dst", c_p), ("rows", c_i32), ("cols", c_i32)]


class TransposeArgs(C.Structure):
    _fields_ = [("n", c_i32), ("pad_", c_i32), ("bf16_mask", C.c_uint64), ("t", TransposeItem * MAX_TENSORS)]


G2_MAX_OUT = 24
G2_MAX_SEG = 32


class G2Out(C.Structure):
    _fields_ = [("y", c_p), ("ldy", c_i64), ("yh", c_p), ("ldyh", c_i64), ("bias", c_p), ("mask", c_p), ("ldmask", c_i64),
                ("bn_partial", c_p), ("M", c_i32), ("N", c_i32), ("act_cols", c_i32), ("accumulate", c_i32), ("mask_bf16", c_i32),
                ("bn_col0", c_i32), ("bn_total_c", c_i32), ("stream_id", c_i32)]


class G2Seg(C.Structure):
    _fields_ = [("a", c_p), ("lda", c_i64), ("b", c_p), ("ldb", c_i64), ("Kr", c_i32), ("out", c_i32)]


class G2Args(C.Structure):
    _fields_ = [("n_out", c_i32), ("n_seg", c_i32), ("mode", c_i32), ("relu", c_i32), ("drop_p", c_f), ("mask_scale", c_f),
                ("tile_cfg", c_i32), ("pad_", c_i32), ("seed", C.c_uint64), ("seed_offset_dev", c_p), ("o", G2Out * G2_MAX_OUT), ("s", G2Seg * G2_MAX_SEG)]


class WShadowItem(C.Structure):
    _fields_ = [("src", c_p), ("dst_h", c_p), ("ld_h", c_i64), ("dst_t", c_p), ("ld_t", c_i64), ("rows", c_i32), ("cols", c_i32)]


class WShadowArgs(C.Structure):
    _fields_ = [("n", c_i32), ("pad_", c_i32), ("t", WShadowItem * MAX_TENSORS)]


class ShadowItem(C.Structure):
    _fields_ = [("src", c_p), ("ld_src", c_i64), ("dst", c_p), ("ld_dst", c_i64), ("rows", c_i64), ("cols", c_i32), ("pad_", c_i32)]


class ShadowArgs(C.Structure):
    _fields_ = [("n", c_i32), ("pad_", c_i32), ("t", ShadowItem * MAX_GROUPS)]


class BwdwGroup(C.Structure):
    _fields_ = [("dz", c_p), ("lddz", c_i64), ("x", c_p), ("ldx", c_i64), ("dw", c_p), ("lddw", c_i64),
                ("db", c_p), ("M", c_i32), ("N", c_i32), ("K", c_i32), ("accumulate", c_i32),
                ("dzh", c_p), ("lddzh", c_i64), ("xh", c_p), ("ldxh", c_i64)]


class LinBwdwArgs(C.Structure):
    _fields_ = [("n_groups", c_i32), ("split_k", c_i32), ("workspace", c_p), ("row_offsets", c_p), ("defer_reduce", c_i32),
                ("pad_", c_i32), ("g", BwdwGroup * MAX_GROUPS)]


class PoolFwdGate(C.Structure):
    _fields_ = [("logits", c_p), ("ld_logits", c_i64), ("out", c_p), ("ld_out", c_i64), ("probs", c_p),
                ("out_h", c_p), ("ld_out_h", c_i64), ("n_sel", c_i32), ("sel", c_i32 * MAX_SEL)]


class PoolFwdArgs(C.Structure):
    _fields_ = [("n_gates", c_i32), ("n_expert", c_i32), ("H", c_i32), ("B", c_i64), ("experts", c_p),
                ("ld_exp", c_i64), ("gate", PoolFwdGate * MAX_GATES)]


class PoolBwdGate(C.Structure):
    _fields_ = [("d_out", c_p), ("ld_dout", c_i64), ("probs", c_p), ("d_logits", c_p), ("ld_dlogits", c_i64),
                ("d_logits_h", c_p), ("ld_dlogits_h", c_i64), ("n_sel", c_i32), ("sel", c_i32 * MAX_SEL)]


class PoolBwdArgs(C.Structure):
    _fields_ = [("n_gates", c_i32), ("n_expert", c_i32), ("H", c_i32), ("B", c_i64), ("experts", c_p),
                ("ld_exp", c_i64), ("d_experts", c_p), ("ld_dexp", c_i64), ("mask_relu", c_i32),
                ("mask_scale", c_f), ("accumulate", c_i32), ("d_experts_h", c_p), ("ld_dexp_h", c_i64), ("gate", PoolBwdGate * MAX_GATES)]


MID_MAX_EXPERT, MID_MAX_GATE = 16, 8


class MidGate1(C.Structure):
    _fields_ = [("logits", c_p), ("ld_logits", c_i64), ("probs", c_p), ("pooled_h", c_p), ("ld_pooled_h", c_i64),
                ("n_sel", c_i32), ("sel", c_i32 * MAX_SEL), ("pad_", c_i32)]


class MidExpert2(C.Structure):
    _fields_ = [("w", c_p), ("ldw", c_i64), ("bias", c_p), ("src", c_i32), ("stream_id", c_i32)]


class MidGate2(C.Structure):
    _fields_ = [("w", c_p), ("ldw", c_i64), ("bias", c_p), ("probs", c_p), ("out", c_p), ("ld_out", c_i64),
                ("out_h", c_p), ("ld_out_h", c_i64), ("src", c_i32), ("n_sel", c_i32), ("sel", c_i32 * MAX_SEL)]


PAIR_MAX_EXPERT = 16
ADAM_CHUNK = 4096


class PairExpert(C.Structure):
    _fields_ = [("x", c_p), ("ldx", c_i64), ("w1", c_p), ("ldw1", c_i64), ("b1", c_p), ("w2", c_p), ("ldw2", c_i64), ("b2", c_p),
                ("h", c_p), ("ldh", c_i64), ("y", c_p), ("ldy", c_i64), ("yh", c_p), ("ldyh", c_i64),
                ("ws", c_p), ("ldws", c_i64), ("bs", c_p), ("ys", c_p), ("ldys", c_i64), ("ns", c_i32),
                ("stream1", c_i32), ("stream2", c_i32), ("pad_", c_i32)]


class ExpertPairArgs(C.Structure):
    _fields_ = [("n_expert", c_i32), ("M", c_i32), ("K1r", c_i32), ("H1", c_i32), ("H2", c_i32), ("relu", c_i32), ("drop_p", c_f),
                ("pad_", c_i32), ("seed1", C.c_uint64), ("seed2", C.c_uint64), ("seed_offset_dev", c_p),
                ("e", PairExpert * PAIR_MAX_EXPERT)]


class CgcMidFwdArgs(C.Structure):
    _fields_ = [("B", c_i64), ("H1", c_i32), ("H2", c_i32), ("n_exp1", c_i32), ("n_gate1", c_i32), ("n_exp2", c_i32), ("n_gate2", c_i32),
                ("ex1", c_p), ("ld_ex1", c_i64), ("ex2", c_p), ("ld_ex2", c_i64), ("relu", c_i32), ("drop_p", c_f),
                ("seed", C.c_uint64), ("seed_offset_dev", c_p),
                ("g1", MidGate1 * MID_MAX_GATE), ("e2", MidExpert2 * MID_MAX_EXPERT), ("g2", MidGate2 * MID_MAX_GATE)]


class MidBGate1(C.Structure):
    _fields_ = [("probs", c_p), ("d_logits", c_p), ("ld_dlogits", c_i64), ("d_logits_h", c_p), ("ld_dlogits_h", c_i64),
                ("n_sel", c_i32), ("sel", c_i32 * MAX_SEL), ("pad_", c_i32)]


class MidBExpert2(C.Structure):
    _fields_ = [("wt", c_p), ("ldwt", c_i64), ("src", c_i32), ("pad_", c_i32)]


class MidBGate2(C.Structure):
    _fields_ = [("d_out", c_p), ("ld_dout", c_i64), ("probs", c_p), ("d_logits", c_p), ("ld_dlogits", c_i64),
                ("d_logits_h", c_p), ("ld_dlogits_h", c_i64), ("wt", c_p), ("ldwt", c_i64), ("src", c_i32), ("n_sel", c_i32),
                ("sel", c_i32 * MAX_SEL)]


class CgcMidBwdArgs(C.Structure):
    _fields_ = [("B", c_i64), ("H1", c_i32), ("H2", c_i32), ("n_exp1", c_i32), ("n_gate1", c_i32), ("n_exp2", c_i32), ("n_gate2", c_i32),
                ("ex1", c_p), ("ld_ex1", c_i64), ("ex2", c_p), ("ld_ex2", c_i64), ("dz1_h", c_p), ("ld_dz1_h", c_i64),
                ("dz2_h", c_p), ("ld_dz2_h", c_i64), ("mask1", c_i32), ("mask2", c_i32), ("scale1", c_f), ("scale2", c_f),
                ("g1", MidBGate1 * MID_MAX_GATE), ("e2", MidBExpert2 * MID_MAX_EXPERT), ("g2", MidBGate2 * MID_MAX_GATE)]


BN_X_BF16, BN_Y_BF16, BN_DY_BF16 = 1, 2, 4


class BnSeg(C.Structure):
    _fields_ = [("x", c_p), ("ldx", c_i64), ("y", c_p), ("ldy", c_i64), ("gamma", c_p), ("beta", c_p),
                ("running_mean", c_p), ("running_var", c_p), ("save_mean", c_p), ("save_invstd", c_p),
                ("num_batches_tracked", c_p), ("yh", c_p), ("ldyh", c_i64), ("C", c_i32), ("row_group", c_i32),
                ("half", c_i32), ("pad_", c_i32)]


class BnFwdArgs(C.Structure):
    _fields_ = [("n_seg", c_i32), ("training", c_i32), ("relu", c_i32), ("skip_le1", c_i32), ("eps", c_f), ("momentum", c_f),
                ("drop_p", c_f), ("seed", C.c_uint64), ("seed_offset_dev", c_p), ("M", c_i64),
                ("row_offsets", c_p), ("workspace", c_p), ("phase", c_i32), ("stats_ready", c_i32), ("exchange", c_p),
                ("s", BnSeg * MAX_BN_SEGS)]


class BnBSeg(C.Structure):
    _fields_ = [("dy", c_p), ("lddy", c_i64), ("y", c_p), ("ldy", c_i64), ("x", c_p), ("ldx", c_i64),
                ("dx", c_p), ("lddx", c_i64), ("gamma", c_p), ("save_mean", c_p), ("save_invstd", c_p),
                ("dgamma", c_p), ("dbeta", c_p), ("dxh", c_p), ("lddxh", c_i64), ("C", c_i32), ("row_group", c_i32), ("accumulate_dx", c_i32),
                ("half", c_i32)]


class BnBwdArgs(C.Structure):
    _fields_ = [("n_seg", c_i32), ("training", c_i32), ("relu", c_i32), ("eps", c_f), ("mask_scale", c_f),
                ("M", c_i64), ("row_offsets", c_p), ("workspace", c_p), ("phase", c_i32), ("pad_", c_i32), ("exchange", c_p),
                ("s", BnBSeg * MAX_BN_SEGS)]


class RowdotGroup(C.Structure):
    _fields_ = [("x", c_p), ("ldx", c_i64), ("w", c_p), ("bias", c_p), ("out", c_p), ("ld_out", c_i64),
                ("logit", c_p), ("ld_logit", c_i64), ("K", c_i32)]


class RowdotFwdArgs(C.Structure):
    _fields_ = [("n_groups", c_i32), ("sigmoid", c_i32), ("n_addend", c_i32), ("addend", c_p * 4),
                ("ld_addend", c_i64 * 4), ("M", c_i64), ("row_offsets", c_p), ("g", RowdotGroup * MAX_GROUPS)]


class RowdotBGroup(C.Structure):
    _fields_ = [("dout", c_p), ("ld_dout", c_i64), ("out", c_p), ("ld_out", c_i64), ("x", c_p), ("ldx", c_i64),
                ("w", c_p), ("dx", c_p), ("lddx", c_i64), ("dw", c_p), ("dbias", c_p), ("dlogit", c_p),
                ("ld_dlogit", c_i64), ("K", c_i32), ("accumulate_dx", c_i32)]


class RowdotBwdArgs(C.Structure):
    _fields_ = [("n_groups", c_i32), ("sigmoid", c_i32), ("M", c_i64), ("row_offsets", c_p), ("workspace", c_p),
                ("bce_group", c_p), ("bce_y_i16", c_p), ("bce_y_f32", c_p), ("bce_loss", c_p), ("bce_partial", c_p),
                ("bce_inv_count", c_f), ("pad_", c_i32), ("g", RowdotBGroup * MAX_GROUPS)]


HEAD_MAX_TOWERS = 8


class HeadTower(C.Structure):
    _fields_ = [("x", c_p), ("ldx", c_i64), ("w", c_p), ("bias", c_p), ("dx", c_p), ("lddx", c_i64), ("dw", c_p), ("dbias", c_p),
                ("K", c_i32), ("accumulate_dx", c_i32)]


class HeadArgs(C.Structure):
    _fields_ = [("n_tower", c_i32), ("sigmoid", c_i32), ("n_addend", c_i32), ("pad_", c_i32), ("M", c_i64),
                ("out", c_p), ("ld_out", c_i64), ("d_out", c_p), ("ld_dout", c_i64),
                ("wide_x", c_p), ("ld_wide", c_i64), ("wide_w", c_p), ("wide_bias", c_p), ("wide_out", c_p), ("ld_wide_out", c_i64),
                ("wide_dx", c_p), ("ld_wide_dx", c_i64), ("wide_dw", c_p), ("wide_dbias", c_p), ("wide_K", c_i32), ("accumulate_wide_dx", c_i32),
                ("addend", c_p * 2), ("ld_addend", c_i64 * 2), ("d_addend", c_p * 2), ("ld_d_addend", c_i64 * 2),
                ("accumulate_d_addend", c_i32 * 2), ("workspace", c_p),
                ("bce_group", c_p), ("bce_y_i16", c_p), ("bce_y_f32", c_p), ("bce_loss", c_p), ("bce_partial", c_p),
                ("bce_inv_count", c_f), ("pad2_", c_i32), ("t", HeadTower * HEAD_MAX_TOWERS)]


TOWER_MAX = 4
TOWER_ROWS = 128
TOWER_ERR_TIMEOUT = 0x40000000


class TowerLayer(C.Structure):
    _fields_ = [("wh", c_p), ("ldwh", c_i64), ("wt", c_p), ("ldwt", c_i64), ("bias", c_p), ("z", c_p), ("ldz", c_i64),
                ("dzh", c_p), ("lddzh", c_i64), ("gamma", c_p), ("beta", c_p), ("running_mean", c_p), ("running_var", c_p),
                ("num_batches_tracked", c_p), ("save_mean", c_p), ("save_invstd", c_p), ("dgamma", c_p), ("dbeta", c_p)]


class TowerDesc(C.Structure):
    _fields_ = [("xh", c_p), ("ldxh", c_i64), ("dx", c_p), ("lddx", c_i64), ("accumulate_dx", c_i32), ("pad_", c_i32),
                ("l1", TowerLayer), ("l2", TowerLayer), ("a1h", c_p), ("lda1h", c_i64), ("a2", c_p), ("lda2", c_i64),
                ("wo", c_p), ("bo", c_p), ("dwo", c_p), ("dbo", c_p), ("dy2", c_p), ("lddy2", c_i64), ("dy1", c_p), ("lddy1", c_i64)]


class TowerArgs(C.Structure):
    _fields_ = [("n_tower", c_i32), ("H0", c_i32), ("H1", c_i32), ("H2", c_i32), ("M", c_i64), ("relu", c_i32), ("sigmoid", c_i32),
                ("drop_p", c_f), ("eps", c_f), ("momentum", c_f), ("pad_", c_i32), ("seed1", C.c_uint64), ("seed2", C.c_uint64),
                ("seed_offset_dev", c_p), ("out", c_p), ("ld_out", c_i64), ("d_out", c_p), ("ld_dout", c_i64),
                ("wide_x", c_p), ("ld_wide", c_i64), ("wide_w", c_p), ("wide_bias", c_p), ("wide_dx", c_p), ("ld_wide_dx", c_i64),
                ("wide_dw", c_p), ("wide_dbias", c_p), ("wide_K", c_i32), ("accumulate_wide_dx", c_i32),
                ("bce_group", c_p), ("bce_y_i16", c_p), ("bce_y_f32", c_p), ("bce_loss", c_p), ("bce_inv_count", c_f), ("pad2_", c_i32),
                ("workspace", c_p), ("err", c_p), ("exchange", c_p * 4), ("t", TowerDesc * TOWER_MAX)]


class StarFuseArgs(C.Structure):
    _fields_ = [("n", c_i32), ("op", c_i32), ("size", c_i64), ("s", c_p), ("ds", c_p), ("accumulate_ds", c_i32), ("pad_", c_i32),
                ("a", c_p * MAX_GROUPS), ("out", c_p * MAX_GROUPS), ("da", c_p * MAX_GROUPS)]


class AdamTensor(C.Structure):
    _fields_ = [("w", c_p), ("g", c_p), ("m", c_p), ("v", c_p), ("n", c_i64), ("l2", c_f), ("n_slabs", c_i32), ("slabs", c_p),
                ("slab_stride", c_i64)]


class AddNArgs(C.Structure):
    _fields_ = [("dst", c_p), ("ld_dst", c_i64), ("rows", c_i64), ("cols", c_i32), ("n", c_i32), ("accumulate", c_i32),
                ("src", c_p * MAX_GROUPS), ("ld_src", c_i64 * MAX_GROUPS)]


class AdamArgs(C.Structure):
    _fields_ = [("n_tensors", c_i32), ("lerp_w", c_f), ("beta2", c_f), ("one_minus_beta2", c_f), ("eps", c_f),
                ("weight_decay", c_f), ("step_scalars", c_p), ("n_scalars", c_i32), ("grad_scale", c_f),
                ("step_dev", c_p), ("reg_sum", c_p), ("reg_seed", c_p), ("t", AdamTensor * MAX_TENSORS)]


# name -> (restype, argtypes); every symbol include/cdcmdr.h declares
_SIGNATURES = {
    "cdc_abi_version": (c_i32, []),
    "cdc_last_error": (C.c_char_p, []),
    "cdc_embed_gather_fwd": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i64, c_p]),
    "cdc_embed_gather_fwd_h": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_i32, c_i32, c_i64, c_p]),
    "cdc_embed_index": (c_i32, [c_p, c_p, c_p, c_p, c_i64, c_i32, c_i64, c_p]),
    "cdc_embed_sort_dedupe": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p]),
    "cdc_embed_sort_dedupe_ids": (c_i32, [c_p, c_p, c_i64, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p]),
    "cdc_embed_segment_sum": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_p]),
    "cdc_embed_segment_sum_direct": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_p]),
    "cdc_embed_segsum_lazy_update": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, AdamHP, c_p, c_i64, c_i32, c_i32, c_i32, c_p]),
    "cdc_embed_grad_dense": (c_i32, [c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i64, c_p]),
    "cdc_embed_adam_touched": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, AdamHP, c_p, c_i64, c_i32, c_i32, c_p]),
    "cdc_embed_adam_dense_pass": (c_i32, [c_p, c_p, c_p, c_i64, AdamHP, c_p, c_p, c_p]),
    "cdc_embed_adam_patch": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_p]),
    "cdc_embed_lazy_catchup": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, AdamHP, c_p, c_p, c_i32, c_i64, c_i32, c_i32, c_p]),
    "cdc_embed_lazy_catchup_gather": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, AdamHP, c_p, c_p, c_p, c_i64, c_i64, c_i32, c_i32, c_p]),
    "cdc_embed_lazy_update": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, AdamHP, c_p, c_p, c_i32, c_i64, c_i32, c_i32, c_p]),
    "cdc_embed_lazy_flush": (c_i32, [c_p, c_p, c_p, c_p, c_i64, c_i32, AdamHP, c_p, c_i32, c_i32, c_i32, c_i32, c_p]),
    "cdc_embed_lazy_flush_bg": (c_i32, [c_p, c_p, c_p, c_p, c_i64, c_i32, AdamHP, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    "cdc_embed_merge_dedupe": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_p]),
    "cdc_eval_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "cdc_eval_metrics": (c_i32, [c_p, c_p, c_p, c_i64, c_i64, c_i32, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "cdc_shard_bucket": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_p]),
    "cdc_shard_expand": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p]),
    "cdc_shard_pack": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_i32, c_i32, c_i32, c_p]),
    "cdc_glinear_fwd": (c_i32, [C.POINTER(LinFwdArgs), c_i32, c_p]),
    "cdc_glinear_bwd_x": (c_i32, [C.POINTER(LinBwdxArgs), c_i32, c_p]),
    "cdc_transpose_multi": (c_i32, [C.POINTER(TransposeArgs), c_p]),
    "cdc_gemm_bf16_nt": (c_i32, [C.POINTER(G2Args), c_p]),
    "cdc_weight_shadows": (c_i32, [C.POINTER(WShadowArgs), c_p]),
    "cdc_shadow_bf16": (c_i32, [C.POINTER(ShadowArgs), c_p]),
    "cdc_glinear_bwd_w": (c_i32, [C.POINTER(LinBwdwArgs), c_i32, c_p]),
    "cdc_glinear_bwd_w_pair": (c_i32, [C.POINTER(LinBwdwArgs), C.POINTER(LinBwdwArgs), c_p, c_p]),
    "cdc_glinear_bwd_w_pair_reduce": (c_i32, [C.POINTER(LinBwdwArgs), C.POINTER(LinBwdwArgs), c_p, c_p]),
    "cdc_gate_pool_fwd": (c_i32, [C.POINTER(PoolFwdArgs), c_p]),
    "cdc_cgc_mid_fwd": (c_i32, [C.POINTER(CgcMidFwdArgs), c_p]),
    "cdc_cgc_mid_bwd": (c_i32, [C.POINTER(CgcMidBwdArgs), c_p]),
    "cdc_cgc_mid_fits": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "cdc_expert_pair_fwd": (c_i32, [C.POINTER(ExpertPairArgs), c_p]),
    "cdc_gate_pool_bwd": (c_i32, [C.POINTER(PoolBwdArgs), c_p]),
    "cdc_bn_fwd": (c_i32, [C.POINTER(BnFwdArgs), c_p]),
    "cdc_bn_bwd": (c_i32, [C.POINTER(BnBwdArgs), c_p]),
    "cdc_rowdot_fwd": (c_i32, [C.POINTER(RowdotFwdArgs), c_p]),
    "cdc_rowdot_bwd": (c_i32, [C.POINTER(RowdotBwdArgs), c_p]),
    "cdc_head_fwd": (c_i32, [C.POINTER(HeadArgs), c_p]),
    "cdc_head_bwd": (c_i32, [C.POINTER(HeadArgs), c_p]),
    "cdc_head_workspace_floats": (c_i64, [C.POINTER(HeadArgs)]),
    "cdc_tower_fwd": (c_i32, [C.POINTER(TowerArgs), c_p]),
    "cdc_tower_bwd": (c_i32, [C.POINTER(TowerArgs), c_p]),
    "cdc_tower_step": (c_i32, [C.POINTER(TowerArgs), c_p]),
    "cdc_tower_workspace_bytes": (c_i64, [C.POINTER(TowerArgs)]),
    "cdc_tower_dp": (c_i32, [C.POINTER(TowerArgs), c_i32, c_p]),
    "cdc_bce_fwd_bwd": (c_i32, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i32, c_f, c_p]),
    "cdc_attn_fwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i32, c_i32, c_i32, c_f, C.c_uint64, c_p, c_p]),
    "cdc_attn_bwd": (c_i32, [c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_i32, c_i32, c_f, C.c_uint64, c_p, c_p]),
    "cdc_add_relu_fwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_p]),
    "cdc_add_relu_bwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i32, c_p, c_i64, c_i32, c_i64, c_i32, c_p]),
    "cdc_sigmoid_gate_fwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_f, c_f, c_f, c_p]),
    "cdc_sigmoid_gate_bwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i32, c_p, c_i64, c_i32, c_i64, c_i32, c_f, c_f, c_f, c_p]),
    "cdc_group_select_fwd": (c_i32, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_i32, c_i32, c_p]),
    "cdc_group_select_bwd": (c_i32, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_i32, c_i32, c_i32, c_p]),
    "cdc_stage_batch": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p, c_p, c_p]),
    "cdc_stage_batch_next": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p, c_p, c_p, c_p, c_i32, c_p, c_p, c_p]),
    "cdc_fm_fwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_i32, c_p]),
    "cdc_fm_bwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_i32, c_i32, c_p]),
    "cdc_bce_mean_fwd_bwd": (c_i32, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i32, c_f, c_p]),
    "cdc_cross_fwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i64, c_i32, c_p]),
    "cdc_cross_bwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_i32, c_p]),
    "cdc_tanh_fwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_p]),
    "cdc_tanh_bwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_i32, c_p]),
    "cdc_cross_combine_fwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_i32, c_p]),
    "cdc_cross_combine_bwd": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i32,
                                      c_p, c_i32, c_p, c_i32, c_p, c_i64, c_i32, c_i32, c_p]),
    "cdc_add_out": (c_i32, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_p]),
    "cdc_copy_or_add": (c_i32, [c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_i32, c_p]),
    "cdc_star_fuse_fwd": (c_i32, [C.POINTER(StarFuseArgs), c_p]),
    "cdc_star_fuse_bwd": (c_i32, [C.POINTER(StarFuseArgs), c_p]),
    "cdc_sum_slices": (c_i32, [c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_i32, c_i32, c_p]),
    "cdc_adam_multi": (c_i32, [C.POINTER(AdamArgs), c_p]),
    "cdc_adam_multi_table": (c_i32, [C.POINTER(AdamArgs), c_p, c_p, c_p, c_i32, c_p]),
    "cdc_embed_segsum_lazy_update_dense": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, AdamHP, c_p, c_i64, c_i32, c_i32, c_i32,
                                                   C.POINTER(AdamArgs), c_p, c_p, c_p, c_i32, c_p]),
    "cdc_step_increment": (c_i32, [c_p, c_p]),
    "cdc_begin_step": (c_i32, [c_p, c_p, c_i32, c_p]),
    "cdc_fill_f32": (c_i32, [c_p, c_f, c_i64, c_p]),
    "cdc_fill_f64": (c_i32, [c_p, C.c_double, c_i64, c_p]),
    "cdc_add_inplace": (c_i32, [c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_p]),
    "cdc_add_n": (c_i32, [c_p, c_p]),
    "cdc_mul_bcast": (c_i32, [c_p, c_p, c_p, c_i64, c_i64, c_p]),
    "cdc_mul_bcast_bwd": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_p]),
    "cdc_group_partition": (c_i32, [c_p, c_p, c_p, c_i64, c_i32, c_p]),
    "cdc_rows_permute": (c_i32, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_i32, c_i32, c_p]),
}

_lib = None


class HipExtensionError(RuntimeError):
    pass


def lib_path():
    return _build.LIB_PATH


def load():
    """Load libcdcmdr.so (building it when absent and hipcc is available). Raises if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not _build.is_fresh():
        # missing, or older than csrc/ + include/ (content hash): rebuild in-tree when a compiler is around
        try:
            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            if not os.path.exists(path):
                raise HipExtensionError(
                    f"libcdcmdr.so is missing at {path} and could not be built ({e}); "
                    "run `python __graft_entry__.py build` — there is no CPU fallback") from e
            import warnings
            warnings.warn(f"libcdcmdr.so is older than its sources and the rebuild failed: {str(e)[:400]}")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise HipExtensionError(f"cannot load {path}: {e}; there is no CPU fallback") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipExtensionError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.cdc_abi_version() != 1:
        raise HipExtensionError(f"ABI version mismatch: library {lib.cdc_abi_version()} != binding 1")
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc, what):
    if rc != 0:
        msg = load().cdc_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


# ---------------------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py / tools): HIP events on the launch stream around every C-ABI call
# ---------------------------------------------------------------------------------------------------------
PROFILE = None      # None, or a list receiving (name, start_event, end_event, flops, bytes)
TRACE = os.environ.get("CDC_TRACE_LAUNCH") == "1"      # development aid: every launch's name on stderr before it is issued


def launch(name, fn, args, stream, flops=0.0, nbytes=0.0):
    """Calls fn(*args, stream); with PROFILE set, brackets it with timing events recorded on the current stream
    (the one every launch of this package goes to)."""
    if TRACE:
        import sys
        sys.stderr.write(f"[cdc launch] {name}\n")
        sys.stderr.flush()
    if PROFILE is None:
        rc = fn(*args, stream)
    else:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args, stream)
        e1.record()
        PROFILE.append((name, e0, e1, flops, nbytes))
    if rc != 0:
        check(rc, name)
