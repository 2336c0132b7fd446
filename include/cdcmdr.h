/*
 * cdcmdr.h — C-ABI of the MI355X (gfx950) multi-domain CTR training hot path.
 *
 * The reference (Chrissie-Law/Causal-Domain-Clustering-for-Multi-Domain-Recommendation,
 * 100 % eager PyTorch) has no FFI of its own; the seam it offers is the nn.Module
 * surface of model/<name>.py.  Every entry point below replaces the ATen work done by one
 * reference function on that path; the reference file:line it stands in for is cited
 * per function.  The host-side mirror of the nn.Module surface (Python) binds these
 * with ctypes — see INTEGRATION.md.
 *
 * Conventions
 *   - raw device pointers + explicit sizes; no torch types cross this boundary;
 *   - the caller owns and pre-allocates every buffer; nothing here allocates, frees
 *     or keeps state between calls;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*);
 *     no call synchronises the device, so every call is hipGraph-capturable;
 *   - return value: 0 = ok, <0 = bad argument (CDC_E_*), >0 = hipError_t;
 *     cdc_last_error() gives a thread-local message for the last non-zero return;
 *   - all floating point storage is fp32.  `prec` selects the arithmetic of the
 *     dense contractions: CDC_PREC_BF16 = bf16 MFMA operands / fp32 accumulate,
 *     CDC_PREC_F32 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32).
 */
#ifndef CDCMDR_H
#define CDCMDR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDC_ABI_VERSION 1

#define CDC_E_BADARG   (-1)
#define CDC_E_TOOBIG   (-2)
#define CDC_E_ALIGN    (-3)

#define CDC_PREC_BF16 0
#define CDC_PREC_F32  1

#define CDC_MAX_GROUPS 32   /* groups / segments per launch (descriptors travel as kernel arguments) */
#define CDC_MAX_TENSORS 48  /* tensors per multi-tensor Adam launch */

int cdc_abi_version(void);
const char* cdc_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Embedding (reference: model/layer.py:129-157 FeaturesEmbedding; dense nn.Embedding grad +
 * whole-table L2 + dense torch.optim.Adam: model/layer.py:31,96-112, run.py:489,720-721)
 * ---------------------------------------------------------------------------------------- */

/* model/layer.py:152-155 — idx = x + offsets computed in x's dtype (int32, wrapping),
 * out[b, f*D:(f+1)*D] = table[idx[b,f], :].   ids [B,F] int32 row-major, offsets [F] int32.
 * An idx outside [0,R) sets *err_flag (device int32, may be NULL) to 1+flat position and
 * yields a zero row (the CPU reference raises IndexError).  idx_out (may be NULL) receives
 * the int32 row indices. */
int cdc_embed_gather_fwd(const int32_t* ids, const int32_t* offsets, const float* table,
                         float* out, int32_t* idx_out, int32_t* err_flag,
                         int64_t B, int32_t F, int32_t D, int64_t R, void* stream);

/* The same gather that also writes the bf16 shadow of `out` (row stride ld_out_h elements; NULL = none) for the contractions
 * that read the embeddings through cdc_gemm_bf16_nt. */
int cdc_embed_gather_fwd_h(const int32_t* ids, const int32_t* offsets, const float* table,
                           float* out, void* out_h, int64_t ld_out_h, int32_t* idx_out, int32_t* err_flag,
                           int64_t B, int32_t F, int32_t D, int64_t R, void* stream);

/* model/layer.py:152 alone: idx_out[b,f] = ids[b,f] + offsets[f] (int32, wrapping). */
int cdc_embed_index(const int32_t* ids, const int32_t* offsets, int32_t* idx_out, int32_t* err_flag, int64_t B, int32_t F,
                    int64_t R, void* stream);   /* out-of-range ids: idx_out = -1 (+ err_flag), skipped downstream */

/* Per-field sort + dedupe of the row indices of one batch (rows of different fields never collide, so each field is
 * sorted on its own).  Keys are (row<<32|b).  Without `scratch` (B <= CDC_SORT_MAX_B): one workgroup per field sorts in
 * LDS.  With `scratch` >= 2*F*B uint64 (any B <= CDC_SORT_MAX_ROWS; required above CDC_SORT_MAX_B): chunks of 1024..16384
 * rows are sorted by separate workgroups and the ~8 sorted runs merged by rank — 3x faster at B = 4096, where one
 * workgroup per field is bound by its CU's LDS bandwidth.  Same output either way.
 *   idx       [B,F] int32 row indices (from cdc_embed_gather_fwd / cdc_embed_index)
 *   uniq_row  [F,B] int32 — field f's unique rows, ascending, first uniq_cnt[f] entries valid
 *   seg_start [F,B+1] int32 — uniq j of field f owns sorted positions [seg_start[f][j], seg_start[f][j+1])
 *   perm      [F,B] int32 — batch row b of each sorted position (ascending b inside a segment)
 *   uniq_cnt  [F]   int32
 */
#define CDC_SORT_MAX_B 16384
#define CDC_SORT_MAX_ROWS 32768
int cdc_embed_sort_dedupe(const int32_t* idx, int32_t* uniq_row, int32_t* seg_start, int32_t* perm,
                          int32_t* uniq_cnt, uint64_t* scratch, int64_t B, int32_t F, void* stream);
/* The same from the raw ids of a batch: row = ids[b,f] + offsets[f] (model/layer.py:139), ids outside [0,R) sort as -1 —
 * cdc_embed_index fused into the sort's first launch.  step_dev != NULL: that launch also does cdc_begin_step's work
 * (++*step_dev, accumulators[0..n_acc) = 0) — for a training step whose first launch this is.  err_flag (may be NULL):
 * device int32, raised to 1 + the flat position of an out-of-range id, like cdc_embed_index. */
int cdc_embed_sort_dedupe_ids(const int32_t* ids, const int32_t* offsets, int64_t R, int32_t* step_dev, double* accumulators,
                              int32_t n_acc, int32_t* err_flag, int32_t* uniq_row, int32_t* seg_start, int32_t* perm,
                              int32_t* uniq_cnt, uint64_t* scratch, int64_t B, int32_t F, void* stream);

/* Per-row gradient of the batch: rowgrad[f, j, :] = sum over unique row j's segment of d_out[b, f*D:(f+1)*D],
 * summed in ascending b (the order aten::embedding_dense_backward uses on the CPU, model/layer.py:140,153) for segments
 * shorter than 64 entries; longer ones are cut into 64/D (from 512 entries on: 256/D) contiguous parts that are summed in
 * ascending b and combined in part order.  rowgrad is [F, B, D] floats.  sorted_scratch is no longer used (the sums read
 * d_out through perm) and may be NULL; the parameter stays for ABI stability. */
int cdc_embed_segment_sum(const float* d_out, const int32_t* seg_start, const int32_t* perm, const int32_t* uniq_cnt,
                          float* sorted_scratch, float* rowgrad, int64_t B, int32_t F, int32_t D, void* stream);
/* The same sums without the sorted copy, for batches whose segments are short (the owner side of the row-sharded table:
 * at most one entry per sending rank, see cdc_shard_bucket): every (unique row, 16-byte chunk) adds its segment straight
 * from d_out in segment order — bit-identical to cdc_embed_segment_sum for segments shorter than 64 entries.  uniq_row (may
 * be NULL): unique rows < 0 (the -1 padding of the lists, one long segment) are skipped and their rowgrad left untouched. */
int cdc_embed_segment_sum_direct(const float* d_out, const int32_t* seg_start, const int32_t* perm, const int32_t* uniq_cnt,
                                 const int32_t* uniq_row, float* rowgrad, int64_t B, int32_t F, int32_t D, void* stream);

/* Dense gradient of the table as the reference's nn.Embedding produces it (drop-in path feeding
 * torch.optim.Adam): grad[uniq_row[f,j], :] += rowgrad[f,j,:]. */
int cdc_embed_grad_dense(const float* rowgrad, const int32_t* uniq_row, const int32_t* uniq_cnt, float* grad,
                         int64_t B, int32_t F, int32_t D, int64_t R, void* stream);

/* Adam hyper-parameters of run.py:720-721 (+ the L2 coefficient of model/layer.py:31), already
 * rounded to fp32 the way torch's CPU Adam rounds its Python-double scalars:
 *   lerp_w = float(1-beta1), beta2 = float(beta2), one_minus_beta2 = float(1-beta2),
 *   l2_twice = 2*float(l2)  (d/dw sum(l2*w^2) as autograd forms it).
 * step_scalars: device table [n_scalars][2] of {step_size_t = lr/(1-beta1^t), sqrt(1-beta2^t)},
 * computed by the host in double and rounded to fp32 (index t, clamped to n_scalars-1 where both
 * have converged).  step_dev points at a device int32 holding the 1-based step t of THIS update. */
typedef struct {
    float lerp_w, beta2, one_minus_beta2, eps, weight_decay, l2_twice;
    const float* step_scalars;
    int32_t n_scalars;
    int32_t fast_replay;            /* lazy replay of L2-only steps: 0 = the exact routine (IEEE divide / sqrt, bit-identical to
                                       the dense pass), 1 = v_rcp_f32 / v_sqrt_f32 (1 ulp each; ~6x fewer instructions; the
                                       replayed trajectory stays within 1e-7 of the exact one, see tests) */
    const float* inv_bc2;           /* fast_replay: device table [n_scalars] of 1/sqrt(1-beta2^t) */
    const float* replay_tab;        /* fast_replay in scaled state (NULL = off): device table [n_scalars][2] of
                                       {A_t = step_size_t * k1 * sqrt(1-beta2^t) / sqrt(k2),  E_t = eps * sqrt(1-beta2^t) / sqrt(k2)} */
    float k1, k2;                   /* the scales of the scaled state, k1 ~ (1-beta1) * (2*l2 + wd),  k2 ~ (1-beta2) * (2*l2 + wd)^2  (both > 0 to
                                       enable): M = m * ik1, V = v * ik2 on the way in; m = M * (k1 + k1_lo), v = V * (k2 + k2_lo) on the way
                                       out, where k + k_lo = 1 / ik to 2^-48 — so that a row's state does not pick up the factor
                                       (ik * k) = 1 + 6e-8 with every replayed segment (round 4: that coherent drift of every row's
                                       moments doubled the spread of the AUC over row orders) */
    float ik1, ik2, k1_lo, k2_lo;
} cdc_adam_hp;

/* Exact dense-Adam semantics for the whole table, in three launches:
 *  (1) touched: for every unique row of the batch: g = rowgrad (cdc_embed_segment_sum) + 2*l2*w + wd*w, Adam
 *      step t, result written to side[f,j,{w,m,v},D] (the table itself is not modified);
 *  (2) dense_pass: every row updated as an untouched row (g = 2*l2*w + wd*w) in one streaming
 *      pass; reg_sum (device double, may be NULL) += sum(w_old^2) (caller multiplies by l2);
 *  (3) patch: side rows copied over the rows touched by the batch. */
int cdc_embed_adam_touched(const float* rowgrad, const int32_t* uniq_row, const int32_t* uniq_cnt,
                           const float* w, const float* m, const float* v, float* side,
                           cdc_adam_hp hp, const int32_t* step_dev,
                           int64_t B, int32_t F, int32_t D, void* stream);
int cdc_embed_adam_dense_pass(float* w, float* m, float* v, int64_t n_elems,
                              cdc_adam_hp hp, const int32_t* step_dev, double* reg_sum, void* stream);
int cdc_embed_adam_patch(const float* side, const int32_t* uniq_row, const int32_t* uniq_cnt,
                         float* w, float* m, float* v, int64_t B, int32_t F, int32_t D, void* stream);

/* Lazy-exact form of the same update: rows carry last[row] = the step their
 * (w,m,v) are valid for.  catchup replays the untouched-row recurrence for the batch's unique rows
 * up to step t-1 (run BEFORE the gather of step t); update applies step t with the batch gradient;
 * flush brings every row to step t_now (dense pass; before state_dict / eval / reg-loss report).
 * reg_ring [ring_len] device doubles: reg_ring[s % ring_len] += sum(w_{s-1}^2) over replayed rows. */
int cdc_embed_lazy_catchup(const int32_t* uniq_row, const int32_t* uniq_cnt,
                           float* w, float* m, float* v, int32_t* last,
                           cdc_adam_hp hp, const int32_t* step_dev, double* reg_ring, int32_t ring_len,
                           int64_t B, int32_t F, int32_t D, void* stream);
int cdc_embed_lazy_update(const float* rowgrad, const int32_t* uniq_row, const int32_t* uniq_cnt,
                          float* w, float* m, float* v, int32_t* last,
                          cdc_adam_hp hp, const int32_t* step_dev, double* reg_ring, int32_t ring_len,
                          int64_t B, int32_t F, int32_t D, void* stream);
/* cdc_embed_lazy_catchup + cdc_embed_gather_fwd in ONE pass over the batch's table rows (needs the sorted lists of
 * cdc_embed_sort_dedupe[_ids]): every unique row is replayed to step t-1, written back, and written to every batch position
 * that looks it up — out [B, F*D] fp32 and, when out_h != NULL, its bf16 shadow (row stride ld_out_h elements).  Rows looked up
 * more than four times are distributed by their whole wave (the row is read from HBM once per step however hot it is); ids
 * outside the table give zero rows.  D/4 must divide 64.  Same replay arithmetic, same bits as the two separate calls. */
int cdc_embed_lazy_catchup_gather(const int32_t* uniq_row, const int32_t* uniq_cnt, const int32_t* seg_start, const int32_t* perm,
                                  float* w, float* m, float* v, int32_t* last, cdc_adam_hp hp, const int32_t* step_dev,
                                  float* out, void* out_h, int64_t ld_out_h, int64_t B, int32_t F, int32_t D, void* stream);
/* cdc_embed_segment_sum + cdc_embed_lazy_update in ONE launch: every row's summed gradient is used for its Adam step t by the
 * thread(s) that formed it (no rowgrad round trip).  Same sums, same arithmetic, identical bits.  short_only = 1: the batch
 * is an owner's merged row lists (segments of at most one entry per sender; rows < 0 are padding and skipped). */
int cdc_embed_segsum_lazy_update(const float* d_out, const int32_t* seg_start, const int32_t* perm, const int32_t* uniq_cnt,
                                 const int32_t* uniq_row, float* w, float* m, float* v, int32_t* last, cdc_adam_hp hp,
                                 const int32_t* step_dev, int64_t B, int32_t F, int32_t D, int32_t short_only, void* stream);
/* brings rows to step target = *step_dev + step_bias: rows with last < target are replayed.
 * period <= 1: all R rows.  period > 1: only slice (target mod period) of the table (ceil(R/period) consecutive rows), so
 * that calling it every step brings every row up to date once per `period` steps — this bounds the gaps the per-batch
 * catch-up sees, spreads the replay work evenly over the steps, and the slice is chosen on the device (graph-replay safe).
 * The slice launch touches no row whose last >= target (the step's own rows after their catch-up). */
int cdc_embed_lazy_flush(float* w, float* m, float* v, int32_t* last, int64_t R, int32_t D,
                         cdc_adam_hp hp, const int32_t* step_dev, int32_t step_bias, int32_t period,
                         int32_t own_mod, int32_t own_rem, void* stream);
/* The same as a BACKGROUND launch: the grid is capped at waves_per_simd workgroups of 256 threads per compute unit (a
 * grid-stride loop covers the slice), its waves keep the lowest issue priority and the loads of the next item are in flight
 * under the replay of the current one.  Issued on a second stream beside the forward/backward (whose kernels raise their
 * priority) the VALU-only replay takes the issue cycles those leave idle instead of every wave slot of the chip
 * (profiles/round3/README.md: 2 waves per SIMD measured best).  waves_per_simd <= 0: the full grid of cdc_embed_lazy_flush. */
int cdc_embed_lazy_flush_bg(float* w, float* m, float* v, int32_t* last, int64_t R, int32_t D,
                            cdc_adam_hp hp, const int32_t* step_dev, int32_t step_bias, int32_t period,
                            int32_t own_mod, int32_t own_rem, int32_t waves_per_simd, void* stream);   /* own_mod > 1: only rows r with r % own_mod == own_rem */

/* Row-sharded table under data parallelism (no counterpart in the reference, which is single-process): row r belongs to
 * rank r % n_rank.  Every kernel that walks unique-row lists skips entries < 0, so an owner can treat the row lists it
 * receives (padded with -1) as a batch and reuse sort / catch-up / segment-sum / update unchanged.
 *   bucket: unique rows of the local batch [F,B] -> send_ids [n_rank][cap][F] (row id or -1), slot_of [F,B]; slots in
 *           ascending row order (deterministic); *overflow (device int32) = max slot+1 that did not fit (0 = fine)
 *   expand: rows received from the owners [n_rank][cap][F][D] -> gathered embeddings out [B, F*D]
 *   pack:   per-unique-row gradients [F][B][D] -> send_grads [n_rank][cap][F][D]
 * (the wire layout [sender][cap][F] read as [n_rank*cap, F] IS the owner's batch: no repacking on either side) */
/* sort_dedupe for a batch made of n_runs runs of B/n_runs rows, each already ascending per field (as unsigned, -1 last):
 * merge by rank + dedupe, no sort.  scratch: F*B uint64. */
int cdc_embed_merge_dedupe(const int32_t* idx, int32_t* uniq_row, int32_t* seg_start, int32_t* perm, int32_t* uniq_cnt,
                           uint64_t* scratch, int64_t B, int32_t F, int32_t n_runs, void* stream);
int cdc_shard_bucket(const int32_t* uniq_row, const int32_t* uniq_cnt, int32_t* send_ids, int32_t* slot_of, int32_t* overflow,
                     int64_t B, int32_t F, int32_t n_rank, int32_t cap, void* stream);
int cdc_shard_expand(const float* rows_recv, const int32_t* uniq_row, const int32_t* uniq_cnt, const int32_t* seg_start,
                     const int32_t* perm, const int32_t* slot_of, float* out, int64_t B, int32_t F, int32_t D,
                     int32_t n_rank, int32_t cap, void* stream);
int cdc_shard_pack(const float* rowgrad, const int32_t* uniq_row, const int32_t* uniq_cnt, const int32_t* slot_of, float* send,
                   int64_t B, int32_t F, int32_t D, int32_t n_rank, int32_t cap, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped linear layers on MFMA (reference: every nn.Linear on the path —
 * model/layer.py:185,193 MultiLayerPerceptron; :275 DNN; model/ple.py:83-94 CGC experts+gates;
 * model/mmoe.py:35-40; model/star.py:90-102 F.linear with W_d*W_s)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float* x;  int64_t ldx;    /* [M,K] activations, row stride ldx */
    const float* w;  int64_t ldw;    /* [N,K] nn.Linear weight layout */
    const float* bias;               /* [N] or NULL */
    float* y;        int64_t ldy;    /* [M,N] */
    int32_t M, N, K;
    int32_t act_cols;                /* columns [0,act_cols) get relu (+dropout) in the epilogue */
    double* bn_partial;              /* NULL, or the workspace of the cdc_bn_fwd launch that normalises y next: the epilogue
                                        also writes that launch's per-(64-row block, column) partial sums of y and y^2
                                        (its statistics pass, saved: cdc_bn_fwd_args.stats_ready).  Needs act_cols == 0,
                                        no row_offsets, CDC_PREC_BF16 or F32 with 64x64 tiles (forced when set). */
    int32_t bn_col0, bn_total_c;     /* y's first column among that launch's columns; its column total */
} cdc_lin_group;

typedef struct {
    int32_t n_groups;
    int32_t relu;                    /* 1: relu on act columns */
    float   drop_p;                  /* dropout probability on act columns (0 = off) */
    uint64_t seed;                   /* dropout stream: element (group,row,col) keyed by seed */
    const int32_t* seed_offset_dev;  /* device int32 added to the seed stream per step (may be NULL) */
    const int32_t* row_offsets;      /* device [n_groups+1] or NULL: ragged rows — group g uses rows
                                        [row_offsets[g], row_offsets[g+1]) of its x / y; M = upper bound */
    cdc_lin_group g[CDC_MAX_GROUPS];
} cdc_lin_fwd_args;

/* y_g = act(x_g · w_gᵀ + bias_g) for every group in one launch. */
int cdc_glinear_fwd(const cdc_lin_fwd_args* a, int32_t prec, void* stream);

/* dX_o = sum over the output's segments s of dZ_s · W_s   (several groups that read the same x
 * reduce into one dX without atomics), then the optional activation mask of the PRODUCING layer:
 * dX = (mask_y > 0) ? dX * mask_scale : 0. */
typedef struct {
    const float* dz; int64_t lddz;   /* [M,N] */
    const float* w;  int64_t ldw;    /* [N,K] */
    const float* wt; int64_t ldwt;   /* optional [K,N] transposed copy of w (cdc_transpose_multi): when every segment of a
                                        launch has one, both operands stream with the reduction index contiguous */
    int32_t N;
    int32_t out;                     /* index of the output this segment reduces into */
    int32_t wt_bf16;                 /* 1: wt points at bf16 elements (cdc_transpose_multi with its bf16_mask bit; ldwt in elements,
                                        a multiple of 8, wt and dz 16-byte aligned, lddz a multiple of 4; CDC_PREC_BF16 only):
                                        half the operand bytes, no conversion in the loop, the same bits in the MFMA */
} cdc_bwdx_seg;
typedef struct {
    float* dx; int64_t lddx;         /* [M,K] */
    const float* mask_y; int64_t ldmask; /* post-activation output of the layer that produced x, or NULL */
    int32_t M, K;
    int32_t mask_cols;               /* mask applies to columns [0,mask_cols) */
    int32_t accumulate;              /* 1: dx += result */
} cdc_bwdx_out;
typedef struct {
    int32_t n_out, n_seg;
    float mask_scale;
    const int32_t* row_offsets;      /* ragged rows per OUTPUT (device [n_out+1]) or NULL */
    cdc_bwdx_out o[CDC_MAX_GROUPS];
    cdc_bwdx_seg s[CDC_MAX_GROUPS];
} cdc_lin_bwdx_args;
int cdc_glinear_bwd_x(const cdc_lin_bwdx_args* a, int32_t prec, void* stream);

/* dst_i[c,r] = src_i[r,c] for up to CDC_MAX_TENSORS row-major matrices in one launch (per-step W^T copies for grad-input).
 * bf16_mask bit i set: dst_i is written as bf16 (round to nearest even, what the MFMA path would round to). */
typedef struct {
    int32_t n;
    int32_t pad_;
    uint64_t bf16_mask;
    struct { const float* src; float* dst; int32_t rows, cols; } t[CDC_MAX_TENSORS];
} cdc_transpose_args;
int cdc_transpose_multi(const cdc_transpose_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * The same contractions with operands that ARE bf16 in memory ("shadows": the bf16 rounding of an fp32 tensor, written once
 * by its producer instead of every time a tile of it is staged).  Arithmetic identical to CDC_PREC_BF16 above: bf16 operands
 * (round to nearest even), fp32 accumulate, fp32 epilogue.  Replaces the same reference lines as cdc_glinear_fwd /
 * cdc_glinear_bwd_x (nn.Linear at model/layer.py:185,193,275; model/ple.py:83-94; model/mmoe.py:35-40 and the two addmm calls
 * autograd issues for forward and grad-input).
 *   out_o[M,N] = epilogue( sum over the output's segments s of A_s[M,Kr] . B_s[N,Kr]^T )
 *   mode 0 (forward):    A = shadow of x, B = shadow of W [N,K];  epilogue: + bias, relu / dropout on columns [0,act_cols),
 *                        optional BatchNorm partial sums (as cdc_lin_group.bn_partial)
 *   mode 1 (grad-input): A = shadow of dZ [M,N_s], B = shadow of W^T [K,N_s] (cdc_weight_shadows);  epilogue: activation mask
 *                        of the layer that produced x on columns [0,act_cols): d = (mask > 0) ? d * mask_scale : 0, then
 *                        optional += into the fp32 destination
 * Kr: the reduction length rounded UP to a multiple of 64; both operands' rows must be readable that far and the padding of
 * at least one of them must be zero (shadows are allocated zero-padded).  Operand pointers 16-byte aligned, row strides
 * multiples of 8 elements.  Every output may be written as fp32 (y), as its bf16 shadow (yh), or both. */
#define CDC_G2_MAX_OUT 24
#define CDC_G2_MAX_SEG 32
typedef struct {
    float* y; int64_t ldy;           /* fp32 destination [M,N] or NULL */
    void*  yh; int64_t ldyh;         /* bf16 destination [M,N] or NULL (row stride in elements) */
    const float* bias;               /* mode 0: [N] or NULL */
    const void* mask; int64_t ldmask;/* mode 1: post-activation output of the layer that produced x (fp32, or bf16 when mask_bf16) or NULL */
    double* bn_partial;              /* mode 0: as cdc_lin_group.bn_partial */
    int32_t M, N;
    int32_t act_cols;                /* mode 0: activation columns; mode 1: masked columns */
    int32_t accumulate;              /* mode 1: y += result (needs y) */
    int32_t mask_bf16;
    int32_t bn_col0, bn_total_c;
    int32_t stream_id;               /* mode 0: distinguishes the dropout streams of the outputs of one launch */
} cdc_g2_out;
typedef struct {
    const void* a; int64_t lda;      /* bf16 [M, >=Kr] */
    const void* b; int64_t ldb;      /* bf16 [N, >=Kr] */
    int32_t Kr;                      /* reduction length, multiple of 64 */
    int32_t out;                     /* output this segment accumulates into */
} cdc_g2_seg;
typedef struct {
    int32_t n_out, n_seg;
    int32_t mode;                    /* 0 forward, 1 grad-input */
    int32_t relu;
    float   drop_p;
    float   mask_scale;
    int32_t tile_cfg;                /* 0 = chosen by the library; 1..10 force a tile shape / ring depth (tuning, tools/gemm2_probe.hip) */
    int32_t pad_;
    uint64_t seed;
    const int32_t* seed_offset_dev;
    cdc_g2_out o[CDC_G2_MAX_OUT];
    cdc_g2_seg s[CDC_G2_MAX_SEG];
} cdc_g2_args;
int cdc_gemm_bf16_nt(const cdc_g2_args* a, void* stream);

/* Weight shadows, once per step before the forward: for every W_i [rows = N, cols = K] (contiguous fp32) the straight bf16
 * copy dst_h [N, ld_h] and/or the transposed bf16 copy dst_t [K, ld_t].  Only the tensor's own elements are written: the
 * destinations' padding (ld_h > K, ld_t > N, several tensors concatenated into one destination) is the caller's, zeroed once. */
typedef struct {
    int32_t n;
    int32_t pad_;
    struct { const float* src; void* dst_h; int64_t ld_h; void* dst_t; int64_t ld_t; int32_t rows, cols; } t[CDC_MAX_TENSORS];
} cdc_wshadow_args;
int cdc_weight_shadows(const cdc_wshadow_args* a, void* stream);

/* dst_i = bf16(src_i) for up to CDC_MAX_GROUPS [rows, cols] views (row strides in elements): the shadow of an activation or
 * gradient whose producer does not write it itself. */
typedef struct {
    int32_t n;
    int32_t pad_;
    struct { const float* src; int64_t ld_src; void* dst; int64_t ld_dst; int64_t rows; int32_t cols; int32_t pad_; } t[CDC_MAX_GROUPS];
} cdc_shadow_args;
int cdc_shadow_bf16(const cdc_shadow_args* a, void* stream);

/* dW_g[N,K] = dZ_gᵀ · X_g ; db_g[N] = column sums of dZ_g (fp32, exact order-fixed reduction). */
typedef struct {
    const float* dz; int64_t lddz;   /* [M,N] */
    const float* x;  int64_t ldx;    /* [M,K] */
    float* dw; int64_t lddw;         /* [N,K] */
    float* db;                       /* [N] or NULL */
    int32_t M, N, K;
    int32_t accumulate;              /* 1: dw += , db += */
    const void* dzh; int64_t lddzh;  /* optional bf16 shadows of dz and x (row strides in elements; rows zero-padded to a multiple */
    const void* xh;  int64_t ldxh;   /* of 64, 128 readable columns past col 0 of a tile).  When every group of a CDC_PREC_BF16 launch
                                        without row_offsets has both, the contraction reads them (direct-to-LDS staging, transposing
                                        LDS reads) and db is summed from the bf16 dz */
} cdc_bwdw_group;
typedef struct {
    int32_t n_groups;
    int32_t split_k;                 /* > 1: the batch rows are cut into split_k slices, each reduced by its own
                                        workgroup into an fp32 slab; a second launch adds the slabs in slice order
                                        (deterministic; no float atomics) */
    float* workspace;                /* split_k > 1: >= split_k * sum_g (N_g*K_g + N_g) floats */
    const int32_t* row_offsets;      /* ragged rows (device [n_groups+1]) or NULL */
    int32_t defer_reduce;            /* split_k > 1 only: 1 = the slab-adding launch is left out — dw / db are NOT written, the consumer
                                        (cdc_adam_tensor.slabs) adds the slabs [dW_0 | db_0 | dW_1 | db_1 | ...] itself; no group may
                                        accumulate */
    int32_t pad_;
    cdc_bwdw_group g[CDC_MAX_GROUPS];
} cdc_lin_bwdw_args;
int cdc_glinear_bwd_w(const cdc_lin_bwdw_args* a, int32_t prec, void* stream);
/* two grad-weight launches of the bf16-shadow form in ONE launch: `wide` (128 x 128 output tiles) and `narrow` (every group N <= 64:
 * 64 x 64 tiles), e.g. a step's two batched launches.  tabs_dev: a DEVICE copy of the two argument blocks, [wide, narrow] (two blocks do
 * not fit a kernel-argument block); the host copies are read for validation and the grid.  Each block: every group with both shadows,
 * no row_offsets, and either split_k <= 1 or defer_reduce (the slab-adding launch is left to the consumer).  Same arithmetic as two
 * cdc_glinear_bwd_w calls. */
int cdc_glinear_bwd_w_pair(const cdc_lin_bwdw_args* wide, const cdc_lin_bwdw_args* narrow, const cdc_lin_bwdw_args* tabs_dev, void* stream);
/* ... and the slabs of both classes added into the gradients by ONE more launch (same sums, same slice order as cdc_glinear_bwd_w's
 * own reduce launch): for callers that need the reduced gradient at once (data parallelism all-reduces it).  Both classes split. */
int cdc_glinear_bwd_w_pair_reduce(const cdc_lin_bwdw_args* wide, const cdc_lin_bwdw_args* narrow, const cdc_lin_bwdw_args* tabs_dev,
                                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Gate softmax + expert pooling (reference: model/ple.py:89-94,105-123; model/mmoe.py:37-40,58-60)
 *   p[b,g,:] = softmax(logits[b, gate g's columns]) ; out[g][b,:] = sum_j p[b,g,j] * experts[b, sel[g][j], :]
 * experts [B, n_expert, H] with row stride ld_exp (expert e at column e*H).
 * ---------------------------------------------------------------------------------------- */
#define CDC_MAX_GATES 16
#define CDC_MAX_SEL   16
typedef struct {
    int32_t n_gates, n_expert, H;
    int64_t B;
    const float* experts; int64_t ld_exp;
    struct {
        const float* logits; int64_t ld_logits;   /* [B, n_sel] */
        float* out; int64_t ld_out;               /* [B, H] */
        float* probs;                             /* [B, n_sel] contiguous (saved for backward) */
        void* out_h; int64_t ld_out_h;            /* optional bf16 shadow of out (cdc_gemm_bf16_nt reads it), or NULL */
        int32_t n_sel;
        int32_t sel[CDC_MAX_SEL];
    } gate[CDC_MAX_GATES];
} cdc_pool_fwd_args;
int cdc_gate_pool_fwd(const cdc_pool_fwd_args* a, void* stream);

typedef struct {
    int32_t n_gates, n_expert, H;
    int64_t B;
    const float* experts; int64_t ld_exp;     /* forward expert outputs (post-activation) */
    float* d_experts; int64_t ld_dexp;        /* [B, n_expert, H]: every expert column written */
    int32_t mask_relu;                        /* 1: d_experts = (experts > 0) ? d * mask_scale : 0 */
    float mask_scale;
    int32_t accumulate;                       /* 1: d_experts += (a second launch for more than CDC_MAX_GATES gates) */
    void* d_experts_h; int64_t ld_dexp_h;     /* optional bf16 shadow of d_experts (the value after the +=), or NULL */
    struct {
        const float* d_out; int64_t ld_dout;  /* [B,H] */
        const float* probs;                   /* [B,n_sel] */
        float* d_logits; int64_t ld_dlogits;  /* [B,n_sel] */
        void* d_logits_h; int64_t ld_dlogits_h;   /* optional bf16 shadow of d_logits, or NULL */
        int32_t n_sel;
        int32_t sel[CDC_MAX_SEL];
    } gate[CDC_MAX_GATES];
} cdc_pool_bwd_args;
int cdc_gate_pool_bwd(const cdc_pool_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * The boundary between two extraction levels of PLE as ONE launch per direction (reference: model/ple.py:96-125 called twice
 * from model/ple.py:54-57): level k's gate softmax + pooling, level k+1's single-layer experts and its gates (nn.Linear
 * (+ReLU +dropout): model/layer.py:185-191; gates model/ple.py:89-94), and level k+1's gate softmax + pooling.  A workgroup
 * owns 16 batch rows: the pooled level-k outputs never leave LDS as fp32 (they are the bf16 A operand of the level-k+1
 * contractions, exactly the rounding cdc_gemm_bf16_nt applies to their shadow), the level-k+1 expert tiles are pooled
 * straight out of LDS.  Arithmetic, rounding points and dropout stream are those of
 *     cdc_gate_pool_fwd -> cdc_gemm_bf16_nt(mode 0) -> cdc_gate_pool_fwd        (forward: bit-identical results)
 *     cdc_gate_pool_bwd -> cdc_gemm_bf16_nt(mode 1) -> cdc_gate_pool_bwd        (backward: same roundings; the row dot
 *                                                                                products of the gate gradients are summed
 *                                                                                by 16 lanes per row instead of H/4)
 * Widths are compile-time: (H1, H2) = (128, 64), the reference's expert_dims ((256,128),(64,)) (config.py:39-42); other
 * shapes take the three separate launches.  sel lists ascending.  What the backward needs is written: probabilities of both
 * levels, the level-k outputs' bf16 shadows (grad-weight operand), the level-k+1 expert activations (fp32).
 * ---------------------------------------------------------------------------------------- */
#define CDC_MID_MAX_EXPERT 16
#define CDC_MID_MAX_GATE 8
typedef struct {
    const float* logits; int64_t ld_logits;   /* [B, n_sel] fp32: this gate's logits */
    float* probs;                             /* [B, n_sel] contiguous (saved for backward) */
    void* pooled_h; int64_t ld_pooled_h;      /* bf16 [B, H1]: the pooled output = an input of the next level */
    int32_t n_sel;
    int32_t sel[CDC_MAX_SEL];
    int32_t pad_;
} cdc_mid_gate1;
typedef struct {
    const void* w; int64_t ldw;               /* bf16 [H2, >= H1] (cdc_weight_shadows' straight copy) */
    const float* bias;                        /* [H2] or NULL */
    int32_t src;                              /* which level-k gate's pooled output this expert reads */
    int32_t stream_id;                        /* dropout stream of this expert (cdc_g2_out.stream_id of the unfused launch) */
} cdc_mid_expert2;
typedef struct {
    const void* w; int64_t ldw;               /* bf16 [n_sel, >= H1] */
    const float* bias;                        /* [n_sel] or NULL */
    float* probs;                             /* [B, n_sel] contiguous */
    float* out; int64_t ld_out;               /* fp32 [B, H2] pooled output, or NULL */
    void* out_h; int64_t ld_out_h;            /* bf16 [B, H2] pooled output, or NULL */
    int32_t src;                              /* which level-k pooled output the gate reads */
    int32_t n_sel;
    int32_t sel[CDC_MAX_SEL];
} cdc_mid_gate2;
typedef struct {
    int64_t B;
    int32_t H1, H2;
    int32_t n_exp1, n_gate1, n_exp2, n_gate2;
    const float* ex1; int64_t ld_ex1;         /* level-k expert activations [B, n_exp1*H1] fp32 */
    float* ex2; int64_t ld_ex2;               /* level-k+1 expert activations [B, n_exp2*H2] fp32 (output) */
    int32_t relu;                             /* level-k+1 experts: relu */
    float drop_p;                             /* ... and dropout (32-bit counter stream of cdc_gemm_bf16_nt) */
    uint64_t seed;
    const int32_t* seed_offset_dev;
    cdc_mid_gate1 g1[CDC_MID_MAX_GATE];
    cdc_mid_expert2 e2[CDC_MID_MAX_EXPERT];
    cdc_mid_gate2 g2[CDC_MID_MAX_GATE];
} cdc_cgc_mid_fwd_args;
int cdc_cgc_mid_fwd(const cdc_cgc_mid_fwd_args* a, void* stream);

typedef struct {
    const float* probs;                       /* [B, n_sel] */
    float* d_logits; int64_t ld_dlogits;      /* [B, n_sel] fp32 (output) */
    void* d_logits_h; int64_t ld_dlogits_h;   /* optional bf16 shadow of d_logits */
    int32_t n_sel;
    int32_t sel[CDC_MAX_SEL];
    int32_t pad_;
} cdc_mid_bgate1;
typedef struct {
    const void* wt; int64_t ldwt;             /* bf16 [H1, >= H2]: the expert's W^T (cdc_weight_shadows' transposed copy) */
    int32_t src;
    int32_t pad_;
} cdc_mid_bexpert2;
typedef struct {
    const float* d_out; int64_t ld_dout;      /* fp32 [B, H2]: gradient of this gate's pooled output */
    const float* probs;                       /* [B, n_sel] */
    float* d_logits; int64_t ld_dlogits;      /* [B, n_sel] fp32 (output) */
    void* d_logits_h; int64_t ld_dlogits_h;   /* optional bf16 shadow */
    const void* wt; int64_t ldwt;             /* bf16 [H1, >= 32]: the gate's W^T, columns >= n_sel zero */
    int32_t src;
    int32_t n_sel;
    int32_t sel[CDC_MAX_SEL];
} cdc_mid_bgate2;
typedef struct {
    int64_t B;
    int32_t H1, H2;
    int32_t n_exp1, n_gate1, n_exp2, n_gate2;
    const float* ex1; int64_t ld_ex1;         /* forward activations (masks and gate-gradient dot products) */
    const float* ex2; int64_t ld_ex2;
    void* dz1_h; int64_t ld_dz1_h;            /* bf16 [B, n_exp1*H1] (output): dZ of the level-k experts' last layer */
    void* dz2_h; int64_t ld_dz2_h;            /* bf16 [B, n_exp2*H2] (output): dZ of the level-k+1 experts */
    int32_t mask1, mask2;                     /* 1: d = (activation > 0) ? d * scale : 0 */
    float scale1, scale2;
    cdc_mid_bgate1 g1[CDC_MID_MAX_GATE];
    cdc_mid_bexpert2 e2[CDC_MID_MAX_EXPERT];
    cdc_mid_bgate2 g2[CDC_MID_MAX_GATE];
} cdc_cgc_mid_bwd_args;
int cdc_cgc_mid_bwd(const cdc_cgc_mid_bwd_args* a, void* stream);
/* 1 if both fused launches fit their LDS budget (150 KB) for these expert / gate counts, 0 if not (use the three launches per
 * direction), CDC_E_BADARG for counts outside [1, CDC_MID_MAX_*]. */
int cdc_cgc_mid_fits(int32_t n_exp1, int32_t n_gate1, int32_t n_exp2, int32_t n_gate2);

/* ------------------------------------------------------------------------------------------
 * Two consecutive BatchNorm-free expert layers as ONE forward launch (csrc/pair.hip; reference: the experts'
 * MultiLayerPerceptron(in, (H1, H2), dropout, output_layer=False, bn=False), model/ple.py:83-88, model/layer.py:185-196:
 * Linear, ReLU, Dropout, Linear, ReLU, Dropout).  For every expert e and its M rows:
 *   h_e = drop(relu(x_e . W1_e^T + b1_e))  -> bf16 [M, H1]  (the second layer's operand; kept for the backward launches)
 *   y_e = drop(relu(h_e . W2_e^T + b2_e))  -> fp32 [M, H2] and / or its bf16 copy
 * and optionally a side output of ns <= 16 columns from the same x_e (a gate's logits, model/ple.py:89-94):
 *   ys_e = x_e . WS_e^T + bs_e             -> fp32 [M, ns]
 * Arithmetic (bf16 operands, fp32 accumulate, K walked in 64-wide slabs), rounding of h and the dropout streams (seed1 / seed2 +
 * stream ids) are those of two cdc_gemm_bf16_nt forward launches with the same arguments: results are bit-identical.
 * Operands as for cdc_gemm_bf16_nt: bf16, 16-byte aligned, rows readable to K1r (a multiple of 64) with zero padding in at
 * least one operand of each product.  Instantiated for (H1, H2) = (256, 128); anything else: CDC_E_BADARG.
 * ---------------------------------------------------------------------------------------- */
#define CDC_PAIR_MAX_EXPERT 16
typedef struct {
    const void* x; int64_t ldx;                        /* bf16 [M, >= K1r] */
    const void* w1; int64_t ldw1; const float* b1;     /* bf16 [H1, >= K1r]; bias fp32 [H1] or NULL */
    const void* w2; int64_t ldw2; const float* b2;     /* bf16 [H2, >= H1];  bias fp32 [H2] or NULL */
    void* h; int64_t ldh;                              /* out bf16 [M, H1] (may be NULL: not kept) */
    float* y; int64_t ldy;                             /* out fp32 [M, H2] (may be NULL if yh) */
    void* yh; int64_t ldyh;                            /* out bf16 [M, H2] (may be NULL) */
    const void* ws; int64_t ldws; const float* bs;     /* side output: bf16 [ns, >= K1r] weight rows (NULL: none), bias or NULL */
    float* ys; int64_t ldys;                           /* out fp32 [M, ns] */
    int32_t ns;
    int32_t stream1, stream2;                          /* dropout stream ids of the two layers (the group's index in the unfused launch) */
    int32_t pad_;
} cdc_pair_expert;
typedef struct {
    int32_t n_expert, M, K1r, H1, H2, relu;
    float drop_p;
    int32_t pad_;
    uint64_t seed1, seed2;
    const int32_t* seed_offset_dev;
    cdc_pair_expert e[CDC_PAIR_MAX_EXPERT];
} cdc_expert_pair_args;
int cdc_expert_pair_fwd(const cdc_expert_pair_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm1d (+ReLU, +dropout) over column segments (reference: model/layer.py:187,199-205 with the
 * batch==1 skip; model/star.py:117-181 MDR_BatchNorm with gamma_d*gamma_s / beta_d+beta_s — the caller
 * passes the combined gamma/beta).  Training: batch mean / biased variance, running stats updated with
 * momentum and the unbiased variance (torch semantics).  Eval: running stats.
 * One segment = C columns sharing one parameter set and one row range.
 * ---------------------------------------------------------------------------------------- */
#define CDC_MAX_BN_SEGS 24
#define CDC_BN_ROWS_PER_BLOCK 64      /* rows per partial-sum block: workspace holds ceil(M/64) partials */
#define CDC_BN_X_BF16 1               /* `half` bits: the operand is stored as bf16 (pointer and ld are those of the bf16 array) */
#define CDC_BN_Y_BF16 2
#define CDC_BN_DY_BF16 4
typedef struct {
    const void* x; int64_t ldx;       /* [M,C] pre-norm; fp32, or bf16 with half & CDC_BN_X_BF16 */
    float* y; int64_t ldy;            /* [M,C] output; may be NULL when yh is given (the output then exists as bf16 only) */
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;  /* updated in training */
    float* save_mean; float* save_invstd;     /* [C] saved for backward (training) */
    int64_t* num_batches_tracked;     /* incremented in training when the segment is normalised (may be NULL) */
    void* yh; int64_t ldyh;           /* optional bf16 copy of y, or NULL */
    int32_t C;
    int32_t row_group;                /* index into row_offsets (ragged rows) */
    int32_t half;                     /* CDC_BN_X_BF16 or 0 */
    int32_t pad_;
} cdc_bn_seg;
typedef struct {
    int32_t n_seg;
    int32_t training;
    int32_t relu;
    int32_t skip_le1;                 /* 0: normalisation skipped when the (group's) batch has exactly 1 row
                                         (model/layer.py:202, star.py:134); 1: when it has <= 1 rows (DNN, layer.py:293) */
    float eps, momentum;
    float drop_p; uint64_t seed; const int32_t* seed_offset_dev;
    int64_t M;                        /* rows (upper bound when ragged) */
    const int32_t* row_offsets;       /* device, or NULL: every segment covers rows [0,M) */
    double* workspace;                /* >= 2 * ceil(M/64) * sum(C) doubles (training) */
    int32_t phase;                    /* 0: statistics + normalisation in one call.  Data parallel (global-batch statistics, as
                                         the reference's single process sees them): 1 = statistics only -> `exchange`;
                                         the caller all-reduces (SUM) `exchange` across ranks; 2 = normalise from `exchange` */
    int32_t stats_ready;              /* 1: the partial sums are already in `workspace` (written by the cdc_glinear_fwd epilogues
                                         that produced every segment's x, see cdc_lin_group.bn_partial): no statistics pass */
    double* exchange;                 /* phases 1/2: [2*sum(C) column sums (x, x^2) | n_seg row counts] doubles */
    cdc_bn_seg s[CDC_MAX_BN_SEGS];
} cdc_bn_fwd_args;
int cdc_bn_fwd(const cdc_bn_fwd_args* a, void* stream);

typedef struct {
    const void* dy; int64_t lddy;     /* grad w.r.t. the post-activation output (bf16 with half & CDC_BN_DY_BF16) */
    const void* y;  int64_t ldy;      /* post-activation output: only its sign is used (relu/dropout mask); bf16 with CDC_BN_Y_BF16 */
    const void* x;  int64_t ldx;      /* pre-norm input (bf16 with CDC_BN_X_BF16) */
    float* dx; int64_t lddx;          /* may be NULL when dxh is given and accumulate_dx is 0 (dx then exists as bf16 only) */
    const float* gamma;
    const float* save_mean; const float* save_invstd;      /* training: batch stats; eval: the running
                                                              mean and 1/sqrt(running_var+eps) */
    float* dgamma; float* dbeta;
    void* dxh; int64_t lddxh;         /* optional bf16 shadow of dx (the value after the +=), or NULL */
    int32_t C;
    int32_t row_group;
    int32_t accumulate_dx;            /* 1: dx += (several segments normalise the same input: STAR's domain_norm) */
    int32_t half;                     /* CDC_BN_X_BF16 | CDC_BN_Y_BF16 | CDC_BN_DY_BF16 */
} cdc_bn_bseg;
typedef struct {
    int32_t n_seg;
    int32_t training;
    int32_t relu;
    float eps;
    float mask_scale;                 /* 1/(1-p) when dropout was on */
    int64_t M;
    const int32_t* row_offsets;
    double* workspace;                /* >= 2 * ceil(M/64) * sum(C) doubles */
    int32_t phase;                    /* as in cdc_bn_fwd_args; exchange = [2*sum(C) sums (dz, dz*xhat) | n_seg row counts] */
    int32_t pad_;
    double* exchange;
    cdc_bn_bseg s[CDC_MAX_BN_SEGS];
} cdc_bn_bwd_args;
int cdc_bn_bwd(const cdc_bn_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row dot products: out[b] = sigmoid?( x[b,:]·w + bias + sum_i addend_i[b] )
 * (reference: model/layer.py:122-126 FeaturesLinear, :193 tower output Linear(→1) followed by
 *  `y_logits += other` and Sigmoid model/layer.py:50-55, :318,326 CrossNetwork.w, dcn.py:42)
 * grouped over towers: group g reads x_g, writes out + g*out_gstride.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float* x; int64_t ldx;
    const float* w; const float* bias;    /* bias: device pointer to 1 float or NULL */
    float* out; int64_t ld_out;           /* out[b*ld_out] */
    float* logit; int64_t ld_logit;       /* optional pre-sigmoid copy (NULL = skip) */
    int32_t K;
} cdc_rowdot_group;
typedef struct {
    int32_t n_groups;
    int32_t sigmoid;
    int32_t n_addend;
    const float* addend[4]; int64_t ld_addend[4];
    int64_t M;
    const int32_t* row_offsets;       /* ragged: group g covers rows [ro[g],ro[g+1]) of x/out/addend */
    cdc_rowdot_group g[CDC_MAX_GROUPS];
} cdc_rowdot_fwd_args;
int cdc_rowdot_fwd(const cdc_rowdot_fwd_args* a, void* stream);

typedef struct {
    const float* dout; int64_t ld_dout;   /* grad w.r.t. out (post-sigmoid if sigmoid) */
    const float* out;  int64_t ld_out;    /* forward output (for sigmoid') */
    const float* x; int64_t ldx;
    const float* w;
    float* dx; int64_t lddx;              /* may be NULL */
    float* dw; float* dbias;              /* [K], [1]; partial sums reduced deterministically */
    float* dlogit; int64_t ld_dlogit;     /* optional: grad w.r.t. the logit (for addends), may be NULL */
    int32_t K;
    int32_t accumulate_dx;
} cdc_rowdot_bgroup;
typedef struct {
    int32_t n_groups;
    int32_t sigmoid;
    int64_t M;
    const int32_t* row_offsets;
    float* workspace;                     /* >= n_groups * CDC_ROWDOT_PARTS * (Kmax+1) floats */
    /* Fused BCELoss(mean) (run.py:484,723; cdc_bce_fwd_bwd's arithmetic): when bce_y_i16 or bce_y_f32 is set, the launch's
     * groups are the columns of one prediction matrix (sigmoid must be 1, no row_offsets), g[].dout is NOT read: for row b
     * and group c, dout = (c == col(b)) ? bce_inv_count * (o - t) / max((1 - o) o, 1e-12) : 0 with col(b) = bce_group[b]
     * (clamped into the launch's columns; column 0 when bce_group is NULL), and *bce_loss = mean BCE of the own columns
     * (ordered sum of per-block partials in bce_partial, >= n_groups * CDC_ROWDOT_PARTS doubles). */
    const int64_t* bce_group;
    const int16_t* bce_y_i16;
    const float* bce_y_f32;
    float* bce_loss;
    double* bce_partial;
    float bce_inv_count;
    int32_t pad_;
    cdc_rowdot_bgroup g[CDC_MAX_GROUPS];
} cdc_rowdot_bwd_args;
#define CDC_ROWDOT_PARTS 256
int cdc_rowdot_bwd(const cdc_rowdot_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * The tower head in one launch per direction (reference: BaseModel.tower_forward, model/layer.py:48-56 — the towers' output
 * Linear(->1) model/layer.py:193, `y_logits += other` for the wide term FeaturesLinear model/layer.py:122-126 and further
 * logits, Sigmoid, cat to [B, n_tower]; backward incl. BCELoss(mean) on the row's own tower, run.py:484,723):
 *   shared[b]  = wide_x[b,:] . wide_w + wide_bias                         (formed once per row; optional)
 *   out[b,t]   = sigmoid?( x_t[b,:] . w_t + bias_t + shared[b] + sum_i addend_i[b] )
 * backward: d_t = d_out[b,t] (or the fused BCE gradient) * out (1 - out);  dx_t = d_t w_t;  dw_t = sum_b d_t x_t;  db_t = sum_b d_t;
 *   dsum = sum_t d_t  ->  d_addend_i (+)= dsum;  wide_dx (+)= dsum wide_w;  wide_dw = sum_b dsum wide_x;  wide_dbias = sum_b dsum.
 * Cross-row sums: CDC_ROWDOT_PARTS row parts in `workspace` (cdc_head_workspace_floats), added in index order (deterministic).
 * ---------------------------------------------------------------------------------------- */
#define CDC_HEAD_MAX_TOWERS 8
typedef struct {
    const float* x; int64_t ldx;          /* [M,K] the tower's last hidden activations */
    const float* w; const float* bias;    /* [K], [1] or NULL */
    float* dx; int64_t lddx;              /* backward: [M,K] or NULL */
    float* dw; float* dbias;              /* backward: [K], [1] (may be NULL) */
    int32_t K;
    int32_t accumulate_dx;
} cdc_head_tower;
typedef struct {
    int32_t n_tower, sigmoid, n_addend, pad_;
    int64_t M;
    float* out; int64_t ld_out;           /* [M, n_tower] forward output (read by the backward) */
    const float* d_out; int64_t ld_dout;  /* backward without the fused loss: grad w.r.t. out */
    const float* wide_x; int64_t ld_wide; /* [M, wide_K] or NULL: the wide term's input (the gathered embeddings) */
    const float* wide_w; const float* wide_bias;
    float* wide_out; int64_t ld_wide_out; /* optional copy of shared[b] ([M,1]), may be NULL */
    float* wide_dx; int64_t ld_wide_dx;   /* backward: [M, wide_K] or NULL */
    float* wide_dw; float* wide_dbias;
    int32_t wide_K, accumulate_wide_dx;
    const float* addend[2]; int64_t ld_addend[2];       /* further logits added to every tower ([M,1] each) */
    float* d_addend[2]; int64_t ld_d_addend[2];         /* backward: their gradients (may be NULL) */
    int32_t accumulate_d_addend[2];
    float* workspace;                     /* backward: >= cdc_head_workspace_floats() floats */
    /* fused BCELoss(mean) as in cdc_rowdot_bwd_args: set bce_y_i16 or bce_y_f32 */
    const int64_t* bce_group;
    const int16_t* bce_y_i16;
    const float* bce_y_f32;
    float* bce_loss;
    double* bce_partial;                  /* >= CDC_ROWDOT_PARTS doubles */
    float bce_inv_count;
    int32_t pad2_;
    cdc_head_tower t[CDC_HEAD_MAX_TOWERS];
} cdc_head_args;
int cdc_head_fwd(const cdc_head_args* a, void* stream);
int cdc_head_bwd(const cdc_head_args* a, void* stream);
int64_t cdc_head_workspace_floats(const cdc_head_args* a);

/* ------------------------------------------------------------------------------------------
 * The towers of a multi-tower model as ONE launch per direction (csrc/tower.hip, round 4).
 * Reference: BaseModel.tower_forward (model/layer.py:35-56) over towers built by build_tower_output as
 * MultiLayerPerceptron(H0, (H1, H2), dropout, output_layer=True) (model/layer.py:178-206):
 *     Linear(H0->H1) -> BatchNorm1d -> ReLU -> Dropout -> Linear(H1->H2) -> BatchNorm1d -> ReLU -> Dropout -> Linear(H2->1)
 *     -> `+= other` (the wide term, FeaturesLinear model/layer.py:122-126) -> Sigmoid -> column t of out [M, n_tower],
 * in training mode (batch statistics); the backward also forms BCELoss(mean) on the row's own tower and its gradient
 * (run.py:484,723) when the bce_* fields are set, like cdc_head_bwd.
 *
 * It replaces cdc_gemm_bf16_nt -> cdc_bn_fwd -> cdc_gemm_bf16_nt -> cdc_bn_fwd -> cdc_head_fwd (5 launches) and
 * cdc_head_bwd (2) -> cdc_bn_bwd (2) -> cdc_gemm_bf16_nt -> cdc_bn_bwd (2) -> cdc_gemm_bf16_nt (8 launches) with the same
 * arithmetic: bf16 MFMA operands / fp32 accumulate for the two contractions and their grad-input products, fp64 column sums
 * for the BatchNorm statistics (forward: per 64-row chunk in the order of cdc_gemm_bf16_nt's statistics epilogue and
 * cdc_bn_fwd's chunk sum, i.e. the same bits), the 32-bit dropout stream of the BatchNorm launches (stream 64 + tower).
 * The grad-weight contractions of the two Linear layers stay with the batched grad-weight launch (cdc_glinear_bwd_w): the
 * backward writes their dZ operands as bf16 (`dzh`), the forward the hidden activation `a1h`.
 *
 * Geometry: a workgroup owns CDC_TOWER_ROWS rows of one tower; grid = n_tower * ceil(M / CDC_TOWER_ROWS) workgroups, which
 * must all be resident (checked against 256 CUs: M <= 256 / n_tower * CDC_TOWER_ROWS, else CDC_E_TOOBIG).  The column sums a
 * BatchNorm needs over all rows are exchanged INSIDE the launch: every workgroup publishes its partial sums with write-through
 * stores (a published value is never all-zero bits), loads the partials of every workgroup of its tower — again while any slot is
 * still empty (bounded) — and adds them up in a fixed order (deterministic; identical in every workgroup).  A poll that runs out
 * sets bit CDC_TOWER_ERR_TIMEOUT in *err (results of that step are then undefined) and ends the remaining polls of the launch.
 * `workspace`: cdc_tower_workspace_bytes() bytes, zeroed once by the caller, private to one (fwd, bwd) pair at a time.
 * Instantiated for H0 in {64, 128}, H1 = 64, H2 = 32 (config.py:39-42: tower_dims (64, 32)); anything else: CDC_E_BADARG.
 * ---------------------------------------------------------------------------------------- */
#define CDC_TOWER_MAX 4
#define CDC_TOWER_ROWS 128
#define CDC_TOWER_ERR_TIMEOUT 0x40000000
typedef struct {
    const void* wh; int64_t ldwh;         /* [N, K64] bf16 weight copy, rows zero-padded to whole 64-element slabs (forward) */
    const void* wt; int64_t ldwt;         /* [K, N64] bf16 transposed copy (backward: grad-input) */
    const float* bias;                    /* [N] */
    float* z; int64_t ldz;                /* [M, N] fp32 pre-normalisation output: written by the forward, read by the backward */
    void* dzh; int64_t lddzh;             /* backward: [M, N] bf16 gradient w.r.t. z (operand of the grad-weight launch) */
    const float* gamma; const float* beta;            /* BatchNorm1d that follows: [N] each */
    float* running_mean; float* running_var;          /* updated by the forward (momentum, unbiased variance) */
    int64_t* num_batches_tracked;                     /* incremented by the forward (may be NULL) */
    float* save_mean; float* save_invstd;             /* [N]: forward -> backward */
    float* dgamma; float* dbeta;                      /* backward (may be NULL) */
} cdc_tower_layer;
typedef struct {
    const void* xh; int64_t ldxh;         /* [M, H0] bf16 copy of the tower's input, readable up to whole 64-element slabs */
    float* dx; int64_t lddx;              /* backward: [M, H0] fp32 gradient w.r.t. the input */
    int32_t accumulate_dx, pad_;
    cdc_tower_layer l1, l2;
    void* a1h; int64_t lda1h;             /* [M, H1] bf16 hidden activation after BatchNorm/ReLU/dropout (forward -> backward, grad-weight) */
    float* a2; int64_t lda2;              /* [M, H2] fp32 second hidden activation (forward -> backward) */
    const float* wo; const float* bo;     /* output Linear(H2 -> 1): [H2], [1] or NULL */
    float* dwo; float* dbo;               /* backward: [H2], [1] (may be NULL) */
    float* dy2; int64_t lddy2;            /* cdc_tower_dp only: [M, H2] / [M, H1] fp32 scratch — the masked gradient w.r.t. a BatchNorm's */
    float* dy1; int64_t lddy1;            /* output, handed from one phase to the next */
} cdc_tower_desc;
typedef struct {
    int32_t n_tower, H0, H1, H2;
    int64_t M;
    int32_t relu, sigmoid;
    float drop_p, eps, momentum;
    int32_t pad_;
    uint64_t seed1, seed2;                /* dropout streams of the two BatchNorm launches this replaces */
    const int32_t* seed_offset_dev;
    float* out; int64_t ld_out;           /* [M, n_tower] */
    const float* d_out; int64_t ld_dout;  /* backward without the fused loss */
    const float* wide_x; int64_t ld_wide; /* [M, wide_K] or NULL: input of the wide term added to every tower's logit */
    const float* wide_w; const float* wide_bias;
    float* wide_dx; int64_t ld_wide_dx;   /* backward: [M, wide_K] or NULL */
    float* wide_dw; float* wide_dbias;
    int32_t wide_K, accumulate_wide_dx;
    /* fused BCELoss(mean) as in cdc_head_args: set bce_y_i16 or bce_y_f32 */
    const int64_t* bce_group;
    const int16_t* bce_y_i16;
    const float* bce_y_f32;
    float* bce_loss;
    float bce_inv_count;
    int32_t pad2_;
    void* workspace;                      /* >= cdc_tower_workspace_bytes(), zeroed once */
    int32_t* err;                         /* device word that receives CDC_TOWER_ERR_TIMEOUT (may be NULL) */
    double* exchange[4];                  /* cdc_tower_dp only: per exchange [2 * n_tower * C column sums | n_tower row counts] doubles
                                             (C = H1, H2, H2, H1): written by the phase in front of it, summed over the ranks by the caller */
    cdc_tower_desc t[CDC_TOWER_MAX];
} cdc_tower_args;
int64_t cdc_tower_workspace_bytes(const cdc_tower_args* a);
int cdc_tower_fwd(const cdc_tower_args* a, void* stream);
int cdc_tower_bwd(const cdc_tower_args* a, void* stream);
/* Both directions of a TRAINING step in one launch, for the fused-loss form only (bce_y_* set: the output gradient does not come
 * from outside, so a workgroup's backward needs nothing but what its own forward produced).  `a` holds what cdc_tower_bwd takes —
 * a superset of cdc_tower_fwd's.  Results bit-identical to cdc_tower_fwd followed by cdc_tower_bwd. */
int cdc_tower_step(const cdc_tower_args* a, void* stream);
/* The same towers under DATA PARALLELISM with global-batch BatchNorm statistics (the reference's single process sees the global
 * batch): the two launches cut at their four exchange points into six phases — 1, 2, 3 forward; 4, 5, 6 backward.  Phases 1, 2, 4, 5
 * end with the LOCAL column sums of the statistics the next phase needs in exchange[0..3] (the last workgroup of a tower to finish
 * adds the tower's partials up, in cdc_bn_fwd's order; nobody waits inside a launch); the caller sums each exchange buffer over
 * the ranks (all-reduce) before the next phase, which normalises with the GLOBAL sums and row count.  Parameter gradients stay
 * local sums (the data-parallel all-reduce of the gradient arena adds the ranks).  Same arithmetic as cdc_tower_fwd / _bwd;
 * a one-rank run gives their results.  M is this rank's row count (may differ between ranks). */
int cdc_tower_dp(const cdc_tower_args* a, int32_t phase, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss (reference: run.py:484,723 — BCELoss(mean) on probabilities gathered by group column,
 * log clamped at -100; backward as aten::binary_cross_entropy_backward with eps 1e-12)
 *   p [B, n_col]; group [B] int64 column per row (NULL => column 0); y int16/float labels.
 *   loss (device float) = mean_b bce(p[b,group[b]], y[b]) * loss_scale
 *   dp [B, n_col] = d loss / d p (zeros elsewhere); inv_count = 1/global_batch.
 * ---------------------------------------------------------------------------------------- */
int cdc_bce_fwd_bwd(const float* p, int64_t ldp, const int64_t* group, const int16_t* y_i16,
                    const float* y_f32, float* loss, float* dp, int64_t lddp,
                    int64_t B, int32_t n_col, float inv_count, void* stream);
/* the same on the MEAN over the n_col tower probabilities of a row (CDC warm-up: cdc.py:100-102 + run.py:616-617);
 * dp[b, c] = bce'(mean_b) / n_col for every column. */
int cdc_bce_mean_fwd_bwd(const float* p, int64_t ldp, const int16_t* y_i16, const float* y_f32, float* loss, float* dp,
                         int64_t lddp, int64_t B, int32_t n_col, float inv_count, void* stream);

/* Second-order factorisation-machine term (reference: model/layer.py:160-175 FactorizationMachine(reduce_sum=True), used by
 * model/dfm.py:33): e [B, F*D] gathered embeddings; out[b] = 0.5 * sum_d((sum_f e)^2 - sum_f e^2);
 * de[b,f,d] (+)= dout[b] * (sum_f' e[b,f',d] - e[b,f,d]). */
int cdc_fm_fwd(const float* e, int64_t lde, float* out, int64_t ldo, int64_t B, int32_t F, int32_t D, void* stream);
int cdc_fm_bwd(const float* e, int64_t lde, const float* dout, int64_t ldd, float* de, int64_t ldde, int64_t B, int32_t F,
               int32_t D, int32_t accumulate, void* stream);

/* Per-row choice among n_group side-by-side feature blocks (reference: model/hinet.py:71-74, `con_feas[mask_g] =
 * specific_feas[g][mask_g]`): out[b,:] = feas[b, group[b]*H:(group[b]+1)*H] (zeros for a group id outside [0,n_group));
 * backward: d_feas gets d_out in the chosen block and nothing elsewhere (zeros are stored unless `accumulate`). */
int cdc_group_select_fwd(const float* feas, int64_t ldf, const int64_t* group, float* out, int64_t ldo, int64_t B,
                         int32_t n_group, int32_t H, void* stream);
int cdc_group_select_bwd(const float* dout, int64_t ldd, const int64_t* group, float* dfeas, int64_t ldf, int64_t B,
                         int32_t n_group, int32_t H, int32_t accumulate, void* stream);

/* Sigmoid gate (reference: model/adasparse.py:52-56 — the pruner: beta 2, alpha 1, eps 0.25; model/pepnet.py:79-80,125 —
 * GateNN's `sigmoid(.) * 2` applied to its input: eps < 0):  pi = beta*sigmoid(alpha*p), pi = 0 where |pi| <= eps,
 * out = a * pi.  Backward: da (+)= dout*pi, dp (+)= dout*a*d(pi)/dp (0 where pruned); da or dp may be NULL (detached). */
int cdc_sigmoid_gate_fwd(const float* a, int64_t lda, const float* p, int64_t ldp, float* out, int64_t ldo, int64_t rows,
                         int32_t cols, float beta, float alpha, float eps, void* stream);
int cdc_sigmoid_gate_bwd(const float* a, int64_t lda, const float* p, int64_t ldp, const float* dout, int64_t lddo, float* da,
                         int64_t ldda, int32_t acc_a, float* dp, int64_t lddp, int32_t acc_p, int64_t rows, int32_t cols,
                         float beta, float alpha, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * CrossNetwork (DCN v1) layer (reference: model/layer.py:321-329): out = x0 * (xl·w) + b + xl
 * ---------------------------------------------------------------------------------------- */
int cdc_cross_fwd(const float* x0, int64_t ld0, const float* xl, int64_t ldl, const float* w, const float* b,
                  float* out, int64_t ldo, float* xw_save, int64_t B, int32_t E, void* stream);
/* d_x0 += , d_xl = , dw = , db =  (workspace >= CDC_ROWDOT_PARTS*2*E floats) */
int cdc_cross_bwd(const float* d_out, int64_t ldo, const float* x0, int64_t ld0, const float* xl, int64_t ldl,
                  const float* w, const float* xw_save, float* d_x0_acc, int64_t ld_dx0, float* d_xl, int64_t ld_dxl,
                  float* dw, float* db, float* workspace, int64_t B, int32_t E, void* stream);

/* ------------------------------------------------------------------------------------------
 * DCN-v2 element-wise pieces (reference: model/layer.py:339-343 CrossNetV2 `x0 * W(x) + b + x`,
 * model/layer.py:384-396 CrossNetMix `tanh`, `x_0 * (uv_x + bias)`, and the `moe_out + x_l` residual :403)
 * The contractions themselves go through cdc_glinear_*.
 * ---------------------------------------------------------------------------------------- */
int cdc_tanh_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int32_t cols, void* stream);
/* dx (=|+=) dy * (1 - y^2) */
int cdc_tanh_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dx, int64_t lddx,
                 int64_t rows, int32_t cols, int32_t accumulate, void* stream);
/* out[:, k*P+e] = x0[:, e] * (u[:, k*P+e] + b1[e]) + b2[e] + r[:, k*P+e]   for k < n_rep, e < P = period
 * (b1, b2 [P] and r [rows, P*n_rep] optional; n_rep > 1 = the experts of one CrossNetMix layer in one launch) */
int cdc_cross_combine_fwd(const float* x0, int64_t ld0, const float* u, int64_t ldu, const float* b1, const float* b2,
                          const float* r, int64_t ldr, float* out, int64_t ldo, int64_t rows, int32_t period, int32_t n_rep,
                          void* stream);
/* d_u = d_out*x0 ; d_x0 += sum_k d_out*(u+b1) ; d_r (=|+=) d_out ; db1 (=|+=) colsum(d_out*x0) ; db2 (=|+=) colsum(d_out)
 * workspace >= CDC_ROWDOT_PARTS * 2 * period floats */
int cdc_cross_combine_bwd(const float* d_out, int64_t lddo, const float* x0, int64_t ld0, const float* u, int64_t ldu,
                          const float* b1, float* d_u, int64_t lddu, float* d_x0_acc, int64_t lddx0, float* d_r, int64_t lddr,
                          int32_t accumulate_r, float* db1, int32_t accumulate_b1, float* db2, int32_t accumulate_b2,
                          float* workspace, int64_t rows, int32_t period, int32_t n_rep, void* stream);
int cdc_add_out(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo,
                int64_t rows, int32_t cols, void* stream);
int cdc_copy_or_add(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t rows, int32_t cols,
                    int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense-parameter Adam, multi-tensor (reference: run.py:720-721 torch.optim.Adam(lr, betas=(0.9,0.99),
 * eps=1e-8, weight_decay=wd) + the L2 term of model/layer.py:96-112 whose gradient is 2*l2*w):
 *   g = grad + 2*l2_i*w + wd*w ; Adam step t ; reg_sum += l2_i * sum(w_old^2) (device double; + *reg_seed once).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float* w; const float* g; float* m; float* v;
    int64_t n;
    float l2;
    int32_t n_slabs;                  /* > 0: the gradient is the sum of n_slabs split-K slabs of a grad-weight launch whose second */
    const float* slabs;               /* launch was deferred (cdc_lin_bwdw_args.defer_reduce): g[i] = ((0 + slabs[i]) + slabs[stride+i]) */
    int64_t slab_stride;              /* + ... in slab order — the sums k_bwd_w_reduce forms, bit for bit; `g` is not read */
} cdc_adam_tensor;
typedef struct {
    int32_t n_tensors;
    float lerp_w, beta2, one_minus_beta2, eps, weight_decay;   /* as in cdc_adam_hp */
    const float* step_scalars; int32_t n_scalars;
    float grad_scale;                 /* g is multiplied by this first (1/world_size after a sum all-reduce) */
    const int32_t* step_dev;
    double* reg_sum;                  /* may be NULL */
    const double* reg_seed;           /* may be NULL: *reg_sum += *reg_seed, once per launch (the lazy table's cached l2*sum(w^2)) */
    cdc_adam_tensor t[CDC_MAX_TENSORS];
} cdc_adam_args;
int cdc_adam_multi(const cdc_adam_args* a, void* stream);
/* the same for ANY number of tensors in one launch: the descriptors (a->t is not read) and the workgroup -> (tensor, chunk of
 * CDC_ADAM_CHUNK elements) map are device arrays the caller builds once per parameter set: workgroup i updates elements
 * [wg_chunk[i] * CDC_ADAM_CHUNK, ...) of tensor wg_tensor[i].  (cdc_adam_multi's limit of CDC_MAX_TENSORS is the 4 KB kernel-argument
 * block; a model with more dense tensors needed several launches.) */
#define CDC_ADAM_CHUNK 4096
int cdc_adam_multi_table(const cdc_adam_args* a, const cdc_adam_tensor* tensors_dev, const int32_t* wg_tensor_dev,
                         const int32_t* wg_chunk_dev, int32_t n_workgroups, void* stream);
/* The two updates that end a training step with the lazy table in ONE launch: cdc_embed_segsum_lazy_update(short_only = 0) on the
 * step's rows and cdc_adam_multi_table on the dense parameters (they touch disjoint memory; each alone is a short launch living on
 * memory latency, one behind the other 20 us + 23 us of the C2 step).  Arguments: those of the two calls.  Same arithmetic and
 * summation orders as the two launches: identical weights and moments; the regularisation sum is added up in another order
 * (double atomics in both forms). */
int cdc_embed_segsum_lazy_update_dense(const float* d_out, const int32_t* seg_start, const int32_t* perm, const int32_t* uniq_cnt,
                                       const int32_t* uniq_row, float* w, float* m, float* v, int32_t* last, cdc_adam_hp hp,
                                       const int32_t* step_dev, int64_t B, int32_t F, int32_t D, int32_t short_only,
                                       const cdc_adam_args* dense, const cdc_adam_tensor* tensors_dev, const int32_t* wg_tensor_dev,
                                       const int32_t* wg_chunk_dev, int32_t n_dense_workgroups, void* stream);

/* ------------------------------------------------------------------------------------------
 * STAR parameter fusion (reference: model/star.py:90-93,100-102,169-176): for every domain g
 *   out_g = a_g * s   (op 0: weights, BatchNorm gamma)   or   out_g = a_g + s   (op 1: biases, BatchNorm beta)
 * backward: da_g = d_out_g * s | d_out_g ;  ds = sum_g d_out_g * a_g | sum_g d_out_g  (fixed order over g).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n;                        /* domains in this launch (<= CDC_MAX_GROUPS) */
    int32_t op;                       /* 0 = multiply, 1 = add */
    int64_t size;                     /* elements per tensor */
    const float* s;                   /* shared tensor */
    float* ds;                        /* backward: gradient of the shared tensor */
    int32_t accumulate_ds;            /* backward: ds += (a further chunk of domains) */
    int32_t pad_;
    const float* a[CDC_MAX_GROUPS];   /* per-domain tensors */
    float* out[CDC_MAX_GROUPS];       /* forward outputs / backward: d_out */
    float* da[CDC_MAX_GROUPS];        /* backward: gradients of the per-domain tensors */
} cdc_star_fuse_args;
int cdc_star_fuse_fwd(const cdc_star_fuse_args* a, void* stream);
int cdc_star_fuse_bwd(const cdc_star_fuse_args* a, void* stream);

/* out[r,c] (=|+=) sum_g in[r, g*cols + c]: fan-in of per-tower gradients of one shared input (fixed order over g) */
int cdc_sum_slices(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows, int32_t cols,
                   int32_t n_slices, int32_t accumulate, void* stream);

/* small utilities used by the step driver */
int cdc_step_increment(int32_t* step_dev, void* stream);                 /* ++*step_dev */
int cdc_begin_step(int32_t* step_dev, double* accumulators, int32_t n_acc, void* stream);
/* one batch into the static buffers of a replayed launch sequence: ids int32 [B,F], labels int16 [B], tower index int64 [B]
 * (group / group_dst may be NULL) — the three tensors run.py:476-479 hands to a step; one launch instead of three copies.
 * field_dims (device int32 [F], may be NULL): an id outside [0, field_dims[f]) sets *alias_flag to 1 + its flat position
 * (atomic max).  Such an id either leaves the table (cdc_embed_gather_fwd flags that one too) or ALIASES a row of another field:
 * the reference gathers that row like any other (model/layer.py:152-153) and its dense backward sums both fields' gradients
 * into one Adam update, while the per-field row lists of this path would update the row once per field — so a training step
 * reports it (TrainStep.check_ids) instead of diverging silently. */
int cdc_stage_batch(const int32_t* ids, const int16_t* y, const int64_t* group, int32_t* ids_dst, int16_t* y_dst,
                    int64_t* group_dst, int64_t B, int32_t F, const int32_t* field_dims, int32_t* alias_flag, void* stream);
/* the same for a step whose row sort runs one step AHEAD (the sort of batch t+1 beside the forward/backward of batch t): this batch as
 * above, the next batch's ids (next_ids, may be NULL) into next_dst — the buffer the look-ahead sort reads — and, with step_dev,
 * cdc_begin_step's work (++*step_dev, accumulators[0..n_acc) = 0), which the sort's first launch does otherwise */
int cdc_stage_batch_next(const int32_t* ids, const int16_t* y, const int64_t* group, int32_t* ids_dst, int16_t* y_dst,
                         int64_t* group_dst, int64_t B, int32_t F, const int32_t* next_ids, int32_t* next_dst,
                         int32_t* step_dev, double* accumulators, int32_t n_acc, const int32_t* field_dims, int32_t* alias_flag,
                         void* stream);
int cdc_fill_f32(float* p, float value, int64_t n, void* stream);
int cdc_fill_f64(double* p, double value, int64_t n, void* stream);
/* dst[r*ld_dst + c] += src[r*ld_src + c]  (gradient fan-in where a kernel cannot accumulate itself) */
int cdc_add_inplace(float* dst, int64_t ld_dst, const float* src, int64_t ld_src, int64_t rows, int32_t cols, void* stream);
/* dst (=|+=) src_0 + src_1 + ... + src_{n-1}, added in list order: the same fan-in from several producers in one launch
 * (the towers' logit gradients into the gradient of a shared addend, model/layer.py FeaturesLinear under every tower). */
typedef struct {
    float* dst; int64_t ld_dst;
    int64_t rows; int32_t cols;
    int32_t n;                        /* sources (<= CDC_MAX_GROUPS) */
    int32_t accumulate;               /* 1: dst += sum, 0: dst = sum */
    const float* src[CDC_MAX_GROUPS];
    int64_t ld_src[CDC_MAX_GROUPS];
} cdc_add_n_args;
int cdc_add_n(const cdc_add_n_args* a, void* stream);
/* out[i] = a[i] * b[i % nb]  (STAR: W_d ⊙ W_s, model/star.py:90) ; and its two gradients */
int cdc_mul_bcast(const float* a, const float* b, float* out, int64_t na, int64_t nb, void* stream);
int cdc_mul_bcast_bwd(const float* d_out, const float* a, const float* b, float* da, float* db,
                      int64_t na, int64_t nb, void* stream);
/* stable partition of batch rows by group id (reference: model/star.py:84-86,113-114):
 *   counts[g], row_offsets[g] (exclusive scan, n_group+1 entries), order[pos] = source row. */
int cdc_group_partition(const int64_t* group, int32_t* row_offsets, int32_t* order,
                        int64_t B, int32_t n_group, void* stream);
/* out[pos,:] = in[order[pos],:]  and the inverse scatter out[order[pos],:] = in[pos,:] */
int cdc_rows_permute(const float* in, int64_t ld_in, const int32_t* order, float* out, int64_t ld_out,
                     int64_t B, int32_t C, int32_t inverse, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Attention branch (reference: model/layer.py:58-84 BaseModel.build_atten / atten_forward — on by default
 * through config.py:24-28).  The linears around it are cdc_glinear_* / cdc_rowdot_* calls on [B*F, .] token buffers; these
 * entry points are the core of torch's nn.MultiheadAttention in between, per sample over its F field tokens:
 *   qkv [B*F, 3A] (row b*F+f; q | k | v, head h in columns h*dh..(h+1)*dh of each third), dh = A/H
 *   probs [B, H, F, F] = softmax_j(q_i*dh^-1/2 . k_j)          (saved for the backward)
 *   out   [B*F, A]: head h's columns = dropout(probs) @ v      (dropout on the probabilities, training only)
 * F <= 64, dh in {4, 8, 16, 32, 64}.  The backward writes every column of dqkv. */
int cdc_attn_fwd(const float* qkv, int64_t ld, float* out, int64_t ldo, float* probs, int64_t B, int32_t F, int32_t A, int32_t H,
                 float drop_p, uint64_t seed, const int32_t* seed_offset_dev, void* stream);
int cdc_attn_bwd(const float* qkv, int64_t ld, const float* probs, const float* dout, int64_t lddo, float* dqkv, int64_t lddq,
                 int64_t B, int32_t F, int32_t A, int32_t H, float drop_p, uint64_t seed, const int32_t* seed_offset_dev,
                 void* stream);
/* out = relu(a + b) (model/layer.py:80-82) and its backward: (out > 0 ? dout : 0) to both addends (stored or added) */
int cdc_add_relu_fwd(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int64_t rows, int32_t cols,
                     void* stream);
int cdc_add_relu_bwd(const float* out, int64_t ldo, const float* dout, int64_t lddo, float* da, int64_t ldda, int32_t acc_a,
                     float* db, int64_t lddb, int32_t acc_b, int64_t rows, int32_t cols, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Evaluation metrics — run.py:684-711: roc_auc_score + log_loss over the whole evaluation set and per
 * domain (evaluate_multi_domain's groupby).  pred f32 [n] (the gathered tower probabilities), label int16 [n] (0/1),
 * domain int32, element i at domain[i*ld_domain] (e.g. the domain column of X: ld = F); NULL when n_domain == 1.
 *   out    [2*(n_domain+1)] doubles: auc of domain 0..n_domain-1, then of ALL rows; then the log-losses in the same order.
 *          A segment with no row or a single class gets NaN for both (the reference's ValueError branch, run.py:699-704).
 *   counts [2*(n_domain+1)] int64: rows per segment, then positives per segment.
 *   err_flag (optional): 1 + index of a row with a NaN score, a label other than 0/1 or a domain outside [0, n_domain).
 * AUC = Mann-Whitney U with mid-ranks, rank sums in integer arithmetic (order-independent); loss in double with sklearn's
 * clipping to the float32 epsilon.  workspace: cdc_eval_workspace_bytes(n, n_domain) bytes, 256-byte aligned. */
int64_t cdc_eval_workspace_bytes(int64_t n, int32_t n_domain);
int cdc_eval_metrics(const float* pred, const int16_t* label, const int32_t* domain, int64_t ld_domain, int64_t n,
                     int32_t n_domain, double* out, int64_t* counts, int32_t* err_flag, void* workspace,
                     int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CDCMDR_H */
