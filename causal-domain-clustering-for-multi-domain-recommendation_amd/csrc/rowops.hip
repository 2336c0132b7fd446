// rowops.hip — the row-wise (HBM / L2 bound) operators between the MFMA contractions:
// gate softmax + expert pooling, BatchNorm(+ReLU+dropout), row dot products (+sigmoid),
// BCE loss, the DCN-v1 cross layer.  One wave (64 lanes) per batch row wherever a row-wise
// reduction is needed; reductions across rows go through order-fixed partial sums (no float atomics).
#include "common.h"

#define ROW_THREADS 256
#define WAVES_PER_BLOCK (ROW_THREADS / 64)

// =================================================================================================
// gate softmax + pooling   (model/ple.py:89-94,105-123 ; model/mmoe.py:37-40,58-60)
// =================================================================================================
__global__ void __launch_bounds__(ROW_THREADS) k_gate_pool_fwd(const cdc_pool_fwd_args a) {
    CDC_PRIO_MAIN();
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= a.B) return;
    const float* ex = a.experts + row * a.ld_exp;
    for (int g = 0; g < a.n_gates; ++g) {
        const auto& G = a.gate[g];
        const float* lg = G.logits + row * G.ld_logits;
        // softmax over n_sel <= 16 logits, every lane redundantly (torch.softmax: max-subtract, exp, normalise)
        float p[CDC_MAX_SEL];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) {
            p[j] = j < G.n_sel ? lg[j] : -INFINITY;
            mx = fmaxf(mx, p[j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) {
            p[j] = j < G.n_sel ? expf(p[j] - mx) : 0.f;
            sum += p[j];
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) p[j] *= inv;
        if (lane < G.n_sel && G.probs) {
            float pv = 0.f;
#pragma unroll
            for (int j = 0; j < CDC_MAX_SEL; ++j) if (j == lane) pv = p[j];
            G.probs[row * G.n_sel + lane] = pv;
        }
        for (int h = lane; h < a.H; h += 64) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < CDC_MAX_SEL; ++j)
                if (j < G.n_sel) acc += p[j] * ex[(int64_t)G.sel[j] * a.H + h];   // same order as torch.sum(dim=1)
            G.out[row * G.ld_out + h] = acc;
            if (G.out_h) reinterpret_cast<__bf16*>(G.out_h)[row * G.ld_out_h + h] = (__bf16)acc;
        }
    }
}

// The same with 16-byte lanes: a row's H values are covered by H/4 lanes (a power of two), so one wave handles 64/(H/4)
// rows and every global access is a float4.  Per element the sum over the selected experts keeps its order, so the
// results equal the scalar kernel's bit for bit.
typedef float pool_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 pool_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pool_store_h4(void* base, int64_t off, const pool_f4& v) {       // 4 x bf16 = one 8-byte store
    pool_h4 h;
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = (__bf16)v[q];
    *reinterpret_cast<pool_h4*>(reinterpret_cast<__bf16*>(base) + off) = h;
}
static bool pool_vec_ok(int H, const void* experts, int64_t ld_exp) {
    const int gl = H / 4;
    return H % 4 == 0 && gl >= 1 && gl <= 64 && (gl & (gl - 1)) == 0 && (((uintptr_t)experts & 15) == 0) && ld_exp % 4 == 0;
}
__global__ void __launch_bounds__(ROW_THREADS) k_gate_pool_fwd_v4(const cdc_pool_fwd_args a) {
    CDC_PRIO_MAIN();
    const int lane = threadIdx.x & 63;
    const int gl = a.H / 4, l = lane % gl;
    const int64_t row = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * (64 / gl) + lane / gl;
    if (row >= a.B) return;
    const pool_f4* ex = reinterpret_cast<const pool_f4*>(a.experts + row * a.ld_exp);
    for (int g = 0; g < a.n_gates; ++g) {
        const auto& G = a.gate[g];
        const float* lg = G.logits + row * G.ld_logits;
        float p[CDC_MAX_SEL];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) {
            p[j] = j < G.n_sel ? lg[j] : -INFINITY;
            mx = fmaxf(mx, p[j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) {
            p[j] = j < G.n_sel ? expf(p[j] - mx) : 0.f;
            sum += p[j];
        }
        const float inv = 1.f / sum;
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) p[j] *= inv;
        if (G.probs) {
            for (int j0 = l; j0 < G.n_sel; j0 += gl) {               // the row's lanes share the n_sel stores
                float pv = 0.f;
#pragma unroll
                for (int j = 0; j < CDC_MAX_SEL; ++j) if (j == j0) pv = p[j];
                G.probs[row * G.n_sel + j0] = pv;
            }
        }
        pool_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j)
            if (j < G.n_sel) acc += p[j] * ex[(int64_t)G.sel[j] * gl + l];
        reinterpret_cast<pool_f4*>(G.out + row * G.ld_out)[l] = acc;
        if (G.out_h) pool_store_h4(G.out_h, row * G.ld_out_h + 4 * l, acc);
    }
}

extern "C" int cdc_gate_pool_fwd(const cdc_pool_fwd_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_gates > 0 && a->n_gates <= CDC_MAX_GATES && a->n_expert > 0 && a->H > 0 && a->B >= 0 && a->experts,
                  CDC_E_BADARG, "gate_pool_fwd: bad argument");
    for (int g = 0; g < a->n_gates; ++g) {
        CDC_CHECK_ARG(a->gate[g].logits && a->gate[g].out && a->gate[g].n_sel > 0 && a->gate[g].n_sel <= CDC_MAX_SEL, CDC_E_BADARG,
                      "gate_pool_fwd: gate %d malformed", g);
        for (int j = 0; j < a->gate[g].n_sel; ++j)
            CDC_CHECK_ARG(a->gate[g].sel[j] >= 0 && a->gate[g].sel[j] < a->n_expert, CDC_E_BADARG, "gate_pool_fwd: gate %d selects expert out of range", g);
    }
    if (a->B == 0) return 0;
    if (pool_vec_ok(a->H, a->experts, a->ld_exp)) {
        bool ok = true;
        for (int g = 0; g < a->n_gates; ++g)
            ok = ok && (((uintptr_t)a->gate[g].out & 15) == 0) && (a->gate[g].ld_out % 4 == 0) &&
                 (!a->gate[g].out_h || ((((uintptr_t)a->gate[g].out_h & 7) == 0) && a->gate[g].ld_out_h % 4 == 0));
        if (ok) {
            const int rows_per_block = WAVES_PER_BLOCK * (64 / (a->H / 4));
            hipLaunchKernelGGL(k_gate_pool_fwd_v4, dim3(cdc_ceil_div(a->B, rows_per_block)), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a);
            CDC_LAUNCH_CHECK("gate_pool_fwd");
            return 0;
        }
    }
    hipLaunchKernelGGL(k_gate_pool_fwd, dim3(cdc_ceil_div(a->B, WAVES_PER_BLOCK)), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("gate_pool_fwd");
    return 0;
}

__global__ void __launch_bounds__(ROW_THREADS) k_gate_pool_bwd(const cdc_pool_bwd_args a) {
    CDC_PRIO_MAIN();
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= a.B) return;
    const float* ex = a.experts + row * a.ld_exp;
    float* dex = a.d_experts + row * a.ld_dexp;
    // (1) gate gradients: dp_j = <d_out, expert_sel[j]> ; d_logit_j = p_j * (dp_j - sum_k p_k dp_k)
    for (int g = 0; g < a.n_gates; ++g) {
        const auto& G = a.gate[g];
        const float* dout = G.d_out + row * G.ld_dout;
        float dp[CDC_MAX_SEL];
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) dp[j] = 0.f;
        for (int h = lane; h < a.H; h += 64) {
            const float d = dout[h];
#pragma unroll
            for (int j = 0; j < CDC_MAX_SEL; ++j)
                if (j < G.n_sel) dp[j] += d * ex[(int64_t)G.sel[j] * a.H + h];
        }
        float dot = 0.f;
        float pj_lane = 0.f, dpj_lane = 0.f;
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) {
            if (j < G.n_sel) {
                dp[j] = wave_sum(dp[j]);
                const float pj = G.probs[row * G.n_sel + j];
                dot += pj * dp[j];
                if (j == lane) { pj_lane = pj; dpj_lane = dp[j]; }
            }
        }
        if (lane < G.n_sel) {
            const float dl = pj_lane * (dpj_lane - dot);
            G.d_logits[row * G.ld_dlogits + lane] = dl;
            if (G.d_logits_h) reinterpret_cast<__bf16*>(G.d_logits_h)[row * G.ld_dlogits_h + lane] = (__bf16)dl;
        }
    }
    // (2) expert gradients: d_expert_e = sum over (gate, j) with sel == e of p * d_out, then the expert's relu/dropout mask.
    // Accumulators live in LDS ([expert][lane] per wave) so that the data-dependent expert index costs one ds op, not a
    // register select chain; each gate's d_out element is read once.
    __shared__ float sacc[WAVES_PER_BLOCK][2 * CDC_MAX_SEL][64];
    float (*acc)[64] = sacc[threadIdx.x >> 6];
    for (int h = lane; h < a.H; h += 64) {
        for (int e = 0; e < a.n_expert; ++e) acc[e][lane] = 0.f;
        for (int g = 0; g < a.n_gates; ++g) {
            const auto& G = a.gate[g];
            const float d = G.d_out[row * G.ld_dout + h];
            for (int j = 0; j < G.n_sel; ++j) acc[G.sel[j]][lane] += G.probs[row * G.n_sel + j] * d;
        }
        for (int e = 0; e < a.n_expert; ++e) {
            float v = acc[e][lane];
            if (a.mask_relu) v = ex[(int64_t)e * a.H + h] > 0.f ? v * a.mask_scale : 0.f;
            float* dst = dex + (int64_t)e * a.H + h;
            v = a.accumulate ? *dst + v : v;
            *dst = v;
            if (a.d_experts_h) reinterpret_cast<__bf16*>(a.d_experts_h)[row * a.ld_dexp_h + (int64_t)e * a.H + h] = (__bf16)v;
        }
    }
}

// 16-byte-lane form of the backward (see k_gate_pool_fwd_v4): H/4 lanes per row, sub-wave xor reductions for the gate
// gradients, float4 expert-gradient accumulators in LDS ([expert][lane] per wave).
__global__ void __launch_bounds__(ROW_THREADS) k_gate_pool_bwd_v4(const cdc_pool_bwd_args a) {
    CDC_PRIO_MAIN();
    extern __shared__ __attribute__((aligned(16))) unsigned char pool_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = a.H / 4, l = lane % gl;
    const int64_t row = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wave) * (64 / gl) + lane / gl;
    const bool live = row < a.B;
    const int64_t r = live ? row : 0;                                     // dead lanes compute on row 0 and store nothing
    const pool_f4* ex = reinterpret_cast<const pool_f4*>(a.experts + r * a.ld_exp);
    pool_f4* acc = reinterpret_cast<pool_f4*>(pool_smem) + (int64_t)wave * a.n_expert * 64;
    for (int e = 0; e < a.n_expert; ++e) acc[e * 64 + lane] = pool_f4{0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < a.n_gates; ++g) {
        const auto& G = a.gate[g];
        const pool_f4 d = reinterpret_cast<const pool_f4*>(G.d_out + r * G.ld_dout)[l];
        // (1) dp_j = <d_out, expert_sel[j]> over the row; d_logit_j = p_j * (dp_j - sum_k p_k dp_k)
        float dp[CDC_MAX_SEL];
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j) {
            dp[j] = 0.f;
            if (j < G.n_sel) {
                const pool_f4 x = ex[(int64_t)G.sel[j] * gl + l];
                float part = d[0] * x[0] + d[1] * x[1] + d[2] * x[2] + d[3] * x[3];
                for (int o = gl >> 1; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                dp[j] = part;
            }
        }
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < CDC_MAX_SEL; ++j)
            if (j < G.n_sel) dot += G.probs[r * G.n_sel + j] * dp[j];
        if (live) {
            for (int j0 = l; j0 < G.n_sel; j0 += gl) {
                float dpj = 0.f;
#pragma unroll
                for (int j = 0; j < CDC_MAX_SEL; ++j) if (j == j0) dpj = dp[j];
                const float dl = G.probs[r * G.n_sel + j0] * (dpj - dot);
                G.d_logits[r * G.ld_dlogits + j0] = dl;
                if (G.d_logits_h) reinterpret_cast<__bf16*>(G.d_logits_h)[r * G.ld_dlogits_h + j0] = (__bf16)dl;
            }
        }
        // (2) every selected expert receives p_j * d_out
        for (int j = 0; j < G.n_sel; ++j) acc[G.sel[j] * 64 + lane] += G.probs[r * G.n_sel + j] * d;
    }
    if (!live) return;
    pool_f4* dex = reinterpret_cast<pool_f4*>(a.d_experts + r * a.ld_dexp);
    for (int e = 0; e < a.n_expert; ++e) {
        pool_f4 v = acc[e * 64 + lane];
        if (a.mask_relu) {
            const pool_f4 x = ex[(int64_t)e * gl + l];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = x[q] > 0.f ? v[q] * a.mask_scale : 0.f;
        }
        pool_f4* dst = dex + (int64_t)e * gl + l;
        if (a.accumulate) v = *dst + v;
        *dst = v;
        if (a.d_experts_h) pool_store_h4(a.d_experts_h, r * a.ld_dexp_h + ((int64_t)e * gl + l) * 4, v);
    }
}

extern "C" int cdc_gate_pool_bwd(const cdc_pool_bwd_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_gates > 0 && a->n_gates <= CDC_MAX_GATES && a->n_expert > 0 && a->n_expert <= 2 * CDC_MAX_SEL && a->H > 0 &&
                      a->B >= 0 && a->experts && a->d_experts, CDC_E_BADARG, "gate_pool_bwd: bad argument (n_expert <= 32)");
    for (int g = 0; g < a->n_gates; ++g)
        CDC_CHECK_ARG(a->gate[g].d_out && a->gate[g].probs && a->gate[g].d_logits && a->gate[g].n_sel > 0 &&
                          a->gate[g].n_sel <= CDC_MAX_SEL, CDC_E_BADARG, "gate_pool_bwd: gate %d malformed", g);
    if (a->B == 0) return 0;
    if (pool_vec_ok(a->H, a->experts, a->ld_exp) && (((uintptr_t)a->d_experts & 15) == 0) && a->ld_dexp % 4 == 0 && a->n_expert <= 16 &&
        (!a->d_experts_h || ((((uintptr_t)a->d_experts_h & 7) == 0) && a->ld_dexp_h % 4 == 0))) {
        bool ok = true;
        for (int g = 0; g < a->n_gates; ++g)
            ok = ok && (((uintptr_t)a->gate[g].d_out & 15) == 0) && (a->gate[g].ld_dout % 4 == 0);
        if (ok) {
            const int rows_per_block = WAVES_PER_BLOCK * (64 / (a->H / 4));
            const size_t lds = (size_t)WAVES_PER_BLOCK * a->n_expert * 64 * sizeof(pool_f4);
            hipLaunchKernelGGL(k_gate_pool_bwd_v4, dim3(cdc_ceil_div(a->B, rows_per_block)), dim3(ROW_THREADS), lds, (hipStream_t)stream, *a);
            CDC_LAUNCH_CHECK("gate_pool_bwd");
            return 0;
        }
    }
    hipLaunchKernelGGL(k_gate_pool_bwd, dim3(cdc_ceil_div(a->B, WAVES_PER_BLOCK)), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("gate_pool_bwd");
    return 0;
}

// =================================================================================================
// BatchNorm1d (+ReLU +dropout)  (model/layer.py:187,199-205 ; model/star.py:117-181)
// block = 64 columns x 64 rows: lane = column, wave w takes rows w, w+4, ...
// =================================================================================================
struct BnTile { int seg, c0, row_lo, M, chunk, col_base; };
// an operand stored as fp32 or (block-uniform choice) as bf16
__device__ __forceinline__ float bn_ld(const void* p, bool half, int64_t i) {
    return half ? (float)reinterpret_cast<const __bf16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}

template <typename Args>
__device__ __forceinline__ bool bn_locate(const Args& a, int n_chunks, BnTile& t) {
    // blockIdx.x -> (segment, 64-column tile, row chunk)
    int tile = blockIdx.x / n_chunks;
    t.chunk = blockIdx.x % n_chunks;
    int64_t col_base = 0;
    const int s = find_group<true>(a.n_seg, tile, [&](int l) { return (a.s[l].C + 63) / 64; }, [&](int l) { return (int64_t)a.s[l].C; },
                                   tile, &col_base);
    if (s < 0) return false;
    t.seg = s; t.c0 = tile * 64; t.col_base = (int)col_base;
    t.row_lo = 0; t.M = (int)a.M;
    if (a.row_offsets) { const int rg = a.s[s].row_group; t.row_lo = a.row_offsets[rg]; t.M = a.row_offsets[rg + 1] - t.row_lo; }
    return true;
}

// pass 1: per (row chunk, column) partial sum and sum of squares in double
__global__ void __launch_bounds__(ROW_THREADS) k_bn_stats(const cdc_bn_fwd_args a, int n_chunks, int total_c) {
    BnTile t;
    if (!bn_locate(a, n_chunks, t)) return;
    const cdc_bn_seg& S = a.s[t.seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = t.c0 + lane;
    double s1 = 0.0, s2 = 0.0;
    const int r_begin = t.chunk * CDC_BN_ROWS_PER_BLOCK;
    const int r_end = min(r_begin + CDC_BN_ROWS_PER_BLOCK, t.M);
    if (c < S.C) {
        constexpr int RPW = CDC_BN_ROWS_PER_BLOCK / WAVES_PER_BLOCK;      // a wave's rows of the chunk: all loads issued before the sums
        const bool xh = S.half & CDC_BN_X_BF16;
        float xv[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int r = r_begin + wave + k * WAVES_PER_BLOCK;
            xv[k] = r < r_end ? bn_ld(S.x, xh, (int64_t)(t.row_lo + r) * S.ldx + c) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {                                  // rows past r_end contribute exact zeros
            const double x = (double)xv[k];
            s1 += x; s2 += x * x;
        }
    }
    __shared__ double sh[2][WAVES_PER_BLOCK][64];
    sh[0][wave][lane] = s1; sh[1][wave][lane] = s2;
    __syncthreads();
    if (wave == 0 && c < S.C) {
        s1 = sh[0][0][lane] + sh[0][1][lane] + sh[0][2][lane] + sh[0][3][lane];
        s2 = sh[1][0][lane] + sh[1][1][lane] + sh[1][2][lane] + sh[1][3][lane];
        double* ws = a.workspace + ((int64_t)t.chunk * total_c + t.col_base + c) * 2;
        ws[0] = s1; ws[1] = s2;
    }
}

// Sum of a column's per-chunk partials (s1, s2): the block's four waves take chunks w, w+4, ... (loads of a wave in flight
// together) and the four wave sums are added in wave order — every block of a column forms the same bits.  ALL threads of the
// block must call this (barrier inside); lanes past the segment's columns read column 0 and ignore the result.
__device__ __forceinline__ void bn_sum_partials(const double* __restrict__ ws, int used, int total_c, int col, double& s1, double& s2) {
    __shared__ double bn_part[2][WAVES_PER_BLOCK][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a1 = 0.0, a2 = 0.0;
#pragma unroll 4
    for (int k = wave; k < used; k += WAVES_PER_BLOCK) {
        const double* p = ws + ((int64_t)k * total_c + col) * 2;
        a1 += p[0]; a2 += p[1];
    }
    bn_part[0][wave][lane] = a1; bn_part[1][wave][lane] = a2;
    __syncthreads();
    s1 = ((bn_part[0][0][lane] + bn_part[0][1][lane]) + bn_part[0][2][lane]) + bn_part[0][3][lane];
    s2 = ((bn_part[1][0][lane] + bn_part[1][1][lane]) + bn_part[1][2][lane]) + bn_part[1][3][lane];
}

// pass 2: finalise stats for the tile's columns (every block redundantly, fixed order), normalise its rows
__global__ void __launch_bounds__(ROW_THREADS) k_bn_apply(const cdc_bn_fwd_args a, int n_chunks, int total_c) {
    BnTile t;
    if (!bn_locate(a, n_chunks, t)) return;
    const cdc_bn_seg& S = a.s[t.seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = t.c0 + lane;
    // rows the statistics are over: this launch's rows, or (phase 2, data parallel) the all-reduced global count
    const bool global_stats = a.training && a.phase == 2 && a.exchange;
    const int Ms = global_stats ? (int)(a.exchange[2 * (int64_t)total_c + t.seg] + 0.5) : t.M;
    // the reference skips BN when the (group's) batch has one row (MLP / MDR_BatchNorm) or at most one row (DNN)
    const bool skip_norm = (Ms == 1) || (a.skip_le1 && Ms <= 1);
    float mean = 0.f, invstd = 1.f;
    double p1 = 0.0, p2 = 0.0;
    if (a.training && !skip_norm && !global_stats)                       // block-uniform
        bn_sum_partials(a.workspace, (t.M + CDC_BN_ROWS_PER_BLOCK - 1) / CDC_BN_ROWS_PER_BLOCK, total_c, t.col_base + (c < S.C ? c : 0), p1, p2);
    if (c < S.C && !skip_norm) {
        if (a.training) {
            double s1 = p1, s2 = p2;
            if (global_stats) {
                s1 = a.exchange[2 * (int64_t)(t.col_base + c)];
                s2 = a.exchange[2 * (int64_t)(t.col_base + c) + 1];
            }
            const double mu = Ms > 0 ? s1 / Ms : 0.0;
            double var = Ms > 0 ? s2 / Ms - mu * mu : 0.0;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            invstd = (float)(1.0 / sqrt(var + (double)a.eps));
            if (t.chunk == 0 && wave == 0) {
                if (S.save_mean) S.save_mean[c] = mean;
                if (S.save_invstd) S.save_invstd[c] = invstd;
                if (S.running_mean && Ms > 0) {
                    const double unbiased = Ms > 1 ? var * ((double)Ms / (double)(Ms - 1)) : var;
                    S.running_mean[c] = (1.f - a.momentum) * S.running_mean[c] + a.momentum * mean;
                    S.running_var[c] = (1.f - a.momentum) * S.running_var[c] + a.momentum * (float)unbiased;
                }
            }
        } else {
            mean = S.running_mean[c];
            invstd = 1.f / sqrtf(S.running_var[c] + a.eps);
        }
    }
    if (a.training && !skip_norm && t.chunk == 0 && t.c0 == 0 && threadIdx.x == 0 && S.num_batches_tracked) *S.num_batches_tracked += 1;
    if (c >= S.C) return;
    const float gam = (skip_norm || !S.gamma) ? 1.f : S.gamma[c];
    const float bet = (skip_norm || !S.beta) ? 0.f : S.beta[c];
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    uint64_t seed = a.seed;
    if (a.drop_p > 0.f && a.seed_offset_dev) seed += (uint64_t)(uint32_t)(*a.seed_offset_dev) * 0xD1342543DE82EF95ull;
    const int r_begin = t.chunk * CDC_BN_ROWS_PER_BLOCK;
    const int r_end = min(r_begin + CDC_BN_ROWS_PER_BLOCK, t.M);
    constexpr int RPW = CDC_BN_ROWS_PER_BLOCK / WAVES_PER_BLOCK;
    const bool xh = S.half & CDC_BN_X_BF16;
    float xv[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {                                      // the wave's 16 loads are in flight together
        const int r = r_begin + wave + k * WAVES_PER_BLOCK;
        xv[k] = r < r_end ? bn_ld(S.x, xh, (int64_t)(t.row_lo + r) * S.ldx + c) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = r_begin + wave + k * WAVES_PER_BLOCK;
        if (r >= r_end) break;
        const int64_t gr = t.row_lo + r;
        float v = xv[k];
        if (!skip_norm) v = (v - mean) * invstd * gam + bet;
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.drop_p > 0.f) {
            const uint64_t e = ((uint64_t)(t.seg + 64) << 56) ^ ((uint64_t)gr * (uint64_t)S.C + (uint64_t)c);
            v = cdc_uniform(seed, e) < a.drop_p ? 0.f : v * keep_scale;
        }
        if (S.y) S.y[gr * S.ldy + c] = v;
        if (S.yh) reinterpret_cast<__bf16*>(S.yh)[gr * S.ldyh + c] = (__bf16)v;
    }
}

// -------------------------------------------------------------------------------------------------
// BatchNorm with 16-byte lanes (all four kernels).  The kernels above give a lane ONE column, i.e. a wave instruction moves 256
// bytes (128 for a bf16 operand): they are bound by the number of memory instructions, not by bytes (dropping a whole 33.5 MB
// output stream did not move k_bn_apply).  Here a lane owns FOUR neighbouring columns of a row: LPR = 8..64 lanes per row
// (tile of 4*LPR columns), 64/LPR rows per wave instruction, the block still covers one 64-row chunk, so the partial-sum
// workspace layout ([chunk][column] pairs of doubles) is the same and the two families mix freely.  The rows' loads are issued
// BEFORE the block sums the chunk partials (which is a chain of L2 reads of its own), and the dropout decisions come from the
// 32-bit stream of csrc/common.h.  Requires C % 4 == 0 and 16-byte (bf16: 8-byte) aligned rows; anything else -> kernels above.
// -------------------------------------------------------------------------------------------------
struct bn_f4 { float v[4]; };
__device__ __forceinline__ bn_f4 bn_ld4(const void* p, bool half, int64_t i) {
    bn_f4 r;
    if (half) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(p) + i);
        r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xFFFF0000u);
        r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xFFFF0000u);
    } else {
        const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
        r.v[0] = f.x; r.v[1] = f.y; r.v[2] = f.z; r.v[3] = f.w;
    }
    return r;
}
__device__ __forceinline__ void bn_st4h(void* p, int64_t i, const float (&v)[4]) {
    union { __bf16 h[4]; uint2 u; } o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o.h[q] = (__bf16)v[q];
    *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(p) + i) = o.u;
}
template <int LS, typename Args>
__device__ __forceinline__ bool bn_locate_v(const Args& a, int n_chunks, BnTile& t) {
    constexpr int TW = 4 << LS;
    int tile = blockIdx.x / n_chunks;
    t.chunk = blockIdx.x % n_chunks;
    int64_t col_base = 0;
    const int s = find_group<true>(a.n_seg, tile, [&](int l) { return (a.s[l].C + TW - 1) / TW; }, [&](int l) { return (int64_t)a.s[l].C; },
                                   tile, &col_base);
    if (s < 0) return false;
    t.seg = s; t.c0 = tile * TW; t.col_base = (int)col_base;
    t.row_lo = 0; t.M = (int)a.M;
    if (a.row_offsets) { const int rg = a.s[s].row_group; t.row_lo = a.row_offsets[rg]; t.M = a.row_offsets[rg + 1] - t.row_lo; }
    return true;
}
// lane geometry of a block (LS = log2 lanes per row)
template <int LS> struct BnGeo {
    static constexpr int LPR = 1 << LS, TW = 4 * LPR, RW = 64 / LPR, ITER = CDC_BN_ROWS_PER_BLOCK / (WAVES_PER_BLOCK * RW);
};
// per-lane partial sums (4 columns) -> workspace [chunk][column] pairs: lanes of the same columns inside the wave by shuffles,
// then the four waves through LDS in wave order
template <int LS>
__device__ __forceinline__ void bn_store_partials(double (&s1)[4], double (&s2)[4], double* __restrict__ ws_chunk, int col0, int C) {
    using G = BnGeo<LS>;
    __shared__ double sh[2][WAVES_PER_BLOCK][G::TW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & (G::LPR - 1);
#pragma unroll
    for (int o = G::LPR; o < 64; o <<= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[q] += __shfl_xor(s1[q], o, 64); s2[q] += __shfl_xor(s2[q], o, 64); }
    }
    if (lane < G::LPR) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { sh[0][wave][cg * 4 + q] = s1[q]; sh[1][wave][cg * 4 + q] = s2[q]; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < G::TW; j += ROW_THREADS) {
        if (col0 + j >= C) continue;
        const double a1 = ((sh[0][0][j] + sh[0][1][j]) + sh[0][2][j]) + sh[0][3][j];
        const double a2 = ((sh[1][0][j] + sh[1][1][j]) + sh[1][2][j]) + sh[1][3][j];
        ws_chunk[(int64_t)j * 2] = a1; ws_chunk[(int64_t)j * 2 + 1] = a2;
    }
}
// the block's TW columns: sums of the per-chunk partials in a fixed order -> LDS (every block of a column forms the same bits)
template <int LS>
__device__ __forceinline__ void bn_sum_partials_v(const double* __restrict__ ws, int used, int total_c, int col_base, int c0, int C,
                                                  double (*out)[BnGeo<LS>::TW]) {
    using G = BnGeo<LS>;
    constexpr int NP = ROW_THREADS / G::TW > 0 ? ROW_THREADS / G::TW : 1;
    __shared__ double part[2][NP][G::TW];
    {
        const int j = threadIdx.x % G::TW, pt = threadIdx.x / G::TW;    // NP threads per column, each takes chunks pt, pt+NP, ...
        double a1 = 0.0, a2 = 0.0;
        if (pt < NP && c0 + j < C) {
#pragma unroll 4
            for (int k = pt; k < used; k += NP) {
                const double* p = ws + ((int64_t)k * total_c + col_base + c0 + j) * 2;
                a1 += p[0]; a2 += p[1];
            }
        }
        if (pt < NP) { part[0][pt][j] = a1; part[1][pt][j] = a2; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < G::TW; j += ROW_THREADS) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int pt = 0; pt < NP; ++pt) { a1 += part[0][pt][j]; a2 += part[1][pt][j]; }
        out[0][j] = a1; out[1][j] = a2;
    }
    __syncthreads();
}

template <int LS>
__global__ void __launch_bounds__(ROW_THREADS) k_bn_stats_v4(const cdc_bn_fwd_args a, int n_chunks, int total_c) {
    CDC_PRIO_MAIN();
    using G = BnGeo<LS>;
    BnTile t;
    if (!bn_locate_v<LS>(a, n_chunks, t)) return;
    const cdc_bn_seg& S = a.s[t.seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & (G::LPR - 1), rs = lane >> LS;
    const int c = t.c0 + cg * 4;
    const int r_begin = t.chunk * CDC_BN_ROWS_PER_BLOCK, r_end = min(r_begin + CDC_BN_ROWS_PER_BLOCK, t.M);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (c < S.C) {
        const bool xh = S.half & CDC_BN_X_BF16;
        bn_f4 xv[G::ITER];
#pragma unroll
        for (int k = 0; k < G::ITER; ++k) {
            const int r = r_begin + (k * WAVES_PER_BLOCK + wave) * G::RW + rs;
            if (r < r_end) xv[k] = bn_ld4(S.x, xh, (int64_t)(t.row_lo + r) * S.ldx + c);
            else xv[k] = bn_f4{{0.f, 0.f, 0.f, 0.f}};
        }
#pragma unroll
        for (int k = 0; k < G::ITER; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) { const double x = (double)xv[k].v[q]; s1[q] += x; s2[q] += x * x; }
    }
    bn_store_partials<LS>(s1, s2, a.workspace + ((int64_t)t.chunk * total_c + t.col_base + t.c0) * 2, t.c0, S.C);
}

template <int LS>
__global__ void __launch_bounds__(ROW_THREADS) k_bn_apply_v4(const cdc_bn_fwd_args a, int n_chunks, int total_c) {
    CDC_PRIO_MAIN();
    using G = BnGeo<LS>;
    BnTile t;
    if (!bn_locate_v<LS>(a, n_chunks, t)) return;
    const cdc_bn_seg& S = a.s[t.seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & (G::LPR - 1), rs = lane >> LS;
    const int c = t.c0 + cg * 4;
    const bool in_c = c < S.C;
    const int r_begin = t.chunk * CDC_BN_ROWS_PER_BLOCK, r_end = min(r_begin + CDC_BN_ROWS_PER_BLOCK, t.M);
    const bool xh = S.half & CDC_BN_X_BF16;
    bn_f4 xv[G::ITER];
#pragma unroll
    for (int k = 0; k < G::ITER; ++k) {                                  // in flight while the statistics are put together
        const int r = r_begin + (k * WAVES_PER_BLOCK + wave) * G::RW + rs;
        if (in_c && r < r_end) xv[k] = bn_ld4(S.x, xh, (int64_t)(t.row_lo + r) * S.ldx + c);
        else xv[k] = bn_f4{{0.f, 0.f, 0.f, 0.f}};
    }
    const bool global_stats = a.training && a.phase == 2 && a.exchange;
    const int Ms = global_stats ? (int)(a.exchange[2 * (int64_t)total_c + t.seg] + 0.5) : t.M;
    const bool skip_norm = (Ms == 1) || (a.skip_le1 && Ms <= 1);
    __shared__ double sums[2][G::TW];
    __shared__ float col_mean[G::TW], col_inv[G::TW];
    if (a.training && !skip_norm && !global_stats)
        bn_sum_partials_v<LS>(a.workspace, (t.M + CDC_BN_ROWS_PER_BLOCK - 1) / CDC_BN_ROWS_PER_BLOCK, total_c, t.col_base, t.c0, S.C, sums);
    for (int j = threadIdx.x; j < G::TW; j += ROW_THREADS) {             // one thread per column of the tile
        const int cj = t.c0 + j;
        float mean = 0.f, invstd = 1.f;
        if (cj < S.C && !skip_norm) {
            if (a.training) {
                double s1 = sums[0][j], s2 = sums[1][j];
                if (global_stats) {
                    s1 = a.exchange[2 * (int64_t)(t.col_base + cj)];
                    s2 = a.exchange[2 * (int64_t)(t.col_base + cj) + 1];
                }
                const double mu = Ms > 0 ? s1 / Ms : 0.0;
                double var = Ms > 0 ? s2 / Ms - mu * mu : 0.0;
                if (var < 0.0) var = 0.0;
                mean = (float)mu;
                invstd = (float)(1.0 / sqrt(var + (double)a.eps));
                if (t.chunk == 0) {
                    if (S.save_mean) S.save_mean[cj] = mean;
                    if (S.save_invstd) S.save_invstd[cj] = invstd;
                    if (S.running_mean && Ms > 0) {
                        const double unbiased = Ms > 1 ? var * ((double)Ms / (double)(Ms - 1)) : var;
                        S.running_mean[cj] = (1.f - a.momentum) * S.running_mean[cj] + a.momentum * mean;
                        S.running_var[cj] = (1.f - a.momentum) * S.running_var[cj] + a.momentum * (float)unbiased;
                    }
                }
            } else {
                mean = S.running_mean[cj];
                invstd = 1.f / sqrtf(S.running_var[cj] + a.eps);
            }
        }
        col_mean[j] = mean; col_inv[j] = invstd;
    }
    if (a.training && !skip_norm && t.chunk == 0 && t.c0 == 0 && threadIdx.x == 0 && S.num_batches_tracked) *S.num_batches_tracked += 1;
    __syncthreads();
    if (!in_c) return;
    float mean[4], scale[4], bet[4], gamv[4];                           // (v - mean) * invstd * gamma + beta, in the scalar kernel's order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        gamv[q] = (skip_norm || !S.gamma) ? 1.f : S.gamma[c + q];
        bet[q] = (skip_norm || !S.beta) ? 0.f : S.beta[c + q];
        mean[q] = col_mean[cg * 4 + q];
        scale[q] = col_inv[cg * 4 + q];
    }
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thr16 = (uint32_t)(a.drop_p * 65536.f + 0.5f);
    const uint32_t seed32 = a.drop_p > 0.f ? g2_seed32(a.seed, a.seed_offset_dev, 64 + t.seg) : 0u;
#pragma unroll
    for (int k = 0; k < G::ITER; ++k) {
        const int r = r_begin + (k * WAVES_PER_BLOCK + wave) * G::RW + rs;
        if (r >= r_end) continue;
        const int64_t gr = t.row_lo + r;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x = xv[k].v[q];
            if (!skip_norm) x = (x - mean[q]) * scale[q] * gamv[q] + bet[q];
            if (a.relu) x = fmaxf(x, 0.f);
            v[q] = x;
        }
        if (a.drop_p > 0.f) {
            const uint32_t h0 = g2_drop_bits(seed32, (int)gr, c >> 1), h1 = g2_drop_bits(seed32, (int)gr, (c >> 1) + 1);
            v[0] = (h0 & 0xFFFFu) < thr16 ? 0.f : v[0] * keep_scale;
            v[1] = (h0 >> 16) < thr16 ? 0.f : v[1] * keep_scale;
            v[2] = (h1 & 0xFFFFu) < thr16 ? 0.f : v[2] * keep_scale;
            v[3] = (h1 >> 16) < thr16 ? 0.f : v[3] * keep_scale;
        }
        if (S.y) *reinterpret_cast<float4*>(S.y + gr * S.ldy + c) = make_float4(v[0], v[1], v[2], v[3]);
        if (S.yh) bn_st4h(S.yh, gr * S.ldyh + c, v);
    }
}

template <int LS>
__global__ void __launch_bounds__(ROW_THREADS) k_bn_bwd_stats_v4(const cdc_bn_bwd_args a, int n_chunks, int total_c) {
    CDC_PRIO_MAIN();
    using G = BnGeo<LS>;
    BnTile t;
    if (!bn_locate_v<LS>(a, n_chunks, t)) return;
    const cdc_bn_bseg& S = a.s[t.seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & (G::LPR - 1), rs = lane >> LS;
    const int c = t.c0 + cg * 4;
    const int r_begin = t.chunk * CDC_BN_ROWS_PER_BLOCK, r_end = min(r_begin + CDC_BN_ROWS_PER_BLOCK, t.M);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (c < S.C && t.M != 1) {
        const bool masked = a.relu || a.mask_scale != 1.f;
        const bool xh = S.half & CDC_BN_X_BF16, yh = S.half & CDC_BN_Y_BF16, dyh = S.half & CDC_BN_DY_BF16;
        bn_f4 dv[G::ITER], yv[G::ITER], xv[G::ITER];
#pragma unroll
        for (int k = 0; k < G::ITER; ++k) {
            const int r = r_begin + (k * WAVES_PER_BLOCK + wave) * G::RW + rs;
            // rows past the group's extent are NOT read (a ragged group near the end of the batch: row_lo + r_begin lies beyond
            // the buffer for the chunks the group does not have — round 2 read them and zeroed the value afterwards)
            const bool ok = r < r_end;
            const int64_t gr = t.row_lo + r;
            dv[k] = yv[k] = xv[k] = bn_f4{{0.f, 0.f, 0.f, 0.f}};
            if (ok) {
                dv[k] = bn_ld4(S.dy, dyh, gr * S.lddy + c);
                if (masked) yv[k] = bn_ld4(S.y, yh, gr * S.ldy + c);
                xv[k] = bn_ld4(S.x, xh, gr * S.ldx + c);
            }
        }
        float mean[4], invstd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { mean[q] = S.save_mean[c + q]; invstd[q] = S.save_invstd[c + q]; }
#pragma unroll
        for (int k = 0; k < G::ITER; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dz = dv[k].v[q];
                if (masked) dz = yv[k].v[q] > 0.f ? dz * a.mask_scale : 0.f;
                const float xhat = (xv[k].v[q] - mean[q]) * invstd[q];
                s1[q] += (double)dz; s2[q] += (double)dz * (double)xhat;
            }
    }
    bn_store_partials<LS>(s1, s2, a.workspace + ((int64_t)t.chunk * total_c + t.col_base + t.c0) * 2, t.c0, S.C);
}

template <int LS>
__global__ void __launch_bounds__(ROW_THREADS) k_bn_bwd_apply_v4(const cdc_bn_bwd_args a, int n_chunks, int total_c) {
    CDC_PRIO_MAIN();
    using G = BnGeo<LS>;
    BnTile t;
    if (!bn_locate_v<LS>(a, n_chunks, t)) return;
    const cdc_bn_bseg& S = a.s[t.seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & (G::LPR - 1), rs = lane >> LS;
    const int c = t.c0 + cg * 4;
    const bool in_c = c < S.C;
    const bool global_stats = a.training && a.phase == 2 && a.exchange;
    const int Ms = global_stats ? (int)(a.exchange[2 * (int64_t)total_c + t.seg] + 0.5) : t.M;
    const bool skip_norm = (Ms == 1);
    const int r_begin = t.chunk * CDC_BN_ROWS_PER_BLOCK, r_end = min(r_begin + CDC_BN_ROWS_PER_BLOCK, t.M);
    const bool masked = a.relu || a.mask_scale != 1.f;
    const bool need_x = !skip_norm && a.training;
    const bool xh = S.half & CDC_BN_X_BF16, yh = S.half & CDC_BN_Y_BF16, dyh = S.half & CDC_BN_DY_BF16;
    bn_f4 dv[G::ITER], yv[G::ITER], xv[G::ITER];
#pragma unroll
    for (int k = 0; k < G::ITER; ++k) {                                  // in flight while the column sums are put together
        const int r = r_begin + (k * WAVES_PER_BLOCK + wave) * G::RW + rs;
        const bool ok = in_c && r < r_end;
        const int64_t gr = t.row_lo + r;
        if (ok) {
            dv[k] = bn_ld4(S.dy, dyh, gr * S.lddy + c);
            if (masked) yv[k] = bn_ld4(S.y, yh, gr * S.ldy + c);
            if (need_x) xv[k] = bn_ld4(S.x, xh, gr * S.ldx + c);
        }
    }
    __shared__ double sums[2][G::TW];
    if (!skip_norm)
        bn_sum_partials_v<LS>(a.workspace, (t.M + CDC_BN_ROWS_PER_BLOCK - 1) / CDC_BN_ROWS_PER_BLOCK, total_c, t.col_base, t.c0, S.C, sums);
    else {
        for (int j = threadIdx.x; j < G::TW; j += ROW_THREADS) { sums[0][j] = 0.0; sums[1][j] = 0.0; }
        __syncthreads();
    }
    if (t.chunk == 0) {
        // parameter gradients stay LOCAL sums: the data-parallel all-reduce of the gradient arena adds the ranks up
        for (int j = threadIdx.x; j < G::TW; j += ROW_THREADS) {
            if (t.c0 + j >= S.C) continue;
            if (S.dbeta) S.dbeta[t.c0 + j] = (float)sums[0][j];
            if (S.dgamma) S.dgamma[t.c0 + j] = (float)sums[1][j];
        }
    }
    if (!in_c) return;
    float gam[4], mean[4], invstd[4], db[4], dg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double s1 = sums[0][cg * 4 + q], s2 = sums[1][cg * 4 + q];
        if (global_stats && !skip_norm) {      // the input gradient needs the sums over the GLOBAL batch
            s1 = a.exchange[2 * (int64_t)(t.col_base + c + q)];
            s2 = a.exchange[2 * (int64_t)(t.col_base + c + q) + 1];
        }
        gam[q] = (skip_norm || !S.gamma) ? 1.f : S.gamma[c + q];
        mean[q] = skip_norm ? 0.f : S.save_mean[c + q];
        invstd[q] = skip_norm ? 1.f : S.save_invstd[c + q];
        db[q] = (float)s1; dg[q] = (float)s2;
    }
    const float invM = Ms > 0 ? 1.f / (float)Ms : 0.f;
#pragma unroll
    for (int k = 0; k < G::ITER; ++k) {
        const int r = r_begin + (k * WAVES_PER_BLOCK + wave) * G::RW + rs;
        if (r >= r_end) continue;
        const int64_t gr = t.row_lo + r;
        float dx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dz = dv[k].v[q];
            if (masked) dz = yv[k].v[q] > 0.f ? dz * a.mask_scale : 0.f;
            if (skip_norm) dx[q] = dz;
            else if (a.training) {
                const float xhat = (xv[k].v[q] - mean[q]) * invstd[q];
                dx[q] = gam[q] * invstd[q] * (dz - invM * (db[q] + xhat * dg[q]));
            } else dx[q] = gam[q] * invstd[q] * dz;
        }
        if (S.dx) {
            float4* dst = reinterpret_cast<float4*>(S.dx + gr * S.lddx + c);
            if (S.accumulate_dx) { const float4 o = *dst; dx[0] = o.x + dx[0]; dx[1] = o.y + dx[1]; dx[2] = o.z + dx[2]; dx[3] = o.w + dx[3]; }
            *dst = make_float4(dx[0], dx[1], dx[2], dx[3]);
        }
        if (S.dxh) bn_st4h(S.dxh, gr * S.lddxh + c, dx);
    }
}

// 16-byte lanes usable?  -> log2(lanes per row) for the launch (3..6), or -1
static inline bool bn_al(const void* p, int64_t ld, bool half) {
    return !p || ((((uintptr_t)p) & (half ? 7 : 15)) == 0 && ld % 4 == 0);
}
static int bn_fwd_ls(const cdc_bn_fwd_args* a) {
    int cmax = 0;
    for (int s = 0; s < a->n_seg; ++s) {
        const cdc_bn_seg& S = a->s[s];
        if (S.C % 4 || !bn_al(S.x, S.ldx, S.half & CDC_BN_X_BF16) || !bn_al(S.y, S.ldy, false) || !bn_al(S.yh, S.ldyh, true)) return -1;
        cmax = std::max(cmax, (int)S.C);
    }
    return cmax >= 256 ? 6 : cmax >= 128 ? 5 : cmax >= 64 ? 4 : 3;
}
static int bn_bwd_ls(const cdc_bn_bwd_args* a) {
    int cmax = 0;
    for (int s = 0; s < a->n_seg; ++s) {
        const cdc_bn_bseg& S = a->s[s];
        if (S.C % 4 || !bn_al(S.x, S.ldx, S.half & CDC_BN_X_BF16) || !bn_al(S.y, S.ldy, S.half & CDC_BN_Y_BF16) ||
            !bn_al(S.dy, S.lddy, S.half & CDC_BN_DY_BF16) || !bn_al(S.dx, S.lddx, false) || !bn_al(S.dxh, S.lddxh, true)) return -1;
        cmax = std::max(cmax, (int)S.C);
    }
    return cmax >= 256 ? 6 : cmax >= 128 ? 5 : cmax >= 64 ? 4 : 3;
}
#define BN_V4_LAUNCH(KERN, LS, GRID, ...)                                                                                   \
    do {                                                                                                                    \
        switch (LS) {                                                                                                       \
            case 6: hipLaunchKernelGGL((KERN<6>), dim3(GRID), dim3(ROW_THREADS), 0, (hipStream_t)stream, __VA_ARGS__); break; \
            case 5: hipLaunchKernelGGL((KERN<5>), dim3(GRID), dim3(ROW_THREADS), 0, (hipStream_t)stream, __VA_ARGS__); break; \
            case 4: hipLaunchKernelGGL((KERN<4>), dim3(GRID), dim3(ROW_THREADS), 0, (hipStream_t)stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL((KERN<3>), dim3(GRID), dim3(ROW_THREADS), 0, (hipStream_t)stream, __VA_ARGS__); break; \
        }                                                                                                                   \
    } while (0)

// data parallel, between the two phases: per-column sums over the row chunks (fixed order) + the segment's row count go to
// the exchange buffer [2*total_c | n_seg] that the caller all-reduces (SUM) across ranks
template <typename Args>
__global__ void __launch_bounds__(ROW_THREADS) k_bn_collect(const Args a, int n_chunks, int total_c) {
    const int gc = blockIdx.x * ROW_THREADS + threadIdx.x;
    if (gc < a.n_seg) {
        int M = (int)a.M;
        if (a.row_offsets) { const int rg = a.s[gc].row_group; M = a.row_offsets[rg + 1] - a.row_offsets[rg]; }
        a.exchange[2 * (int64_t)total_c + gc] = (double)M;
    }
    if (gc >= total_c) return;
    int s = 0, base = 0;
    for (; s < a.n_seg; ++s) { if (gc < base + a.s[s].C) break; base += a.s[s].C; }
    int M = (int)a.M;
    if (a.row_offsets) { const int rg = a.s[s].row_group; M = a.row_offsets[rg + 1] - a.row_offsets[rg]; }
    const int used = (M + CDC_BN_ROWS_PER_BLOCK - 1) / CDC_BN_ROWS_PER_BLOCK;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < used; ++k) {
        const double* ws = a.workspace + ((int64_t)k * total_c + gc) * 2;
        s1 += ws[0]; s2 += ws[1];
    }
    a.exchange[2 * (int64_t)gc] = s1;
    a.exchange[2 * (int64_t)gc + 1] = s2;
}

extern "C" int cdc_bn_fwd(const cdc_bn_fwd_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_seg > 0 && a->n_seg <= CDC_MAX_BN_SEGS && a->M >= 0, CDC_E_BADARG, "bn_fwd: bad argument");
    CDC_CHECK_ARG(!a->training || a->workspace, CDC_E_BADARG, "bn_fwd: training needs a workspace");
    int total_c = 0, col_tiles = 0;
    for (int s = 0; s < a->n_seg; ++s) {
        const cdc_bn_seg& S = a->s[s];
        CDC_CHECK_ARG(S.x && (S.y || S.yh) && S.C > 0 && S.ldx >= S.C && (!S.y || S.ldy >= S.C) && (!S.yh || S.ldyh >= S.C) &&
                          (S.half & ~CDC_BN_X_BF16) == 0, CDC_E_BADARG, "bn_fwd: segment %d malformed", s);
        CDC_CHECK_ARG(a->training || (S.running_mean && S.running_var), CDC_E_BADARG, "bn_fwd: eval needs running stats");
        total_c += S.C;
        col_tiles += (S.C + 63) / 64;
    }
    CDC_CHECK_ARG(a->phase >= 0 && a->phase <= 2 && (a->phase == 0 || (a->exchange && a->training)), CDC_E_BADARG,
                  "bn_fwd: phases 1/2 need training mode and an exchange buffer");
    if (a->M == 0) return 0;
    const int n_chunks = (int)cdc_ceil_div(a->M, CDC_BN_ROWS_PER_BLOCK);
    const int64_t grid = (int64_t)col_tiles * n_chunks;
    CDC_CHECK_ARG(grid < (1ll << 31), CDC_E_TOOBIG, "bn_fwd: grid too large");
    const int ls = bn_fwd_ls(a);                                         // 16-byte lanes (k_bn_*_v4) when every segment allows them
    int64_t grid_v = 0;
    if (ls >= 0)
        for (int s = 0; s < a->n_seg; ++s) grid_v += (int64_t)cdc_ceil_div(a->s[s].C, 4 << ls) * n_chunks;
    if (a->training && a->phase != 2 && !a->stats_ready) {
        if (ls >= 0) BN_V4_LAUNCH(k_bn_stats_v4, ls, grid_v, *a, n_chunks, total_c);
        else hipLaunchKernelGGL(k_bn_stats, dim3(grid), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a, n_chunks, total_c);
        CDC_LAUNCH_CHECK("bn_stats");
    }
    if (a->phase == 1) {
        hipLaunchKernelGGL(k_bn_collect<cdc_bn_fwd_args>, dim3(cdc_ceil_div(std::max(total_c, a->n_seg), ROW_THREADS)), dim3(ROW_THREADS), 0,
                           (hipStream_t)stream, *a, n_chunks, total_c);
        CDC_LAUNCH_CHECK("bn_collect");
        return 0;
    }
    if (ls >= 0) BN_V4_LAUNCH(k_bn_apply_v4, ls, grid_v, *a, n_chunks, total_c);
    else hipLaunchKernelGGL(k_bn_apply, dim3(grid), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a, n_chunks, total_c);
    CDC_LAUNCH_CHECK("bn_apply");
    return 0;
}

// backward pass 1: partial sums of dz (-> dbeta) and dz * xhat (-> dgamma), dz = activation-masked dy
__global__ void __launch_bounds__(ROW_THREADS) k_bn_bwd_stats(const cdc_bn_bwd_args a, int n_chunks, int total_c) {
    BnTile t;
    if (!bn_locate(a, n_chunks, t)) return;
    const cdc_bn_bseg& S = a.s[t.seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = t.c0 + lane;
    double s1 = 0.0, s2 = 0.0;
    const int r_begin = t.chunk * CDC_BN_ROWS_PER_BLOCK;
    const int r_end = min(r_begin + CDC_BN_ROWS_PER_BLOCK, t.M);
    if (c < S.C && t.M != 1) {
        const float mean = S.save_mean[c], invstd = S.save_invstd[c];
        constexpr int RPW = CDC_BN_ROWS_PER_BLOCK / WAVES_PER_BLOCK;
        const bool masked = a.relu || a.mask_scale != 1.f;
        const bool xh = S.half & CDC_BN_X_BF16, yh = S.half & CDC_BN_Y_BF16, dyh = S.half & CDC_BN_DY_BF16;
        float dv[RPW], yv[RPW], xv[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {                                  // 48 loads in flight per lane instead of 3
            const int r = r_begin + wave + k * WAVES_PER_BLOCK;
            const bool ok = r < r_end;
            const int64_t gr = t.row_lo + r;                             // (only dereferenced for rows of the group: see k_bn_bwd_stats_v4)
            dv[k] = ok ? bn_ld(S.dy, dyh, gr * S.lddy + c) : 0.f;
            yv[k] = (ok && masked) ? bn_ld(S.y, yh, gr * S.ldy + c) : 1.f;
            xv[k] = ok ? bn_ld(S.x, xh, gr * S.ldx + c) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {                                  // rows past r_end carry dz = 0: exact zeros in both sums
            float dz = dv[k];
            if (masked) dz = yv[k] > 0.f ? dz * a.mask_scale : 0.f;
            const float xhat = (xv[k] - mean) * invstd;
            s1 += (double)dz; s2 += (double)dz * (double)xhat;
        }
    }
    __shared__ double sh[2][WAVES_PER_BLOCK][64];
    sh[0][wave][lane] = s1; sh[1][wave][lane] = s2;
    __syncthreads();
    if (wave == 0 && c < S.C) {
        s1 = sh[0][0][lane] + sh[0][1][lane] + sh[0][2][lane] + sh[0][3][lane];
        s2 = sh[1][0][lane] + sh[1][1][lane] + sh[1][2][lane] + sh[1][3][lane];
        double* ws = a.workspace + ((int64_t)t.chunk * total_c + t.col_base + c) * 2;
        ws[0] = s1; ws[1] = s2;
    }
}

__global__ void __launch_bounds__(ROW_THREADS) k_bn_bwd_apply(const cdc_bn_bwd_args a, int n_chunks, int total_c) {
    BnTile t;
    if (!bn_locate(a, n_chunks, t)) return;
    const cdc_bn_bseg& S = a.s[t.seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = t.c0 + lane;
    const bool global_stats = a.training && a.phase == 2 && a.exchange;
    const int Ms = global_stats ? (int)(a.exchange[2 * (int64_t)total_c + t.seg] + 0.5) : t.M;
    const bool skip_norm = (Ms == 1);
    double s1 = 0.0, s2 = 0.0;
    if (!skip_norm)                                                      // block-uniform; every thread takes part (barrier inside)
        bn_sum_partials(a.workspace, (t.M + CDC_BN_ROWS_PER_BLOCK - 1) / CDC_BN_ROWS_PER_BLOCK, total_c, t.col_base + (c < S.C ? c : 0), s1, s2);
    if (c >= S.C) return;
    if (t.chunk == 0 && wave == 0) {
        // parameter gradients stay LOCAL sums: the data-parallel all-reduce of the gradient arena adds the ranks up
        if (S.dbeta) S.dbeta[c] = (float)s1;
        if (S.dgamma) S.dgamma[c] = (float)s2;
    }
    if (global_stats && !skip_norm) {          // the input gradient needs the sums over the GLOBAL batch
        s1 = a.exchange[2 * (int64_t)(t.col_base + c)];
        s2 = a.exchange[2 * (int64_t)(t.col_base + c) + 1];
    }
    const float gam = (skip_norm || !S.gamma) ? 1.f : S.gamma[c];
    const float mean = skip_norm ? 0.f : S.save_mean[c];
    const float invstd = skip_norm ? 1.f : S.save_invstd[c];
    const float invM = Ms > 0 ? 1.f / (float)Ms : 0.f;
    const float db = (float)s1, dg = (float)s2;
    const int r_begin = t.chunk * CDC_BN_ROWS_PER_BLOCK;
    const int r_end = min(r_begin + CDC_BN_ROWS_PER_BLOCK, t.M);
    constexpr int RPW = CDC_BN_ROWS_PER_BLOCK / WAVES_PER_BLOCK;
    const bool masked = a.relu || a.mask_scale != 1.f;
    const bool need_x = !skip_norm && a.training;
    const bool xh = S.half & CDC_BN_X_BF16, yh = S.half & CDC_BN_Y_BF16, dyh = S.half & CDC_BN_DY_BF16;
    float dv[RPW], yv[RPW], xv[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = r_begin + wave + k * WAVES_PER_BLOCK;
        const bool ok = r < r_end;
        const int64_t gr = t.row_lo + r;
        dv[k] = ok ? bn_ld(S.dy, dyh, gr * S.lddy + c) : 0.f;
        yv[k] = (ok && masked) ? bn_ld(S.y, yh, gr * S.ldy + c) : 1.f;
        xv[k] = (ok && need_x) ? bn_ld(S.x, xh, gr * S.ldx + c) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int r = r_begin + wave + k * WAVES_PER_BLOCK;
        if (r >= r_end) break;
        const int64_t gr = t.row_lo + r;
        float dz = dv[k];
        if (masked) dz = yv[k] > 0.f ? dz * a.mask_scale : 0.f;
        float dx;
        if (skip_norm) dx = dz;
        else if (a.training) {
            const float xhat = (xv[k] - mean) * invstd;
            dx = gam * invstd * (dz - invM * (db + xhat * dg));
        } else dx = gam * invstd * dz;
        if (S.dx) {
            float* dst = S.dx + gr * S.lddx + c;
            if (S.accumulate_dx) dx = *dst + dx;
            *dst = dx;
        }
        if (S.dxh) reinterpret_cast<__bf16*>(S.dxh)[gr * S.lddxh + c] = (__bf16)dx;
    }
}

extern "C" int cdc_bn_bwd(const cdc_bn_bwd_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_seg > 0 && a->n_seg <= CDC_MAX_BN_SEGS && a->M >= 0 && a->workspace, CDC_E_BADARG, "bn_bwd: bad argument");
    int total_c = 0, col_tiles = 0;
    for (int s = 0; s < a->n_seg; ++s) {
        const cdc_bn_bseg& S = a->s[s];
        CDC_CHECK_ARG(S.dy && S.y && S.x && (S.dx || (S.dxh && !S.accumulate_dx)) && S.save_mean && S.save_invstd && S.C > 0, CDC_E_BADARG,
                      "bn_bwd: segment %d malformed", s);
        total_c += S.C;
        col_tiles += (S.C + 63) / 64;
    }
    CDC_CHECK_ARG(a->phase >= 0 && a->phase <= 2 && (a->phase == 0 || (a->exchange && a->training)), CDC_E_BADARG,
                  "bn_bwd: phases 1/2 need training mode and an exchange buffer");
    if (a->M == 0) return 0;
    const int n_chunks = (int)cdc_ceil_div(a->M, CDC_BN_ROWS_PER_BLOCK);
    const int64_t grid = (int64_t)col_tiles * n_chunks;
    CDC_CHECK_ARG(grid < (1ll << 31), CDC_E_TOOBIG, "bn_bwd: grid too large");
    const int ls = bn_bwd_ls(a);
    int64_t grid_v = 0;
    if (ls >= 0)
        for (int s = 0; s < a->n_seg; ++s) grid_v += (int64_t)cdc_ceil_div(a->s[s].C, 4 << ls) * n_chunks;
    if (a->phase != 2) {
        if (ls >= 0) BN_V4_LAUNCH(k_bn_bwd_stats_v4, ls, grid_v, *a, n_chunks, total_c);
        else hipLaunchKernelGGL(k_bn_bwd_stats, dim3(grid), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a, n_chunks, total_c);
        CDC_LAUNCH_CHECK("bn_bwd_stats");
    }
    if (a->phase == 1) {
        hipLaunchKernelGGL(k_bn_collect<cdc_bn_bwd_args>, dim3(cdc_ceil_div(std::max(total_c, a->n_seg), ROW_THREADS)), dim3(ROW_THREADS), 0,
                           (hipStream_t)stream, *a, n_chunks, total_c);
        CDC_LAUNCH_CHECK("bn_bwd_collect");
        return 0;
    }
    if (ls >= 0) BN_V4_LAUNCH(k_bn_bwd_apply_v4, ls, grid_v, *a, n_chunks, total_c);
    else hipLaunchKernelGGL(k_bn_bwd_apply, dim3(grid), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a, n_chunks, total_c);
    CDC_LAUNCH_CHECK("bn_bwd_apply");
    return 0;
}

// =================================================================================================
// row dot products (+addends, +sigmoid)  (model/layer.py:122-126, :193 + :50-55, dcn.py:42)
// =================================================================================================
__global__ void __launch_bounds__(ROW_THREADS) k_rowdot_fwd(const cdc_rowdot_fwd_args a) {
    CDC_PRIO_MAIN();
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.y;
    const cdc_rowdot_group& G = a.g[g];
    int row_lo = 0, M = (int)a.M;
    if (a.row_offsets) { row_lo = a.row_offsets[g]; M = a.row_offsets[g + 1] - row_lo; }
    const int64_t r = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= M) return;
    const int64_t gr = row_lo + r;
    const float* x = G.x + gr * G.ldx;
    float acc = 0.f;
    for (int k = lane; k < G.K; k += 64) acc += x[k] * G.w[k];
    acc = wave_sum(acc);
    if (lane == 0) {
        if (G.bias) acc += G.bias[0];
        for (int i = 0; i < a.n_addend; ++i) acc += a.addend[i][gr * a.ld_addend[i]];
        if (G.logit) G.logit[gr * G.ld_logit] = acc;
        if (a.sigmoid) acc = 1.f / (1.f + expf(-acc));
        G.out[gr * G.ld_out] = acc;
    }
}

extern "C" int cdc_rowdot_fwd(const cdc_rowdot_fwd_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_groups > 0 && a->n_groups <= CDC_MAX_GROUPS && a->M >= 0 && a->n_addend >= 0 && a->n_addend <= 4,
                  CDC_E_BADARG, "rowdot_fwd: bad argument");
    for (int g = 0; g < a->n_groups; ++g)
        CDC_CHECK_ARG(a->g[g].x && a->g[g].w && a->g[g].out && a->g[g].K > 0, CDC_E_BADARG, "rowdot_fwd: group %d malformed", g);
    if (a->M == 0) return 0;
    hipLaunchKernelGGL(k_rowdot_fwd, dim3(cdc_ceil_div(a->M, WAVES_PER_BLOCK), a->n_groups), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("rowdot_fwd");
    return 0;
}

// backward: rows are split into CDC_ROWDOT_PARTS contiguous parts; each part's block writes its partial
// (dw[K], dbias) to the workspace; a second launch adds the parts in order.
__global__ void __launch_bounds__(ROW_THREADS) k_rowdot_bwd(const cdc_rowdot_bwd_args a, int kmax) {
    CDC_PRIO_MAIN();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.y, part = blockIdx.x;
    const cdc_rowdot_bgroup& G = a.g[g];
    int row_lo = 0, M = (int)a.M;
    if (a.row_offsets) { row_lo = a.row_offsets[g]; M = a.row_offsets[g + 1] - row_lo; }
    const int per = (M + CDC_ROWDOT_PARTS - 1) / CDC_ROWDOT_PARTS;
    const int r_begin = part * per, r_end = min(r_begin + per, M);
    extern __shared__ float sh[];      // [WAVES_PER_BLOCK][kmax + 1]
    float* mine = sh + wave * (kmax + 1);
    for (int k = lane; k <= kmax; k += 64) mine[k] = 0.f;
    float db = 0.f;
    double loss_part = 0.0;
    const bool bce = a.bce_y_i16 != nullptr || a.bce_y_f32 != nullptr;
    // four rows per round: their d (and sigmoid outputs) are fetched together, then their x rows — the launch is a few
    // dependent round trips long, so what counts is how many loads each of them carries
    constexpr int RB = 4;
    for (int r0 = r_begin + wave; r0 < r_end; r0 += RB * WAVES_PER_BLOCK) {
        float d[RB];
        int64_t gr[RB];
        if (bce) {
            // the loss and its gradient formed here instead of by a cdc_bce_fwd_bwd launch in between (same float operations)
            float o[RB], t[RB];
            int64_t col[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = r0 + q * WAVES_PER_BLOCK;
                gr[q] = row_lo + min(r, r_end - 1);
                o[q] = G.out[gr[q] * G.ld_out];
                t[q] = a.bce_y_i16 ? (float)a.bce_y_i16[gr[q]] : a.bce_y_f32[gr[q]];
                col[q] = a.bce_group ? a.bce_group[gr[q]] : 0;
            }
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                if (col[q] < 0 || col[q] >= a.n_groups) col[q] = 0;
                const bool own = col[q] == g && r0 + q * WAVES_PER_BLOCK < r_end;
                const float x = o[q];
                if (own) loss_part += (double)((t[q] - 1.f) * fmaxf(log1pf(-x), -100.f) - t[q] * fmaxf(logf(x), -100.f));
                const float dout = own ? a.bce_inv_count * (x - t[q]) / fmaxf((1.f - x) * x, 1e-12f) : 0.f;
                d[q] = dout * x * (1.f - x);
            }
        } else {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = r0 + q * WAVES_PER_BLOCK;
                gr[q] = row_lo + min(r, r_end - 1);
                d[q] = G.dout[gr[q] * G.ld_dout];
            }
            if (a.sigmoid) {
                float o[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) o[q] = G.out[gr[q] * G.ld_out];
#pragma unroll
                for (int q = 0; q < RB; ++q) d[q] = d[q] * o[q] * (1.f - o[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            if (r0 + q * WAVES_PER_BLOCK >= r_end) d[q] = 0.f;          // rows past the part: contribute nothing, store nothing
            else if (lane == 0 && G.dlogit) G.dlogit[gr[q] * G.ld_dlogit] = d[q];
            db += d[q];
        }
        for (int k = lane; k < G.K; k += 64) {
            float xv[RB], old[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) xv[q] = G.x[gr[q] * G.ldx + k];
            if (G.dx && G.accumulate_dx) {
#pragma unroll
                for (int q = 0; q < RB; ++q) old[q] = G.dx[gr[q] * G.lddx + k];
            }
            const float wk = G.dx ? G.w[k] : 0.f;
            float acc = mine[k];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const bool live = r0 + q * WAVES_PER_BLOCK < r_end;
                if (live) {
                    acc += d[q] * xv[q];                                // ascending row order, as before
                    if (G.dx) {
                        const float v = d[q] * wk;
                        G.dx[gr[q] * G.lddx + k] = G.accumulate_dx ? old[q] + v : v;
                    }
                }
            }
            mine[k] = acc;
        }
    }
    if (lane == 0) mine[kmax] = db;
    __shared__ double loss_w[WAVES_PER_BLOCK];
    if (bce && lane == 0) loss_w[wave] = loss_part;                      // every lane of a wave holds the same sum
    __syncthreads();
    if (bce && threadIdx.x == 0)
        a.bce_partial[(int64_t)g * CDC_ROWDOT_PARTS + part] = ((loss_w[0] + loss_w[1]) + loss_w[2]) + loss_w[3];
    float* ws = a.workspace + ((int64_t)g * CDC_ROWDOT_PARTS + part) * (kmax + 1);
    for (int k = threadIdx.x; k <= kmax; k += ROW_THREADS) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_PER_BLOCK; ++w) s += sh[w * (kmax + 1) + k];
        ws[k] = s;
    }
}
// one wave per output element k: lane l adds parts l, l+64, ... in ascending order, a butterfly adds the 64 lane sums
__global__ void __launch_bounds__(ROW_THREADS) k_rowdot_bwd_final(const cdc_rowdot_bwd_args a, int kmax) {
    const int g = blockIdx.y;
    const cdc_rowdot_bgroup& G = a.g[g];
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x == (kmax + WAVES_PER_BLOCK) / WAVES_PER_BLOCK) {
        // the extra block of a launch with the fused BCE: group 0's first wave adds the per-block loss partials in index order
        if (g != 0 || threadIdx.x >= 64 || !a.bce_loss) return;
        const int n = a.n_groups * CDC_ROWDOT_PARTS;
        double s = 0.0;
        for (int i = lane; i < n; i += 64) s += a.bce_partial[i];
        s = wave_sum_d(s);
        if (lane == 0) *a.bce_loss = (float)(s * (double)a.bce_inv_count);
        return;
    }
    const int k = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (k > kmax) return;
    const float* ws = a.workspace + (int64_t)g * CDC_ROWDOT_PARTS * (kmax + 1) + k;
    float v[CDC_ROWDOT_PARTS / 64];
#pragma unroll
    for (int i = 0; i < CDC_ROWDOT_PARTS / 64; ++i) v[i] = ws[(int64_t)(lane + 64 * i) * (kmax + 1)];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CDC_ROWDOT_PARTS / 64; ++i) s += v[i];
    s = wave_sum(s);
    if (lane != 0) return;
    if (k < G.K) { if (G.dw) G.dw[k] = s; }
    else if (k == kmax && G.dbias) G.dbias[0] = s;
}

extern "C" int cdc_rowdot_bwd(const cdc_rowdot_bwd_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_groups > 0 && a->n_groups <= CDC_MAX_GROUPS && a->M >= 0 && a->workspace, CDC_E_BADARG, "rowdot_bwd: bad argument");
    int kmax = 0;
    for (int g = 0; g < a->n_groups; ++g) {
        CDC_CHECK_ARG((a->g[g].dout || a->bce_y_i16 || a->bce_y_f32) && a->g[g].x && a->g[g].w && a->g[g].K > 0 &&
                          (!a->sigmoid || a->g[g].out), CDC_E_BADARG, "rowdot_bwd: group %d malformed", g);
        kmax = std::max(kmax, a->g[g].K);
    }
    CDC_CHECK_ARG((size_t)WAVES_PER_BLOCK * (kmax + 1) * 4 <= 64 * 1024, CDC_E_TOOBIG, "rowdot_bwd: K too large");
    if (a->bce_y_i16 || a->bce_y_f32)
        CDC_CHECK_ARG(a->sigmoid && !a->row_offsets && a->bce_loss && a->bce_partial && a->bce_inv_count > 0.f, CDC_E_BADARG,
                      "rowdot_bwd: the fused BCE needs sigmoid outputs, dense rows, a loss pointer and the partial-sum buffer");
    hipLaunchKernelGGL(k_rowdot_bwd, dim3(CDC_ROWDOT_PARTS, a->n_groups), dim3(ROW_THREADS), WAVES_PER_BLOCK * (kmax + 1) * sizeof(float),
                       (hipStream_t)stream, *a, kmax);
    CDC_LAUNCH_CHECK("rowdot_bwd");
    const bool bce = a->bce_y_i16 || a->bce_y_f32;
    hipLaunchKernelGGL(k_rowdot_bwd_final, dim3(cdc_ceil_div(kmax + 1, WAVES_PER_BLOCK) + (bce ? 1 : 0), a->n_groups), dim3(ROW_THREADS), 0,
                       (hipStream_t)stream, *a, kmax);
    CDC_LAUNCH_CHECK("rowdot_bwd_final");
    return 0;
}

// =================================================================================================
// BCE on probabilities, mean reduction, + its gradient  (run.py:484,723; aten binary_cross_entropy)
// =================================================================================================
__global__ void __launch_bounds__(1024) k_bce(const float* __restrict__ p, int64_t ldp, const int64_t* __restrict__ group,
                                              const int16_t* __restrict__ y_i16, const float* __restrict__ y_f32,
                                              float* __restrict__ loss, float* __restrict__ dp, int64_t lddp, int64_t B,
                                              int32_t n_col, float inv_count) {
    double acc = 0.0;
    // one workgroup (the loss is one ordered sum), so the launch is as long as its chain of dependent loads: four rows per
    // thread and round, their tower indices fetched together, then their probabilities and labels
    constexpr int RB = 4;
    for (int64_t b0 = threadIdx.x; b0 < B; b0 += (int64_t)RB * blockDim.x) {
        int64_t col[RB];
        float x[RB], t[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int64_t b = b0 + (int64_t)q * blockDim.x;
            col[q] = (b < B && group) ? group[b] : 0;
            if (col[q] < 0 || col[q] >= n_col) col[q] = 0;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int64_t b = b0 + (int64_t)q * blockDim.x;
            x[q] = b < B ? p[b * ldp + col[q]] : 0.5f;
            t[q] = b < B ? (y_i16 ? (float)y_i16[b] : y_f32[b]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int64_t b = b0 + (int64_t)q * blockDim.x;
            if (b >= B) continue;
            const float l = (t[q] - 1.f) * fmaxf(log1pf(-x[q]), -100.f) - t[q] * fmaxf(logf(x[q]), -100.f);
            acc += (double)l;
            if (dp) {
                for (int c = 0; c < n_col; ++c) dp[b * lddp + c] = 0.f;
                dp[b * lddp + col[q]] = inv_count * (x[q] - t[q]) / fmaxf((1.f - x[q]) * x[q], 1e-12f);
            }
        }
    }
    __shared__ double sh[16];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
        *loss = (float)(s * (double)inv_count);
    }
}

extern "C" int cdc_bce_fwd_bwd(const float* p, int64_t ldp, const int64_t* group, const int16_t* y_i16, const float* y_f32,
                               float* loss, float* dp, int64_t lddp, int64_t B, int32_t n_col, float inv_count, void* stream) {
    CDC_CHECK_ARG(p && loss && (y_i16 || y_f32) && B > 0 && n_col > 0 && ldp >= n_col && (!dp || lddp >= n_col), CDC_E_BADARG,
                  "bce_fwd_bwd: bad argument");
    hipLaunchKernelGGL(k_bce, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, ldp, group, y_i16, y_f32, loss, dp, lddp, B, n_col, inv_count);
    CDC_LAUNCH_CHECK("bce_fwd_bwd");
    return 0;
}

// BCE on the MEAN of the tower probabilities (CDC warm-up, cdc.py:100-102 `torch.mean(y_cat, dim=1)` + run.py:616):
// loss = mean_b bce(mean_c p[b,c], y[b]); d loss / d p[b,c] = bce'(mean) / n_col for every column.
__global__ void __launch_bounds__(1024) k_bce_mean(const float* __restrict__ p, int64_t ldp, const int16_t* __restrict__ y_i16,
                                                   const float* __restrict__ y_f32, float* __restrict__ loss, float* __restrict__ dp,
                                                   int64_t lddp, int64_t B, int32_t n_col, float inv_count) {
    double acc = 0.0;
    const float inv_c = 1.f / (float)n_col;
    for (int64_t b = threadIdx.x; b < B; b += blockDim.x) {
        float sum = 0.f;
        for (int c = 0; c < n_col; ++c) sum += p[b * ldp + c];
        const float x = sum / (float)n_col;
        const float t = y_i16 ? (float)y_i16[b] : y_f32[b];
        const float l = (t - 1.f) * fmaxf(log1pf(-x), -100.f) - t * fmaxf(logf(x), -100.f);
        acc += (double)l;
        if (dp) {
            const float g = inv_count * (x - t) / fmaxf((1.f - x) * x, 1e-12f) * inv_c;
            for (int c = 0; c < n_col; ++c) dp[b * lddp + c] = g;
        }
    }
    __shared__ double sh[16];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
        *loss = (float)(s * (double)inv_count);
    }
}
extern "C" int cdc_bce_mean_fwd_bwd(const float* p, int64_t ldp, const int16_t* y_i16, const float* y_f32, float* loss, float* dp,
                                    int64_t lddp, int64_t B, int32_t n_col, float inv_count, void* stream) {
    CDC_CHECK_ARG(p && loss && (y_i16 || y_f32) && B > 0 && n_col > 0 && ldp >= n_col && (!dp || lddp >= n_col), CDC_E_BADARG,
                  "bce_mean_fwd_bwd: bad argument");
    hipLaunchKernelGGL(k_bce_mean, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, ldp, y_i16, y_f32, loss, dp, lddp, B, n_col, inv_count);
    CDC_LAUNCH_CHECK("bce_mean_fwd_bwd");
    return 0;
}

// =================================================================================================
// second-order factorisation-machine term (model/layer.py:160-175, reduce_sum=True):
//   out[b] = 0.5 * sum_d ( (sum_f e[b,f,d])^2 - sum_f e[b,f,d]^2 )        d out / d e[b,f,d] = sum_f' e[b,f',d] - e[b,f,d]
// one wave per row; lane l owns dimension d = l (+64, ...); the field sums run in field order like torch.sum(dim=1)
// =================================================================================================
__global__ void __launch_bounds__(ROW_THREADS) k_fm_fwd(const float* __restrict__ e, int64_t lde, float* __restrict__ out, int64_t ldo,
                                                        int64_t B, int32_t F, int32_t D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* er = e + row * lde;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) {
        float s = 0.f, q = 0.f;
        for (int f = 0; f < F; ++f) {
            const float v = er[f * D + d];
            s += v;
            q += v * v;
        }
        acc += s * s - q;
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row * ldo] = 0.5f * acc;
}
__global__ void __launch_bounds__(ROW_THREADS) k_fm_bwd(const float* __restrict__ e, int64_t lde, const float* __restrict__ dout,
                                                        int64_t ldd, float* __restrict__ de, int64_t ldde, int64_t B, int32_t F,
                                                        int32_t D, int32_t accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* er = e + row * lde;
    float* dr = de + row * ldde;
    const float g = dout[row * ldd];
    for (int d = lane; d < D; d += 64) {
        float s = 0.f;
        for (int f = 0; f < F; ++f) s += er[f * D + d];
        for (int f = 0; f < F; ++f) {
            const float v = g * (s - er[f * D + d]);
            dr[f * D + d] = accumulate ? dr[f * D + d] + v : v;
        }
    }
}
extern "C" int cdc_fm_fwd(const float* e, int64_t lde, float* out, int64_t ldo, int64_t B, int32_t F, int32_t D, void* stream) {
    CDC_CHECK_ARG(e && out && B >= 0 && F > 0 && D > 0 && lde >= (int64_t)F * D && ldo >= 1, CDC_E_BADARG, "fm_fwd: bad argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_fm_fwd, dim3(cdc_ceil_div(B, WAVES_PER_BLOCK)), dim3(ROW_THREADS), 0, (hipStream_t)stream, e, lde, out, ldo, B, F, D);
    CDC_LAUNCH_CHECK("fm_fwd");
    return 0;
}
extern "C" int cdc_fm_bwd(const float* e, int64_t lde, const float* dout, int64_t ldd, float* de, int64_t ldde, int64_t B, int32_t F,
                          int32_t D, int32_t accumulate, void* stream) {
    CDC_CHECK_ARG(e && dout && de && B >= 0 && F > 0 && D > 0 && lde >= (int64_t)F * D && ldde >= (int64_t)F * D && ldd >= 1, CDC_E_BADARG,
                  "fm_bwd: bad argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_fm_bwd, dim3(cdc_ceil_div(B, WAVES_PER_BLOCK)), dim3(ROW_THREADS), 0, (hipStream_t)stream, e, lde, dout, ldd, de, ldde,
                       B, F, D, accumulate);
    CDC_LAUNCH_CHECK("fm_bwd");
    return 0;
}

// =================================================================================================
// per-row choice among n_group side-by-side feature blocks (model/hinet.py:71-74: con_feas[mask_g] = specific_feas[g][mask_g]):
//   out[b, :] = feas[b, group[b]*H : (group[b]+1)*H]   (zeros when group[b] is outside [0, n_group))
// backward: the chosen block receives d_out, the other blocks nothing (stored as zeros when this is the first writer)
// =================================================================================================
__global__ void __launch_bounds__(256) k_group_select_fwd(const float* __restrict__ feas, int64_t ldf, const int64_t* __restrict__ group,
                                                          float* __restrict__ out, int64_t ldo, int64_t B, int32_t n_group, int32_t H) {
    const int64_t total = B * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int h = (int)(i - b * H);
        const int64_t g = group[b];
        out[b * ldo + h] = (g >= 0 && g < n_group) ? feas[b * ldf + g * H + h] : 0.f;
    }
}
__global__ void __launch_bounds__(256) k_group_select_bwd(const float* __restrict__ dout, int64_t ldd, const int64_t* __restrict__ group,
                                                          float* __restrict__ dfeas, int64_t ldf, int64_t B, int32_t n_group, int32_t H,
                                                          int32_t accumulate) {
    const int64_t total = B * (int64_t)n_group * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / ((int64_t)n_group * H);
        const int c = (int)(i - b * (int64_t)n_group * H);
        const int g = c / H, h = c - g * H;
        const float d = group[b] == g ? dout[b * ldd + h] : 0.f;
        float* dst = dfeas + b * ldf + c;
        if (accumulate) { if (d != 0.f) *dst += d; }
        else *dst = d;
    }
}
extern "C" int cdc_group_select_fwd(const float* feas, int64_t ldf, const int64_t* group, float* out, int64_t ldo, int64_t B,
                                    int32_t n_group, int32_t H, void* stream) {
    CDC_CHECK_ARG(feas && group && out && B >= 0 && n_group > 0 && H > 0 && ldf >= (int64_t)n_group * H && ldo >= H, CDC_E_BADARG,
                  "group_select_fwd: bad argument");
    if (B == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(B * H, 256), 8192);
    hipLaunchKernelGGL(k_group_select_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, feas, ldf, group, out, ldo, B, n_group, H);
    CDC_LAUNCH_CHECK("group_select_fwd");
    return 0;
}
extern "C" int cdc_group_select_bwd(const float* dout, int64_t ldd, const int64_t* group, float* dfeas, int64_t ldf, int64_t B,
                                    int32_t n_group, int32_t H, int32_t accumulate, void* stream) {
    CDC_CHECK_ARG(dout && group && dfeas && B >= 0 && n_group > 0 && H > 0 && ldf >= (int64_t)n_group * H && ldd >= H, CDC_E_BADARG,
                  "group_select_bwd: bad argument");
    if (B == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(B * (int64_t)n_group * H, 256), 8192);
    hipLaunchKernelGGL(k_group_select_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, ldd, group, dfeas, ldf, B, n_group, H,
                       accumulate);
    CDC_LAUNCH_CHECK("group_select_bwd");
    return 0;
}

// =================================================================================================
// sigmoid gate:  pi = beta * sigmoid(alpha * p);  pi = 0 where |pi| <= eps;  out = a * pi
//   AdaSparse's pruner (model/adasparse.py:52-56: beta 2, alpha 1, eps 0.25) and PEPNet's GateNN output applied to its
//   input (model/pepnet.py:79-80,125: beta 2, alpha 1, no threshold: eps < 0).
// backward: d a = d out * pi;  d p = d out * a * beta*alpha*s*(1-s) where the gate is not pruned, 0 where it is.
// =================================================================================================
__global__ void __launch_bounds__(256) k_sigmoid_gate_fwd(const float* __restrict__ a, int64_t lda, const float* __restrict__ p, int64_t ldp,
                                                          float* __restrict__ out, int64_t ldo, int64_t rows, int32_t cols, float beta,
                                                          float alpha, float eps) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        float pi = beta / (1.f + expf(-alpha * p[r * ldp + c]));
        if (fabsf(pi) - eps <= 0.f) pi = 0.f;
        out[r * ldo + c] = a[r * lda + c] * pi;
    }
}
__global__ void __launch_bounds__(256) k_sigmoid_gate_bwd(const float* __restrict__ a, int64_t lda, const float* __restrict__ p, int64_t ldp,
                                                          const float* __restrict__ dout, int64_t lddo, float* __restrict__ da, int64_t ldda,
                                                          int32_t acc_a, float* __restrict__ dp, int64_t lddp, int32_t acc_p, int64_t rows,
                                                          int32_t cols, float beta, float alpha, float eps) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float s = 1.f / (1.f + expf(-alpha * p[r * ldp + c]));
        float pi = beta * s;
        const bool pruned = fabsf(pi) - eps <= 0.f;
        if (pruned) pi = 0.f;
        const float g = dout[r * lddo + c];
        if (da) {
            float* d = da + r * ldda + c;
            const float v = g * pi;
            *d = acc_a ? *d + v : v;
        }
        if (dp) {
            float* d = dp + r * lddp + c;
            const float v = pruned ? 0.f : g * a[r * lda + c] * beta * alpha * s * (1.f - s);
            *d = acc_p ? *d + v : v;
        }
    }
}
extern "C" int cdc_sigmoid_gate_fwd(const float* a, int64_t lda, const float* p, int64_t ldp, float* out, int64_t ldo, int64_t rows,
                                    int32_t cols, float beta, float alpha, float eps, void* stream) {
    CDC_CHECK_ARG(a && p && out && rows >= 0 && cols > 0 && lda >= cols && ldp >= cols && ldo >= cols, CDC_E_BADARG, "sigmoid_gate_fwd: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, 256), 8192);
    hipLaunchKernelGGL(k_sigmoid_gate_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, lda, p, ldp, out, ldo, rows, cols, beta, alpha, eps);
    CDC_LAUNCH_CHECK("sigmoid_gate_fwd");
    return 0;
}
extern "C" int cdc_sigmoid_gate_bwd(const float* a, int64_t lda, const float* p, int64_t ldp, const float* dout, int64_t lddo, float* da,
                                    int64_t ldda, int32_t acc_a, float* dp, int64_t lddp, int32_t acc_p, int64_t rows, int32_t cols,
                                    float beta, float alpha, float eps, void* stream) {
    CDC_CHECK_ARG(a && p && dout && (da || dp) && rows >= 0 && cols > 0 && lda >= cols && ldp >= cols && lddo >= cols &&
                      (!da || ldda >= cols) && (!dp || lddp >= cols), CDC_E_BADARG, "sigmoid_gate_bwd: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, 256), 8192);
    hipLaunchKernelGGL(k_sigmoid_gate_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, lda, p, ldp, dout, lddo, da, ldda, acc_a, dp,
                       lddp, acc_p, rows, cols, beta, alpha, eps);
    CDC_LAUNCH_CHECK("sigmoid_gate_bwd");
    return 0;
}

// =================================================================================================
// DCN-v1 cross layer  (model/layer.py:321-329):  out = x0 * (xl . w) + b + xl
// =================================================================================================
__global__ void __launch_bounds__(ROW_THREADS) k_cross_fwd(const float* __restrict__ x0, int64_t ld0, const float* __restrict__ xl,
                                                           int64_t ldl, const float* __restrict__ w, const float* __restrict__ b,
                                                           float* __restrict__ out, int64_t ldo, float* __restrict__ xw_save,
                                                           int64_t B, int32_t E) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* xr = xl + row * ldl;
    float s = 0.f;
    for (int k = lane; k < E; k += 64) s += xr[k] * w[k];
    s = wave_sum(s);
    if (lane == 0 && xw_save) xw_save[row] = s;
    const float* x0r = x0 + row * ld0;
    for (int k = lane; k < E; k += 64) out[row * ldo + k] = x0r[k] * s + b[k] + xr[k];
}

extern "C" int cdc_cross_fwd(const float* x0, int64_t ld0, const float* xl, int64_t ldl, const float* w, const float* b,
                             float* out, int64_t ldo, float* xw_save, int64_t B, int32_t E, void* stream) {
    CDC_CHECK_ARG(x0 && xl && w && b && out && B >= 0 && E > 0 && ld0 >= E && ldl >= E && ldo >= E, CDC_E_BADARG, "cross_fwd: bad argument");
    if (B == 0) return 0;
    hipLaunchKernelGGL(k_cross_fwd, dim3(cdc_ceil_div(B, WAVES_PER_BLOCK)), dim3(ROW_THREADS), 0, (hipStream_t)stream, x0, ld0, xl, ldl,
                       w, b, out, ldo, xw_save, B, E);
    CDC_LAUNCH_CHECK("cross_fwd");
    return 0;
}

// d_s = <d_out, x0>; d_x0 += d_out*s; d_xl = d_out + d_s*w; dw = sum_b d_s*xl; db = sum_b d_out
__global__ void __launch_bounds__(ROW_THREADS) k_cross_bwd(const float* __restrict__ d_out, int64_t ldo, const float* __restrict__ x0,
                                                           int64_t ld0, const float* __restrict__ xl, int64_t ldl,
                                                           const float* __restrict__ w, const float* __restrict__ xw_save,
                                                           float* __restrict__ d_x0_acc, int64_t ld_dx0, float* __restrict__ d_xl,
                                                           int64_t ld_dxl, float* __restrict__ workspace, int64_t B, int32_t E) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = blockIdx.x;
    const int64_t per = (B + CDC_ROWDOT_PARTS - 1) / CDC_ROWDOT_PARTS;
    const int64_t r_begin = part * per, r_end = min(r_begin + per, B);
    extern __shared__ float sh[];      // [WAVES][2E]
    float* mine = sh + wave * 2 * E;
    for (int k = lane; k < 2 * E; k += 64) mine[k] = 0.f;
    for (int64_t r = r_begin + wave; r < r_end; r += WAVES_PER_BLOCK) {
        const float* dor = d_out + r * ldo;
        const float* x0r = x0 + r * ld0;
        const float* xlr = xl + r * ldl;
        float ds = 0.f;
        for (int k = lane; k < E; k += 64) ds += dor[k] * x0r[k];
        ds = wave_sum(ds);
        const float s = xw_save[r];
        for (int k = lane; k < E; k += 64) {
            const float d = dor[k];
            d_x0_acc[r * ld_dx0 + k] += d * s;
            d_xl[r * ld_dxl + k] = d + ds * w[k];
            mine[k] += ds * xlr[k];
            mine[E + k] += d;
        }
    }
    __syncthreads();
    float* ws = workspace + (int64_t)part * 2 * E;
    for (int k = threadIdx.x; k < 2 * E; k += ROW_THREADS) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < WAVES_PER_BLOCK; ++wv) s += sh[wv * 2 * E + k];
        ws[k] = s;
    }
}
__global__ void __launch_bounds__(ROW_THREADS) k_cross_bwd_final(const float* __restrict__ workspace, float* __restrict__ dw,
                                                                 float* __restrict__ db, int32_t E) {
    const int k = blockIdx.x * ROW_THREADS + threadIdx.x;
    if (k >= 2 * E) return;
    float s = 0.f;
#pragma unroll 16
    for (int p = 0; p < CDC_ROWDOT_PARTS; ++p) s += workspace[(int64_t)p * 2 * E + k];
    if (k < E) dw[k] = s; else db[k - E] = s;
}

extern "C" int cdc_cross_bwd(const float* d_out, int64_t ldo, const float* x0, int64_t ld0, const float* xl, int64_t ldl,
                             const float* w, const float* xw_save, float* d_x0_acc, int64_t ld_dx0, float* d_xl, int64_t ld_dxl,
                             float* dw, float* db, float* workspace, int64_t B, int32_t E, void* stream) {
    CDC_CHECK_ARG(d_out && x0 && xl && w && xw_save && d_x0_acc && d_xl && dw && db && workspace && B > 0 && E > 0, CDC_E_BADARG,
                  "cross_bwd: bad argument");
    CDC_CHECK_ARG((size_t)WAVES_PER_BLOCK * 2 * E * 4 <= 64 * 1024, CDC_E_TOOBIG, "cross_bwd: E too large");
    hipLaunchKernelGGL(k_cross_bwd, dim3(CDC_ROWDOT_PARTS), dim3(ROW_THREADS), WAVES_PER_BLOCK * 2 * E * sizeof(float), (hipStream_t)stream,
                       d_out, ldo, x0, ld0, xl, ldl, w, xw_save, d_x0_acc, ld_dx0, d_xl, ld_dxl, workspace, B, E);
    CDC_LAUNCH_CHECK("cross_bwd");
    hipLaunchKernelGGL(k_cross_bwd_final, dim3(cdc_ceil_div(2 * E, ROW_THREADS)), dim3(ROW_THREADS), 0, (hipStream_t)stream, workspace, dw, db, E);
    CDC_LAUNCH_CHECK("cross_bwd_final");
    return 0;
}

// =================================================================================================
// dense-parameter Adam, multi-tensor (run.py:720-721 + the L2 term of model/layer.py:96-112)
// =================================================================================================
#include "adam_dense.h"

__global__ void __launch_bounds__(ROW_THREADS) k_adam_multi(const cdc_adam_args a) {
    // Which tensor does this workgroup's chunk belong to?  Walking the argument block one tensor at a time costs a scalar
    // cache miss per tensor (12 us for the last of 48); instead lane i reads tensor i's size, a wave scan turns the chunk
    // counts into offsets and a ballot names the tensor — one vector load for the whole search, the same in every wave.
    const int lane = threadIdx.x & 63;
    const int64_t n_l = lane < a.n_tensors ? a.t[lane].n : 0;
    const int nc = (int)((n_l + ADAM_CHUNK - 1) / ADAM_CHUNK);
    int inc = nc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    const int exc = inc - nc;
    const unsigned long long owner = __ballot((int)blockIdx.x >= exc && (int)blockIdx.x < inc);
    if (owner == 0ull) return;
    const int ti = __builtin_amdgcn_readfirstlane(__ffsll((long long)owner) - 1);
    const int chunk = (int)blockIdx.x - __builtin_amdgcn_readlane(exc, ti);
    const cdc_adam_tensor& T = a.t[ti];
    const AdamHdr h = {a.lerp_w, a.beta2, a.one_minus_beta2, a.eps, a.weight_decay, a.grad_scale, a.step_scalars, a.n_scalars, a.step_dev,
                       a.reg_sum, a.reg_seed};
    adam_chunk<ADAM_CHUNK>(h, T, chunk, blockIdx.x == 0);
}

// any number of tensors in ONE launch: the tensor descriptors and the workgroup -> (tensor, chunk) map live in device memory (built
// once per parameter set by the caller); the 48-tensor limit above is the 4 KB kernel-argument block, and two launches cost the
// C2 step 8 us more than one
__global__ void __launch_bounds__(ROW_THREADS) k_adam_multi_tab(const AdamHdr h, const cdc_adam_tensor* __restrict__ tab,
                                                                const int32_t* __restrict__ wg_tensor, const int32_t* __restrict__ wg_chunk) {
    const int ti = __builtin_amdgcn_readfirstlane(wg_tensor[blockIdx.x]);
    const int chunk = __builtin_amdgcn_readfirstlane(wg_chunk[blockIdx.x]);
    const cdc_adam_tensor T = tab[ti];
    adam_chunk<ADAM_CHUNK>(h, T, chunk, blockIdx.x == 0);
}

extern "C" int cdc_adam_multi(const cdc_adam_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_tensors > 0 && a->n_tensors <= CDC_MAX_TENSORS && a->step_dev && a->step_scalars && a->n_scalars > 0,
                  CDC_E_BADARG, "adam_multi: bad argument");
    int64_t chunks = 0;
    for (int i = 0; i < a->n_tensors; ++i) {
        CDC_CHECK_ARG(a->t[i].w && a->t[i].m && a->t[i].v && a->t[i].n > 0, CDC_E_BADARG, "adam_multi: tensor %d malformed", i);
        CDC_CHECK_ARG(a->t[i].n_slabs <= 0 || (a->t[i].slabs && a->t[i].n_slabs <= 256 && a->t[i].slab_stride >= a->t[i].n), CDC_E_BADARG,
                      "adam_multi: tensor %d: gradient slabs malformed", i);
        chunks += cdc_ceil_div(a->t[i].n, ADAM_CHUNK);
    }
    CDC_CHECK_ARG(chunks < (1ll << 31), CDC_E_TOOBIG, "adam_multi: too many chunks");
    hipLaunchKernelGGL(k_adam_multi, dim3(chunks), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("adam_multi");
    return 0;
}

extern "C" int cdc_adam_multi_table(const cdc_adam_args* a, const cdc_adam_tensor* tensors_dev, const int32_t* wg_tensor_dev,
                                    const int32_t* wg_chunk_dev, int32_t n_workgroups, void* stream) {
    CDC_CHECK_ARG(a && tensors_dev && wg_tensor_dev && wg_chunk_dev && n_workgroups > 0 && a->step_dev && a->step_scalars && a->n_scalars > 0,
                  CDC_E_BADARG, "adam_multi_table: bad argument");
    const AdamHdr h = {a->lerp_w, a->beta2, a->one_minus_beta2, a->eps, a->weight_decay, a->grad_scale, a->step_scalars, a->n_scalars,
                       a->step_dev, a->reg_sum, a->reg_seed};
    hipLaunchKernelGGL(k_adam_multi_tab, dim3(n_workgroups), dim3(ROW_THREADS), 0, (hipStream_t)stream, h, tensors_dev, wg_tensor_dev, wg_chunk_dev);
    CDC_LAUNCH_CHECK("adam_multi_table");
    return 0;
}

// =================================================================================================
// DCN-v2 element-wise pieces (model/layer.py:339-343 CrossNetV2, :372-407 CrossNetMix)
// =================================================================================================
__global__ void __launch_bounds__(ROW_THREADS) k_tanh_fwd(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                          int64_t rows, int32_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        y[r * ldy + c] = tanhf(x[r * ldx + c]);
    }
}
extern "C" int cdc_tanh_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int32_t cols, void* stream) {
    CDC_CHECK_ARG(x && y && rows >= 0 && cols > 0, CDC_E_BADARG, "tanh_fwd: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, ROW_THREADS), 4096);
    hipLaunchKernelGGL(k_tanh_fwd, dim3(blocks), dim3(ROW_THREADS), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols);
    CDC_LAUNCH_CHECK("tanh_fwd");
    return 0;
}
// dx (=|+=) dy * (1 - y^2)
__global__ void __launch_bounds__(ROW_THREADS) k_tanh_bwd(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y,
                                                          int64_t ldy, float* __restrict__ dx, int64_t lddx, int64_t rows,
                                                          int32_t cols, int32_t accumulate) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float yv = y[r * ldy + c];
        const float v = dy[r * lddy + c] * (1.f - yv * yv);
        float* dst = dx + r * lddx + c;
        *dst = accumulate ? *dst + v : v;
    }
}
extern "C" int cdc_tanh_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dx, int64_t lddx, int64_t rows,
                            int32_t cols, int32_t accumulate, void* stream) {
    CDC_CHECK_ARG(dy && y && dx && rows >= 0 && cols > 0, CDC_E_BADARG, "tanh_bwd: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, ROW_THREADS), 4096);
    hipLaunchKernelGGL(k_tanh_bwd, dim3(blocks), dim3(ROW_THREADS), 0, (hipStream_t)stream, dy, lddy, y, ldy, dx, lddx, rows, cols, accumulate);
    CDC_LAUNCH_CHECK("tanh_bwd");
    return 0;
}

// out[:, k*P + e] = x0[:, e] * (u[:, k*P + e] + b1[e]) + b2[e] + r[:, k*P + e]     (b1, b2, r optional; P = period,
// n_rep blocks of P columns share x0 / b1 / b2: the n experts of a CrossNetMix layer in one launch)
__global__ void __launch_bounds__(ROW_THREADS) k_cross_combine_fwd(const float* __restrict__ x0, int64_t ld0, const float* __restrict__ u,
                                                                   int64_t ldu, const float* __restrict__ b1,
                                                                   const float* __restrict__ b2, const float* __restrict__ r,
                                                                   int64_t ldr, float* __restrict__ out, int64_t ldo, int64_t rows,
                                                                   int32_t period, int32_t n_rep) {
    const int cols = period * n_rep;
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cols;
        const int c = (int)(i - row * cols);
        const int e = c % period;
        float uv = u[row * ldu + c];
        if (b1) uv += b1[e];
        float v = x0[row * ld0 + e] * uv;
        if (b2) v += b2[e];
        if (r) v += r[row * ldr + c];
        out[row * ldo + c] = v;
    }
}
extern "C" int cdc_cross_combine_fwd(const float* x0, int64_t ld0, const float* u, int64_t ldu, const float* b1, const float* b2,
                                     const float* r, int64_t ldr, float* out, int64_t ldo, int64_t rows, int32_t period,
                                     int32_t n_rep, void* stream) {
    CDC_CHECK_ARG(x0 && u && out && rows >= 0 && period > 0 && n_rep > 0, CDC_E_BADARG, "cross_combine_fwd: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * period * n_rep, ROW_THREADS), 4096);
    hipLaunchKernelGGL(k_cross_combine_fwd, dim3(blocks), dim3(ROW_THREADS), 0, (hipStream_t)stream, x0, ld0, u, ldu, b1, b2, r, ldr, out,
                       ldo, rows, period, n_rep);
    CDC_LAUNCH_CHECK("cross_combine_fwd");
    return 0;
}
// d_u = d_out*x0 ; d_x0 += sum_k d_out*(u+b1) ; d_r (=|+=) d_out ; db1 = colsum(d_out*x0) ; db2 = colsum(d_out)
// thread = base column e (owns every replica k of it: no race on d_x0), CDC_ROWDOT_PARTS row parts ->
// workspace partials [parts][2][period] -> ordered final sum
__global__ void __launch_bounds__(ROW_THREADS) k_cross_combine_bwd(const float* __restrict__ d_out, int64_t lddo, const float* __restrict__ x0,
                                                                   int64_t ld0, const float* __restrict__ u, int64_t ldu,
                                                                   const float* __restrict__ b1, float* __restrict__ d_u, int64_t lddu,
                                                                   float* __restrict__ d_x0_acc, int64_t lddx0, float* __restrict__ d_r,
                                                                   int64_t lddr, int32_t accumulate_r, float* __restrict__ workspace,
                                                                   int64_t rows, int32_t period, int32_t n_rep) {
    const int part = blockIdx.y;
    const int64_t per = (rows + CDC_ROWDOT_PARTS - 1) / CDC_ROWDOT_PARTS;
    const int64_t r_begin = part * per, r_end = min(r_begin + per, rows);
    for (int e = blockIdx.x * ROW_THREADS + threadIdx.x; e < period; e += gridDim.x * ROW_THREADS) {
        float s1 = 0.f, s2 = 0.f;
        const float bv = b1 ? b1[e] : 0.f;
        for (int64_t row = r_begin; row < r_end; ++row) {
            const float xv = x0[row * ld0 + e];
            float dx0 = 0.f;
            for (int k = 0; k < n_rep; ++k) {
                const int c = k * period + e;
                const float d = d_out[row * lddo + c];
                const float g = d * xv;
                d_u[row * lddu + c] = g;
                dx0 += d * (u[row * ldu + c] + bv);
                if (d_r) { float* dst = d_r + row * lddr + c; *dst = accumulate_r ? *dst + d : d; }
                s1 += g; s2 += d;
            }
            d_x0_acc[row * lddx0 + e] += dx0;
        }
        workspace[((int64_t)part * 2 + 0) * period + e] = s1;
        workspace[((int64_t)part * 2 + 1) * period + e] = s2;
    }
}
__global__ void __launch_bounds__(ROW_THREADS) k_cross_combine_bwd_final(const float* __restrict__ workspace, float* __restrict__ db1,
                                                                         float* __restrict__ db2, int32_t period, int32_t acc1,
                                                                         int32_t acc2) {
    const int c = blockIdx.x * ROW_THREADS + threadIdx.x;
    if (c >= period) return;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 16
    for (int p = 0; p < CDC_ROWDOT_PARTS; ++p) {
        s1 += workspace[((int64_t)p * 2 + 0) * period + c];
        s2 += workspace[((int64_t)p * 2 + 1) * period + c];
    }
    if (db1) db1[c] = acc1 ? db1[c] + s1 : s1;
    if (db2) db2[c] = acc2 ? db2[c] + s2 : s2;
}
extern "C" int cdc_cross_combine_bwd(const float* d_out, int64_t lddo, const float* x0, int64_t ld0, const float* u, int64_t ldu,
                                     const float* b1, float* d_u, int64_t lddu, float* d_x0_acc, int64_t lddx0, float* d_r,
                                     int64_t lddr, int32_t accumulate_r, float* db1, int32_t accumulate_b1, float* db2,
                                     int32_t accumulate_b2, float* workspace, int64_t rows, int32_t period, int32_t n_rep,
                                     void* stream) {
    CDC_CHECK_ARG(d_out && x0 && u && d_u && d_x0_acc && workspace && rows > 0 && period > 0 && n_rep > 0, CDC_E_BADARG,
                  "cross_combine_bwd: bad argument");
    dim3 grid((unsigned)cdc_ceil_div(period, ROW_THREADS), CDC_ROWDOT_PARTS);
    hipLaunchKernelGGL(k_cross_combine_bwd, grid, dim3(ROW_THREADS), 0, (hipStream_t)stream, d_out, lddo, x0, ld0, u, ldu, b1, d_u, lddu,
                       d_x0_acc, lddx0, d_r, lddr, accumulate_r, workspace, rows, period, n_rep);
    CDC_LAUNCH_CHECK("cross_combine_bwd");
    if (db1 || db2) {
        hipLaunchKernelGGL(k_cross_combine_bwd_final, dim3(cdc_ceil_div(period, ROW_THREADS)), dim3(ROW_THREADS), 0, (hipStream_t)stream,
                           workspace, db1, db2, period, accumulate_b1, accumulate_b2);
        CDC_LAUNCH_CHECK("cross_combine_bwd_final");
    }
    return 0;
}

// out = a + b
__global__ void __launch_bounds__(ROW_THREADS) k_add_out(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                         float* __restrict__ out, int64_t ldo, int64_t rows, int32_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        out[r * ldo + c] = a[r * lda + c] + b[r * ldb + c];
    }
}
extern "C" int cdc_add_out(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int64_t rows,
                           int32_t cols, void* stream) {
    CDC_CHECK_ARG(a && b && out && rows >= 0 && cols > 0, CDC_E_BADARG, "add_out: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, ROW_THREADS), 4096);
    hipLaunchKernelGGL(k_add_out, dim3(blocks), dim3(ROW_THREADS), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo, rows, cols);
    CDC_LAUNCH_CHECK("add_out");
    return 0;
}
// dst (=|+=) src   (gradient routing of identity / residual edges)
__global__ void __launch_bounds__(ROW_THREADS) k_copy_or_add(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src,
                                                             int64_t lds, int64_t rows, int32_t cols, int32_t accumulate) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float v = src[r * lds + c];
        float* p = dst + r * ldd + c;
        *p = accumulate ? *p + v : v;
    }
}
extern "C" int cdc_copy_or_add(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t rows, int32_t cols,
                               int32_t accumulate, void* stream) {
    CDC_CHECK_ARG(dst && src && rows >= 0 && cols > 0, CDC_E_BADARG, "copy_or_add: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, ROW_THREADS), 4096);
    hipLaunchKernelGGL(k_copy_or_add, dim3(blocks), dim3(ROW_THREADS), 0, (hipStream_t)stream, dst, ldd, src, lds, rows, cols, accumulate);
    CDC_LAUNCH_CHECK("copy_or_add");
    return 0;
}


// =================================================================================================
// STAR parameter fusion (model/star.py:90-93,100-102,169-176)
// =================================================================================================
__global__ void __launch_bounds__(ROW_THREADS) k_star_fuse_fwd(const cdc_star_fuse_args a) {
    const int g = blockIdx.y;
    const float* ag = a.a[g];
    float* og = a.out[g];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.size; i += (int64_t)gridDim.x * blockDim.x)
        og[i] = a.op == 0 ? ag[i] * a.s[i] : ag[i] + a.s[i];
}
extern "C" int cdc_star_fuse_fwd(const cdc_star_fuse_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n > 0 && a->n <= CDC_MAX_GROUPS && a->size > 0 && a->s && (a->op == 0 || a->op == 1), CDC_E_BADARG,
                  "star_fuse_fwd: bad argument");
    for (int g = 0; g < a->n; ++g) CDC_CHECK_ARG(a->a[g] && a->out[g], CDC_E_BADARG, "star_fuse_fwd: domain %d malformed", g);
    dim3 grid((unsigned)std::min<int64_t>(cdc_ceil_div(a->size, ROW_THREADS), 1024), a->n);
    hipLaunchKernelGGL(k_star_fuse_fwd, grid, dim3(ROW_THREADS), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("star_fuse_fwd");
    return 0;
}
__global__ void __launch_bounds__(ROW_THREADS) k_star_fuse_bwd(const cdc_star_fuse_args a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.size; i += (int64_t)gridDim.x * blockDim.x) {
        const float sv = a.s[i];
        float acc = 0.f;
        for (int g = 0; g < a.n; ++g) {
            const float d = a.out[g][i];
            if (a.da[g]) a.da[g][i] = a.op == 0 ? d * sv : d;
            acc += a.op == 0 ? d * a.a[g][i] : d;
        }
        if (a.ds) a.ds[i] = a.accumulate_ds ? a.ds[i] + acc : acc;
    }
}
extern "C" int cdc_star_fuse_bwd(const cdc_star_fuse_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n > 0 && a->n <= CDC_MAX_GROUPS && a->size > 0 && a->s && (a->op == 0 || a->op == 1), CDC_E_BADARG,
                  "star_fuse_bwd: bad argument");
    for (int g = 0; g < a->n; ++g) CDC_CHECK_ARG(a->a[g] && a->out[g], CDC_E_BADARG, "star_fuse_bwd: domain %d malformed", g);
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(a->size, ROW_THREADS), 2048);
    hipLaunchKernelGGL(k_star_fuse_bwd, dim3(blocks), dim3(ROW_THREADS), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("star_fuse_bwd");
    return 0;
}

// out[r, c] (=|+=) sum_g in[r, g*cols + c]   (fixed order over g)
__global__ void __launch_bounds__(ROW_THREADS) k_sum_slices(const float* __restrict__ in, int64_t ld_in, float* __restrict__ out,
                                                            int64_t ld_out, int64_t rows, int32_t cols, int32_t n_slices,
                                                            int32_t accumulate) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        float acc = 0.f;
        for (int g = 0; g < n_slices; ++g) acc += in[r * ld_in + (int64_t)g * cols + c];
        float* dst = out + r * ld_out + c;
        *dst = accumulate ? *dst + acc : acc;
    }
}
extern "C" int cdc_sum_slices(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows, int32_t cols,
                              int32_t n_slices, int32_t accumulate, void* stream) {
    CDC_CHECK_ARG(in && out && rows >= 0 && cols > 0 && n_slices > 0, CDC_E_BADARG, "sum_slices: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, ROW_THREADS), 4096);
    hipLaunchKernelGGL(k_sum_slices, dim3(blocks), dim3(ROW_THREADS), 0, (hipStream_t)stream, in, ld_in, out, ld_out, rows, cols, n_slices,
                       accumulate);
    CDC_LAUNCH_CHECK("sum_slices");
    return 0;
}
