// tower.hip — the towers' hidden layers + output layer + sigmoid in ONE forward launch (see cdc_tower_args in cdcmdr.h).
//
// grid = ceil(M/64) x n_tower workgroups of 256 threads; workgroup (rb, t) owns rows [64 rb, 64 rb + 64) of tower t, wave w its
// 16-row slice.  Per layer: z = x W^T + b with v_mfma_f32_16x16x32_bf16 — the A fragments straight from the bf16 input (layer 0:
// global memory, later layers: the previous layer's activations in LDS), the B fragments straight from the bf16 weight copy (a
// tower's weights are a few KB, served by L2) — then the BatchNorm statistics of the 64 rows as fp64 column sums -> `partial`,
// a grid-wide barrier, every workgroup adds the row blocks' partials in index order (same bits everywhere), normalise + ReLU
// + dropout in registers, next layer.  The last layer's activations meet the output weights in registers.
#include "common.h"

typedef __bf16 tw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float tw_f32x4 __attribute__((ext_vector_type(4)));

#define TW_THREADS 256
#define TW_ROWS 64
#define TW_NT (CDC_TOWER_MAX_DIM / 16)           /* 16-column output tiles per layer, at most */
#define TW_LDH (CDC_TOWER_MAX_DIM + 8)           /* bf16 elements per LDS row: 272 bytes, 16-byte aligned, rows shifted by 4 banks */

// All workgroups of the launch: arrive, then wait until `target` have.  What crosses workgroups (the partial sums) is written and
// read with agent-scope ATOMIC stores / loads, which go to the level all XCDs share, and the counter is an atomic: no
// __threadfence() anywhere — an agent-scope fence writes back / invalidates a whole XCD's L2 and made this launch take 100 us.
// Thread 0 spins with a sleep and a give-up bound (a workgroup that can never arrive must not hang the device).
__device__ __forceinline__ void tower_barrier(int32_t* cnt, int target, int32_t* err) {
    __syncthreads();                                                     // (waits for the workgroup's own stores)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { if (err) atomicExch(err, 1); break; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void tower_put(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double tower_get(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(TW_THREADS) k_tower_fwd(const cdc_tower_args a) {
    __shared__ __attribute__((aligned(16))) __bf16 htile[TW_ROWS][TW_LDH];
    __shared__ double wsum[2][TW_THREADS / 64][CDC_TOWER_MAX_DIM];
    __shared__ double csum[2][CDC_TOWER_MAX_DIM];
    __shared__ float col_mean[CDC_TOWER_MAX_DIM], col_scale[CDC_TOWER_MAX_DIM], col_beta[CDC_TOWER_MAX_DIM];
    __shared__ float wide_sh[TW_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x % a.n_tower, rb = blockIdx.x / a.n_tower;
    const int NB = (int)((a.M + TW_ROWS - 1) / TW_ROWS);
    const int n_blocks = NB * a.n_tower;
    const cdc_tower_desc& T = a.t[t];
    const int M = (int)a.M;
    const int r_blk = rb * TW_ROWS;
    const int frow = lane & 15, fk = (lane >> 4) * 8;                    // operand fragments: row / column of the 16x32 piece
    const int crow = (lane >> 4) * 4, ccol = lane & 15;                  // accumulator fragments: rows crow..crow+3, column ccol
    const int r_a = min(r_blk + wave * 16 + frow, M - 1);                // (rows past M: a valid row is read, its results are dropped)
    const bool stats = a.training && M > 1;                              // the reference skips BatchNorm for a batch of one row
    // development aid: workgroup 0 leaves wall-clock marks (100 MHz) of its phases behind the error flag (err[2..], as int64)
    long long* marks = (blockIdx.x == 0 && tid == 0 && a.err) ? reinterpret_cast<long long*>(a.err + 2) : nullptr;
    int n_mark = 0;
#define TW_MARK() do { if (marks && n_mark < 12) marks[n_mark++] = (long long)wall_clock64(); } while (0)
    TW_MARK();

    // ---- the wide term (wide_K MACs per row).  Training: the row block's n_tower workgroups share its 64 rows out (row rl goes to
    // tower rl % n_tower), publish the logits in wide_out (agent-scope stores) and pick all of them up after the first barrier —
    // a third of the bytes per workgroup at three towers, and one round of loads instead of four.  Eval (no barrier in this
    // launch): every workgroup forms all of its rows.
    const bool wide_shared = a.wide_x && stats && a.wide_out && a.n_tower > 1;
    if (a.wide_x) {
        const int KJ = (a.wide_K + 63) / 64;
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = lane + 64 * j; wv[j] = (j < KJ && k < a.wide_K) ? a.wide_w[k] : 0.f; }
        const int step = wide_shared ? a.n_tower : 1;
        int first = wave * 16;                                           // the wave's rows rl = first, first + step, ... < wave*16 + 16
        if (wide_shared) { const int m = first % a.n_tower; first += (t - m + a.n_tower) % a.n_tower; }
        for (int base = first; base < wave * 16 + 16; base += 6 * step) {            // six rows per round: all loads of a round in flight
            float xv[6][8];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int rl = base + q * step;
                const int r = min(r_blk + min(rl, TW_ROWS - 1), M - 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = lane + 64 * j;
                    xv[q][j] = (rl < wave * 16 + 16 && j < KJ && k < a.wide_K) ? a.wide_x[(int64_t)r * a.ld_wide + k] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int rl = base + q * step;
                if (rl >= wave * 16 + 16) break;                         // wave-uniform
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += xv[q][j] * wv[j];     // ascending k per lane, then the wave sum (csrc/head.hip's order)
                const int r = min(r_blk + rl, M - 1);
                for (int k = lane + 512; k < a.wide_K; k += 64) acc += a.wide_x[(int64_t)r * a.ld_wide + k] * a.wide_w[k];
                acc = wave_sum(acc);
                if (a.wide_bias) acc += a.wide_bias[0];
                if (lane == 0) {
                    if (wide_shared) {
                        if (r_blk + rl < M) __hip_atomic_store(a.wide_out + (int64_t)(r_blk + rl) * a.ld_wide_out, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        wide_sh[rl] = acc;
                    }
                }
            }
        }
        if (!wide_shared && t == 0 && a.wide_out) {
            __syncthreads();
            if (tid < TW_ROWS && r_blk + tid < M) a.wide_out[(int64_t)(r_blk + tid) * a.ld_wide_out] = wide_sh[tid];
        }
    }
    TW_MARK();                                                           // 1: wide term done
    tw_f32x4 acc[TW_NT];
    float yv[TW_NT][4];
    for (int l = 0; l < a.n_layer; ++l) {
        const cdc_tower_layer& Lr = T.l[l];
        const int K = l == 0 ? a.K0 : a.H[l - 1];
        const int H = a.H[l];
        const int KS = (K + 31) / 32, NT = H / 16;
#pragma unroll
        for (int nt = 0; nt < TW_NT; ++nt) acc[nt] = tw_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) {
            tw_bf16x8 af;
            if (l == 0) af = *reinterpret_cast<const tw_bf16x8*>(reinterpret_cast<const __bf16*>(T.xh) + (int64_t)r_a * T.ldxh + ks * 32 + fk);
            else af = *reinterpret_cast<const tw_bf16x8*>(&htile[wave * 16 + frow][ks * 32 + fk]);
#pragma unroll
            for (int nt = 0; nt < TW_NT; ++nt) {
                if (nt >= NT) break;
                const tw_bf16x8 bf = *reinterpret_cast<const tw_bf16x8*>(reinterpret_cast<const __bf16*>(Lr.wh) + (int64_t)(nt * 16 + frow) * Lr.ldwh + ks * 32 + fk);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[nt], 0, 0, 0);
            }
        }
        TW_MARK();                                                       // 2/6: contractions done
        // bias, the pre-norm output (read by the backward), and the statistics of the block's rows
        double s1[TW_NT], s2[TW_NT];
#pragma unroll
        for (int nt = 0; nt < TW_NT; ++nt) {
            s1[nt] = 0.0; s2[nt] = 0.0;
            if (nt >= NT) continue;
            const int col = nt * 16 + ccol;
            const float bv = Lr.bias ? Lr.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r_blk + wave * 16 + crow + i;
                const float z = acc[nt][i] + bv;
                acc[nt][i] = z;
                if (r < M) {
                    Lr.z[(int64_t)r * Lr.ldz + col] = z;
                    s1[nt] += (double)z; s2[nt] += (double)z * (double)z;
                }
            }
        }
        if (stats) {
#pragma unroll
            for (int nt = 0; nt < TW_NT; ++nt) {
                if (nt >= NT) break;
                s1[nt] += __shfl_xor(s1[nt], 16, 64); s2[nt] += __shfl_xor(s2[nt], 16, 64);
                s1[nt] += __shfl_xor(s1[nt], 32, 64); s2[nt] += __shfl_xor(s2[nt], 32, 64);
                if (lane < 16) { wsum[0][wave][nt * 16 + lane] = s1[nt]; wsum[1][wave][nt * 16 + lane] = s2[nt]; }
            }
            __syncthreads();
            double* mine = a.partial + (((int64_t)l * NB + rb) * a.n_tower + t) * (2 * CDC_TOWER_MAX_DIM);
            if (tid < H) {
                tower_put(mine + 2 * tid, ((wsum[0][0][tid] + wsum[0][1][tid]) + wsum[0][2][tid]) + wsum[0][3][tid]);
                tower_put(mine + 2 * tid + 1, ((wsum[1][0][tid] + wsum[1][1][tid]) + wsum[1][2][tid]) + wsum[1][3][tid]);
            }
            TW_MARK();                                                   // 3/7: partials written
            tower_barrier(a.sync + l, n_blocks, a.err);
            TW_MARK();                                                   // 4/8: barrier passed
            // the tower's column sums over all row blocks: TPC = 256 / Hp threads per column (Hp = H rounded up to a power of two),
            // thread (j, q) adds blocks q, q + TPC, ... in ascending order — sixteen loads in flight at a time — then the TPC parts
            // are added in index order: every workgroup forms the same bits
            {
                int hp = 16;
                while (hp < H) hp <<= 1;
                const int TPC = TW_THREADS / hp;
                const int j = tid % hp, q = tid / hp;
                double a1 = 0.0, a2 = 0.0;
                if (j < H) {
                    const double* base = a.partial + ((int64_t)l * NB * a.n_tower + t) * (2 * CDC_TOWER_MAX_DIM) + 2 * j;
                    const int64_t bstride = (int64_t)a.n_tower * (2 * CDC_TOWER_MAX_DIM);
                    for (int b0 = q; b0 < NB; b0 += 16 * TPC) {
                        double v1[16], v2[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int b = b0 + u * TPC;
                            const double* p = base + (int64_t)min(b, NB - 1) * bstride;
                            v1[u] = tower_get(p); v2[u] = tower_get(p + 1);
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            if (b0 + u * TPC < NB) { a1 += v1[u]; a2 += v2[u]; }
                    }
                }
                double* flat1 = &wsum[0][0][0];
                double* flat2 = &wsum[1][0][0];
                flat1[q * hp + j] = a1; flat2[q * hp + j] = a2;
                __syncthreads();
                if (tid < H) {
                    double c1 = 0.0, c2 = 0.0;
                    for (int qq = 0; qq < TPC; ++qq) { c1 += flat1[qq * hp + tid]; c2 += flat2[qq * hp + tid]; }
                    csum[0][tid] = c1; csum[1][tid] = c2;
                }
                __syncthreads();
            }
            if (l == 0 && wide_shared) {                                 // everybody's share of the wide term has been published
                if (tid < TW_ROWS) {
                    const int r = min(r_blk + tid, M - 1);
                    wide_sh[tid] = __hip_atomic_load(a.wide_out + (int64_t)r * a.ld_wide_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            }
        }
        TW_MARK();                                                       // 5/9: column sums gathered
        if (tid < H) {
            float mean = 0.f, invstd = 1.f;
            if (a.training) {
                if (stats) {
                    const double mu = csum[0][tid] / M;
                    double var = csum[1][tid] / M - mu * mu;
                    if (var < 0.0) var = 0.0;
                    mean = (float)mu;
                    invstd = (float)(1.0 / sqrt(var + (double)a.eps));
                    if (rb == 0) {
                        if (Lr.save_mean) Lr.save_mean[tid] = mean;
                        if (Lr.save_invstd) Lr.save_invstd[tid] = invstd;
                        if (Lr.running_mean) {
                            const double unbiased = var * ((double)M / (double)(M - 1));
                            Lr.running_mean[tid] = (1.f - a.momentum) * Lr.running_mean[tid] + a.momentum * mean;
                            Lr.running_var[tid] = (1.f - a.momentum) * Lr.running_var[tid] + a.momentum * (float)unbiased;
                        }
                    }
                }
            } else {
                mean = Lr.running_mean[tid];
                invstd = 1.f / sqrtf(Lr.running_var[tid] + a.eps);
            }
            const bool norm = a.training ? stats : true;
            col_mean[tid] = mean;
            col_scale[tid] = norm ? invstd * (Lr.gamma ? Lr.gamma[tid] : 1.f) : 1.f;
            col_beta[tid] = (norm && Lr.beta) ? Lr.beta[tid] : 0.f;
            if (!norm) col_mean[tid] = 0.f;
        }
        if (stats && rb == 0 && tid == 0 && Lr.num_batches_tracked) *Lr.num_batches_tracked += 1;
        __syncthreads();
        // normalise + ReLU + dropout in the accumulator registers; outputs for the backward; bf16 tile for the next layer
        const bool drop = a.training && a.drop_p > 0.f;
        const float keep_scale = drop ? 1.f / (1.f - a.drop_p) : 1.f;
        const uint32_t thr16 = (uint32_t)(a.drop_p * 65536.f + 0.5f);
        const uint32_t seed32 = drop ? g2_seed32(a.seed + (uint64_t)l * 0x9E3779B97F4A7C15ull, a.seed_offset_dev, 128 + t) : 0u;
#pragma unroll
        for (int nt = 0; nt < TW_NT; ++nt) {
            if (nt >= NT) break;
            const int col = nt * 16 + ccol;
            const float mean = col_mean[col], sc = col_scale[col], be = col_beta[col];
            const float gam = Lr.gamma ? Lr.gamma[col] : 1.f;
            (void)gam;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r_blk + wave * 16 + crow + i;
                float v = (acc[nt][i] - mean) * sc + be;
                v = fmaxf(v, 0.f);
                if (drop) {
                    const uint32_t h = g2_drop_bits(seed32, r, col >> 1);
                    v = ((col & 1) ? (h >> 16) : (h & 0xFFFFu)) < thr16 ? 0.f : v * keep_scale;
                }
                yv[nt][i] = v;
                if (r < M) {
                    if (Lr.y) Lr.y[(int64_t)r * Lr.ldy + col] = v;
                    if (Lr.yh) reinterpret_cast<__bf16*>(Lr.yh)[(int64_t)r * Lr.ldyh + col] = (__bf16)v;
                }
                if (l + 1 < a.n_layer) htile[wave * 16 + crow + i][col] = (__bf16)v;
            }
        }
        if (l + 1 < a.n_layer) {
            const int Kn = (H + 31) / 32 * 32;                           // the next layer reads whole 32-column pieces: zero the rest
            for (int c = H + (tid % 32); c < Kn; c += 32)
                for (int r = tid / 32; r < TW_ROWS; r += TW_THREADS / 32) htile[r][c] = (__bf16)0.f;
            __syncthreads();
        }
    }

    // ---- output layer: logit = y_last . w_out + b_out (+ wide + addends), sigmoid -------------------------------------------
    {
        const int H = a.H[a.n_layer - 1], NT = H / 16;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < TW_NT; ++nt) {
            if (nt >= NT) break;
            const float w = T.w_out[nt * 16 + ccol];
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] += yv[nt][i] * w;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] += __shfl_xor(p[i], o, 64);
        if (ccol == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = wave * 16 + crow + i, r = r_blk + rl;
                if (r >= M) continue;
                float v = p[i];
                if (T.b_out) v += T.b_out[0];
                if (a.wide_x) v += wide_sh[rl];
                for (int q = 0; q < a.n_addend; ++q) v += a.addend[q][(int64_t)r * a.ld_addend[q]];
                if (a.sigmoid) v = 1.f / (1.f + expf(-v));
                a.out[(int64_t)r * a.ld_out + t] = v;
            }
        }
    }
    TW_MARK();                                                           // 10: outputs written
    // ---- leave the counters zero for the next launch: the last workgroup to get here resets them ------------------------------
    if (stats) {
        __syncthreads();
        if (tid == 0) {
            const int done = atomicAdd(a.sync + CDC_TOWER_MAX_LAYERS, 1);
            if (done == n_blocks - 1) {
                for (int l = 0; l <= CDC_TOWER_MAX_LAYERS; ++l) atomicExch(a.sync + l, 0);
            }
        }
    }
}

extern "C" int64_t cdc_tower_fwd_workspace_doubles(int64_t M, int32_t n_tower, int32_t n_layer) {
    if (M < 0 || n_tower <= 0 || n_layer <= 0) return -1;
    return (int64_t)n_layer * cdc_ceil_div(std::max<int64_t>(M, 1), TW_ROWS) * n_tower * 2 * CDC_TOWER_MAX_DIM;
}

static int tower_capacity() {
    static int cap = -1;
    if (cap < 0) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tower_fwd, TW_THREADS, 0) != hipSuccess) return 0;
        cap = per_cu * prop.multiProcessorCount;
    }
    return cap;
}
extern "C" int cdc_tower_fwd_fits(int64_t M, int32_t n_tower) {
    if (M <= 0 || n_tower <= 0 || n_tower > CDC_TOWER_MAX) return 0;
    const int64_t blocks = cdc_ceil_div(M, TW_ROWS) * n_tower;
    // half of what the device could hold: the launch must not depend on having the chip to itself
    return blocks * 2 <= tower_capacity() ? 1 : 0;
}

extern "C" int cdc_tower_fwd(const cdc_tower_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_tower > 0 && a->n_tower <= CDC_TOWER_MAX && a->n_layer > 0 && a->n_layer <= CDC_TOWER_MAX_LAYERS && a->M >= 0 &&
                      a->out && a->ld_out >= a->n_tower && a->n_addend >= 0 && a->n_addend <= 2, CDC_E_BADARG, "tower_fwd: bad argument");
    CDC_CHECK_ARG(a->K0 > 0 && a->K0 % 16 == 0 && a->K0 <= CDC_TOWER_MAX_DIM, CDC_E_BADARG, "tower_fwd: K0 must be a multiple of 16 up to %d", CDC_TOWER_MAX_DIM);
    for (int l = 0; l < a->n_layer; ++l)
        CDC_CHECK_ARG(a->H[l] > 0 && a->H[l] % 16 == 0 && a->H[l] <= CDC_TOWER_MAX_DIM, CDC_E_BADARG, "tower_fwd: layer %d width", l);
    CDC_CHECK_ARG(!a->training || a->M <= 1 || (a->partial && a->sync), CDC_E_BADARG, "tower_fwd: training needs partial and sync");
    CDC_CHECK_ARG(!a->wide_x || (a->wide_w && a->wide_K > 0 && a->ld_wide >= a->wide_K), CDC_E_BADARG, "tower_fwd: wide term malformed");
    for (int t = 0; t < a->n_tower; ++t) {
        const cdc_tower_desc& T = a->t[t];
        CDC_CHECK_ARG(T.xh && T.w_out && (((uintptr_t)T.xh) & 15) == 0 && T.ldxh % 8 == 0, CDC_E_BADARG, "tower_fwd: tower %d input", t);
        for (int l = 0; l < a->n_layer; ++l) {
            const cdc_tower_layer& L = T.l[l];
            CDC_CHECK_ARG(L.wh && L.z && (((uintptr_t)L.wh) & 15) == 0 && L.ldwh % 8 == 0 && (a->training || (L.running_mean && L.running_var)),
                          CDC_E_BADARG, "tower_fwd: tower %d layer %d malformed", t, l);
        }
    }
    if (a->M == 0) return 0;
    const int64_t blocks = cdc_ceil_div(a->M, TW_ROWS) * a->n_tower;
    if (a->training && a->M > 1)
        CDC_CHECK_ARG(blocks <= tower_capacity(), CDC_E_TOOBIG, "tower_fwd: %lld workgroups cannot be resident at once", (long long)blocks);
    hipLaunchKernelGGL(k_tower_fwd, dim3((unsigned)blocks), dim3(TW_THREADS), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("tower_fwd");
    return 0;
}
