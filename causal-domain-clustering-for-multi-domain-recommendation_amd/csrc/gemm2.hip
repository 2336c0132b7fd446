// gemm2.hip — the bf16 contraction path of the grouped linear layers, operands ALREADY bf16 in memory.
//
// Same arithmetic as gemm.hip's CDC_PREC_BF16 path (v_mfma_f32_16x16x32_bf16: bf16 operands rounded to nearest even from
// the fp32 values, fp32 accumulate, fp32 bias / activation epilogue) for every nn.Linear of the reference's path
// (model/layer.py:185,193,275; model/ple.py:83-94; model/mmoe.py:35-40) and its two autograd GEMMs that contract over
// a memory-contiguous index (forward: x·Wᵀ over K; grad-input: dZ·W over N with the per-step Wᵀ copy) — but the rounding
// happens ONCE, where a tensor is produced (a "shadow" copy written by the producer's epilogue, or by cdc_shadow_bf16 /
// cdc_weight_shadows), not every time a tile of it is staged.  What that buys on gfx950:
//   * half the operand bytes from L2, no v_cvt_pk_bf16_f32 in the K loop;
//   * tiles go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction in full 128-byte lines), no
//     register staging: 128x128 (or 128x64) output tiles with BK = 64 at ~100 VGPRs, two workgroups per CU;
//   * LDS image = [rows][64 bf16] with the 16-byte chunk index XORed by (row & 7) — applied to the SOURCE address of the
//     direct load and to the fragment read, the LDS destination stays lane-linear — so every ds_read_b128 of an MFMA
//     fragment is bank-conflict free;
//   * two LDS stages: the loads of K-slab t+1 are in flight under the MFMAs of slab t, one barrier per slab.
// K-slabs are whole: every operand row must be READABLE up to the next multiple of 64 elements and the padding of at
// least one operand of each product must be zero (the shadows are allocated zero-padded for that).
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#ifndef G2_PROBE
#define G2_PROBE 0      /* tools/gemm2_probe.hip: 1 = no epilogue, 2 = no MFMAs, 4 = no reloads, 8 = no global stores, 16 = no fragment reads */
#endif
#define G2_THREADS 256
#define G2_BK 64
#define G2_ROW_BYTES (G2_BK * 2)        /* one LDS row: 64 bf16 = 128 B = eight 16-byte chunks */

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ int g2_xcd_remap(int bid, int nblk) {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}

// Dropout decisions of this path: one 32-bit counter hash (lowbias32) per PAIR of neighbouring columns, 16 bits per element —
// keep iff bits >= round(p * 65536).  (The 64-bit mix of cdc_uniform costs ~150 issue cycles per element: 4 us per 128x128
// tile of the level-1 launch; this one ~25.)  Never regenerated in backward: the mask is read off the saved output.
// (g2_hash32 / g2_seed32 / g2_drop_bits: csrc/common.h)

template <int BM, int BN, int NSTAGE>
struct G2Cfg {
    static constexpr int MT = BM / 32, NT = BN / 32;                // 16x16 MFMA tiles per wave (wave tile BM/2 x BN/2, waves 2 x 2)
    static constexpr int A_BYTES = BM * G2_ROW_BYTES;
    static constexpr int B_BYTES = BN * G2_ROW_BYTES;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int CS = BN + 4;                               // epilogue tile row stride (floats)
    static constexpr int OUT_BYTES = BM * CS * 4;
    static constexpr int Q_BYTES = 4 * 64 * 2 * 8;                  // BatchNorm partial-sum scratch behind the epilogue tile
    static constexpr int SMEM = NSTAGE * STAGE > OUT_BYTES + Q_BYTES ? NSTAGE * STAGE : OUT_BYTES + Q_BYTES;
    static constexpr int A_PER_WAVE = BM / 32;                      // direct-load instructions per wave and stage (8 rows each)
    static constexpr int B_PER_WAVE = BN / 32;
    static constexpr int LOADS = A_PER_WAVE + B_PER_WAVE;           // per wave and slab: what one vmcnt unit of a slab is
    static constexpr int BLOCKS_PER_CU = SMEM <= 53 * 1024 ? 3 : (SMEM <= 80 * 1024 ? 2 : 1);
};

template <int N> __device__ __forceinline__ void g2_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int NSTAGE>
__global__ void __launch_bounds__(G2_THREADS, (G2Cfg<BM, BN, NSTAGE>::BLOCKS_PER_CU)) k_g2_nt(const cdc_g2_args a) {
    CDC_PRIO_MAIN();
    typedef G2Cfg<BM, BN, NSTAGE> Cfg;
    constexpr int G2_BM = BM;                                        // (the epilogue below predates the BM template parameter)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile = g2_xcd_remap(blockIdx.x, gridDim.x);
    const int o = find_group<false>(a.n_out, tile, [&](int l) { return ((a.o[l].M + BM - 1) / BM) * ((a.o[l].N + BN - 1) / BN); },
                                    [](int) { return (int64_t)0; }, tile, nullptr);
    if (o < 0) return;
    const cdc_g2_out& O = a.o[o];
    const int tn_cnt = (O.N + BN - 1) / BN;
    const int i0 = (tile / tn_cnt) * BM, j0 = (tile % tn_cnt) * BN;
    const int M = O.M, N = O.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

    f32x4_t acc[Cfg::MT][Cfg::NT];
#pragma unroll
    for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // direct loads: one wave instruction = 8 rows x 128 B; lane l lands on (row l>>3, physical chunk l&7) and therefore
    // fetches the logical chunk (l&7) ^ (row&7) of its row
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    // fragment reads: lane -> row (l&15) of a 16-row block, logical chunk ks*4 + (l>>4)
    const int frow = lane & 15;
    const int fx = lane & 7;                                         // (row & 7) of every row this lane reads

    // segments of this output: lane l holds segment l's descriptor (one vector load for the whole list).  The slabs of all
    // segments form ONE sequence through the ring: the loads of the next segment's first slabs are in flight under the
    // MFMAs of this segment's last ones (grad-input of a layer whose 12 groups share one input restarts nothing 12 times).
    unsigned long long todo;
    uint64_t v_a, v_b;
    int64_t v_lda, v_ldb;
    int v_nk;
    int total = 0;                                                   // slabs of this output, all segments
    {
        const bool has = lane < a.n_seg;
        const cdc_g2_seg& L = a.s[has ? lane : 0];
        v_a = (uint64_t)L.a; v_b = (uint64_t)L.b; v_lda = L.lda; v_ldb = L.ldb;
        const bool mine = has && L.out == o;
        v_nk = mine ? L.Kr / G2_BK : 0;
        todo = __ballot(mine);
        int t = v_nk;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        total = __builtin_amdgcn_readfirstlane(t);
    }
    auto lane64 = [](uint64_t v, int l) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    };
    const __bf16* pa[Cfg::A_PER_WAVE];
    const __bf16* pb[Cfg::B_PER_WAVE];
    int left = 0;                                                    // slabs of the segment being issued that are not issued yet
    int issued = 0, istage = 0;
    auto issue_next = [&]() {
        if (left == 0) {                                             // next segment (uniform): its row pointers
            const int s = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            todo &= todo - 1;
            const __bf16* A = reinterpret_cast<const __bf16*>(lane64(v_a, s));
            const __bf16* B = reinterpret_cast<const __bf16*>(lane64(v_b, s));
            const int64_t lda = (int64_t)lane64((uint64_t)v_lda, s), ldb = (int64_t)lane64((uint64_t)v_ldb, s);
            left = __builtin_amdgcn_readlane(v_nk, s);
#pragma unroll
            for (int q = 0; q < Cfg::A_PER_WAVE; ++q) {
                int r = i0 + wave * (BM / 4) + q * 8 + lrow;
                r = r < M ? r : M - 1;                               // rows past the extent only feed accumulators nobody stores
                pa[q] = A + (int64_t)r * lda + lchunk * 8;
            }
#pragma unroll
            for (int q = 0; q < Cfg::B_PER_WAVE; ++q) {
                int r = j0 + wave * (BN / 4) + q * 8 + lrow;
                r = r < N ? r : N - 1;
                pb[q] = B + (int64_t)r * ldb + lchunk * 8;
            }
        }
        unsigned char* as = smem + istage * Cfg::STAGE + wave * (BM / 4) * G2_ROW_BYTES;
        unsigned char* bs = smem + istage * Cfg::STAGE + Cfg::A_BYTES + wave * (BN / 4) * G2_ROW_BYTES;
#pragma unroll
        for (int q = 0; q < Cfg::A_PER_WAVE; ++q) { glds16(pa[q], as + q * 8 * G2_ROW_BYTES); pa[q] += G2_BK; }
#pragma unroll
        for (int q = 0; q < Cfg::B_PER_WAVE; ++q) { glds16(pb[q], bs + q * 8 * G2_ROW_BYTES); pb[q] += G2_BK; }
        --left;
        ++issued;
        istage = istage + 1 == NSTAGE ? 0 : istage + 1;
    };
    // ring of NSTAGE LDS slabs, NSTAGE-1 of them in flight: slab t is waited for with a COUNTED vmcnt (the younger slabs stay
    // in flight across the barrier), the slab freed by the barrier (read during t-1) is refilled right after it
#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p)
        if (issued < total) issue_next();
    int stage = 0;
    for (int t = 0; t < total; ++t) {
        if (NSTAGE > 2 && total - t - 1 >= NSTAGE - 2) g2_wait_vmcnt<Cfg::LOADS * (NSTAGE - 2)>();
        else g2_wait_vmcnt<0>();                                     // (the last slabs: fewer are in flight)
        __builtin_amdgcn_s_barrier();                                // everyone's share of slab t has landed; slab t-1's reads are over
        if (issued < total && !(G2_PROBE & 4)) issue_next();
        const unsigned char* As = smem + stage * Cfg::STAGE;
        const unsigned char* Bs = As + Cfg::A_BYTES;
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
        bf16x8_t af[2][Cfg::MT], bfr[2][Cfg::NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                             // all fragment reads of the slab first: the second half's land under the first half's MFMAs
            const int coff = ((ks * 4 + (lane >> 4)) ^ fx) << 4;
            if ((G2_PROBE & 16) && t > 0) {
#pragma unroll
                for (int mt = 0; mt < Cfg::MT; ++mt) af[ks][mt] = (bf16x8_t)(__bf16)(float)(lane + mt);
#pragma unroll
                for (int nt = 0; nt < Cfg::NT; ++nt) bfr[ks][nt] = (bf16x8_t)(__bf16)(float)(lane - nt);
            } else {
#pragma unroll
                for (int mt = 0; mt < Cfg::MT; ++mt)
                    af[ks][mt] = *reinterpret_cast<const bf16x8_t*>(As + (wm + mt * 16 + frow) * G2_ROW_BYTES + coff);
#pragma unroll
                for (int nt = 0; nt < Cfg::NT; ++nt)
                    bfr[ks][nt] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn + nt * 16 + frow) * G2_ROW_BYTES + coff);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (G2_PROBE & 2) {
#pragma unroll
                for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < Cfg::NT; ++nt) {
                        acc[mt][nt][0] += (float)af[ks][mt][0] + (float)bfr[ks][nt][0];
                        acc[mt][nt][1] += (float)af[ks][mt][7] + (float)bfr[ks][nt][7];
                    }
            } else {
#pragma unroll
                for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < Cfg::NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][mt], bfr[ks][nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: accumulators -> LDS tile -> whole rows out (fp32 and/or the bf16 shadow) ------------------------------
    if (G2_PROBE & 1) {
        float sacc = 0.f;
#pragma unroll
        for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < Cfg::NT; ++nt) sacc += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
        if (sacc == 123.456f) O.y[0] = sacc;
        return;
    }
    __syncthreads();
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mt = 0; mt < Cfg::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ct[(wm + mt * 16 + (lane >> 4) * 4 + r) * Cfg::CS + wn + nt * 16 + (lane & 15)] = acc[mt][nt][r];
    __syncthreads();

    const bool fwd = a.mode == 0;
    if (fwd && O.bn_partial) {
        // statistics pass of the BatchNorm that normalises y next (cdc_bn_fwd_args.stats_ready): per (64-row block, column)
        // sums of y and y^2 in double; wave q takes rows 16q..16q+15 of a block, lane = column, quarters added in order
        static_assert(CDC_BN_ROWS_PER_BLOCK == 64, "partial-sum blocks are 64 rows");
        double (*quarter)[64][2] = reinterpret_cast<double (*)[64][2]>(smem + Cfg::OUT_BYTES);   // all LDS in the one dynamic array
#pragma unroll
        for (int cc = 0; cc < BN / 64; ++cc) {
            const int col = j0 + cc * 64 + lane;
            const float bv = (col < N && O.bias) ? O.bias[col] : 0.f;
#pragma unroll
            for (int h = 0; h < G2_BM / 64; ++h) {
                const int r_lo = i0 + h * 64;
                if (r_lo >= M) break;                                 // uniform
                const int rows = min(64, M - r_lo);
                double s1 = 0.0, s2 = 0.0;
                for (int r = wave * 16; r < min(wave * 16 + 16, rows); ++r) {
                    const double x = (double)(ct[(h * 64 + r) * Cfg::CS + cc * 64 + lane] + bv);
                    s1 += x; s2 += x * x;
                }
                quarter[wave][lane][0] = s1; quarter[wave][lane][1] = s2;
                __syncthreads();
                if (wave == 0 && col < N) {
                    double* ws = O.bn_partial + ((int64_t)(r_lo / 64) * O.bn_total_c + O.bn_col0 + col) * 2;
                    ws[0] = ((quarter[0][lane][0] + quarter[1][lane][0]) + quarter[2][lane][0]) + quarter[3][lane][0];
                    ws[1] = ((quarter[0][lane][1] + quarter[1][lane][1]) + quarter[2][lane][1]) + quarter[3][lane][1];
                }
                __syncthreads();
            }
        }
    }

    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thr16 = (uint32_t)(a.drop_p * 65536.f + 0.5f);
    const uint32_t seed32 = (fwd && a.drop_p > 0.f) ? g2_seed32(a.seed, a.seed_offset_dev, O.stream_id) : 0u;
    constexpr int C8 = BN / 8;                                       // 8-column pieces per tile row
    constexpr int ROWS_PER_PASS = G2_THREADS / C8;
    const int c8 = (tid % C8) * 8, lr0 = tid / C8;
    const int col = j0 + c8;
    const bool vec_y = !O.y || (((((uintptr_t)O.y) & 15) == 0) && (O.ldy % 4 == 0));
    const bool vec_h = !O.yh || (((((uintptr_t)O.yh) & 15) == 0) && (O.ldyh % 8 == 0));
    const bool vec_m = !O.mask || (((((uintptr_t)O.mask) & 15) == 0) && (O.mask_bf16 ? (O.ldmask % 8 == 0) : (O.ldmask % 4 == 0)));
    // per THREAD: its 8 columns are all inside the output and on one side of the activation boundary -> vector path;
    // a piece that straddles N or act_cols (or unaligned destinations) -> element-wise path; pieces past N: nothing to do
    if (col >= N) return;
    const int row_end = min(G2_BM, M - i0);
    const bool act_all = col + 8 <= O.act_cols, act_none = col >= O.act_cols;
    float* const yp = O.y ? O.y + (int64_t)i0 * O.ldy + col : nullptr;
    __bf16* const hp = O.yh ? reinterpret_cast<__bf16*>(O.yh) + (int64_t)i0 * O.ldyh + col : nullptr;

    if (col + 8 <= N && vec_y && vec_h && vec_m && (act_all || act_none)) {
        // ---- fast path (every tile of a whole-slab problem): straight-line code, 8 columns per thread, all decisions uniform
        if (fwd) {
            f32x4_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
            if (O.bias) { b0 = *reinterpret_cast<const f32x4_t*>(O.bias + col); b1 = *reinterpret_cast<const f32x4_t*>(O.bias + col + 4); }
            const bool relu = act_all && a.relu, drop = act_all && a.drop_p > 0.f;
            const uint32_t ckey = seed32 + (uint32_t)(col >> 1) * 0x85EBCA77U;
#pragma unroll 2
            for (int lr = lr0; lr < row_end; lr += ROWS_PER_PASS) {
                f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS + c8) + b0;
                f32x4_t hi = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS + c8 + 4) + b1;
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { lo[q] = fmaxf(lo[q], 0.f); hi[q] = fmaxf(hi[q], 0.f); }
                }
                if (drop) {
                    const uint32_t rkey = ckey + (uint32_t)(i0 + lr) * 0x9E3779B1U;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const uint32_t h0 = g2_hash32(rkey + (uint32_t)pr * 0x85EBCA77U);
                        const uint32_t h1 = g2_hash32(rkey + (uint32_t)(pr + 2) * 0x85EBCA77U);
                        lo[2 * pr] = (h0 & 0xFFFFu) < thr16 ? 0.f : lo[2 * pr] * keep_scale;
                        lo[2 * pr + 1] = (h0 >> 16) < thr16 ? 0.f : lo[2 * pr + 1] * keep_scale;
                        hi[2 * pr] = (h1 & 0xFFFFu) < thr16 ? 0.f : hi[2 * pr] * keep_scale;
                        hi[2 * pr + 1] = (h1 >> 16) < thr16 ? 0.f : hi[2 * pr + 1] * keep_scale;
                    }
                }
                if (G2_PROBE & 8) { if (lo[0] != 123.456f) continue; }
                if (yp) {
                    *reinterpret_cast<f32x4_t*>(yp + (int64_t)lr * O.ldy) = lo;
                    *reinterpret_cast<f32x4_t*>(yp + (int64_t)lr * O.ldy + 4) = hi;
                }
                if (hp) {
                    bf16x8_t h8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { h8[q] = (__bf16)lo[q]; h8[4 + q] = (__bf16)hi[q]; }
                    *reinterpret_cast<bf16x8_t*>(hp + (int64_t)lr * O.ldyh) = h8;
                }
            }
        } else if (act_all && O.mask != nullptr && O.mask_bf16 && !O.accumulate) {
            // grad-input whose activation mask is a bf16 shadow (the expert stacks): ALL of the thread's mask vectors are fetched
            // before the first is used — one at a time, every pass of the row loop waited a memory latency (the layer-2 grad-input
            // launch: 2 K slabs of MFMA work behind 8 such waits)
            constexpr int PASSES = G2_BM / ROWS_PER_PASS;
            bf16x8_t m8[PASSES];
#pragma unroll
            for (int p_ = 0; p_ < PASSES; ++p_) {
                const int lr = lr0 + p_ * ROWS_PER_PASS;
                m8[p_] = lr < row_end ? *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const __bf16*>(O.mask) + (int64_t)(i0 + lr) * O.ldmask + col)
                                      : bf16x8_t{};
            }
#pragma unroll
            for (int p_ = 0; p_ < PASSES; ++p_) {
                const int lr = lr0 + p_ * ROWS_PER_PASS;
                if (lr >= row_end) break;
                f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS + c8);
                f32x4_t hi = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS + c8 + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    lo[q] = (float)m8[p_][q] > 0.f ? lo[q] * a.mask_scale : 0.f;
                    hi[q] = (float)m8[p_][4 + q] > 0.f ? hi[q] * a.mask_scale : 0.f;
                }
                if (G2_PROBE & 8) { if (lo[0] != 123.456f) continue; }
                if (yp) {
                    *reinterpret_cast<f32x4_t*>(yp + (int64_t)lr * O.ldy) = lo;
                    *reinterpret_cast<f32x4_t*>(yp + (int64_t)lr * O.ldy + 4) = hi;
                }
                if (hp) {
                    bf16x8_t h8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { h8[q] = (__bf16)lo[q]; h8[4 + q] = (__bf16)hi[q]; }
                    *reinterpret_cast<bf16x8_t*>(hp + (int64_t)lr * O.ldyh) = h8;
                }
            }
        } else {
            const bool masked = act_all && O.mask != nullptr;
#pragma unroll 2
            for (int lr = lr0; lr < row_end; lr += ROWS_PER_PASS) {
                f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS + c8);
                f32x4_t hi = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS + c8 + 4);
                if (masked) {
                    if (O.mask_bf16) {
                        const bf16x8_t m8 = *reinterpret_cast<const bf16x8_t*>(reinterpret_cast<const __bf16*>(O.mask) + (int64_t)(i0 + lr) * O.ldmask + col);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            lo[q] = (float)m8[q] > 0.f ? lo[q] * a.mask_scale : 0.f;
                            hi[q] = (float)m8[4 + q] > 0.f ? hi[q] * a.mask_scale : 0.f;
                        }
                    } else {
                        const float* mp = reinterpret_cast<const float*>(O.mask) + (int64_t)(i0 + lr) * O.ldmask + col;
                        const f32x4_t m0 = *reinterpret_cast<const f32x4_t*>(mp), m1 = *reinterpret_cast<const f32x4_t*>(mp + 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            lo[q] = m0[q] > 0.f ? lo[q] * a.mask_scale : 0.f;
                            hi[q] = m1[q] > 0.f ? hi[q] * a.mask_scale : 0.f;
                        }
                    }
                }
                if (O.accumulate) {
                    lo = *reinterpret_cast<const f32x4_t*>(yp + (int64_t)lr * O.ldy) + lo;
                    hi = *reinterpret_cast<const f32x4_t*>(yp + (int64_t)lr * O.ldy + 4) + hi;
                }
                if (G2_PROBE & 8) { if (lo[0] != 123.456f) continue; }
                if (yp) {
                    *reinterpret_cast<f32x4_t*>(yp + (int64_t)lr * O.ldy) = lo;
                    *reinterpret_cast<f32x4_t*>(yp + (int64_t)lr * O.ldy + 4) = hi;
                }
                if (hp) {
                    bf16x8_t h8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { h8[q] = (__bf16)lo[q]; h8[4 + q] = (__bf16)hi[q]; }
                    *reinterpret_cast<bf16x8_t*>(hp + (int64_t)lr * O.ldyh) = h8;
                }
            }
        }
        return;
    }

    // ---- element-wise path: pieces that straddle an edge, unaligned destinations
    for (int lr = lr0; lr < row_end; lr += ROWS_PER_PASS) {
        const int row = i0 + lr;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ct[lr * Cfg::CS + c8 + q];
        if (fwd) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (col + q >= N) continue;
                float x = v[q] + (O.bias ? O.bias[col + q] : 0.f);
                if (col + q < O.act_cols) {
                    if (a.relu) x = fmaxf(x, 0.f);
                    if (a.drop_p > 0.f) {
                        const uint32_t h = g2_drop_bits(seed32, row, (col + q) >> 1);
                        const uint32_t bits = ((col + q) & 1) ? (h >> 16) : (h & 0xFFFFu);
                        x = bits < thr16 ? 0.f : x * keep_scale;
                    }
                }
                v[q] = x;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (col + q >= N) continue;
                if (O.mask && col + q < O.act_cols) {
                    const float mk = O.mask_bf16 ? (float)reinterpret_cast<const __bf16*>(O.mask)[(int64_t)row * O.ldmask + col + q]
                                                 : reinterpret_cast<const float*>(O.mask)[(int64_t)row * O.ldmask + col + q];
                    v[q] = mk > 0.f ? v[q] * a.mask_scale : 0.f;
                }
                if (O.accumulate) v[q] = O.y[(int64_t)row * O.ldy + col + q] + v[q];
            }
        }
        if (G2_PROBE & 8) { if (v[0] != 123.456f) continue; }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (col + q >= N) continue;
            if (O.y) O.y[(int64_t)row * O.ldy + col + q] = v[q];
            if (O.yh) reinterpret_cast<__bf16*>(O.yh)[(int64_t)row * O.ldyh + col + q] = (__bf16)v[q];
        }
    }
}

extern "C" int cdc_gemm_bf16_nt(const cdc_g2_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_out > 0 && a->n_out <= CDC_G2_MAX_OUT && a->n_seg > 0 && a->n_seg <= CDC_G2_MAX_SEG, CDC_E_BADARG,
                  "gemm_bf16_nt: bad counts");
    CDC_CHECK_ARG(a->mode == 0 || a->mode == 1, CDC_E_BADARG, "gemm_bf16_nt: mode must be 0 (forward) or 1 (grad-input)");
    CDC_CHECK_ARG(a->drop_p >= 0.f && a->drop_p < 1.f, CDC_E_BADARG, "gemm_bf16_nt: dropout p out of range");
    int max_n = 0, max_m = 0, max_kr = 0;
    for (int o = 0; o < a->n_out; ++o) {
        const cdc_g2_out& O = a->o[o];
        CDC_CHECK_ARG((O.y || O.yh) && O.M >= 0 && O.N > 0 && (!O.y || O.ldy >= O.N) && (!O.yh || O.ldyh >= O.N), CDC_E_BADARG,
                      "gemm_bf16_nt: output %d malformed (M=%d N=%d)", o, O.M, O.N);
        CDC_CHECK_ARG(!(O.accumulate && !O.y), CDC_E_BADARG, "gemm_bf16_nt: output %d accumulates without an fp32 destination", o);
        CDC_CHECK_ARG(!O.bn_partial || (a->mode == 0 && O.act_cols == 0 && O.bn_col0 >= 0 && O.bn_col0 + O.N <= O.bn_total_c), CDC_E_BADARG,
                      "gemm_bf16_nt: output %d cannot feed BatchNorm partial sums", o);
        if (O.N > max_n) max_n = O.N;
        if (O.M > max_m) max_m = O.M;
    }
    for (int s = 0; s < a->n_seg; ++s) {
        const cdc_g2_seg& S = a->s[s];
        CDC_CHECK_ARG(S.a && S.b && S.out >= 0 && S.out < a->n_out && S.Kr > 0 && S.Kr % G2_BK == 0 && S.lda >= S.Kr && S.ldb >= S.Kr,
                      CDC_E_BADARG, "gemm_bf16_nt: segment %d malformed (Kr=%d must be a positive multiple of 64 covered by both rows)", s, S.Kr);
        CDC_CHECK_ARG(((((uintptr_t)S.a) | ((uintptr_t)S.b)) & 15) == 0 && S.lda % 8 == 0 && S.ldb % 8 == 0, CDC_E_ALIGN,
                      "gemm_bf16_nt: segment %d operands must be 16-byte aligned with row strides that are multiples of 8 elements", s);
    }
    {   // the longest reduction of any output (all its segments together): what the ring depth / tile choice looks at
        int64_t per_out[CDC_G2_MAX_OUT] = {0};
        for (int s = 0; s < a->n_seg; ++s) per_out[a->s[s].out] += a->s[s].Kr;
        for (int o = 0; o < a->n_out; ++o)
            if (per_out[o] > max_kr) max_kr = (int)std::min<int64_t>(per_out[o], 1 << 30);
    }
    if (max_m == 0) return 0;
    auto tiles = [&](int bm, int bn) {
        int64_t t = 0;
        for (int o = 0; o < a->n_out; ++o) t += cdc_ceil_div(a->o[o].M, bm) * cdc_ceil_div(a->o[o].N, bn);
        return t;
    };
    int cfg = a->tile_cfg;
    if (cfg <= 0 || cfg > 10) {
        // tile shape: narrow outputs (towers, gates) take 64-wide column tiles; few tiles -> 64-row tiles so that the chip fills;
        // ring depth: deep (1 workgroup per CU) only when the K loop is long enough to use it
        const bool narrow = max_n <= 64;
        const int64_t t128 = tiles(128, narrow ? 64 : 128);
        const bool short_rows = t128 < 256;
        // measured on the C2 launches (tools/gemm2_probe.hip, profiles/round2): 128x128 with two workgroups per CU wins whenever
        // it fills the chip; under-filled launches take 64-row tiles, and 64x64 when the reduction is long (grad-input of the
        // first layer: 4096 x 416 outputs over K = 2068 — more, smaller tiles beat deeper rings)
        if (narrow) cfg = short_rows ? 9 : 7;
        else if (short_rows) cfg = max_kr >= 1024 ? 9 : 10;
        else cfg = 1;
    }
    hipStream_t st = (hipStream_t)stream;
#define G2_LAUNCH(BM_, BN_, NS_)                                                                                              \
    do {                                                                                                                      \
        typedef G2Cfg<BM_, BN_, NS_> C_;                                                                                      \
        static bool attr_done = false;                                                                                       \
        if (!attr_done) {                                                                                                     \
            (void)hipFuncSetAttribute((const void*)k_g2_nt<BM_, BN_, NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, C_::SMEM); \
            attr_done = true;                                                                                                 \
        }                                                                                                                     \
        const int64_t grid = tiles(BM_, BN_);                                                                                 \
        CDC_CHECK_ARG(grid < (1ll << 31), CDC_E_TOOBIG, "gemm_bf16_nt: grid too large");                                      \
        hipLaunchKernelGGL((k_g2_nt<BM_, BN_, NS_>), dim3((unsigned)grid), dim3(G2_THREADS), C_::SMEM, st, *a);               \
    } while (0)
    switch (cfg) {
        case 1: G2_LAUNCH(128, 128, 2); break;
        case 2: G2_LAUNCH(128, 128, 3); break;
        case 3: G2_LAUNCH(128, 128, 4); break;
        case 4: G2_LAUNCH(64, 128, 3); break;
        case 5: G2_LAUNCH(64, 128, 4); break;
        case 6: G2_LAUNCH(128, 64, 2); break;
        case 7: G2_LAUNCH(128, 64, 3); break;
        case 8: G2_LAUNCH(64, 64, 4); break;
        case 9: G2_LAUNCH(64, 64, 3); break;
        default: G2_LAUNCH(64, 128, 2); break;
    }
#undef G2_LAUNCH
    CDC_LAUNCH_CHECK("gemm_bf16_nt");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// grad-weight from the shadows: dW[N,K] = dZ^T . X, db[N] = column sums of dZ — the contraction runs over the BATCH rows, which
// are the slow index of both operands.  Both tiles are staged in their natural layout ([64 batch rows][128 (or 64) columns],
// direct-to-LDS, whole 256-byte (128-byte) rows) and the MFMA fragments — which need the reduction index contiguous — come out
// of the CDNA4 transposing LDS read ds_read_b64_tr_b16.  LDS image: 16-byte chunk index XORed with 2*(row&7) (256-byte rows;
// 2*((row>>1)&3) for 128-byte rows): every transposing read is bank-conflict free.  The k order inside one MFMA is a
// permutation of the 32 staged rows (lane group g, element j <-> row 4g+j | 16+4g+(j-4)), the same for both operands.
// db: one more MFMA per 16 output rows against an all-ones fragment (column 0 of the product = the column sums).
// split_k > 1: (tile, row slice) pairs write fp32 slabs that k_bwd_w_reduce (gemm.hip) adds in slice order.
// ---------------------------------------------------------------------------------------------------------------------
typedef short g2_s16x4 __attribute__((ext_vector_type(4)));

template <int TW> struct G2Tn {                                      // one operand tile of TW columns x 64 batch rows
    static constexpr int ROW_BYTES = TW * 2;
    static constexpr int ROWS_PER_INSTR = 1024 / ROW_BYTES;          // 4 (TW 128) or 8 (TW 64)
    static constexpr int CHUNKS = ROW_BYTES / 16;                    // 16 or 8
    static constexpr int INSTR = 64 / ROWS_PER_INSTR;                // per slab
    static constexpr int PER_WAVE = INSTR / 4;
    static constexpr int BYTES = 64 * ROW_BYTES;
    __device__ static __forceinline__ int swz(int row) { return TW == 128 ? 2 * (row & 7) : 2 * ((row >> 1) & 3); }
};

template <int TW>
__device__ __forceinline__ bf16x8_t g2_tr_fragment(const unsigned char* tile, int row_base, int col0, int lane) {
    typedef G2Tn<TW> T;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row = row_base + 4 * g + q;
    const int c = (col0 + 4 * p) >> 3;
    const unsigned char* a0 = tile + row * T::ROW_BYTES + ((c ^ T::swz(row)) << 4) + ((p & 1) << 3);
    const unsigned char* a1 = a0 + 16 * T::ROW_BYTES;               // row + 16: the same (row & 7), the same swizzle
    typedef g2_s16x4 __attribute__((address_space(3))) * lds_ptr;
    const g2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a0);
    const g2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a1);
    union { g2_s16x4 h[2]; bf16x8_t v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}

// (the body is a device function so that one launch can run two tile classes: k_g2_tn_dual below; `a` may live in the kernel
// argument block or in device memory)
template <int BMO, int BNO, int NSTAGE>
__device__ __forceinline__ void g2_tn_body(const cdc_lin_bwdw_args& a, const int64_t slab_stride, const int bid, const int nblk,
                                           unsigned char* const smem) {
    typedef G2Tn<BMO> TA;
    typedef G2Tn<BNO> TB;
    constexpr int STAGE = TA::BYTES + TB::BYTES;
    constexpr int MT = BMO / 32, NT = BNO / 32;
    constexpr int LOADS = TA::PER_WAVE + TB::PER_WAVE;
    const int S = a.split_k > 1 ? a.split_k : 1;
    const int id = g2_xcd_remap(bid, nblk);
    const int split = id % S;
    int tile = id / S;
    int64_t g_off = 0;
    const int g = find_group<true>(a.n_groups, tile, [&](int l) { return ((a.g[l].N + BMO - 1) / BMO) * ((a.g[l].K + BNO - 1) / BNO); },
                                   [&](int l) { return (int64_t)a.g[l].N * a.g[l].K + a.g[l].N; }, tile, &g_off);
    if (g < 0) return;
    const cdc_bwdw_group& G = a.g[g];
    const int tn_cnt = (G.K + BNO - 1) / BNO;
    const int M = G.M;
    int chunk = ((M + S - 1) / S + 63) / 64 * 64;
    if (chunk < 64) chunk = 64;
    const int r0 = split * chunk;
    int rn = M - r0;
    if (rn > chunk) rn = chunk;
    if (rn < 0) rn = 0;
    const int total = (rn + 63) / 64;                                // slabs of 64 batch rows (rows past M are the shadows' zero padding)
    const int i0 = (tile / tn_cnt) * BMO, j0 = (tile % tn_cnt) * BNO;   // i over N (dW rows), j over K (dW columns)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (BMO / 2), wn = (wave & 1) * (BNO / 2);

    const __bf16* pa[TA::PER_WAVE];
    const __bf16* pb[TB::PER_WAVE];
    {
        const __bf16* dz = reinterpret_cast<const __bf16*>(G.dzh) + (int64_t)r0 * G.lddzh + i0;
        const __bf16* x = reinterpret_cast<const __bf16*>(G.xh) + (int64_t)r0 * G.ldxh + j0;
#pragma unroll
        for (int q = 0; q < TA::PER_WAVE; ++q) {
            const int row = (wave * TA::PER_WAVE + q) * TA::ROWS_PER_INSTR + lane / TA::CHUNKS;
            const int c = (lane % TA::CHUNKS) ^ TA::swz(row);
            pa[q] = dz + (int64_t)row * G.lddzh + c * 8;
        }
#pragma unroll
        for (int q = 0; q < TB::PER_WAVE; ++q) {
            const int row = (wave * TB::PER_WAVE + q) * TB::ROWS_PER_INSTR + lane / TB::CHUNKS;
            const int c = (lane % TB::CHUNKS) ^ TB::swz(row);
            pb[q] = x + (int64_t)row * G.ldxh + c * 8;
        }
    }
    const int64_t adv_a = 64 * G.lddzh, adv_b = 64 * G.ldxh;
    int issued = 0, istage = 0;
    auto issue_next = [&]() {
        unsigned char* as = smem + istage * STAGE + wave * TA::PER_WAVE * 1024;
        unsigned char* bs = smem + istage * STAGE + TA::BYTES + wave * TB::PER_WAVE * 1024;
#pragma unroll
        for (int q = 0; q < TA::PER_WAVE; ++q) { glds16(pa[q], as + q * 1024); pa[q] += adv_a; }
#pragma unroll
        for (int q = 0; q < TB::PER_WAVE; ++q) { glds16(pb[q], bs + q * 1024); pb[q] += adv_b; }
        ++issued;
        istage = istage + 1 == NSTAGE ? 0 : istage + 1;
    };
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t dbacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) dbacc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool want_db = (G.db != nullptr) && (j0 == 0) && (wn == 0);      // wave-uniform
    bf16x8_t ones;
#pragma unroll
    for (int q = 0; q < 8; ++q) ones[q] = (__bf16)1.0f;

#pragma unroll
    for (int p = 0; p < NSTAGE - 1; ++p)
        if (issued < total) issue_next();
    int stage = 0;
    for (int t = 0; t < total; ++t) {
        if (NSTAGE > 2 && total - t - 1 >= NSTAGE - 2) g2_wait_vmcnt<LOADS * (NSTAGE - 2)>();
        else g2_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (issued < total) issue_next();
        const unsigned char* As = smem + stage * STAGE;
        const unsigned char* Bs = As + TA::BYTES;
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t fa[MT], fb[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = g2_tr_fragment<BMO>(As, ks * 32, wm + mt * 16, lane);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[nt] = g2_tr_fragment<BNO>(Bs, ks * 32, wn + nt * 16, lane);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
            if (want_db) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) dbacc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], ones, dbacc[mt], 0, 0, 0);
            }
        }
    }
    float* slab = S > 1 ? a.workspace + (int64_t)split * slab_stride + g_off : nullptr;
    if (want_db && (lane & 15) == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = i0 + wm + mt * 16 + (lane >> 4) * 4 + r;
                if (n >= G.N) continue;
                if (slab) slab[(int64_t)G.N * G.K + n] = dbacc[mt][r];
                else G.db[n] = G.accumulate ? G.db[n] + dbacc[mt][r] : dbacc[mt][r];
            }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = j0 + wn + nt * 16 + (lane & 15);
            if (col >= G.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wm + mt * 16 + (lane >> 4) * 4 + r;
                if (row >= G.N) continue;
                const float val = acc[mt][nt][r];
                if (slab) slab[(int64_t)row * G.K + col] = val;
                else {
                    float* dst = G.dw + (int64_t)row * G.lddw + col;
                    *dst = G.accumulate ? *dst + val : val;
                }
            }
        }
}

// called by cdc_glinear_bwd_w (gemm.hip) when every group of a CDC_PREC_BF16 launch carries its shadows
template <int BMO, int BNO, int NSTAGE>
__global__ void __launch_bounds__(G2_THREADS, 2) k_g2_tn(const cdc_lin_bwdw_args a, int64_t slab_stride) {
    CDC_PRIO_MAIN();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    g2_tn_body<BMO, BNO, NSTAGE>(a, slab_stride, blockIdx.x, gridDim.x, smem);
}
// the wide class (128 x 128 tiles) and the narrow class (64 x 64 tiles) of a step's batched grad-weight contractions in ONE launch:
// workgroups [0, n_wide) run the first body on tabs[0], the rest the second on tabs[1] (argument blocks in device memory: two of
// them do not fit the 4 KB kernel-argument block).  Nothing of the two classes depends on the other; as two launches the second
// one's 14 us sat behind the first one's tail.
__global__ void __launch_bounds__(G2_THREADS, 2) k_g2_tn_dual(const cdc_lin_bwdw_args* __restrict__ tabs, int64_t slab_wide, int64_t slab_narrow,
                                                              int n_wide) {
    CDC_PRIO_MAIN();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x < n_wide) g2_tn_body<128, 128, 2>(tabs[0], slab_wide, blockIdx.x, n_wide, smem);
    else g2_tn_body<64, 64, 3>(tabs[1], slab_narrow, (int)blockIdx.x - n_wide, (int)gridDim.x - n_wide, smem);
}

int g2_launch_bwd_w(const cdc_lin_bwdw_args* a, int64_t slab_stride, hipStream_t st) {
    int max_n = 0, max_k = 0;
    for (int g = 0; g < a->n_groups; ++g) {
        const cdc_bwdw_group& G = a->g[g];
        CDC_CHECK_ARG(((((uintptr_t)G.dzh) | ((uintptr_t)G.xh)) & 15) == 0 && G.lddzh % 8 == 0 && G.ldxh % 8 == 0, CDC_E_ALIGN,
                      "glinear_bwd_w: group %d: shadows must be 16-byte aligned with row strides that are multiples of 8 elements", g);
        if (G.N > max_n) max_n = G.N;
        if (G.K > max_k) max_k = G.K;
    }
    const int S = a->split_k > 1 ? a->split_k : 1;
    // 64x64 output tiles when no group has more than 64 output rows (gates, narrow layers: a 128-row tile would be mostly
    // padding), whatever their K; the caller keeps such groups in launches of their own (plan.py: _emit_deferred_dw)
    const bool small = max_n <= 64;
    (void)max_k;
    int64_t t = 0;
    for (int g = 0; g < a->n_groups; ++g)
        t += small ? cdc_ceil_div(a->g[g].N, 64) * cdc_ceil_div(a->g[g].K, 64) : cdc_ceil_div(a->g[g].N, 128) * cdc_ceil_div(a->g[g].K, 128);
    const int64_t grid = t * S;
    CDC_CHECK_ARG(grid < (1ll << 31), CDC_E_TOOBIG, "glinear_bwd_w: grid too large");
    if (grid == 0) return 0;
    if (small) hipLaunchKernelGGL((k_g2_tn<64, 64, 3>), dim3((unsigned)grid), dim3(G2_THREADS), 3 * (G2Tn<64>::BYTES * 2), st, *a, slab_stride);
    else       hipLaunchKernelGGL((k_g2_tn<128, 128, 2>), dim3((unsigned)grid), dim3(G2_THREADS), 2 * (G2Tn<128>::BYTES * 2), st, *a, slab_stride);
    CDC_LAUNCH_CHECK("glinear_bwd_w(shadows)");
    return 0;
}
static int64_t g2_tn_tiles(const cdc_lin_bwdw_args* a, int T) {
    int64_t t = 0;
    for (int g = 0; g < a->n_groups; ++g) t += cdc_ceil_div(a->g[g].N, T) * cdc_ceil_div(a->g[g].K, T);
    return t;
}
int g2_launch_bwd_w_dual(const cdc_lin_bwdw_args* wide, const cdc_lin_bwdw_args* narrow, const cdc_lin_bwdw_args* tabs_dev, int64_t slab_wide,
                         int64_t slab_narrow, hipStream_t st) {
    for (int w = 0; w < 2; ++w) {
        const cdc_lin_bwdw_args* a = w ? narrow : wide;
        for (int g = 0; g < a->n_groups; ++g) {
            const cdc_bwdw_group& G = a->g[g];
            CDC_CHECK_ARG(G.dzh && G.xh && ((((uintptr_t)G.dzh) | ((uintptr_t)G.xh)) & 15) == 0 && G.lddzh % 8 == 0 && G.ldxh % 8 == 0, CDC_E_ALIGN,
                          "glinear_bwd_w_pair: group %d of the %s class: bf16 shadows, 16-byte aligned, row strides multiples of 8", g, w ? "narrow" : "wide");
            CDC_CHECK_ARG(w ? G.N <= 64 : true, CDC_E_BADARG, "glinear_bwd_w_pair: group %d of the narrow class has %d output rows", g, G.N);
        }
    }
    const int64_t nw = g2_tn_tiles(wide, 128) * (wide->split_k > 1 ? wide->split_k : 1);
    const int64_t nn = g2_tn_tiles(narrow, 64) * (narrow->split_k > 1 ? narrow->split_k : 1);
    CDC_CHECK_ARG(nw + nn < (1ll << 31) && nw > 0 && nn > 0, CDC_E_TOOBIG, "glinear_bwd_w_pair: grid out of range");
    constexpr int LDS_W = 2 * (G2Tn<128>::BYTES * 2), LDS_N = 3 * (G2Tn<64>::BYTES * 2);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_g2_tn_dual, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_W > LDS_N ? LDS_W : LDS_N);
        attr_done = true;
    }
    hipLaunchKernelGGL(k_g2_tn_dual, dim3((unsigned)(nw + nn)), dim3(G2_THREADS), LDS_W > LDS_N ? LDS_W : LDS_N, st, tabs_dev, slab_wide, slab_narrow,
                       (int)nw);
    CDC_LAUNCH_CHECK("glinear_bwd_w_pair");
    return 0;
}
bool g2_bwd_w_uses_small_tiles(const cdc_lin_bwdw_args* a) {
    for (int g = 0; g < a->n_groups; ++g)
        if (a->g[g].N > 64) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// shadows
// ---------------------------------------------------------------------------------------------------------------------
// weights: for every tensor W [rows = N, cols = K] (nn.Linear layout, contiguous): the straight bf16 copy [N, ld_h] (forward's
// B operand) and the transposed bf16 copy [K, ld_t] (grad-input's B operand) in one pass over W, 32x32 tiles through LDS.
// Columns / rows of the destinations beyond the tensor are never written (allocated zero: the K / N padding of the slabs).
__global__ void __launch_bounds__(256) k_weight_shadows(const cdc_wshadow_args a) {
    __shared__ float tile[32][33];
    int blk = blockIdx.x;
    const int ti = find_group<false>(a.n, blk, [&](int l) { return ((a.t[l].rows + 31) / 32) * ((a.t[l].cols + 31) / 32); },
                                     [](int) { return (int64_t)0; }, blk, nullptr);
    if (ti < 0) return;
    const int rows = a.t[ti].rows, cols = a.t[ti].cols;
    const int tc = (cols + 31) / 32;
    const int r0 = (blk / tc) * 32, c0 = (blk % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const float* src = a.t[ti].src;
    __bf16* dh = reinterpret_cast<__bf16*>(a.t[ti].dst_h);
    __bf16* dt = reinterpret_cast<__bf16*>(a.t[ti].dst_t);
    const int64_t ld_h = a.t[ti].ld_h, ld_t = a.t[ti].ld_t;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        const float v = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
        tile[k][tx] = v;
        if (dh && r < rows && c < cols) dh[(int64_t)r * ld_h + c] = (__bf16)v;
    }
    if (!dt) return;
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < rows) dt[(int64_t)c * ld_t + r] = (__bf16)tile[tx][k];
    }
}
extern "C" int cdc_weight_shadows(const cdc_wshadow_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n > 0 && a->n <= CDC_MAX_TENSORS, CDC_E_BADARG, "weight_shadows: bad count");
    int64_t blocks = 0;
    for (int i = 0; i < a->n; ++i) {
        CDC_CHECK_ARG(a->t[i].src && (a->t[i].dst_h || a->t[i].dst_t) && a->t[i].rows > 0 && a->t[i].cols > 0, CDC_E_BADARG,
                      "weight_shadows: tensor %d malformed", i);
        CDC_CHECK_ARG((!a->t[i].dst_h || a->t[i].ld_h >= a->t[i].cols) && (!a->t[i].dst_t || a->t[i].ld_t >= a->t[i].rows), CDC_E_BADARG,
                      "weight_shadows: tensor %d: destination row stride too small", i);
        blocks += cdc_ceil_div(a->t[i].rows, 32) * cdc_ceil_div(a->t[i].cols, 32);
    }
    hipLaunchKernelGGL(k_weight_shadows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("weight_shadows");
    return 0;
}

// activations / gradients whose producer does not write its shadow itself: dst[r, c] = bf16(src[r, c]) for up to
// CDC_MAX_GROUPS [rows, cols] views in one launch (8 columns per thread when everything is aligned)
__global__ void __launch_bounds__(256) k_shadow_bf16(const cdc_shadow_args a) {
    int blk = blockIdx.x;
    const int ti = find_group<false>(a.n, blk, [&](int l) { return (int)(((int64_t)a.t[l].rows * ((a.t[l].cols + 7) / 8) + 255) / 256); },
                                     [](int) { return (int64_t)0; }, blk, nullptr);
    if (ti < 0) return;
    const int cols = a.t[ti].cols;
    const int64_t rows = a.t[ti].rows;
    const int c8n = (cols + 7) / 8;
    const int64_t i = (int64_t)blk * 256 + threadIdx.x;
    if (i >= rows * c8n) return;
    const int64_t r = i / c8n;
    const int c = (int)(i % c8n) * 8;
    const float* s = a.t[ti].src + r * a.t[ti].ld_src + c;
    __bf16* d = reinterpret_cast<__bf16*>(a.t[ti].dst) + r * a.t[ti].ld_dst + c;
    const bool vec = c + 7 < cols && ((((uintptr_t)s) & 15) == 0) && ((((uintptr_t)d) & 15) == 0);
    if (vec) {
        const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(s), hi = *reinterpret_cast<const f32x4_t*>(s + 4);
        bf16x8_t h;
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = (__bf16)lo[q]; h[4 + q] = (__bf16)hi[q]; }
        *reinterpret_cast<bf16x8_t*>(d) = h;
    } else {
        for (int q = 0; q < 8 && c + q < cols; ++q) d[q] = (__bf16)s[q];
    }
}
extern "C" int cdc_shadow_bf16(const cdc_shadow_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n > 0 && a->n <= CDC_MAX_GROUPS, CDC_E_BADARG, "shadow_bf16: bad count");
    int64_t blocks = 0;
    for (int i = 0; i < a->n; ++i) {
        CDC_CHECK_ARG(a->t[i].src && a->t[i].dst && a->t[i].rows >= 0 && a->t[i].cols > 0 && a->t[i].ld_src >= a->t[i].cols &&
                          a->t[i].ld_dst >= a->t[i].cols, CDC_E_BADARG, "shadow_bf16: view %d malformed", i);
        blocks += cdc_ceil_div((int64_t)a->t[i].rows * cdc_ceil_div(a->t[i].cols, 8), 256);
    }
    if (blocks == 0) return 0;
    CDC_CHECK_ARG(blocks < (1ll << 31), CDC_E_TOOBIG, "shadow_bf16: grid too large");
    hipLaunchKernelGGL(k_shadow_bf16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("shadow_bf16");
    return 0;
}
