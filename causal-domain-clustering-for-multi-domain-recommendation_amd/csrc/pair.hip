// pair.hip — two consecutive BatchNorm-free expert layers as ONE forward launch (include/cdcmdr.h: cdc_expert_pair_fwd).
//
// Reference: MultiLayerPerceptron(input_dims, (H1, H2), dropout, output_layer=False, bn=False) of every PLE expert
// (model/ple.py:83-88, model/layer.py:185-196): Linear -> ReLU -> Dropout -> Linear -> ReLU -> Dropout, n_expert of them side by
// side on the same input, plus the level's gate projections of that input (model/ple.py:89-94).  As two grouped launches
// (gemm2.hip) the [B, n_expert*H1] hidden activation is written by one launch and read straight back by the next, whose K = H1
// = 256 gives it four slabs per tile: prologue, epilogue and launch ramp are most of its time.  Here a workgroup owns 128 batch
// rows of ONE expert and walks both layers:
//   phase 1  [128 x H1] = x[128 x K] . W1^T over K in 64-wide slabs, operands global -> LDS directly (2-slab ring), 8 waves as
//            2 x 4 wave tiles of 64 x 64; a side output of <= 16 columns (a gate's logits: same x, own weight rows) rides along as
//            one more MFMA tile per wave;
//   between  W2 (H2 x H1 bf16 = 64 KB) is fetched whole into LDS while the accumulators get bias / ReLU / dropout in REGISTERS
//            (a lane pair swaps half its rows by DPP so that each lane owns both columns of a dropout pair: one hash per two
//            elements, packed bf16x2 LDS writes) and land as the bf16 A-operand image of phase 2 — the same rounding the shadow
//            of the unfused launch gets; the tile goes out to memory once, from LDS, in whole rows, for the backward launches;
//   phase 2  [128 x H2] = h . W2^T, every operand already in LDS: four slabs back to back, no barrier; epilogue as gemm2's.
// Arithmetic, accumulation order and dropout streams are those of the two gemm2 launches it replaces: results are bit-identical
// (tests/test_gpu_pair.py).  LDS 128 KB, one workgroup per CU; blocks are numbered so that the 8 experts of a row block share an
// XCD (its L2 holds 4 row blocks of x and all of W1).
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define PAIR_THREADS 512
#define PAIR_BM 128
#define PAIR_BK 64
#define PAIR_ROWB (PAIR_BK * 2)          /* one LDS row of a slab image: 64 bf16 = 128 B = eight 16-byte chunks */
#define PAIR_SIDE 16                     /* columns of the side output's MFMA tile */
#define PAIR_KARG __attribute__((address_space(4)))
#ifndef PAIR_NSTAGE
#define PAIR_NSTAGE 2                    /* LDS ring depth of phase 1 (tools/pair_probe.hip: 3 stages measured no faster: 23.4 vs 22.9 us) */
#endif
#ifndef PAIR_PROBE
#define PAIR_PROBE 0                     /* tools/pair_probe.hip: 1 no dropout hash in epilogue 1, 2 no store of h, 4 no phase 2, 8 no epilogue 2,
                                            16 no phase 1, 32 no epilogue 1 */
#endif

#ifndef PAIR_TRACE
#define PAIR_TRACE 0                     /* probe builds only (tools/pair_trace.py): thread 0 of workgroup 0 stamps the 100 MHz wall clock at phase boundaries */
#endif
#if PAIR_TRACE
__device__ unsigned long long g_pair_stamps[16];
#define PAIR_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_pair_stamps[(i)] = wall_clock64(); } while (0)
extern "C" int cdc_debug_pair_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_stamps), sizeof(unsigned long long) * 16); }
#else
#define PAIR_STAMP(i) do { } while (0)
#endif

template <int H1, int H2>
struct PairCfg {
    static constexpr int A_BYTES = PAIR_BM * PAIR_ROWB;
    static constexpr int B_BYTES = H1 * PAIR_ROWB;
    static constexpr int S_BYTES = PAIR_SIDE * PAIR_ROWB;
    static constexpr int STAGE = A_BYTES + B_BYTES + S_BYTES;
    static constexpr int KS2 = H1 / PAIR_BK;                        // K slabs of phase 2
    static constexpr int H_BYTES = KS2 * A_BYTES;                   // hidden tile: KS2 slab images of [128 rows][128 B]
    static constexpr int W2_SLAB = H2 * PAIR_ROWB;
    static constexpr int W2_BYTES = KS2 * W2_SLAB;
    static constexpr int CS = H2 + 4;                               // epilogue-2 tile row stride (floats)
    static constexpr int OUT_BYTES = PAIR_BM * CS * 4;
    static constexpr int SMEM_A = PAIR_NSTAGE * STAGE > H_BYTES + W2_BYTES ? PAIR_NSTAGE * STAGE : H_BYTES + W2_BYTES;
    static constexpr int SMEM = SMEM_A > OUT_BYTES ? SMEM_A : OUT_BYTES;
    static constexpr int MT1 = 4, NT1 = H1 / 64;                    // phase 1: waves 2 (rows) x 4 (columns), wave tile 64 x H1/4
    static constexpr int MT2 = 2, NT2 = H2 / 32;                    // phase 2: waves 4 x 2, wave tile 32 x H2/2
    static_assert(H1 == 256 && H2 == 128, "wave tilings are laid out for (256, 128) (config.py:39-42)");
};

__device__ __forceinline__ void pair_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int pair_xcd_remap(int bid, int nblk) {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}
__device__ __forceinline__ float pair_swap1(float v) {              // the value of lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <int H1, int H2>
__global__ void __launch_bounds__(PAIR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) k_pair_fwd(const cdc_expert_pair_args a_by_value) {
    CDC_PRIO_MAIN();
    (void)a_by_value;
    PAIR_STAMP(0);
    const PAIR_KARG cdc_expert_pair_args& a = *(const PAIR_KARG cdc_expert_pair_args*)__builtin_amdgcn_kernarg_segment_ptr();
    typedef PairCfg<H1, H2> Cfg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ne = a.n_expert, M = a.M;
    const int tile = pair_xcd_remap(blockIdx.x, gridDim.x);
    const int e = __builtin_amdgcn_readfirstlane(tile % ne);
    const int i0 = (tile / ne) * PAIR_BM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // descriptor fields -> locals (reads through the kernarg pointer are not treated as invariant across stores)
    const PAIR_KARG cdc_pair_expert& E = a.e[e];
    const __bf16* const X = reinterpret_cast<const __bf16*>(E.x);
    const __bf16* const W1 = reinterpret_cast<const __bf16*>(E.w1);
    const __bf16* const W2 = reinterpret_cast<const __bf16*>(E.w2);
    const __bf16* const WS = reinterpret_cast<const __bf16*>(E.ws);
    const int64_t ldx = E.ldx, ldw1 = E.ldw1, ldw2 = E.ldw2, ldws = E.ldws;
    const float* const b1 = E.b1;
    const float* const b2 = E.b2;
    const float* const bs = E.bs;
    __bf16* const Hout = reinterpret_cast<__bf16*>(E.h);
    float* const Y = E.y;
    __bf16* const Yh = reinterpret_cast<__bf16*>(E.yh);
    float* const YS = E.ys;
    const int64_t ldh = E.ldh, ldy = E.ldy, ldyh = E.ldyh, ldys = E.ldys;
    const int ns = WS ? E.ns : 0;
    const int stream1 = E.stream1, stream2 = E.stream2;
    const int nk = a.K1r / PAIR_BK;
    const bool relu = a.relu != 0;
    const float drop_p = a.drop_p;
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t thr16 = (uint32_t)(drop_p * 65536.f + 0.5f);
    const uint32_t seed32_1 = drop_p > 0.f ? g2_seed32(a.seed1, a.seed_offset_dev, stream1) : 0u;
    const uint32_t seed32_2 = drop_p > 0.f ? g2_seed32(a.seed2, a.seed_offset_dev, stream2) : 0u;

    // direct loads: one wave instruction = 8 rows x 128 B; lane l lands on (row l>>3, physical chunk l&7) and fetches the logical
    // chunk (l&7) ^ (row&7) of its row; fragment reads undo the XOR (gemm2.hip)
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const int frow = lane & 15;
    const int fx = lane & 7;

    // ---- phase 1 --------------------------------------------------------------------------------------------------------
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * (H1 / 4);
    f32x4_t acc[Cfg::MT1][Cfg::NT1];
#pragma unroll
    for (int mt = 0; mt < Cfg::MT1; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NT1; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t accs = {0.f, 0.f, 0.f, 0.f};

    const __bf16* pa[2];
    const __bf16* pb[H1 / 64];
    const __bf16* ps;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int r = i0 + wave * 16 + q * 8 + lrow;
        r = r < M ? r : M - 1;                                       // rows past the extent only feed accumulators nobody stores
        pa[q] = X + (int64_t)r * ldx + lchunk * 8;
    }
#pragma unroll
    for (int q = 0; q < H1 / 64; ++q) pb[q] = W1 + (int64_t)(wave * (H1 / 8) + q * 8 + lrow) * ldw1 + lchunk * 8;
    {
        int r = (wave & 1) * 8 + lrow;
        r = r < ns ? r : (ns > 0 ? ns - 1 : 0);
        ps = ns > 0 ? WS + (int64_t)r * ldws + lchunk * 8 : X;
    }
    const bool side_loader = ns > 0 && wave < 2;                    // uniform
    auto issue = [&](int stage) {
        unsigned char* base = smem + stage * Cfg::STAGE;
        unsigned char* as = base + wave * 16 * PAIR_ROWB;
        unsigned char* bsm = base + Cfg::A_BYTES + wave * (H1 / 8) * PAIR_ROWB;
#pragma unroll
        for (int q = 0; q < 2; ++q) { pair_glds16(pa[q], as + q * 8 * PAIR_ROWB); pa[q] += PAIR_BK; }
#pragma unroll
        for (int q = 0; q < H1 / 64; ++q) { pair_glds16(pb[q], bsm + q * 8 * PAIR_ROWB); pb[q] += PAIR_BK; }
        if (side_loader) { pair_glds16(ps, base + Cfg::A_BYTES + Cfg::B_BYTES + wave * 8 * PAIR_ROWB); ps += PAIR_BK; }
    };
    // bias of this lane's columns (both columns of its dropout pair), one pair per column tile; side bias
    const int colpair = (lane & 15) & ~1;
    float bias_lo[Cfg::NT1], bias_hi[Cfg::NT1];
#pragma unroll
    for (int nt = 0; nt < Cfg::NT1; ++nt) {
        bias_lo[nt] = b1 ? b1[wn + nt * 16 + colpair] : 0.f;
        bias_hi[nt] = b1 ? b1[wn + nt * 16 + colpair + 1] : 0.f;
    }
    const float bias_s = (ns > 0 && bs && (lane & 15) < ns) ? bs[lane & 15] : 0.f;

    // ring of PAIR_NSTAGE slabs, PAIR_NSTAGE - 1 of them in flight: slab t is waited for with a COUNTED vmcnt (6 direct loads per wave
    // and slab, 7 for the two waves that fetch the side rows), the slab freed by the barrier is refilled right after it
    const int nk1 = (PAIR_PROBE & 16) ? 0 : nk;
    PAIR_STAMP(1);
#pragma unroll
    for (int p = 0; p < PAIR_NSTAGE - 1; ++p)
        if (p < nk1) issue(p);
    int stage = 0, fill = PAIR_NSTAGE - 1;
    for (int t = 0; t < nk1; ++t) {
        if (PAIR_NSTAGE > 2 && nk1 - 1 - t >= PAIR_NSTAGE - 2) {
            if (side_loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * (PAIR_NSTAGE - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (PAIR_NSTAGE - 2)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                // slab t has landed everywhere; slab t-1's reads are over
        if (t + PAIR_NSTAGE - 1 < nk1) {
            issue(fill);
            fill = fill + 1 == PAIR_NSTAGE ? 0 : fill + 1;
        }
        const unsigned char* As = smem + stage * Cfg::STAGE;
        const unsigned char* Bs = As + Cfg::A_BYTES;
        const unsigned char* Ss = Bs + Cfg::B_BYTES;
        stage = stage + 1 == PAIR_NSTAGE ? 0 : stage + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + (lane >> 4)) ^ fx) << 4;
            bf16x8_t af[Cfg::MT1], bfr[Cfg::NT1];
#pragma unroll
            for (int mt = 0; mt < Cfg::MT1; ++mt) af[mt] = *reinterpret_cast<const bf16x8_t*>(As + (wm + mt * 16 + frow) * PAIR_ROWB + coff);
#pragma unroll
            for (int nt = 0; nt < Cfg::NT1; ++nt) bfr[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn + nt * 16 + frow) * PAIR_ROWB + coff);
            if (ns > 0) {                                            // side tile: rows 16*wave .. +15 of the block
                const bf16x8_t sa = *reinterpret_cast<const bf16x8_t*>(As + (wave * 16 + frow) * PAIR_ROWB + coff);
                const bf16x8_t sf = *reinterpret_cast<const bf16x8_t*>(Ss + frow * PAIR_ROWB + coff);
                accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, sf, accs, 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < Cfg::MT1; ++mt)
#pragma unroll
                for (int nt = 0; nt < Cfg::NT1; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    PAIR_STAMP(2);
    __syncthreads();                                                 // the ring is free
    PAIR_STAMP(3);

    // ---- W2 whole into LDS behind the hidden tile's image: 8 direct loads per wave, in flight under the register epilogue ----
    unsigned char* const Hs = smem;
    unsigned char* const W2s = smem + Cfg::H_BYTES;
    {
        constexpr int PER_WAVE = Cfg::KS2 * (H2 / 8) / 8;            // wave instructions (8 rows each) per wave
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) {
            const int blk = wave * PER_WAVE + q;                     // 8-row block index over (slab, rows)
            const int s = blk / (H2 / 8), r = (blk % (H2 / 8)) * 8 + lrow;
            pair_glds16(W2 + (int64_t)r * ldw2 + s * PAIR_BK + lchunk * 8, W2s + s * Cfg::W2_SLAB + (blk % (H2 / 8)) * 8 * PAIR_ROWB);
        }
    }
    // side output: logits = acc + bias, no activation (the gates' softmax is the pooling launch's)
    if (ns > 0 && YS && (lane & 15) < ns) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + wave * 16 + (lane >> 4) * 4 + r;
            if (row < M) YS[(int64_t)row * ldys + (lane & 15)] = accs[r] + bias_s;
        }
    }

    // ---- epilogue 1 in registers: bias, ReLU, dropout, bf16 -> the A-operand image of phase 2 --------------------------------
    {
        const bool odd = lane & 1;
        const int rsub = (lane >> 4) * 4 + (odd ? 2 : 0);            // this lane's two rows inside a 16-row tile after the swap
        unsigned char* const hslab = Hs + (wave & 3) * Cfg::A_BYTES; // columns wn .. wn+63 are K slab (wave & 3) of phase 2
        const bool drop = drop_p > 0.f && !(PAIR_PROBE & 1);
#pragma unroll
        for (int mt = 0; mt < (PAIR_PROBE & 32 ? 0 : Cfg::MT1); ++mt) {
            const int r_in = wm + mt * 16 + rsub;                    // row inside the block (and r_in + 1)
            const uint32_t rkey0 = seed32_1 + (uint32_t)(i0 + r_in) * 0x9E3779B1U;
#pragma unroll
            for (int nt = 0; nt < Cfg::NT1; ++nt) {
                const f32x4_t c = acc[mt][nt];
                // even lane keeps rows 0,1 and takes the neighbour's (column + 1); odd lane keeps rows 2,3 and takes column - 1
                const float r0 = pair_swap1(odd ? c[0] : c[2]);
                const float r1 = pair_swap1(odd ? c[1] : c[3]);
                float v0lo = odd ? r0 : c[0], v0hi = odd ? c[2] : r0;
                float v1lo = odd ? r1 : c[1], v1hi = odd ? c[3] : r1;
                v0lo += bias_lo[nt]; v0hi += bias_hi[nt]; v1lo += bias_lo[nt]; v1hi += bias_hi[nt];
                if (relu) { v0lo = fmaxf(v0lo, 0.f); v0hi = fmaxf(v0hi, 0.f); v1lo = fmaxf(v1lo, 0.f); v1hi = fmaxf(v1hi, 0.f); }
                const int cc = nt * 16 + colpair;                    // column inside the slab (even)
                if (drop) {
                    const uint32_t ck = (uint32_t)((wn + cc) >> 1) * 0x85EBCA77U;
                    const uint32_t h0 = g2_hash32(rkey0 + ck);
                    const uint32_t h1 = g2_hash32(rkey0 + 0x9E3779B1U + ck);
                    v0lo = (h0 & 0xFFFFu) < thr16 ? 0.f : v0lo * keep_scale;
                    v0hi = (h0 >> 16) < thr16 ? 0.f : v0hi * keep_scale;
                    v1lo = (h1 & 0xFFFFu) < thr16 ? 0.f : v1lo * keep_scale;
                    v1hi = (h1 >> 16) < thr16 ? 0.f : v1hi * keep_scale;
                }
                bf16x2_t p0, p1;
                p0[0] = (__bf16)v0lo; p0[1] = (__bf16)v0hi; p1[0] = (__bf16)v1lo; p1[1] = (__bf16)v1hi;
                const int boff = (cc & 7) * 2;
                *reinterpret_cast<bf16x2_t*>(hslab + r_in * PAIR_ROWB + ((((cc >> 3)) ^ (r_in & 7)) << 4) + boff) = p0;
                *reinterpret_cast<bf16x2_t*>(hslab + (r_in + 1) * PAIR_ROWB + ((((cc >> 3)) ^ ((r_in + 1) & 7)) << 4) + boff) = p1;
            }
        }
    }
    PAIR_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's share of W2
    __syncthreads();
    PAIR_STAMP(5);

    // ---- the hidden tile out to memory, whole rows from LDS (read by grad-weight and as the activation mask of grad-input) ----
    if (Hout && !(PAIR_PROBE & 2)) {
        constexpr int CH = H1 / 8;                                   // 16-byte chunks per row
#pragma unroll
        for (int p = 0; p < PAIR_BM * CH / PAIR_THREADS; ++p) {
            const int idx = p * PAIR_THREADS + tid;
            const int row = idx / CH, c16 = idx % CH;
            if (i0 + row < M) {
                const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(Hs + (c16 >> 3) * Cfg::A_BYTES + row * PAIR_ROWB + (((c16 & 7) ^ (row & 7)) << 4));
                *reinterpret_cast<bf16x8_t*>(Hout + (int64_t)(i0 + row) * ldh + c16 * 8) = v;
            }
        }
    }

    PAIR_STAMP(6);
    // ---- phase 2: every operand in LDS -----------------------------------------------------------------------------------
    const int wm2 = (wave >> 1) * 32, wn2 = (wave & 1) * (H2 / 2);
    f32x4_t acc2[Cfg::MT2][Cfg::NT2];
#pragma unroll
    for (int mt = 0; mt < Cfg::MT2; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NT2; ++nt) acc2[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < (PAIR_PROBE & 4 ? 0 : Cfg::KS2); ++s) {
        const unsigned char* As = Hs + s * Cfg::A_BYTES;
        const unsigned char* Bs = W2s + s * Cfg::W2_SLAB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + (lane >> 4)) ^ fx) << 4;
            bf16x8_t af[Cfg::MT2], bfr[Cfg::NT2];
#pragma unroll
            for (int mt = 0; mt < Cfg::MT2; ++mt) af[mt] = *reinterpret_cast<const bf16x8_t*>(As + (wm2 + mt * 16 + frow) * PAIR_ROWB + coff);
#pragma unroll
            for (int nt = 0; nt < Cfg::NT2; ++nt) bfr[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn2 + nt * 16 + frow) * PAIR_ROWB + coff);
#pragma unroll
            for (int mt = 0; mt < Cfg::MT2; ++mt)
#pragma unroll
                for (int nt = 0; nt < Cfg::NT2; ++nt)
                    acc2[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc2[mt][nt], 0, 0, 0);
        }
    }

    PAIR_STAMP(7);
    if (PAIR_PROBE & 8) { if (acc2[0][0][0] == 123.456f && Y) Y[0] = 1.f; return; }
    // ---- epilogue 2: accumulators -> LDS tile -> whole rows out (gemm2.hip's fast path: 8 columns per thread) -------------------
    __syncthreads();
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mt = 0; mt < Cfg::MT2; ++mt)
#pragma unroll
        for (int nt = 0; nt < Cfg::NT2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ct[(wm2 + mt * 16 + (lane >> 4) * 4 + r) * Cfg::CS + wn2 + nt * 16 + (lane & 15)] = acc2[mt][nt][r];
    __syncthreads();
    PAIR_STAMP(8);
    constexpr int C8 = H2 / 8;
    constexpr int ROWS_PER_PASS = PAIR_THREADS / C8;
    const int c8 = (tid % C8) * 8, lr0 = tid / C8;
    const int row_end = min(PAIR_BM, M - i0);
    f32x4_t q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
    if (b2) { q0 = *reinterpret_cast<const f32x4_t*>(b2 + c8); q1 = *reinterpret_cast<const f32x4_t*>(b2 + c8 + 4); }
    const bool drop = drop_p > 0.f;
    const uint32_t ckey = seed32_2 + (uint32_t)(c8 >> 1) * 0x85EBCA77U;
    float* const yp = Y ? Y + (int64_t)i0 * ldy + c8 : nullptr;
    __bf16* const hp = Yh ? Yh + (int64_t)i0 * ldyh + c8 : nullptr;
#pragma unroll 2
    for (int lr = lr0; lr < row_end; lr += ROWS_PER_PASS) {
        f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS + c8) + q0;
        f32x4_t hi = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS + c8 + 4) + q1;
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
        if (yp) {
            *reinterpret_cast<f32x4_t*>(yp + (int64_t)lr * ldy) = lo;
            *reinterpret_cast<f32x4_t*>(yp + (int64_t)lr * ldy + 4) = hi;
        }
        if (hp) {
            bf16x8_t h8;
#pragma unroll
            for (int q = 0; q < 4; ++q) { h8[q] = (__bf16)lo[q]; h8[4 + q] = (__bf16)hi[q]; }
            *reinterpret_cast<bf16x8_t*>(hp + (int64_t)lr * ldyh) = h8;
        }
    }
    PAIR_STAMP(9);
}

extern "C" int cdc_expert_pair_fwd(const cdc_expert_pair_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n_expert > 0 && a->n_expert <= CDC_PAIR_MAX_EXPERT && a->M >= 0, CDC_E_BADARG, "expert_pair_fwd: bad counts");
    CDC_CHECK_ARG(a->H1 == 256 && a->H2 == 128, CDC_E_BADARG, "expert_pair_fwd: instantiated for hidden widths (256, 128), got (%d, %d)",
                  a->H1, a->H2);
    CDC_CHECK_ARG(a->K1r > 0 && a->K1r % PAIR_BK == 0, CDC_E_BADARG, "expert_pair_fwd: K1r=%d must be a positive multiple of 64", a->K1r);
    CDC_CHECK_ARG(a->drop_p >= 0.f && a->drop_p < 1.f, CDC_E_BADARG, "expert_pair_fwd: dropout p out of range");
    for (int e = 0; e < a->n_expert; ++e) {
        const cdc_pair_expert& E = a->e[e];
        CDC_CHECK_ARG(E.x && E.w1 && E.w2 && (E.y || E.yh) && E.ldx >= a->K1r && E.ldw1 >= a->K1r && E.ldw2 >= a->H1, CDC_E_BADARG,
                      "expert_pair_fwd: expert %d malformed", e);
        CDC_CHECK_ARG((!E.h || E.ldh >= a->H1) && (!E.y || E.ldy >= a->H2) && (!E.yh || E.ldyh >= a->H2), CDC_E_BADARG,
                      "expert_pair_fwd: expert %d: output row stride below its width", e);
        CDC_CHECK_ARG(((((uintptr_t)E.x) | ((uintptr_t)E.w1) | ((uintptr_t)E.w2) | ((uintptr_t)E.h) | ((uintptr_t)E.y) | ((uintptr_t)E.yh) |
                        ((uintptr_t)E.b2)) & 15) == 0 &&
                          E.ldx % 8 == 0 && E.ldw1 % 8 == 0 && E.ldw2 % 8 == 0 && (!E.h || E.ldh % 8 == 0) && (!E.y || E.ldy % 4 == 0) &&
                          (!E.yh || E.ldyh % 8 == 0),
                      CDC_E_ALIGN, "expert_pair_fwd: expert %d: operands must be 16-byte aligned with row strides of whole 16-byte chunks", e);
        if (E.ws) {
            CDC_CHECK_ARG(E.ns > 0 && E.ns <= PAIR_SIDE && E.ys && E.ldys >= E.ns && E.ldws >= a->K1r && E.ldws % 8 == 0 &&
                              (((uintptr_t)E.ws) & 15) == 0,
                          CDC_E_BADARG, "expert_pair_fwd: expert %d: side output malformed (ns=%d)", e, E.ns);
        }
    }
    if (a->M == 0) return 0;
    typedef PairCfg<256, 128> C_;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_pair_fwd<256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, C_::SMEM);
        attr_done = true;
    }
    const int64_t grid = cdc_ceil_div(a->M, PAIR_BM) * a->n_expert;
    CDC_CHECK_ARG(grid < (1ll << 31), CDC_E_TOOBIG, "expert_pair_fwd: grid too large");
    hipLaunchKernelGGL((k_pair_fwd<256, 128>), dim3((unsigned)grid), dim3(PAIR_THREADS), C_::SMEM, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("expert_pair_fwd");
    return 0;
}
