// gemm.hip — grouped linear layers on the gfx950 matrix cores.
//
// Stands in for every nn.Linear / F.linear on the reference's training path
// (model/layer.py:185,193,275; model/ple.py:83-94; model/mmoe.py:35-40; model/star.py:90-102)
// and for the three ATen addmm calls autograd issues for each of them (forward, grad-input,
// grad-weight).  Storage is fp32; the contraction runs either on bf16 MFMA with fp32
// accumulation (v_mfma_f32_16x16x32_bf16, operands rounded while staging into LDS) or on the
// exact fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// One 256-thread workgroup (4 waves, 2x2) owns a BM x BN output tile of one group; all groups of
// a layer go out in ONE launch (descriptors travel as kernel arguments).  Global -> register ->
// LDS staging with the next K-slab's loads in flight under the current slab's MFMAs.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define GEMM_THREADS 256
#ifndef CDC_GEMM_PROBE
#define CDC_GEMM_PROBE 0      /* tools/gemm_probe.hip: 1 = no reloads, 2 = no MFMAs, 4 = no LDS stores, 8 = no epilogue */
#endif
#define GEMM_BK 32           // reduction depth staged per barrier pair (64 measured slower: register pressure halves occupancy)
// LDS row strides chosen so the 16 lanes of a ds_read_b128 group land on disjoint bank quads:
// bf16: BK + 8 elements (80 B at BK 32, 144 B at BK 64); fp32: BK + 1
template <bool BF16, int BK> struct LdsElem;
template <int BK> struct LdsElem<true, BK> { typedef __bf16 type; static constexpr int stride = BK + 8; };
template <int BK> struct LdsElem<false, BK> { typedef float type; static constexpr int stride = BK + 1; };

// A tile operand: element (i, r) lives at p[i*s_i + r*s_r]; i in [0,I) (tile rows), r in [0,Rn) (reduction).
struct Operand {
    const float* p;
    int64_t s_i, s_r;
    int I, Rn;      // valid extents
    bool half = false;   // p really points at bf16 elements (strides in elements): a copy its producer already rounded
};

// ---- staging: global -> registers (fp32) -------------------------------------------------------
// RC = true : reduction index contiguous in memory (s_r == 1): thread loads 4 consecutive r
// RC = false: tile-row index contiguous (s_i == 1): thread loads 4 consecutive i at one r
template <int ROWS, bool RC, int BK>
struct Stage {
    static constexpr int PER = ROWS * BK / 4 / GEMM_THREADS;   // float4 loads per thread
    f32x4_t v[PER];

    __device__ __forceinline__ void load(const Operand& op, int i0, int r0, int tid) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            int i, r;
            if (RC) { r = (tid % (BK / 4)) * 4; i = tid / (BK / 4) + p * (GEMM_THREADS / (BK / 4)); }
            else    { i = (tid % (ROWS / 4)) * 4; r = tid / (ROWS / 4) + p * (GEMM_THREADS / (ROWS / 4)); }
            const int gi = i0 + i, gr = r0 + r;
            f32x4_t val = {0.f, 0.f, 0.f, 0.f};
            if (RC) {
                if (gi < op.I) {
                    const float* src = op.p + (int64_t)gi * op.s_i + gr;
                    if (gr + 3 < op.Rn && (((uintptr_t)src) & 15) == 0) val = *reinterpret_cast<const f32x4_t*>(src);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (gr + q < op.Rn) val[q] = src[q];
                    }
                }
            } else {
                if (gr < op.Rn) {
                    const float* src = op.p + (int64_t)gr * op.s_r + gi;
                    if (gi + 3 < op.I && (((uintptr_t)src) & 15) == 0) val = *reinterpret_cast<const f32x4_t*>(src);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (gi + q < op.I) val[q] = src[q];
                    }
                }
            }
            v[p] = val;
        }
    }

    template <typename T>
    __device__ __forceinline__ void store(T* lds, int stride, int tid) const {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            int i, r;
            if (RC) { r = (tid % (BK / 4)) * 4; i = tid / (BK / 4) + p * (GEMM_THREADS / (BK / 4)); }
            else    { i = (tid % (ROWS / 4)) * 4; r = tid / (ROWS / 4) + p * (GEMM_THREADS / (ROWS / 4)); }
            if (RC) {
                if constexpr (sizeof(T) == 2) {
                    bf16x4_t b = __builtin_convertvector(v[p], bf16x4_t);
                    *reinterpret_cast<bf16x4_t*>(lds + i * stride + r) = b;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) lds[i * stride + r + q] = v[p][q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) lds[(i + q) * stride + r] = (T)v[p][q];
            }
        }
    }
};

// ---- lean staging for reduction-contiguous operands ---------------------------------------------------------------
// PMC counters showed the generic Stage spending ~400 VALU instructions per K-slab and wave on index arithmetic, bounds
// predicates and alignment tests, against 16 MFMAs.  Everything that does not depend on the slab is hoisted here: one
// pointer per pass (rows past the operand's extent are CLAMPED to its last row — they only feed accumulator rows/columns
// the epilogue never stores), unconditional 16-byte loads for every full slab, pointer += BK.  Needs: 16-byte aligned base,
// row stride a multiple of 4 floats; the (rare) partial last slab takes a guarded path.
template <int ROWS, int BK>
struct LeanStage {
    static constexpr int PER = ROWS * BK / 4 / GEMM_THREADS;
    static constexpr int TPR = BK / 4;                    // threads per tile row
    static constexpr int RSTEP = GEMM_THREADS / TPR;      // tile rows per pass
    f32x4_t v[PER];
    const float* ptr[PER];

    __device__ __forceinline__ void init(const Operand& op, int i0, int tid) {
        const int r = (tid % TPR) * 4, i = tid / TPR;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            int gi = i0 + i + p * RSTEP;
            gi = gi < op.I ? gi : op.I - 1;
            ptr[p] = op.p + (int64_t)gi * op.s_i + r;
        }
    }
    __device__ __forceinline__ void load_full() {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            v[p] = *reinterpret_cast<const f32x4_t*>(ptr[p]);
            ptr[p] += BK;
        }
    }
    __device__ __forceinline__ void stride2() {}
    __device__ __forceinline__ void load2() {                                  // every other slab (two stages in flight)
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            v[p] = *reinterpret_cast<const f32x4_t*>(ptr[p]);
            ptr[p] += 2 * BK;
        }
    }
    __device__ __forceinline__ void load_tail(int r0, int Rn, int tid) {       // partial last slab: element guards
        const int r = r0 + (tid % TPR) * 4;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            f32x4_t val = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) if (r + q < Rn) val[q] = ptr[p][q];
            v[p] = val;
        }
    }
    template <typename T>
    __device__ __forceinline__ void store(T* lds, int stride, int tid) const {
        const int r = (tid % TPR) * 4, i = tid / TPR;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            T* dst = lds + (i + p * RSTEP) * stride + r;
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<bf16x4_t*>(dst) = __builtin_convertvector(v[p], bf16x4_t);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = v[p][q];
            }
        }
    }
};
__device__ __forceinline__ bool lean_ok(const Operand& op) {
    return !op.half && op.s_r == 1 && op.I > 0 && ((((uintptr_t)op.p) & 15) == 0) && (op.s_i % 4 == 0);
}
// The same for an operand that already IS bf16 in memory (the per-step W^T copies): 16-byte loads of eight elements, half
// the bytes from L2 and no conversion in the loop — the bits are those the fp32 path's convert would have produced.
template <int ROWS, int BK>
struct LeanStageH {
    static constexpr int TPR = BK / 8;                    // threads per tile row
    static constexpr int RSTEP = GEMM_THREADS / TPR;      // tile rows per pass
    static constexpr int PER = (ROWS * BK / 8 + GEMM_THREADS - 1) / GEMM_THREADS;
    bf16x8_t v[PER];
    const __bf16* ptr[PER];

    __device__ __forceinline__ void init(const Operand& op, int i0, int tid) {
        const int r = (tid % TPR) * 8, i = tid / TPR;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            int gi = i0 + i + p * RSTEP;
            gi = gi < op.I ? gi : op.I - 1;
            ptr[p] = reinterpret_cast<const __bf16*>(op.p) + (int64_t)gi * op.s_i + r;
        }
    }
    __device__ __forceinline__ void load_full() {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            v[p] = *reinterpret_cast<const bf16x8_t*>(ptr[p]);
            ptr[p] += BK;
        }
    }
    __device__ __forceinline__ void stride2() {}
    __device__ __forceinline__ void load2() {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            v[p] = *reinterpret_cast<const bf16x8_t*>(ptr[p]);
            ptr[p] += 2 * BK;
        }
    }
    __device__ __forceinline__ void load_tail(int r0, int Rn, int tid) {
        const int r = r0 + (tid % TPR) * 8;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            bf16x8_t val;
#pragma unroll
            for (int q = 0; q < 8; ++q) val[q] = (r + q < Rn) ? ptr[p][q] : (__bf16)0.f;
            v[p] = val;
        }
    }
    template <typename T>
    __device__ __forceinline__ void store(T* lds, int stride, int tid) const {
        static_assert(sizeof(T) == 2, "a bf16 operand feeds the bf16 MFMA path only");
        const int r = (tid % TPR) * 8, i = tid / TPR;
#pragma unroll
        for (int p = 0; p < PER; ++p)
            if (ROWS * BK / 8 >= GEMM_THREADS || i + p * RSTEP < ROWS)
                *reinterpret_cast<bf16x8_t*>(lds + (i + p * RSTEP) * stride + r) = v[p];
    }
};
__device__ __forceinline__ bool lean_ok_h(const Operand& op) {
    return op.half && op.s_r == 1 && op.I > 0 && ((((uintptr_t)op.p) & 15) == 0) && (op.s_i % 8 == 0);
}

// ---- one K-slab of MFMAs for this wave ------------------------------------------------------------
template <bool BF16, int MT, int NT, int BK>
__device__ __forceinline__ void mma_slab(const typename LdsElem<BF16, BK>::type* As, const typename LdsElem<BF16, BK>::type* Bs,
                                         int wm, int wn, int lane, f32x4_t (&acc)[MT][NT]) {
    constexpr int S = LdsElem<BF16, BK>::stride;
    const int lr = lane & 15, lk = lane >> 4;
    if constexpr (BF16) {
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8_t a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8_t*>(As + (wm + mt * 16 + lr) * S + ks * 32 + lk * 8);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn + nt * 16 + lr) * S + ks * 32 + lk * 8);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            float a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = As[(wm + mt * 16 + lr) * S + kk * 4 + lk];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = Bs[(wn + nt * 16 + lr) * S + kk * 4 + lk];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    }
}

// ---- the lean loop with a B operand that is bf16 in memory (the host side guarantees lean_ok(A) and lean_ok_h(B)) ----
template <int BM, int BN, int DEPTH>
__device__ __forceinline__ void gemm_accumulate_half_b(const Operand& A, const Operand& B, int i0, int j0, unsigned char* smem,
                                                       f32x4_t (&acc)[BM / 32][BN / 32]) {
    constexpr int BK = GEMM_BK;
    typedef __bf16 T;
    constexpr int S = LdsElem<true, BK>::stride;
    T* As = reinterpret_cast<T*>(smem);
    T* Bs = As + BM * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    const int nk = (A.Rn + BK - 1) / BK;
    const int nfull = A.Rn / BK;
    if constexpr (DEPTH == 2) {
        if (nfull == nk && nk >= 2) {
            LeanStage<BM, BK> a0, a1;
            LeanStageH<BN, BK> b0, b1;
            a0.init(A, i0, tid); b0.init(B, j0, tid);
            a1 = a0; b1 = b0;
#pragma unroll
            for (int p = 0; p < LeanStage<BM, BK>::PER; ++p) a1.ptr[p] += BK;
#pragma unroll
            for (int p = 0; p < LeanStageH<BN, BK>::PER; ++p) b1.ptr[p] += BK;
            a0.load2(); b0.load2(); a1.load2(); b1.load2();
            for (int kt = 0; kt < nk; kt += 2) {
                a0.store(As, S, tid); b0.store(Bs, S, tid);
                __syncthreads();
                if (kt + 2 < nk) { a0.load2(); b0.load2(); }
                mma_slab<true, BM / 32, BN / 32, BK>(As, Bs, wm, wn, lane, acc);
                __syncthreads();
                if (kt + 1 < nk) {
                    a1.store(As, S, tid); b1.store(Bs, S, tid);
                    __syncthreads();
                    if (kt + 3 < nk) { a1.load2(); b1.load2(); }
                    mma_slab<true, BM / 32, BN / 32, BK>(As, Bs, wm, wn, lane, acc);
                    __syncthreads();
                }
            }
            return;
        }
    }
    LeanStage<BM, BK> la;
    LeanStageH<BN, BK> lb;
    la.init(A, i0, tid);
    lb.init(B, j0, tid);
    if (nfull > 0) { la.load_full(); lb.load_full(); }
    else if (nk > 0) { la.load_tail(0, A.Rn, tid); lb.load_tail(0, A.Rn, tid); }
    for (int kt = 0; kt < nk; ++kt) {
        la.store(As, S, tid);
        lb.store(Bs, S, tid);
        __syncthreads();
        if (kt + 1 < nfull) { la.load_full(); lb.load_full(); }
        else if (kt + 1 < nk) { la.load_tail((kt + 1) * BK, A.Rn, tid); lb.load_tail((kt + 1) * BK, A.Rn, tid); }
        mma_slab<true, BM / 32, BN / 32, BK>(As, Bs, wm, wn, lane, acc);
        __syncthreads();
    }
}

// ---- accumulate A(i,r)·B(j,r) over r for one (A,B) operand pair into acc ---------------------------
template <bool BF16, int BM, int BN, bool A_RC, bool B_RC, bool COLSUM, int DEPTH = 1>
__device__ __forceinline__ void gemm_accumulate(const Operand& A, const Operand& B, int i0, int j0, unsigned char* smem,
                                                f32x4_t (&acc)[BM / 32][BN / 32], f32x4_t* colsum) {
    constexpr int BK = GEMM_BK;
    typedef typename LdsElem<BF16, BK>::type T;
    constexpr int S = LdsElem<BF16, BK>::stride;
    T* As = reinterpret_cast<T*>(smem);
    T* Bs = As + BM * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    const int nk = (A.Rn + BK - 1) / BK;
    if constexpr (A_RC && B_RC && !COLSUM) {
        if constexpr (BF16) {
            if (B.half) {                                  // uniform over the workgroup; eligibility checked by the launcher
                gemm_accumulate_half_b<BM, BN, DEPTH>(A, B, i0, j0, smem, acc);
                return;
            }
        }
        if (lean_ok(A) && lean_ok(B)) {                    // uniform over the workgroup
            const int nfull = A.Rn / BK;
            if constexpr (DEPTH == 2) {
                // two slabs in flight (only full slabs take this path; a partial last slab falls back to depth 1 below)
                if (nfull == nk && nk >= 2) {
                    LeanStage<BM, BK> a0, a1;
                    LeanStage<BN, BK> b0, b1;
                    a0.init(A, i0, tid); b0.init(B, j0, tid);
                    a1 = a0; b1 = b0;
#pragma unroll
                    for (int p = 0; p < LeanStage<BM, BK>::PER; ++p) { a1.ptr[p] += BK; a0.stride2(); }
#pragma unroll
                    for (int p = 0; p < LeanStage<BN, BK>::PER; ++p) { b1.ptr[p] += BK; b0.stride2(); }
                    a0.load2(); b0.load2(); a1.load2(); b1.load2();
                    for (int kt = 0; kt < nk; kt += 2) {
                        a0.store(As, S, tid); b0.store(Bs, S, tid);
                        __syncthreads();
                        if (kt + 2 < nk) { a0.load2(); b0.load2(); }
                        mma_slab<BF16, BM / 32, BN / 32, BK>(As, Bs, wm, wn, lane, acc);
                        __syncthreads();
                        if (kt + 1 < nk) {
                            a1.store(As, S, tid); b1.store(Bs, S, tid);
                            __syncthreads();
                            if (kt + 3 < nk) { a1.load2(); b1.load2(); }
                            mma_slab<BF16, BM / 32, BN / 32, BK>(As, Bs, wm, wn, lane, acc);
                            __syncthreads();
                        }
                    }
                    return;
                }
            }
            LeanStage<BM, BK> la;
            LeanStage<BN, BK> lb;
            la.init(A, i0, tid);
            lb.init(B, j0, tid);
            if (nfull > 0) { la.load_full(); lb.load_full(); }
            else if (nk > 0) { la.load_tail(0, A.Rn, tid); lb.load_tail(0, A.Rn, tid); }
            for (int kt = 0; kt < nk; ++kt) {
                if (!(CDC_GEMM_PROBE & 4) || kt == 0) {
                    la.store(As, S, tid);
                    lb.store(Bs, S, tid);
                }
                __syncthreads();
                if (!(CDC_GEMM_PROBE & 1)) {
                    if (kt + 1 < nfull) { la.load_full(); lb.load_full(); }
                    else if (kt + 1 < nk) { la.load_tail((kt + 1) * BK, A.Rn, tid); lb.load_tail((kt + 1) * BK, A.Rn, tid); }
                }
                if (!(CDC_GEMM_PROBE & 2)) mma_slab<BF16, BM / 32, BN / 32, BK>(As, Bs, wm, wn, lane, acc);
                __syncthreads();
            }
            if (CDC_GEMM_PROBE & 2) {                      // keep the loads alive when the MFMAs are probed away
#pragma unroll
                for (int p = 0; p < LeanStage<BM, BK>::PER; ++p) acc[0][0] += la.v[p];
#pragma unroll
                for (int p = 0; p < LeanStage<BN, BK>::PER; ++p) acc[0][0] += lb.v[p];
            }
            return;
        }
    }
    Stage<BM, A_RC, BK> sa;
    Stage<BN, B_RC, BK> sb;
    if (nk > 0) { sa.load(A, i0, 0, tid); sb.load(B, j0, 0, tid); }
    for (int kt = 0; kt < nk; ++kt) {
        if (COLSUM) {
            // bias gradient: column sums of the A operand (dZ) in fp32, taken before the bf16 rounding
#pragma unroll
            for (int p = 0; p < Stage<BM, A_RC, BK>::PER; ++p) colsum[0] += sa.v[p];
        }
        sa.store(As, S, tid);
        sb.store(Bs, S, tid);
        __syncthreads();
        if (kt + 1 < nk) { sa.load(A, i0, (kt + 1) * BK, tid); sb.load(B, j0, (kt + 1) * BK, tid); }
        mma_slab<BF16, BM / 32, BN / 32, BK>(As, Bs, wm, wn, lane, acc);
        __syncthreads();
    }
}

// ---- epilogue staging: the wave's accumulators go through LDS so that global rows are written as full float4 runs ----
// (an MFMA accumulator lane owns 4 rows x 1 column: stored directly, every wave store covers four 64-byte pieces of four
// different rows and every element pays its own 64-bit address arithmetic; the probe build showed that direct epilogue
// costing 15 of the 37 us of the level-1 launch)
template <int BM, int BN>
struct OutTile {
    static constexpr int CS = BN + 4;                                      // row stride (floats): 16-lane groups land on disjoint banks
    static constexpr bool staged = (size_t)BM * CS * sizeof(float) <= 48 * 1024;
    static constexpr int F4_PER_ROW = BN / 4;
    static constexpr int PER = BM * F4_PER_ROW / GEMM_THREADS;
    __device__ static __forceinline__ void put(float* ct, const f32x4_t (&acc)[BM / 32][BN / 32], int wm, int wn, int lane) {
#pragma unroll
        for (int mt = 0; mt < BM / 32; ++mt)
#pragma unroll
            for (int nt = 0; nt < BN / 32; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ct[(wm + mt * 16 + (lane >> 4) * 4 + r) * CS + wn + nt * 16 + (lane & 15)] = acc[mt][nt][r];
    }
};

// XCD-aware block remap (bijective): blocks that share an XCD (bid % 8) get a contiguous run of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
}

// =================================================================================================
// forward: y = act(x · wᵀ + bias)
// =================================================================================================
template <bool BF16, int BM, int BN, int DEPTH>
__global__ void __launch_bounds__(GEMM_THREADS) k_glinear_fwd(const cdc_lin_fwd_args a) {
    CDC_PRIO_MAIN();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int g = find_group<false>(a.n_groups, tile, [&](int l) { return ((a.g[l].M + BM - 1) / BM) * ((a.g[l].N + BN - 1) / BN); },
                                    [](int) { return (int64_t)0; }, tile, nullptr);
    if (g < 0) return;
    const cdc_lin_group& G = a.g[g];
    const int tn_cnt = (G.N + BN - 1) / BN;
    int row_lo = 0, M = G.M;
    if (a.row_offsets) { row_lo = a.row_offsets[g]; M = a.row_offsets[g + 1] - row_lo; }
    const int i0 = (tile / tn_cnt) * BM, j0 = (tile % tn_cnt) * BN;
    if (i0 >= M) return;

    Operand A{G.x + (int64_t)row_lo * G.ldx, G.ldx, 1, M, G.K};
    Operand B{G.w, G.ldw, 1, G.N, G.K};
    f32x4_t acc[BM / 32][BN / 32];
#pragma unroll
    for (int mt = 0; mt < BM / 32; ++mt)
#pragma unroll
        for (int nt = 0; nt < BN / 32; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    gemm_accumulate<BF16, BM, BN, true, true, false, DEPTH>(A, B, i0, j0, smem, acc, nullptr);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    if (CDC_GEMM_PROBE & 8) {
        if (acc[0][0][0] == 123.456f) G.y[0] = acc[1][1][1];
        return;
    }
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    uint64_t seed = a.seed;
    if (a.drop_p > 0.f && a.seed_offset_dev) seed += (uint64_t)(uint32_t)(*a.seed_offset_dev) * 0xD1342543DE82EF95ull;
    float* y = G.y + (int64_t)row_lo * G.ldy;
    auto finish = [&](float val, int row, int col, float bv) -> float {
        val += bv;
        if (col < G.act_cols) {
            if (a.relu) val = fmaxf(val, 0.f);
            if (a.drop_p > 0.f) {
                const uint64_t e = ((uint64_t)g << 56) ^ ((uint64_t)(row_lo + row) * (uint64_t)G.N + (uint64_t)col);
                val = cdc_uniform(seed, e) < a.drop_p ? 0.f : val * keep_scale;
            }
        }
        return val;
    };
    typedef OutTile<BM, BN> OT;
    if constexpr (OT::staged) {
        float* ct = reinterpret_cast<float*>(smem);                      // the operand staging area is free after the last barrier
        OT::put(ct, acc, wm, wn, lane);
        __syncthreads();
        if (G.bn_partial) {                                               // uniform over the workgroup
            // the statistics pass of the BatchNorm that normalises y next: per (64-row block, column) sums of y and y^2 in
            // double, written where k_bn_stats would put them (block index = row / CDC_BN_ROWS_PER_BLOCK).  Wave q takes rows
            // 16q..16q+15 of a block, lane = column; the four quarter sums are added in order by wave 0.
            static_assert(CDC_BN_ROWS_PER_BLOCK == 64 && BM % 64 == 0 && BN % 64 == 0, "partial-sum blocks are 64 rows");
            __shared__ double quarter[4][64][2];
#pragma unroll
            for (int cc = 0; cc < BN / 64; ++cc) {
                const int col = j0 + cc * 64 + lane;
                const float bv = (col < G.N && G.bias) ? G.bias[col] : 0.f;
#pragma unroll
                for (int h = 0; h < BM / 64; ++h) {
                    const int r_lo = i0 + h * 64;
                    if (r_lo >= M) break;                                 // uniform
                    const int rows = min(64, M - r_lo);
                    double s1 = 0.0, s2 = 0.0;
                    for (int r = wave * 16; r < min(wave * 16 + 16, rows); ++r) {
                        const double x = (double)(ct[(h * 64 + r) * OT::CS + cc * 64 + lane] + bv);
                        s1 += x; s2 += x * x;
                    }
                    quarter[wave][lane][0] = s1; quarter[wave][lane][1] = s2;
                    __syncthreads();
                    if (wave == 0 && col < G.N) {
                        double* ws = G.bn_partial + ((int64_t)(r_lo / 64) * G.bn_total_c + G.bn_col0 + col) * 2;
                        ws[0] = ((quarter[0][lane][0] + quarter[1][lane][0]) + quarter[2][lane][0]) + quarter[3][lane][0];
                        ws[1] = ((quarter[0][lane][1] + quarter[1][lane][1]) + quarter[2][lane][1]) + quarter[3][lane][1];
                    }
                    __syncthreads();
                }
            }
        }
        const bool vec = ((((uintptr_t)y) & 15) == 0) && (G.ldy % 4 == 0);
#pragma unroll
        for (int p = 0; p < OT::PER; ++p) {
            const int idx = threadIdx.x + p * GEMM_THREADS;
            const int lr = idx / OT::F4_PER_ROW, c4 = (idx % OT::F4_PER_ROW) * 4;
            const int row = i0 + lr, col = j0 + c4;
            if (row >= M || col >= G.N) continue;
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(ct + lr * OT::CS + c4);
            float* dst = y + (int64_t)row * G.ldy + col;
            if (vec && col + 3 < G.N) {
                f32x4_t o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = finish(v[q], row, col + q, G.bias ? G.bias[col + q] : 0.f);
                *reinterpret_cast<f32x4_t*>(dst) = o;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (col + q < G.N) dst[q] = finish(v[q], row, col + q, G.bias ? G.bias[col + q] : 0.f);
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < BM / 32; ++mt)
#pragma unroll
            for (int nt = 0; nt < BN / 32; ++nt) {
                const int col = j0 + wn + nt * 16 + (lane & 15);
                if (col >= G.N) continue;
                const float bv = G.bias ? G.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + wm + mt * 16 + (lane >> 4) * 4 + r;
                    if (row >= M) continue;
                    y[(int64_t)row * G.ldy + col] = finish(acc[mt][nt][r], row, col, bv);
                }
            }
    }
}

// =================================================================================================
// grad-input: dX_o = sum_s dZ_s · W_s  (+ activation mask of the producing layer)
// =================================================================================================
template <bool BF16, int BM, int BN, bool WT, int DEPTH>
__global__ void __launch_bounds__(GEMM_THREADS) k_glinear_bwd_x(const cdc_lin_bwdx_args a) {
    CDC_PRIO_MAIN();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int o = find_group<false>(a.n_out, tile, [&](int l) { return ((a.o[l].M + BM - 1) / BM) * ((a.o[l].K + BN - 1) / BN); },
                                    [](int) { return (int64_t)0; }, tile, nullptr);
    if (o < 0) return;
    const cdc_bwdx_out& O = a.o[o];
    const int tn_cnt = (O.K + BN - 1) / BN;
    int row_lo = 0, M = O.M;
    if (a.row_offsets) { row_lo = a.row_offsets[o]; M = a.row_offsets[o + 1] - row_lo; }
    const int i0 = (tile / tn_cnt) * BM, j0 = (tile % tn_cnt) * BN;
    if (i0 >= M) return;

    f32x4_t acc[BM / 32][BN / 32];
#pragma unroll
    for (int mt = 0; mt < BM / 32; ++mt)
#pragma unroll
        for (int nt = 0; nt < BN / 32; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // lane l fetches segment l's descriptor (one round of vector loads for the whole list instead of a scalar cache miss or
    // two per segment, each in front of that segment's first tile load); the loop then reads them lane by lane
    unsigned long long todo;
    uint64_t v_dz, v_w;
    int64_t v_lddz, v_ldw;
    int v_n, v_half;
    {
        const int l = threadIdx.x & 63;
        const bool has = l < a.n_seg;
        const cdc_bwdx_seg& L = a.s[has ? l : 0];
        v_dz = (uint64_t)L.dz; v_lddz = L.lddz; v_n = L.N; v_half = L.wt_bf16;
        v_w = WT ? (uint64_t)L.wt : (uint64_t)L.w;
        v_ldw = WT ? L.ldwt : L.ldw;
        todo = __ballot(has && L.out == o);
    }
    auto lane64 = [](uint64_t v, int l) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    };
    while (todo) {
        const int s = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        todo &= todo - 1;
        const float* dz = reinterpret_cast<const float*>(lane64(v_dz, s));
        const float* wp = reinterpret_cast<const float*>(lane64(v_w, s));
        const int64_t lddz = (int64_t)lane64((uint64_t)v_lddz, s), ldw = (int64_t)lane64((uint64_t)v_ldw, s);
        const int N = __builtin_amdgcn_readlane(v_n, s);
        Operand A{dz + (int64_t)row_lo * lddz, lddz, 1, M, N};           // (i=row, r=n)  r contiguous
        if (WT) {
            Operand B{wp, ldw, 1, O.K, N};                                // W^T [K,N]: (j=k, r=n)  r contiguous
            B.half = __builtin_amdgcn_readlane(v_half, s) != 0;          // ... stored as bf16 by cdc_transpose_multi
            gemm_accumulate<BF16, BM, BN, true, true, false, DEPTH>(A, B, i0, j0, smem, acc, nullptr);
        } else {
            Operand B{wp, 1, ldw, O.K, N};                                // W [N,K]:   (j=k, r=n)  j contiguous
            gemm_accumulate<BF16, BM, BN, true, false, false>(A, B, i0, j0, smem, acc, nullptr);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
    float* dx = O.dx + (int64_t)row_lo * O.lddx;
    const float* mk = O.mask_y ? O.mask_y + (int64_t)row_lo * O.ldmask : nullptr;
    typedef OutTile<BM, BN> OT;
    if constexpr (OT::staged) {
        float* ct = reinterpret_cast<float*>(smem);
        OT::put(ct, acc, wm, wn, lane);
        __syncthreads();
        const bool vec = ((((uintptr_t)dx) & 15) == 0) && (O.lddx % 4 == 0) &&
                         (!mk || (((((uintptr_t)mk) & 15) == 0) && (O.ldmask % 4 == 0)));
#pragma unroll
        for (int p = 0; p < OT::PER; ++p) {
            const int idx = threadIdx.x + p * GEMM_THREADS;
            const int lr = idx / OT::F4_PER_ROW, c4 = (idx % OT::F4_PER_ROW) * 4;
            const int row = i0 + lr, col = j0 + c4;
            if (row >= M || col >= O.K) continue;
            f32x4_t v = *reinterpret_cast<const f32x4_t*>(ct + lr * OT::CS + c4);
            float* dst = dx + (int64_t)row * O.lddx + col;
            if (vec && col + 3 < O.K) {
                if (mk) {
                    const f32x4_t m4 = *reinterpret_cast<const f32x4_t*>(mk + (int64_t)row * O.ldmask + col);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < O.mask_cols) v[q] = m4[q] > 0.f ? v[q] * a.mask_scale : 0.f;
                }
                if (O.accumulate) {
                    const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dst);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = d4[q] + v[q];
                }
                *reinterpret_cast<f32x4_t*>(dst) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (col + q >= O.K) continue;
                    float val = v[q];
                    if (mk && col + q < O.mask_cols) val = mk[(int64_t)row * O.ldmask + col + q] > 0.f ? val * a.mask_scale : 0.f;
                    dst[q] = O.accumulate ? dst[q] + val : val;
                }
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < BM / 32; ++mt)
#pragma unroll
            for (int nt = 0; nt < BN / 32; ++nt) {
                const int col = j0 + wn + nt * 16 + (lane & 15);
                if (col >= O.K) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i0 + wm + mt * 16 + (lane >> 4) * 4 + r;
                    if (row >= M) continue;
                    float val = acc[mt][nt][r];
                    if (mk && col < O.mask_cols) val = mk[(int64_t)row * O.ldmask + col] > 0.f ? val * a.mask_scale : 0.f;
                    float* dst = dx + (int64_t)row * O.lddx + col;
                    *dst = O.accumulate ? *dst + val : val;
                }
            }
    }
}

// =================================================================================================
// grad-weight: dW_g = dZ_gᵀ · X_g ; db_g = colsum(dZ_g).  The reduction runs over the batch rows, the output is
// small: with split_k > 1 each (tile, row slice) pair is its own workgroup writing an fp32 slab, and
// k_bwd_w_reduce adds the slabs in slice order.
// =================================================================================================
template <bool BF16, int BM, int BN>
__global__ void __launch_bounds__(GEMM_THREADS) k_glinear_bwd_w(const cdc_lin_bwdw_args a, int64_t slab_stride) {
    CDC_PRIO_MAIN();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = a.split_k > 1 ? a.split_k : 1;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id % S;
    int tile = id / S;
    int64_t g_off = 0;
    const int g = find_group<true>(a.n_groups, tile, [&](int l) { return ((a.g[l].N + BM - 1) / BM) * ((a.g[l].K + BN - 1) / BN); },
                                   [&](int l) { return (int64_t)a.g[l].N * a.g[l].K + a.g[l].N; }, tile, &g_off);
    if (g < 0) return;
    const cdc_bwdw_group& G = a.g[g];
    const int tn_cnt = (G.K + BN - 1) / BN;
    int row_lo = 0, M = G.M;
    if (a.row_offsets) { row_lo = a.row_offsets[g]; M = a.row_offsets[g + 1] - row_lo; }
    // this workgroup's slice of the batch rows (multiple of BK so every slice but the last is whole K-slabs)
    int chunk = ((M + S - 1) / S + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    if (chunk < GEMM_BK) chunk = GEMM_BK;
    const int r0 = split * chunk;
    int rn = M - r0;
    if (rn > chunk) rn = chunk;
    if (rn < 0) rn = 0;
    const int i0 = (tile / tn_cnt) * BM, j0 = (tile % tn_cnt) * BN;   // i over N (dW rows), j over K (dW cols)

    Operand A{G.dz + (int64_t)(row_lo + r0) * G.lddz, 1, G.lddz, G.N, rn};     // (i=n, r=b)  i contiguous
    Operand B{G.x + (int64_t)(row_lo + r0) * G.ldx, 1, G.ldx, G.K, rn};        // (j=k, r=b)  j contiguous
    f32x4_t acc[BM / 32][BN / 32];
#pragma unroll
    for (int mt = 0; mt < BM / 32; ++mt)
#pragma unroll
        for (int nt = 0; nt < BN / 32; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t colsum = {0.f, 0.f, 0.f, 0.f};
    const bool want_db = (G.db != nullptr) && (j0 == 0);
    if (want_db) gemm_accumulate<BF16, BM, BN, false, false, true>(A, B, i0, j0, smem, acc, &colsum);
    else         gemm_accumulate<BF16, BM, BN, false, false, false>(A, B, i0, j0, smem, acc, nullptr);

    float* slab = S > 1 ? a.workspace + (int64_t)split * slab_stride + g_off : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (want_db) {
        // threads tid % (BM/4) share the same 4 columns; fixed-order reduction through LDS
        float* red = reinterpret_cast<float*>(smem);                       // GEMM_THREADS * 4 floats
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) red[tid * 4 + q] = colsum[q];
        __syncthreads();
        if (tid < BM) {
            const int c4 = tid / 4, q = tid % 4;
            float sum = 0.f;
            for (int t = c4; t < GEMM_THREADS; t += BM / 4) sum += red[t * 4 + q];
            const int n = i0 + tid;
            if (n < G.N) {
                if (slab) slab[(int64_t)G.N * G.K + n] = sum;
                else G.db[n] = G.accumulate ? G.db[n] + sum : sum;
            }
        }
    }
    const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
#pragma unroll
    for (int mt = 0; mt < BM / 32; ++mt)
#pragma unroll
        for (int nt = 0; nt < BN / 32; ++nt) {
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

// (the body is a device function so that one launch can reduce the slabs of two grad-weight launches: k_bwd_w_reduce_dual below; `a` may
// live in the kernel argument block or in device memory)
__device__ __forceinline__ void bwd_w_reduce_body(const cdc_lin_bwdw_args& a, const int64_t slab_stride, const int64_t total, const int bid,
                                                  const int nblk) {
    // one flat index space over all groups: a slab IS the concatenation [dW_0 | db_0 | dW_1 | db_1 | ...]
    __shared__ int64_t first[CDC_MAX_GROUPS + 1];                      // first flat index of every group (scan by wave 0)
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int64_t sz = lane < a.n_groups ? (int64_t)a.g[lane].N * a.g[lane].K + a.g[lane].N : 0;
        int64_t inc = sz;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane < a.n_groups) first[lane + 1] = inc;
        if (lane == 0) first[0] = 0;
    }
    __syncthreads();
    // 16-byte lanes when every group's flat start, its dW size and the slab stride are multiples of 4 floats and the
    // destinations are dense: the same sums in the same order, a quarter of the load instructions
    bool vec4 = (slab_stride % 4 == 0) && ((((uintptr_t)a.workspace) & 15) == 0);
    for (int g = 0; g < a.n_groups && vec4; ++g) {
        const cdc_bwdw_group& G = a.g[g];
        vec4 = first[g] % 4 == 0 && ((int64_t)G.N * G.K) % 4 == 0 && G.lddw == G.K && ((((uintptr_t)G.dw) & 15) == 0) &&
               (!G.db || ((((uintptr_t)G.db) & 15) == 0));
    }
    if (vec4) {
        const int64_t total4 = (total + 3) / 4;
        for (int64_t q = (int64_t)bid * blockDim.x + threadIdx.x; q < total4; q += (int64_t)nblk * blockDim.x) {
            const int64_t e = q * 4;
            int g = 0;
            while (g + 1 < a.n_groups && e >= first[g + 1]) ++g;
            const cdc_bwdw_group& G = a.g[g];
            const int64_t local = e - first[g];
            const int64_t nk = (int64_t)G.N * G.K;
            const int64_t gsz = nk + G.N;
            const int n_valid = (int)min((int64_t)4, gsz - local);       // a group whose N is not a multiple of 4 ends mid-piece
            if (n_valid == 4) {
                f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
                int s = 0;
                for (; s + 4 <= a.split_k; s += 4) {
                    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(a.workspace + (int64_t)s * slab_stride + e);
                    const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(a.workspace + (int64_t)(s + 1) * slab_stride + e);
                    const f32x4_t v2 = *reinterpret_cast<const f32x4_t*>(a.workspace + (int64_t)(s + 2) * slab_stride + e);
                    const f32x4_t v3 = *reinterpret_cast<const f32x4_t*>(a.workspace + (int64_t)(s + 3) * slab_stride + e);
                    sum += v0; sum += v1; sum += v2; sum += v3;
                }
                for (; s < a.split_k; ++s) sum += *reinterpret_cast<const f32x4_t*>(a.workspace + (int64_t)s * slab_stride + e);
                float* dst = local < nk ? G.dw + local : (G.db ? G.db + (local - nk) : nullptr);
                if (dst) {
                    if (G.accumulate) sum += *reinterpret_cast<const f32x4_t*>(dst);
                    *reinterpret_cast<f32x4_t*>(dst) = sum;
                }
            } else {
                for (int k = 0; k < n_valid; ++k) {
                    float sum = 0.f;
                    for (int s = 0; s < a.split_k; ++s) sum += a.workspace[(int64_t)s * slab_stride + e + k];
                    const int64_t lk = local + k;
                    float* dst = lk < nk ? G.dw + lk : (G.db ? G.db + (lk - nk) : nullptr);
                    if (dst) *dst = G.accumulate ? *dst + sum : sum;
                }
                // the remaining lanes of this piece belong to the next group: its flat start is a multiple of 4, so none do
            }
        }
        return;
    }
    for (int64_t e = (int64_t)bid * blockDim.x + threadIdx.x; e < total; e += (int64_t)nblk * blockDim.x) {
        // slabs are added in slice order (deterministic); four loads are in flight per round
        float sum = 0.f;
        int s = 0;
        for (; s + 4 <= a.split_k; s += 4) {
            const float v0 = a.workspace[(int64_t)s * slab_stride + e], v1 = a.workspace[(int64_t)(s + 1) * slab_stride + e];
            const float v2 = a.workspace[(int64_t)(s + 2) * slab_stride + e], v3 = a.workspace[(int64_t)(s + 3) * slab_stride + e];
            sum += v0; sum += v1; sum += v2; sum += v3;
        }
        for (; s < a.split_k; ++s) sum += a.workspace[(int64_t)s * slab_stride + e];
        int g = 0;
        while (g + 1 < a.n_groups && e >= first[g + 1]) ++g;
        const int64_t local = e - first[g];
        const cdc_bwdw_group& G = a.g[g];
        const int64_t nk = (int64_t)G.N * G.K;
        if (local < nk) {
            float* dst = G.lddw == G.K ? G.dw + local : G.dw + (local / G.K) * G.lddw + (local % G.K);
            *dst = G.accumulate ? *dst + sum : sum;
        } else if (G.db) {
            float* dst = G.db + (local - nk);
            *dst = G.accumulate ? *dst + sum : sum;
        }
    }
}
__global__ void __launch_bounds__(256) k_bwd_w_reduce(const cdc_lin_bwdw_args a, int64_t slab_stride, int64_t total) {
    CDC_PRIO_MAIN();
    bwd_w_reduce_body(a, slab_stride, total, blockIdx.x, gridDim.x);
}
// the slabs of the wide and of the narrow class of a step's batched grad-weight contractions (k_g2_tn_dual) summed by ONE launch: workgroups
// [0, n_first) take tabs[0], the others tabs[1]
__global__ void __launch_bounds__(256) k_bwd_w_reduce_dual(const cdc_lin_bwdw_args* __restrict__ tabs, int64_t slab0, int64_t total0, int64_t slab1,
                                                           int64_t total1, int n_first) {
    CDC_PRIO_MAIN();
    if ((int)blockIdx.x < n_first) bwd_w_reduce_body(tabs[0], slab0, total0, blockIdx.x, n_first);
    else bwd_w_reduce_body(tabs[1], slab1, total1, (int)blockIdx.x - n_first, (int)gridDim.x - n_first);
}

// =================================================================================================
// grad-weight on bf16 MFMA without transposing stores: both operands are staged in their NATURAL layout
// ([batch row][column], coalesced 16-B loads, 8-B LDS stores) and the MFMA fragments — which need the reduction
// (batch-row) index contiguous — are fetched with the CDNA4 transposing LDS read ds_read_b64_tr_b16.
// k-order inside one MFMA is a permutation of the 32 staged rows (same for A and B, so the sum is unchanged):
// lane group g, element j  <->  row 4g + j (j < 4) | 16 + 4g + (j - 4): each half-wave then reads 8 consecutive rows,
// which the 160-byte row stride spreads over all 64 banks.
// =================================================================================================
#define TR_BK 64
#define TR_STRIDE 80                 /* bf16 elements per LDS row: 64 + 16 pad = 160 B */
typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8_t tr_fragment(const __bf16* tile, int row_base, int col0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const __bf16* a0 = tile + (row_base + 4 * g + q) * TR_STRIDE + col0 + 4 * p;
    const __bf16* a1 = a0 + 16 * TR_STRIDE;
    typedef s16x4_t __attribute__((address_space(3))) * lds_ptr;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a1);
    union { s16x4_t h[2]; bf16x8_t v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}

__global__ void __launch_bounds__(GEMM_THREADS) k_glinear_bwd_w_tr(const cdc_lin_bwdw_args a, int64_t slab_stride) {
    CDC_PRIO_MAIN();
    __shared__ __attribute__((aligned(16))) __bf16 As[TR_BK * TR_STRIDE];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[TR_BK * TR_STRIDE];
    const int S = a.split_k > 1 ? a.split_k : 1;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id % S;
    int tile = id / S;
    int64_t g_off = 0;
    const int g = find_group<true>(a.n_groups, tile, [&](int l) { return ((a.g[l].N + 63) / 64) * ((a.g[l].K + 63) / 64); },
                                   [&](int l) { return (int64_t)a.g[l].N * a.g[l].K + a.g[l].N; }, tile, &g_off);
    if (g < 0) return;
    const cdc_bwdw_group& G = a.g[g];
    const int tn_cnt = (G.K + 63) / 64;
    int row_lo = 0, M = G.M;
    if (a.row_offsets) { row_lo = a.row_offsets[g]; M = a.row_offsets[g + 1] - row_lo; }
    int chunk = ((M + S - 1) / S + TR_BK - 1) / TR_BK * TR_BK;
    if (chunk < TR_BK) chunk = TR_BK;
    const int r0 = split * chunk;
    int rn = M - r0;
    if (rn > chunk) rn = chunk;
    if (rn < 0) rn = 0;
    const int i0 = (tile / tn_cnt) * 64, j0 = (tile % tn_cnt) * 64;    // i over N (dW rows), j over K (dW cols)
    const float* dz = G.dz + (int64_t)(row_lo + r0) * G.lddz;
    const float* x = G.x + (int64_t)(row_lo + r0) * G.ldx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    // staging: thread -> 4 consecutive columns (c4) of rows r_, r_+16, r_+32, r_+48
    const int c4 = (tid & 15) * 4, rr = tid >> 4;
    const bool vec_a = (G.lddz % 4 == 0) && ((((uintptr_t)dz) & 15) == 0) && (i0 + c4 + 3 < G.N);
    const bool vec_b = (G.ldx % 4 == 0) && ((((uintptr_t)x) & 15) == 0) && (j0 + c4 + 3 < G.K);
    f32x4_t ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = k0 + rr + 16 * p;
            f32x4_t va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (r < rn) {
                const float* pa = dz + (int64_t)r * G.lddz + i0 + c4;
                const float* pb = x + (int64_t)r * G.ldx + j0 + c4;
                if (vec_a) va = *reinterpret_cast<const f32x4_t*>(pa);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (i0 + c4 + q < G.N) va[q] = pa[q];
                }
                if (vec_b) vb = *reinterpret_cast<const f32x4_t*>(pb);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (j0 + c4 + q < G.K) vb[q] = pb[q];
                }
            }
            ra[p] = va; rb[p] = vb;
        }
    };
    f32x4_t acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t colsum = {0.f, 0.f, 0.f, 0.f};
    const bool want_db = (G.db != nullptr) && (j0 == 0);
    const int nk = (rn + TR_BK - 1) / TR_BK;
    if (nk > 0) load(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (want_db) colsum += ra[p];                               // bias gradient in fp32, before the bf16 rounding
            const int r = rr + 16 * p;
            *reinterpret_cast<bf16x4_t*>(As + r * TR_STRIDE + c4) = __builtin_convertvector(ra[p], bf16x4_t);
            *reinterpret_cast<bf16x4_t*>(Bs + r * TR_STRIDE + c4) = __builtin_convertvector(rb[p], bf16x4_t);
        }
        __syncthreads();
        if (kt + 1 < nk) load((kt + 1) * TR_BK);
#pragma unroll
        for (int ks = 0; ks < TR_BK / 32; ++ks) {
            bf16x8_t fa[2], fb[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) fa[mt] = tr_fragment(As, ks * 32, wm + mt * 16, lane);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) fb[nt] = tr_fragment(Bs, ks * 32, wn + nt * 16, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
    float* slab = S > 1 ? a.workspace + (int64_t)split * slab_stride + g_off : nullptr;
    if (want_db) {
        __shared__ float red[GEMM_THREADS * 4];
#pragma unroll
        for (int q = 0; q < 4; ++q) red[tid * 4 + q] = colsum[q];
        __syncthreads();
        if (tid < 64) {
            const int cc = tid / 4, q = tid % 4;
            float sum = 0.f;
            for (int t = cc; t < GEMM_THREADS; t += 16) sum += red[t * 4 + q];
            const int n = i0 + tid;
            if (n < G.N) {
                if (slab) slab[(int64_t)G.N * G.K + n] = sum;
                else G.db[n] = G.accumulate ? G.db[n] + sum : sum;
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
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

// =================================================================================================
// host side
// =================================================================================================
template <bool BF16, int BM, int BN>
static constexpr size_t lds_bytes() {
    size_t tiles = (size_t)(BM + BN) * LdsElem<BF16, GEMM_BK>::stride * sizeof(typename LdsElem<BF16, GEMM_BK>::type);
    size_t red = (size_t)GEMM_THREADS * 4 * sizeof(float);
    size_t out = OutTile<BM, BN>::staged ? (size_t)BM * OutTile<BM, BN>::CS * sizeof(float) : 0;
    size_t m = tiles > red ? tiles : red;
    return m > out ? m : out;
}

static bool pick_big_tiles(int64_t tiles64) {
    // Measured on the C2 launches (M = 4096): 64x64 tiles beat 128x128 even for the 2048-tile level-1 launch (0.097 vs
    // 0.101 ms forward, 0.126 vs 0.153 ms grad-input per step) — these launches wait on loads, and four resident
    // workgroups per CU hide more of that than two.  128x128 (a quarter of the L2 traffic per flop) only once the
    // small tiles already oversubscribe the chip many times over.
    return tiles64 >= 16384;
}

extern "C" int cdc_glinear_fwd(const cdc_lin_fwd_args* a, int32_t prec, void* stream) {
    CDC_CHECK_ARG(a && a->n_groups > 0 && a->n_groups <= CDC_MAX_GROUPS, CDC_E_BADARG, "glinear_fwd: bad group count");
    CDC_CHECK_ARG(prec == CDC_PREC_BF16 || prec == CDC_PREC_F32, CDC_E_BADARG, "glinear_fwd: bad precision");
    CDC_CHECK_ARG(a->drop_p >= 0.f && a->drop_p < 1.f, CDC_E_BADARG, "glinear_fwd: dropout p out of range");
    int64_t t64 = 0, t128 = 0;
    bool fused_bn = false;
    for (int g = 0; g < a->n_groups; ++g) {
        const cdc_lin_group& G = a->g[g];
        CDC_CHECK_ARG(G.x && G.w && G.y && G.M >= 0 && G.N > 0 && G.K > 0 && G.ldx >= G.K && G.ldw >= G.K && G.ldy >= G.N,
                      CDC_E_BADARG, "glinear_fwd: group %d malformed (M=%d N=%d K=%d)", g, G.M, G.N, G.K);
        t64 += cdc_ceil_div(G.M, 64) * cdc_ceil_div(G.N, 64);
        t128 += cdc_ceil_div(G.M, 128) * cdc_ceil_div(G.N, 128);
        if (G.bn_partial) {
            CDC_CHECK_ARG(G.act_cols == 0 && !a->row_offsets && G.bn_col0 >= 0 && G.bn_col0 + G.N <= G.bn_total_c, CDC_E_BADARG,
                          "glinear_fwd: group %d cannot feed BatchNorm partial sums (activation, ragged rows or bad columns)", g);
            fused_bn = true;
        }
    }
    if (t64 == 0) return 0;
    const bool big = !fused_bn && pick_big_tiles(t64);       // the partial sums are written by the LDS-staged 64x64 epilogue
    const int64_t grid = big ? t128 : t64;
    CDC_CHECK_ARG(grid < (1ll << 31), CDC_E_TOOBIG, "glinear_fwd: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (prec == CDC_PREC_BF16) {
        // one slab in flight: a second one slowed the forward down (0.102 vs 0.097 ms per step at 64x64)
        if (big) hipLaunchKernelGGL((k_glinear_fwd<true, 128, 128, 1>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<true, 128, 128>()), st, *a);
        else     hipLaunchKernelGGL((k_glinear_fwd<true, 64, 64, 1>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<true, 64, 64>()), st, *a);
    } else {
        if (big) hipLaunchKernelGGL((k_glinear_fwd<false, 128, 128, 1>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<false, 128, 128>()), st, *a);
        else     hipLaunchKernelGGL((k_glinear_fwd<false, 64, 64, 1>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<false, 64, 64>()), st, *a);
    }
    CDC_LAUNCH_CHECK("glinear_fwd");
    return 0;
}

template <bool WT>
static void launch_bwd_x(const cdc_lin_bwdx_args* a, int32_t prec, bool big, int64_t grid, hipStream_t st) {
    if (prec == CDC_PREC_BF16) {
        // 64x64: two slabs in flight (0.114 vs 0.126 ms per step); 128x128: one (two cost 25 %)
        if (big) hipLaunchKernelGGL((k_glinear_bwd_x<true, 128, 128, WT, 1>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<true, 128, 128>()), st, *a);
        else     hipLaunchKernelGGL((k_glinear_bwd_x<true, 64, 64, WT, WT ? 2 : 1>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<true, 64, 64>()), st, *a);
    } else {
        if (big) hipLaunchKernelGGL((k_glinear_bwd_x<false, 128, 128, WT, 1>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<false, 128, 128>()), st, *a);
        else     hipLaunchKernelGGL((k_glinear_bwd_x<false, 64, 64, WT, 1>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<false, 64, 64>()), st, *a);
    }
}

extern "C" int cdc_glinear_bwd_x(const cdc_lin_bwdx_args* a, int32_t prec, void* stream) {
    CDC_CHECK_ARG(a && a->n_out > 0 && a->n_out <= CDC_MAX_GROUPS && a->n_seg > 0 && a->n_seg <= CDC_MAX_GROUPS, CDC_E_BADARG,
                  "glinear_bwd_x: bad counts");
    CDC_CHECK_ARG(prec == CDC_PREC_BF16 || prec == CDC_PREC_F32, CDC_E_BADARG, "glinear_bwd_x: bad precision");
    int64_t t64 = 0, t128 = 0;
    for (int o = 0; o < a->n_out; ++o) {
        const cdc_bwdx_out& O = a->o[o];
        CDC_CHECK_ARG(O.dx && O.M >= 0 && O.K > 0 && O.lddx >= O.K, CDC_E_BADARG, "glinear_bwd_x: output %d malformed", o);
        t64 += cdc_ceil_div(O.M, 64) * cdc_ceil_div(O.K, 64);
        t128 += cdc_ceil_div(O.M, 128) * cdc_ceil_div(O.K, 128);
    }
    bool all_wt = true;
    for (int s = 0; s < a->n_seg; ++s) {
        const cdc_bwdx_seg& S = a->s[s];
        CDC_CHECK_ARG(S.dz && S.w && S.N > 0 && S.out >= 0 && S.out < a->n_out && S.lddz >= S.N && S.ldw >= a->o[S.out].K,
                      CDC_E_BADARG, "glinear_bwd_x: segment %d malformed", s);
        CDC_CHECK_ARG(!S.wt || S.ldwt >= S.N, CDC_E_BADARG, "glinear_bwd_x: segment %d transposed copy malformed", s);
        all_wt = all_wt && S.wt != nullptr;
        if (S.wt_bf16)
            CDC_CHECK_ARG(prec == CDC_PREC_BF16 && S.wt && S.ldwt % 8 == 0 && (((uintptr_t)S.wt | (uintptr_t)S.dz) & 15) == 0 &&
                              S.lddz % 4 == 0, CDC_E_BADARG, "glinear_bwd_x: segment %d: bf16 transposed copy not eligible", s);
    }
    for (int s = 0; s < a->n_seg; ++s)
        CDC_CHECK_ARG(!a->s[s].wt_bf16 || all_wt, CDC_E_BADARG, "glinear_bwd_x: bf16 transposed copies need every segment transposed");
    if (t64 == 0) return 0;
    const bool big = pick_big_tiles(t64);
    const int64_t grid = big ? t128 : t64;
    CDC_CHECK_ARG(grid < (1ll << 31), CDC_E_TOOBIG, "glinear_bwd_x: grid too large");
    if (all_wt) launch_bwd_x<true>(a, prec, big, grid, (hipStream_t)stream);
    else launch_bwd_x<false>(a, prec, big, grid, (hipStream_t)stream);
    CDC_LAUNCH_CHECK("glinear_bwd_x");
    return 0;
}

// dst[c, r] = src[r, c] for a list of matrices: 32x32 tiles through LDS (padded), one launch for the whole list
__global__ void __launch_bounds__(256) k_transpose_multi(const cdc_transpose_args a) {
    __shared__ float tile[32][33];
    int blk = blockIdx.x;
    const int ti = find_group<false>(a.n, blk, [&](int l) { return ((a.t[l].rows + 31) / 32) * ((a.t[l].cols + 31) / 32); },
                                     [](int) { return (int64_t)0; }, blk, nullptr);
    if (ti < 0) return;
    const int tc = (a.t[ti].cols + 31) / 32;
    const int rows = a.t[ti].rows, cols = a.t[ti].cols;
    const int r0 = (blk / tc) * 32, c0 = (blk % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < rows && c < cols) ? a.t[ti].src[(int64_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < cols && r < rows) {
            if ((a.bf16_mask >> ti) & 1ull) reinterpret_cast<__bf16*>(a.t[ti].dst)[(int64_t)c * rows + r] = (__bf16)tile[tx][k];
            else a.t[ti].dst[(int64_t)c * rows + r] = tile[tx][k];
        }
    }
}
extern "C" int cdc_transpose_multi(const cdc_transpose_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->n > 0 && a->n <= CDC_MAX_TENSORS, CDC_E_BADARG, "transpose_multi: bad count");
    int64_t blocks = 0;
    for (int i = 0; i < a->n; ++i) {
        CDC_CHECK_ARG(a->t[i].src && a->t[i].dst && a->t[i].rows > 0 && a->t[i].cols > 0, CDC_E_BADARG, "transpose_multi: tensor %d malformed", i);
        blocks += cdc_ceil_div(a->t[i].rows, 32) * cdc_ceil_div(a->t[i].cols, 32);
    }
    hipLaunchKernelGGL(k_transpose_multi, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("transpose_multi");
    return 0;
}

extern "C" int cdc_glinear_bwd_w_pair(const cdc_lin_bwdw_args* wide, const cdc_lin_bwdw_args* narrow, const cdc_lin_bwdw_args* tabs_dev,
                                      void* stream) {
    CDC_CHECK_ARG(wide && narrow && tabs_dev, CDC_E_BADARG, "glinear_bwd_w_pair: null argument");
    int64_t slab[2] = {0, 0};
    for (int w = 0; w < 2; ++w) {
        const cdc_lin_bwdw_args* a = w ? narrow : wide;
        CDC_CHECK_ARG(a->n_groups > 0 && a->n_groups <= CDC_MAX_GROUPS && !a->row_offsets && a->split_k <= 256, CDC_E_BADARG,
                      "glinear_bwd_w_pair: bad group count / ragged rows / split");
        CDC_CHECK_ARG(a->split_k <= 1 || (a->workspace && a->defer_reduce), CDC_E_BADARG,
                      "glinear_bwd_w_pair: a split launch of the pair leaves its slabs to the consumer (defer_reduce)");
        for (int g = 0; g < a->n_groups; ++g) {
            const cdc_bwdw_group& G = a->g[g];
            CDC_CHECK_ARG(G.dw && G.M >= 0 && G.N > 0 && G.K > 0 && G.lddw >= G.K && !(a->defer_reduce && G.accumulate), CDC_E_BADARG,
                          "glinear_bwd_w_pair: group %d malformed", g);
            slab[w] += (int64_t)G.N * G.K + G.N;
        }
    }
    return g2_launch_bwd_w_dual(wide, narrow, tabs_dev, slab[0], slab[1], (hipStream_t)stream);
}
// the same, and the slabs of both classes summed into the gradients by ONE more launch (callers that need the reduced gradient right
// away: data parallelism all-reduces it; autograd hands it on) — two launches where cdc_glinear_bwd_w twice issues four
extern "C" int cdc_glinear_bwd_w_pair_reduce(const cdc_lin_bwdw_args* wide, const cdc_lin_bwdw_args* narrow, const cdc_lin_bwdw_args* tabs_dev,
                                             void* stream) {
    const int rc = cdc_glinear_bwd_w_pair(wide, narrow, tabs_dev, stream);
    if (rc) return rc;
    CDC_CHECK_ARG(wide->split_k > 1 && narrow->split_k > 1, CDC_E_BADARG, "glinear_bwd_w_pair_reduce: both classes must be split (else cdc_glinear_bwd_w_pair)");
    int64_t slab[2] = {0, 0};
    for (int w = 0; w < 2; ++w) {
        const cdc_lin_bwdw_args* a = w ? narrow : wide;
        for (int g = 0; g < a->n_groups; ++g) slab[w] += (int64_t)a->g[g].N * a->g[g].K + a->g[g].N;
    }
    const int b0 = (int)std::min<int64_t>(cdc_ceil_div(slab[0], 1024), 4096), b1 = (int)std::min<int64_t>(cdc_ceil_div(slab[1], 1024), 4096);
    hipLaunchKernelGGL(k_bwd_w_reduce_dual, dim3(b0 + b1), dim3(256), 0, (hipStream_t)stream, tabs_dev, slab[0], slab[0], slab[1], slab[1], b0);
    CDC_LAUNCH_CHECK("bwd_w_reduce_dual");
    return 0;
}

extern "C" int cdc_glinear_bwd_w(const cdc_lin_bwdw_args* a, int32_t prec, void* stream) {
    CDC_CHECK_ARG(a && a->n_groups > 0 && a->n_groups <= CDC_MAX_GROUPS, CDC_E_BADARG, "glinear_bwd_w: bad group count");
    CDC_CHECK_ARG(prec == CDC_PREC_BF16 || prec == CDC_PREC_F32, CDC_E_BADARG, "glinear_bwd_w: bad precision");
    CDC_CHECK_ARG(a->split_k <= 1 || a->workspace, CDC_E_BADARG, "glinear_bwd_w: split_k needs a workspace");
    CDC_CHECK_ARG(a->split_k <= 256, CDC_E_BADARG, "glinear_bwd_w: split_k too large");
    int64_t t64 = 0, slab = 0;
    for (int g = 0; g < a->n_groups; ++g) {
        const cdc_bwdw_group& G = a->g[g];
        CDC_CHECK_ARG(G.dz && G.x && G.dw && G.M >= 0 && G.N > 0 && G.K > 0 && G.lddz >= G.N && G.ldx >= G.K && G.lddw >= G.K,
                      CDC_E_BADARG, "glinear_bwd_w: group %d malformed", g);
        CDC_CHECK_ARG(!(a->defer_reduce && G.accumulate), CDC_E_BADARG, "glinear_bwd_w: group %d accumulates, its slabs cannot be left to the consumer", g);
        t64 += cdc_ceil_div(G.N, 64) * cdc_ceil_div(G.K, 64);
        slab += (int64_t)G.N * G.K + G.N;
    }
    const int S = a->split_k > 1 ? a->split_k : 1;
    const int64_t grid = t64 * S;
    CDC_CHECK_ARG(grid < (1ll << 31), CDC_E_TOOBIG, "glinear_bwd_w: grid too large");
    hipStream_t st = (hipStream_t)stream;
    bool shadows = prec == CDC_PREC_BF16 && !a->row_offsets;
    for (int g = 0; g < a->n_groups && shadows; ++g) shadows = a->g[g].dzh != nullptr && a->g[g].xh != nullptr;
    if (shadows) {
        const int rc = g2_launch_bwd_w(a, slab, st);
        if (rc) return rc;
    } else if (prec == CDC_PREC_BF16)
        hipLaunchKernelGGL(k_glinear_bwd_w_tr, dim3(grid), dim3(GEMM_THREADS), 0, st, *a, slab);
    else
        hipLaunchKernelGGL((k_glinear_bwd_w<false, 64, 64>), dim3(grid), dim3(GEMM_THREADS), (lds_bytes<false, 64, 64>()), st, *a, slab);
    CDC_LAUNCH_CHECK("glinear_bwd_w");
    if (S > 1 && !a->defer_reduce) {
        int64_t total = 0;
        for (int g = 0; g < a->n_groups; ++g) total += (int64_t)a->g[g].N * a->g[g].K + a->g[g].N;
        int blocks = (int)std::min<int64_t>(cdc_ceil_div(total, 1024), 4096);       // four floats per thread (one 16-byte piece when aligned)
        hipLaunchKernelGGL(k_bwd_w_reduce, dim3(blocks), dim3(256), 0, st, *a, slab, total);
        CDC_LAUNCH_CHECK("bwd_w_reduce");
    }
    return 0;
}
