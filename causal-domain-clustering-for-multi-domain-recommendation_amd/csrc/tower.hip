// tower.hip — the towers of a multi-tower model in ONE forward and ONE backward launch (round 4).
//
// Reference: BaseModel.tower_forward (model/layer.py:35-56) over MultiLayerPerceptron(H0, (H1, H2), dropout, output_layer=True)
// towers (model/layer.py:178-206): per tower Linear -> BatchNorm1d -> ReLU -> Dropout -> Linear -> BatchNorm1d -> ReLU -> Dropout ->
// Linear(->1), `+= other` (the wide term, model/layer.py:122-126), Sigmoid; in the training step BCELoss(mean) on the row's own
// tower (run.py:484,723) and the whole backward of the chain.
//
// Before (round 3): 13 launches of 5.5-15 us each for 0.3 % of the step's flops — two bf16 contractions, two BatchNorm
// normalisations, the head, and the mirror chain (head backward + ordered final, 2 x (statistics, normalise), two grad-input
// contractions): 111 us of the 305 us main chain of a C2 step (profiles/round3/step_timeline.txt).  Every one of them only
// re-reads what its predecessor wrote; what forces launch boundaries is BatchNorm's batch statistics — a sum over ALL rows
// between any two layers.  Here a workgroup owns CDC_TOWER_ROWS rows of one tower and keeps them in LDS from the first
// contraction to the sigmoid (forward) and from the loss gradient to the input gradient (backward); the column sums are
// exchanged among the workgroups of a tower INSIDE the launch:
//     publish partial sums (write-through stores; a published value is never all-zero bits) ... every workgroup loads ALL
//     partials of its tower (write-through loads, a batch in flight together), loads again while any of them is still an empty
//     slot (bounded), and adds them up in the same fixed order.
// Every 4- or 8-byte value is its own flag (aligned accesses of that size are single-copy atomic): no arrival counter, no wait for
// the stores' acknowledgement in front of one, no ordering between different words to rely on, no fence, no reliance on placement
// (the first version of this round used MI355X_MICROARCH.md's counter hand-off — publish, drain, "ONE lane ... agent-scope atomic
// add", "sc1 load poll of that counter", gather: 5.5-6.5 us per exchange against 3-4 now).  Record arrays exist twice, chosen by a
// launch count: a workgroup clears its own slots of the other copy at entry (tw_nz / tw_empty / tw_poll / tw_finish below).  Every
// word another workgroup reads is stored and loaded through tw_st/tw_ld (relaxed agent-scope atomics = global_store/load ... sc1),
// nothing else.  Work that needs no exchange runs while the others' sums arrive: the wide term's dot products (forward), the wide
// term's gradients (backward).  A poll that runs out sets CDC_TOWER_ERR_TIMEOUT in *err and a give-up word that ends the other
// polls of the launch at once, so the grid always drains.  Residency: at most 256 workgroups of 256 threads, one per CU, <= 128
// VGPRs — they fit beside the background replay slice (2 x 64 VGPRs per SIMD, no LDS); kernels of the other queue that hold a
// CU finish without depending on this launch.
//
// Arithmetic = that of the launches replaced (csrc/gemm2.hip, csrc/rowops.hip k_bn_*_v4, csrc/head.hip): bf16 operands rounded
// to nearest even from fp32, v_mfma_f32_16x16x32_bf16 over K in ascending 32-wide steps, bias added in fp32 afterwards; BatchNorm
// statistics as fp64 sums per 64-row chunk (wave q adds rows 16q..16q+15, quarters in order) and chunks added as NP interleaved
// partial sums in order — the bits of cdc_gemm_bf16_nt's statistics epilogue + cdc_bn_fwd; (x - mean) * invstd * gamma + beta;
// the 32-bit dropout stream of the BatchNorm launches (stream 64 + tower, one hash per column pair).  The backward's column
// sums are fp64 sums per row block in a fixed order of their own (not the bits of k_bn_bwd_stats_v4's shuffle tree).
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define TW_THREADS 256
#define TW_ROWS CDC_TOWER_ROWS
#define TW_KARG __attribute__((address_space(4)))
#define TW_GLOBAL __attribute__((address_space(1)))
#define TW_SLAB (TW_ROWS * 128)              /* one 64-column bf16 slab of a 128-row operand tile */
#define TW_SPIN_LIMIT 200000u                /* polls of ~1 us: a wait gives up after ~0.2 s */
#define TW_HDR_BYTES 8192
#define TW_LINE 32                           /* ints per 128-byte line: every counter on a line of its own */
// header lines
#define TW_FEPOCH 0                          /* launches of the forward body so far: its parity chooses the record copy */
#define TW_MUTE 1                            /* tests only: workgroup (value - 1) of the next forward publishes nothing (the others give up) */
#define TW_FDONE 8
#define TW_BEPOCH 9                          /* the same for the backward body */
#define TW_BDONE 14
#define TW_POISON 15
// the split form (data parallel, cdc_tower_dp): one arrival counter per tower and exchange — the LAST workgroup to arrive adds the
// tower's partials up and puts the counter back to zero (nobody waits)
#define TW_S(e, t) (16 + 4 * (e) + (t))
#ifndef TW_TRACE
#define TW_TRACE 0          /* 1 (probe builds only): thread 0 of workgroup 0 stamps the 100 MHz wall clock at every phase boundary
                               into header bytes [4096, 8192): forward stamps [0, 16), backward stamps [16, 32) */
#endif
#if TW_TRACE
#define TW_STAMP(i) do { if (blockIdx.x == TW_TRACE - 1 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.workspace) + 4096)[(i)] = wall_clock64(); } while (0)
#else
#define TW_STAMP(i) do { } while (0)
#endif

// ---- inter-workgroup accessors: relaxed agent-scope atomics on GLOBAL pointers (global_load / global_store ... sc1)
template <typename T> __device__ __forceinline__ T tw_ld(const T* p) {
    return __hip_atomic_load((const TW_GLOBAL T*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ void tw_st(T* p, T v) {
    __hip_atomic_store((TW_GLOBAL T*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int tw_add(int* p, int v) {
    return __hip_atomic_fetch_add((TW_GLOBAL int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tw_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- the in-launch exchange of the monolithic kernels (cdc_tower_fwd / _bwd / _step): SELF-VALIDATING records.
// A published value is never all-zero bits (tw_nz: +0 goes out as -0, which adds like +0), an unpublished slot is all-zero bits.
// A reader loads the records it needs (a batch in flight together) and loads again while any of them is still empty: every 4- or
// 8-byte value is its own flag (naturally aligned accesses of that size are single-copy atomic), so there is no arrival counter, no
// wait for the write-through stores to be acknowledged before one, and no ordering between different values to rely on.
// Two copies of every record array, chosen by the parity of a launch count in the header (TW_FEPOCH / TW_BEPOCH, advanced by the
// last workgroup through the body): a workgroup publishes into copy e & 1 and, at entry, clears ITS OWN slots of copy (e + 1) & 1 —
// whose readers ran in the previous launch and whose next readers run in the next one.  A zeroed workspace is a valid start, and a
// launch that gave up (bounded polls, below) leaves nothing behind that the launches after it could trip over.
__device__ __forceinline__ double tw_nz(double v) { return v == 0.0 ? -0.0 : v; }
__device__ __forceinline__ float tw_nz(float v) { return v == 0.f ? -0.f : v; }
__device__ __forceinline__ bool tw_empty(double v) { return __double_as_longlong(v) == 0ll; }
__device__ __forceinline__ bool tw_empty(float v) { return __float_as_uint(v) == 0u; }
// one more round of a poll loop: false = give up (~0.2-0.3 s of polls, or another workgroup has given up: the launch's results are
// void and *err says so)
__device__ __forceinline__ bool tw_spin(unsigned& spins, int* hdr, int32_t* err) {
    int* poison = hdr + TW_POISON * TW_LINE;
    __builtin_amdgcn_s_sleep(2);
    ++spins;
    if ((spins & 63u) == 0u && tw_ld(poison) != 0) return false;
    if (spins >= TW_SPIN_LIMIT) {
        tw_st(poison, 1);
        if (err) __hip_atomic_fetch_or((TW_GLOBAL int32_t*)err, (int32_t)CDC_TOWER_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    return true;
}
// one value per lane (inactive lanes pass p = nullptr): loaded again while any lane of the wave still sees an empty slot
template <typename T>
__device__ __forceinline__ T tw_poll(const T* p, int* hdr, int32_t* err) {
    T v = p ? tw_ld(p) : (T)1;
    unsigned spins = 0;
    while (__any(tw_empty(v))) {
        if (!tw_spin(spins, hdr, err)) break;
        if (p) v = tw_ld(p);
    }
    return v;
}
// the last workgroup through `done_line` advances the direction's launch count (the next launch publishes into the other copy) and
// puts the give-up word back (all others are past their last poll: a workgroup adds here after it)
__device__ __forceinline__ void tw_finish(int* hdr, int done_line, int epoch_line, int n_wg, int epoch) {
    const int old = tw_add(hdr + done_line * TW_LINE, 1);
    if (old == n_wg - 1) {
        tw_st(hdr + done_line * TW_LINE, 0);
        tw_st(hdr + TW_POISON * TW_LINE, 0);
        tw_st(hdr + epoch_line * TW_LINE, epoch + 1);
    }
}

// ---- workspace layout (bytes from the start of `workspace`)
struct TwLayout {
    int64_t st1, st2;        // forward: per 64-row chunk column sums (x, x^2): [chunk][n_tower*H][2] doubles
    int64_t b2, b1;          // backward: per row block column sums (dz, dz*xhat): [block][n_tower*H][2] doubles
    int64_t hd;              // backward: head weight-gradient partials [block][n_tower][H2 + 4] floats (slot H2 = bias)
    int64_t wd;              // backward: wide weight-gradient partials [n_tower*block][WD_LD] floats (slot wide_K = bias)
    int64_t loss;            // backward: [workgroup] doubles (a row's loss is added by the workgroup of its OWN tower)
    int64_t wide;            // forward: the wide term of every row [M] floats (each row formed by ONE workgroup, read by all towers)
    int64_t copy;            // the record arrays above exist twice: copy 1 sits `copy` bytes behind copy 0 (the split form uses copy 0)
    int64_t total;
    int wd_ld;
};
__host__ __device__ inline TwLayout tw_layout(int n_tower, int H1, int H2, int64_t M, int wide_K) {
    TwLayout L;
    const int64_t nchunk = (M + 63) / 64, G = (M + TW_ROWS - 1) / TW_ROWS;
    int64_t o = TW_HDR_BYTES;
    L.st1 = o; o += nchunk * n_tower * H1 * 16;
    L.st2 = o; o += nchunk * n_tower * H2 * 16;
    L.b2 = o; o += G * n_tower * H2 * 16;
    L.b1 = o; o += G * n_tower * H1 * 16;
    L.hd = o; o += G * n_tower * (H2 + 4) * 4;
    L.wd_ld = (wide_K + 1 + 3) / 4 * 4;
    L.wd = o; o += (int64_t)n_tower * G * L.wd_ld * 4;
    o = (o + 15) / 16 * 16;
    L.loss = o; o += (int64_t)n_tower * G * 8;
    L.wide = o; o += (M + 3) / 4 * 16;
    o = (o + 127) / 128 * 128;
    L.copy = o - TW_HDR_BYTES;
    L.total = TW_HDR_BYTES + 2 * L.copy;
    return L;
}

// ---- operand tiles in LDS: rows of 128 bytes (64 bf16), the 16-byte chunk index XORed by (row & 7) as in csrc/gemm2.hip
__device__ __forceinline__ void tw_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const TW_GLOBAL void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// rows [0, n_rows) of a row-major bf16 matrix (row stride ld elements, 64-column slab `slab`) -> tile; n_rows a multiple of 32;
// rows >= valid are fetched from row valid-1 (their products are never stored)
__device__ __forceinline__ void tw_load_tile(const __bf16* src, int64_t ld, int slab, int n_rows, int valid, unsigned char* tile,
                                             int wave, int lane) {
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
    const int per_wave = n_rows / 4;
    for (int q = 0; q < per_wave / 8; ++q) {
        const int r0 = wave * per_wave + q * 8;
        int r = r0 + lrow;
        r = r < valid ? r : valid - 1;
        tw_glds16(src + (int64_t)r * ld + slab * 64 + lchunk * 8, tile + r0 * 128);
    }
}
__device__ __forceinline__ bf16x8_t tw_frag(const unsigned char* tile, int row, int chunk) {
    return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}
__device__ __forceinline__ void tw_put8(unsigned char* tile, int row, int chunk, const float (&v)[8]) {
    bf16x8_t h;
#pragma unroll
    for (int q = 0; q < 8; ++q) h[q] = (__bf16)v[q];
    *reinterpret_cast<bf16x8_t*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4)) = h;
}

// C[128 x NT*16] += A[128 x KS*32] * B^T, wave w owns rows 32w..32w+31; A, B tiles in LDS (slabs of 64 columns)
template <int NT, int KS>
__device__ __forceinline__ void tw_mfma(const unsigned char* A, const unsigned char* Bt, int b_slab_bytes, f32x4_t (&acc)[2][NT], int wave, int lane) {
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const unsigned char* As = A + (ks >> 1) * TW_SLAB;
        const unsigned char* Bs = Bt + (ks >> 1) * b_slab_bytes;
        const int chunk = (ks & 1) * 4 + fq;
        bf16x8_t af[2], bfr[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[mt] = tw_frag(As, wave * 32 + mt * 16 + frow, chunk);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bfr[nt] = tw_frag(Bs, nt * 16 + frow, chunk);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
    }
}
template <int NT>
__device__ __forceinline__ void tw_acc_to_tile(const f32x4_t (&acc)[2][NT], float* ct, int cs, int wave, int lane) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ct[(wave * 32 + mt * 16 + (lane >> 4) * 4 + r) * cs + nt * 16 + (lane & 15)] = acc[mt][nt][r];
}

// ---- forward statistics of one layer: this workgroup's (at most two) 64-row chunks -> workspace, in the order of
// cdc_gemm_bf16_nt's statistics epilogue (csrc/gemm2.hip): wave q adds rows 16q..16q+15 of the chunk, quarters in order.
// All threads call (barriers inside).  Waves 0 and 1 store; both have drained when the call returns.
template <int C, bool DRAIN = true>
__device__ __forceinline__ void tw_fwd_chunk_sums(const float* ct, int cs, int row0, int M, double* quarter /*[2][4][64][2]*/, double* ws,
                                                  int total_c, int col0, int wave, int lane, bool publish = true) {
    static_assert(TW_ROWS == 128, "two 64-row chunks per block");
    const int rows0 = min(64, M - row0), rows1 = min(64, M - row0 - 64);          // rows of the two chunks (rows1 may be <= 0)
    double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
    if (lane < C) {
#pragma unroll 4
        for (int r = wave * 16; r < wave * 16 + 16; ++r) {                         // two independent chains: the chunks' sums interleave
            if (r < rows0) { const double x = (double)ct[r * cs + lane]; s1[0] += x; s2[0] += x * x; }
            if (r < rows1) { const double x = (double)ct[(64 + r) * cs + lane]; s1[1] += x; s2[1] += x * x; }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) { quarter[((h * 4 + wave) * 64 + lane) * 2] = s1[h]; quarter[((h * 4 + wave) * 64 + lane) * 2 + 1] = s2[h]; }
    __syncthreads();
    if (publish && wave < 2 && lane < C && (wave == 0 || rows1 > 0)) {              // wave h publishes chunk h
        const int h = wave;
        double* p = ws + ((int64_t)(row0 / 64 + h) * total_c + col0 + lane) * 2;
        const double* q = quarter + (h * 4 * 64 + lane) * 2;
        tw_st(p, tw_nz(((q[0] + q[64 * 2]) + q[2 * 64 * 2]) + q[3 * 64 * 2]));
        tw_st(p + 1, tw_nz(((q[1] + q[64 * 2 + 1]) + q[2 * 64 * 2 + 1]) + q[3 * 64 * 2 + 1]));
    }
    if (DRAIN) tw_drain();                                                         // (split form: every wave, also what it stored before this call)
    __syncthreads();                                                               // (`quarter` is free again)
}
// Sum over `n_parts` published partial records of the tower's C columns, as cdc_bn_fwd's bn_sum_partials_v does: NP = 256 / C
// threads per column take records pt, pt + NP, ... and the NP sums are added in order.  out[2][C] doubles in LDS.  All threads call.
// POLL (hdr != nullptr): the monolithic kernels' exchange — a batch is loaded again while any of its records is still unpublished.
template <int C>
__device__ __forceinline__ void tw_gather_sums(const double* ws, int n_parts, int total_c, int col0, double* part /*[2][NP][C]*/, double* out, int tid,
                                               int* hdr = nullptr, int32_t* err = nullptr) {
    constexpr int NP = TW_THREADS / C, BATCH = 16;
    const int j = tid % C, pt = tid / C;
    double a1 = 0.0, a2 = 0.0;
    unsigned spins = 0;
    for (int k0 = pt; k0 < n_parts; k0 += NP * BATCH) {                  // a batch's 32 loads are in flight together (one round trip
        double v1[BATCH], v2[BATCH];                                     // to L2 per batch, not per record); the adds keep the order
        for (;;) {
            bool bad = false;
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int k = k0 + b * NP;
                const double* p = ws + ((int64_t)(k < n_parts ? k : pt) * total_c + col0 + j) * 2;
                v1[b] = tw_ld(p); v2[b] = tw_ld(p + 1);
            }
            if (!hdr) break;
#pragma unroll
            for (int b = 0; b < BATCH; ++b) bad = bad || tw_empty(v1[b]) || tw_empty(v2[b]);
            if (!__any(bad)) break;
            if (!tw_spin(spins, hdr, err)) break;
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b)
            if (k0 + b * NP < n_parts) { a1 += v1[b]; a2 += v2[b]; }
    }
    part[(0 * NP + pt) * C + j] = a1; part[(1 * NP + pt) * C + j] = a2;
    __syncthreads();
    if (tid < C) {
        double b1 = 0.0, b2 = 0.0;
#pragma unroll
        for (int q = 0; q < NP; ++q) { b1 += part[(0 * NP + q) * C + tid]; b2 += part[(1 * NP + q) * C + tid]; }
        out[tid] = b1; out[C + tid] = b2;
    }
    __syncthreads();
}
// batch mean / biased variance -> col_mean, col_inv (LDS); the tower's first workgroup also writes what the module keeps
template <int C>
__device__ __forceinline__ void tw_finish_stats(const double* sums, int Ms, float eps, float momentum, bool writer, float* save_mean,
                                                float* save_invstd, float* running_mean, float* running_var, int64_t* nbt, float* col_mean,
                                                float* col_inv, int tid) {
    if (tid < C) {
        const double s1 = sums[tid], s2 = sums[C + tid];
        const double mu = s1 / Ms;
        double var = s2 / Ms - mu * mu;
        if (var < 0.0) var = 0.0;
        const float mean = (float)mu;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (writer) {
            if (save_mean) save_mean[tid] = mean;
            if (save_invstd) save_invstd[tid] = invstd;
            if (running_mean) {
                const double unbiased = Ms > 1 ? var * ((double)Ms / (double)(Ms - 1)) : var;
                running_mean[tid] = (1.f - momentum) * running_mean[tid] + momentum * mean;
                running_var[tid] = (1.f - momentum) * running_var[tid] + momentum * (float)unbiased;
            }
        }
        col_mean[tid] = mean; col_inv[tid] = invstd;
    }
    if (writer && tid == 0 && nbt) *nbt += 1;
    __syncthreads();
}
// normalise + ReLU + dropout of 8 neighbouring columns (k_bn_apply_v4's arithmetic and dropout decisions)
__device__ __forceinline__ void tw_bn_apply8(float (&v)[8], const float* col_mean, const float* col_inv, const float* gamma, const float* beta, int c,
                                             bool relu, float drop_p, float keep_scale, uint32_t thr16, uint32_t seed32, int grow) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float x = (v[q] - col_mean[c + q]) * col_inv[c + q] * gamma[c + q] + beta[c + q];
        if (relu) x = fmaxf(x, 0.f);
        v[q] = x;
    }
    if (drop_p > 0.f) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const uint32_t h = g2_drop_bits(seed32, grow, (c >> 1) + pr);
            v[2 * pr] = (h & 0xFFFFu) < thr16 ? 0.f : v[2 * pr] * keep_scale;
            v[2 * pr + 1] = (h >> 16) < thr16 ? 0.f : v[2 * pr + 1] * keep_scale;
        }
    }
}

// the small per-column vectors of a tower, staged in LDS at kernel entry (a global load at the point of use is a round trip of
// its own on the critical path of a workgroup): offsets in floats
#define TW_PAR_B1 0          /* bias, gamma, beta of layer 1 (64 each) — backward: save_mean, gamma * save_invstd, save_invstd */
#define TW_PAR_G1 64
#define TW_PAR_BE1 128
#define TW_PAR_B2 192        /* the same of layer 2 */
#define TW_PAR_G2 256
#define TW_PAR_BE2 320
#define TW_PAR_WO 384        /* head weight [H2] and bias [1] */
#define TW_PAR_FLOATS 456

// =================================================================================================
// forward
// =================================================================================================
template <int NK0, int H1, int H2>
struct TwFwdCfg {
    static constexpr int CS1 = H1 + 4, CS2 = H2 + 4;
    static constexpr int XS = 0;                                     // [NK0][128][64] bf16
    static constexpr int W1S = XS + NK0 * TW_SLAB;                   // [NK0][H1][64] bf16
    static constexpr int W2S = W1S + NK0 * H1 * 128;                 // [H1/64][H2][64] bf16
    static constexpr int A1S = W2S + (H1 / 64) * H2 * 128;           // [H1/64][128][64] bf16
    static constexpr int CT = A1S + (H1 / 64) * TW_SLAB;             // [128][CS1] fp32 (layer 2: [128][CS2])
    static constexpr int PART = CT + TW_ROWS * CS1 * 4;              // [2][8][64] doubles (also the chunk quarters)
    static constexpr int SUMS = PART + 2 * 8 * 64 * 8;               // [2][64] doubles
    static constexpr int CMEAN = SUMS + 2 * 64 * 8;                  // [64] floats
    static constexpr int CINV = CMEAN + 64 * 4;
    static constexpr int WIDE = CINV + 64 * 4;                       // [128] floats
    static constexpr int PAR = WIDE + TW_ROWS * 4;                   // the layers' small vectors, fetched at entry (TW_PAR_*)
    static constexpr int SMEM = PAR + TW_PAR_FLOATS * 4;
};

// `keep` (cdc_tower_step only, else null): LDS floats handed to the backward body — [mean1 64 | invstd1 64 | mean2 64 | invstd2 64 |
// the rows' logit gradients 128 | "this workgroup forms the row's wide gradient" flags 128 (ints) | the rows' loss terms 128 doubles]:
// the head has a row's output in a register, so the loss and its gradient (the fused-BCE form: a row's OWN tower only) are formed
// there — the backward body starts without its round trips to the labels, the tower column and the outputs
#define TW_KEEP_D 256
#define TW_KEEP_MINE 384
#define TW_KEEP_LOSS 512
#define TW_KEEP_FLOATS 768
template <int NK0, int H1, int H2>
__device__ __forceinline__ void tw_fwd_body(const TW_KARG cdc_tower_args& a, unsigned char* smem, float* keep) {
    typedef TwFwdCfg<NK0, H1, H2> Cfg;
    static_assert(H1 % 64 == 0 && H1 <= 64 && H2 % 16 == 0 && H2 <= 64 && H2 % 8 == 0, "instantiated shapes");
#if TW_TRACE
    const unsigned long long t_top = wall_clock64();                   // (before the first read of the argument block)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = (int)a.M, n_tower = a.n_tower;
    const int G = (M + TW_ROWS - 1) / TW_ROWS;
    const int t = (int)blockIdx.x / G, jb = (int)blockIdx.x - t * G;
    const int row0 = jb * TW_ROWS, rows = min(TW_ROWS, M - row0);
    const TW_KARG cdc_tower_desc& T = a.t[t];
    int* hdr = reinterpret_cast<int*>(a.workspace);
    const TwLayout L = tw_layout(n_tower, H1, H2, M, a.wide_x ? a.wide_K : 0);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(a.workspace);
    int32_t* err = a.err;
    const bool writer = jb == 0;
    // the record copy of this launch; this workgroup's slots of the OTHER copy are cleared for the next launch
    const int epoch = __builtin_amdgcn_readfirstlane(tw_ld(hdr + TW_FEPOCH * TW_LINE));
    const bool mute = __builtin_amdgcn_readfirstlane(tw_ld(hdr + TW_MUTE * TW_LINE)) == (int)blockIdx.x + 1;
    unsigned char* const rec = wsb + (epoch & 1) * L.copy;
    {
        unsigned char* const nxt = wsb + ((epoch + 1) & 1) * L.copy;
        const int c0 = row0 / 64;                                        // this block's two 64-row chunks
        for (int i = tid; i < 2 * 2 * H1; i += TW_THREADS) {
            const int h = i / (2 * H1), k = i - h * 2 * H1;
            if ((c0 + h) * 64 < M) tw_st(reinterpret_cast<double*>(nxt + L.st1) + ((int64_t)(c0 + h) * n_tower * H1 + t * H1) * 2 + k, 0.0);
        }
        for (int i = tid; i < 2 * 2 * H2; i += TW_THREADS) {
            const int h = i / (2 * H2), k = i - h * 2 * H2;
            if ((c0 + h) * 64 < M) tw_st(reinterpret_cast<double*>(nxt + L.st2) + ((int64_t)(c0 + h) * n_tower * H2 + t * H2) * 2 + k, 0.0);
        }
        if (a.wide_x)
            for (int i = t + n_tower * tid; i < rows; i += n_tower * TW_THREADS) tw_st(reinterpret_cast<float*>(nxt + L.wide) + row0 + i, 0.f);
    }

    unsigned char* XS = smem + Cfg::XS;
    unsigned char* W1S = smem + Cfg::W1S;
    unsigned char* W2S = smem + Cfg::W2S;
    unsigned char* A1S = smem + Cfg::A1S;
    float* ct = reinterpret_cast<float*>(smem + Cfg::CT);
    double* part = reinterpret_cast<double*>(smem + Cfg::PART);
    double* sums = reinterpret_cast<double*>(smem + Cfg::SUMS);
    float* col_mean = reinterpret_cast<float*>(smem + Cfg::CMEAN);
    float* col_inv = reinterpret_cast<float*>(smem + Cfg::CINV);
    float* wide_s = reinterpret_cast<float*>(smem + Cfg::WIDE);
    float* par_s = reinterpret_cast<float*>(smem + Cfg::PAR);
    {
        float v = 0.f;
        const int k = tid & 63, which = tid >> 6;
        if (which == 0 && k < H1) { par_s[TW_PAR_B1 + k] = T.l1.bias[k]; par_s[TW_PAR_G1 + k] = T.l1.gamma[k]; par_s[TW_PAR_BE1 + k] = T.l1.beta[k]; }
        if (which == 1 && k < H2) { par_s[TW_PAR_B2 + k] = T.l2.bias[k]; par_s[TW_PAR_G2 + k] = T.l2.gamma[k]; par_s[TW_PAR_BE2 + k] = T.l2.beta[k]; }
        if (which == 2 && k < H2) par_s[TW_PAR_WO + k] = T.wo[k];
        if (which == 2 && k == H2) par_s[TW_PAR_WO + H2] = T.bo ? T.bo[0] : 0.f;
        (void)v;
    }

    // cdc_tower_step: the head thread of a row (tid = 2 row) fetches the row's tower column and label now, for the loss at the end
    int k_own = 0;
    float k_tgt = 0.f;
    if (keep && (tid & 1) == 0 && (tid >> 1) < rows) {
        const int64_t r = row0 + (tid >> 1);
        int64_t c = a.bce_group ? a.bce_group[r] : 0;
        if (c < 0 || c >= n_tower) c = 0;
        k_own = (int)c;
        k_tgt = a.bce_y_i16 ? (float)a.bce_y_i16[r] : a.bce_y_f32[r];
    }

    // ---- operands of both contractions: global -> LDS, all in flight at once
    {
        const __bf16* xh = reinterpret_cast<const __bf16*>(T.xh) + (int64_t)row0 * T.ldxh;
        const __bf16* w1 = reinterpret_cast<const __bf16*>(T.l1.wh);
        const __bf16* w2 = reinterpret_cast<const __bf16*>(T.l2.wh);
        const int64_t ldx = T.ldxh, ldw1 = T.l1.ldwh, ldw2 = T.l2.ldwh;
#pragma unroll
        for (int s = 0; s < NK0; ++s) {
            tw_load_tile(xh, ldx, s, TW_ROWS, rows, XS + s * TW_SLAB, wave, lane);
            tw_load_tile(w1, ldw1, s, H1, H1, W1S + s * H1 * 128, wave, lane);
        }
#pragma unroll
        for (int s = 0; s < H1 / 64; ++s) tw_load_tile(w2, ldw2, s, H2, H2, W2S + s * H2 * 128, wave, lane);
    }
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thr16 = (uint32_t)(a.drop_p * 65536.f + 0.5f);
    const bool relu = a.relu != 0;
    const float drop_p = a.drop_p;
    TW_STAMP(0);
#if TW_TRACE
    if (blockIdx.x == TW_TRACE - 1 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.workspace) + 4096)[15] = t_top;
#endif
    tw_drain();
    __syncthreads();
    TW_STAMP(1);

    // ================= layer 1: Z1 = X W1^T + b1
    {
        f32x4_t acc[2][H1 / 16];
        tw_mfma<H1 / 16, NK0 * 2>(XS, W1S, H1 * 128, acc, wave, lane);
        tw_acc_to_tile<H1 / 16>(acc, ct, Cfg::CS1, wave, lane);
    }
    __syncthreads();
    {
        constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
        const int c = (tid % C8) * 8, lr0 = tid / C8;
        const float* bias = par_s + TW_PAR_B1;
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bias + c), b1 = *reinterpret_cast<const f32x4_t*>(bias + c + 4);
        float* z = T.l1.z; const int64_t ldz = T.l1.ldz;
#pragma unroll
        for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
            f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS1 + c) + b0;
            f32x4_t hi = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS1 + c + 4) + b1;
            *reinterpret_cast<f32x4_t*>(ct + lr * Cfg::CS1 + c) = lo;
            *reinterpret_cast<f32x4_t*>(ct + lr * Cfg::CS1 + c + 4) = hi;
            if (lr < rows) {
                *reinterpret_cast<f32x4_t*>(z + (int64_t)(row0 + lr) * ldz + c) = lo;
                *reinterpret_cast<f32x4_t*>(z + (int64_t)(row0 + lr) * ldz + c + 4) = hi;
            }
        }
    }
    __syncthreads();
    tw_fwd_chunk_sums<H1, false>(ct, Cfg::CS1, row0, M, part, reinterpret_cast<double*>(rec + L.st1), n_tower * H1, t * H1, wave, lane, !mute);
    TW_STAMP(2);

    // ---- while the other workgroups' sums arrive: the wide term (model/layer.py:122-126) of this block's rows l = t, t + n_tower, ...
    // (the other towers' workgroups of the block form the rest; everyone reads all of them in front of the head): a wave per row,
    // 16-byte lanes, the loads of twelve rows in flight together
    if (a.wide_x) {
        const float* wx = a.wide_x; const int64_t ldw = a.ld_wide; const int K4 = a.wide_K >> 2;
        float* pub = reinterpret_cast<float*>(rec + L.wide);
        f32x4_t wv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) wv[q] = (lane + 64 * q) < K4 ? *reinterpret_cast<const f32x4_t*>(a.wide_w + 4 * (lane + 64 * q)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        const float wb = a.wide_bias ? a.wide_bias[0] : 0.f;
        const int n_mine = (rows - t + n_tower - 1) / n_tower;           // rows t, t + n_tower, ... < rows
        for (int i0 = wave; i0 < n_mine; i0 += 4 * 12) {
            f32x4_t xv[12][2];
#pragma unroll
            for (int b = 0; b < 12; ++b) {
                const int i = i0 + 4 * b;
                const int64_t gr = row0 + t + (int64_t)n_tower * (i < n_mine ? i : i0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    xv[b][q] = (lane + 64 * q) < K4 ? *reinterpret_cast<const f32x4_t*>(wx + gr * ldw + 4 * (lane + 64 * q)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            float sv[12];
#pragma unroll
            for (int b = 0; b < 12; ++b) {
                float s_ = 0.f;
#pragma unroll
                for (int q = 0; q < 2; ++q) s_ += (xv[b][q][0] * wv[q][0] + xv[b][q][1] * wv[q][1]) + (xv[b][q][2] * wv[q][2] + xv[b][q][3] * wv[q][3]);
                sv[b] = s_;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)                             // twelve independent butterflies, step by step (a shuffle is ~100 cycles
#pragma unroll
                for (int b = 0; b < 12; ++b) sv[b] += __shfl_xor(sv[b], o, 64);   // of latency: one row after the other they add up)
#pragma unroll
            for (int b = 0; b < 12; ++b) {
                const int i = i0 + 4 * b;
                if (lane == 0 && i < n_mine) tw_st(pub + row0 + t + n_tower * i, tw_nz(sv[b] + wb));
            }
        }
    }
    TW_STAMP(3);
    TW_STAMP(4);
    tw_gather_sums<H1>(reinterpret_cast<const double*>(rec + L.st1), (M + 63) / 64, n_tower * H1, t * H1, part, sums, tid, hdr, err);
    tw_finish_stats<H1>(sums, M, a.eps, a.momentum, writer, T.l1.save_mean, T.l1.save_invstd, T.l1.running_mean, T.l1.running_var,
                        T.l1.num_batches_tracked, col_mean, col_inv, tid);
    if (keep && tid < H1) { keep[tid] = col_mean[tid]; keep[64 + tid] = col_inv[tid]; }
    TW_STAMP(5);

    // ---- A1 = dropout(relu(bn(Z1))): bf16 into the A operand image of layer 2 and into its global copy
    {
        constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
        const int c8 = tid % C8, c = c8 * 8, lr0 = tid / C8;
        const uint32_t seed32 = drop_p > 0.f ? g2_seed32(a.seed1, a.seed_offset_dev, 64 + t) : 0u;
        const float* gamma = par_s + TW_PAR_G1; const float* beta = par_s + TW_PAR_BE1;
        __bf16* a1h = reinterpret_cast<__bf16*>(T.a1h); const int64_t lda = T.lda1h;
#pragma unroll
        for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ct[lr * Cfg::CS1 + c + q];
            tw_bn_apply8(v, col_mean, col_inv, gamma, beta, c, relu, drop_p, keep_scale, thr16, seed32, row0 + lr);
            tw_put8(A1S + (c8 >> 3) * TW_SLAB, lr, c8 & 7, v);
            if (lr < rows) {
                bf16x8_t h;
#pragma unroll
                for (int q = 0; q < 8; ++q) h[q] = (__bf16)v[q];
                *reinterpret_cast<bf16x8_t*>(a1h + (int64_t)(row0 + lr) * lda + c) = h;
            }
        }
    }
    __syncthreads();

    // ================= layer 2: Z2 = A1 W2^T + b2
    {
        f32x4_t acc[2][H2 / 16];
        tw_mfma<H2 / 16, H1 / 32>(A1S, W2S, H2 * 128, acc, wave, lane);
        tw_acc_to_tile<H2 / 16>(acc, ct, Cfg::CS2, wave, lane);
    }
    __syncthreads();
    {
        constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
        const int c = (tid % C8) * 8, lr0 = tid / C8;
        const float* bias = par_s + TW_PAR_B2;
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bias + c), b1 = *reinterpret_cast<const f32x4_t*>(bias + c + 4);
        float* z = T.l2.z; const int64_t ldz = T.l2.ldz;
#pragma unroll
        for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
            f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS2 + c) + b0;
            f32x4_t hi = *reinterpret_cast<const f32x4_t*>(ct + lr * Cfg::CS2 + c + 4) + b1;
            *reinterpret_cast<f32x4_t*>(ct + lr * Cfg::CS2 + c) = lo;
            *reinterpret_cast<f32x4_t*>(ct + lr * Cfg::CS2 + c + 4) = hi;
            if (lr < rows) {
                *reinterpret_cast<f32x4_t*>(z + (int64_t)(row0 + lr) * ldz + c) = lo;
                *reinterpret_cast<f32x4_t*>(z + (int64_t)(row0 + lr) * ldz + c + 4) = hi;
            }
        }
    }
    __syncthreads();
    tw_fwd_chunk_sums<H2, false>(ct, Cfg::CS2, row0, M, part, reinterpret_cast<double*>(rec + L.st2), n_tower * H2, t * H2, wave, lane);
    TW_STAMP(6);
    TW_STAMP(7);
    tw_gather_sums<H2>(reinterpret_cast<const double*>(rec + L.st2), (M + 63) / 64, n_tower * H2, t * H2, part, sums, tid, hdr, err);
    if (a.wide_x && wave < (rows + 63) / 64) {                           // (behind the sums: the block's other workgroups published these long ago)
        const float v = tw_poll<float>(tid < rows ? reinterpret_cast<const float*>(rec + L.wide) + row0 + tid : nullptr, hdr, err);
        if (tid < rows) wide_s[tid] = v;
    }
    tw_finish_stats<H2>(sums, M, a.eps, a.momentum, writer, T.l2.save_mean, T.l2.save_invstd, T.l2.running_mean, T.l2.running_var,
                        T.l2.num_batches_tracked, col_mean, col_inv, tid);
    if (keep && tid < H2) { keep[128 + tid] = col_mean[tid]; keep[192 + tid] = col_inv[tid]; }
    TW_STAMP(8);
    // ---- A2 = dropout(relu(bn(Z2))) in fp32: kept in the tile for the head, written out for the backward
    {
        constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
        const int c = (tid % C8) * 8, lr0 = tid / C8;
        const uint32_t seed32 = drop_p > 0.f ? g2_seed32(a.seed2, a.seed_offset_dev, 64 + t) : 0u;
        const float* gamma = par_s + TW_PAR_G2; const float* beta = par_s + TW_PAR_BE2;
        float* a2 = T.a2; const int64_t lda = T.lda2;
#pragma unroll
        for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ct[lr * Cfg::CS2 + c + q];
            tw_bn_apply8(v, col_mean, col_inv, gamma, beta, c, relu, drop_p, keep_scale, thr16, seed32, row0 + lr);
#pragma unroll
            for (int q = 0; q < 8; ++q) ct[lr * Cfg::CS2 + c + q] = v[q];
            if (lr < rows) {
                *reinterpret_cast<f32x4_t*>(a2 + (int64_t)(row0 + lr) * lda + c) = f32x4_t{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4_t*>(a2 + (int64_t)(row0 + lr) * lda + c + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
            }
        }
    }
    __syncthreads();
    // ---- the head: Linear(H2 -> 1) + wide term + sigmoid; two threads per row, each half of the row's columns in ascending order
    {
        const int lr = tid >> 1, hf = tid & 1;
        const float* arow = ct + lr * Cfg::CS2 + hf * (H2 / 2);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < H2 / 2; k += 4) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(arow + k);
            const f32x4_t w = *reinterpret_cast<const f32x4_t*>(par_s + TW_PAR_WO + hf * (H2 / 2) + k);
            acc += v[0] * w[0]; acc += v[1] * w[1]; acc += v[2] * w[2]; acc += v[3] * w[3];
        }
        acc += __shfl_xor(acc, 1, 64);
        if (hf == 0 && lr < rows) {
            if (T.bo) acc += par_s[TW_PAR_WO + H2];
            if (a.wide_x) acc += wide_s[lr];
            if (a.sigmoid) acc = 1.f / (1.f + expf(-acc));
            a.out[(int64_t)(row0 + lr) * a.ld_out + t] = acc;
        }
        if (keep && hf == 0) {                                           // (cdc_tower_step: the arithmetic of tw_bwd_body's phase (1), fused-BCE form)
            float d_t = 0.f;
            double lp = 0.0;
            int mine = 0;
            if (lr < rows && k_own == t) {
                const float o = acc, tgt = k_tgt;
                lp = (double)((tgt - 1.f) * fmaxf(log1pf(-o), -100.f) - tgt * fmaxf(logf(o), -100.f));
                const float dout = a.bce_inv_count * (o - tgt) / fmaxf((1.f - o) * o, 1e-12f);
                d_t = a.sigmoid ? dout * o * (1.f - o) : dout;
                mine = 1;
            }
            keep[TW_KEEP_D + lr] = d_t;
            reinterpret_cast<int*>(keep)[TW_KEEP_MINE + lr] = mine;
            reinterpret_cast<double*>(keep + TW_KEEP_LOSS)[lr] = lp;
        }
    }
    TW_STAMP(9);
    if (tid == 0) tw_finish(hdr, TW_FDONE, TW_FEPOCH, n_tower * G, epoch);
}

// =================================================================================================
// backward
// =================================================================================================
template <int NK0, int H1, int H2>
struct TwBwdCfg {
    static constexpr int H0 = NK0 * 64;
    static constexpr int CSX = H0 + 4, CS1 = H1 + 4, CS2 = H2 + 4;
    static constexpr int WT2S = 0;                                   // [H1][64] bf16: W2^T rows (K = H2, zero padded)
    static constexpr int WT1S = WT2S + H1 * 128;                     // [H0][64] bf16: W1^T rows (K = H1)
    static constexpr int AS = WT1S + H0 * 128;                       // [128][64] bf16: the A operand (dZ2, then dZ1)
    static constexpr int P = AS + TW_SLAB;                           // fp32 tiles, see the phases
    static constexpr int P_L1 = 2 * TW_ROWS * CS1 * 4;               // [dZ | xhat] of layer 1
    static constexpr int P_L2 = 3 * TW_ROWS * CS2 * 4;               // [dZ | xhat | a2] of layer 2
    static constexpr int P_X = TW_ROWS * CSX * 4;                    // dX
    static constexpr int P_BYTES = (P_L1 > P_L2 ? (P_L1 > P_X ? P_L1 : P_X) : (P_L2 > P_X ? P_L2 : P_X));
    static constexpr int PART = P + P_BYTES;                         // [2][8][64] doubles
    static constexpr int SUMS = PART + 2 * 8 * 64 * 8;               // [2][64] doubles
    static constexpr int DS = SUMS + 2 * 64 * 8;                     // d[128], dsum[128] floats, own list [128] ints, loss [128] doubles
    static constexpr int DSUM = DS + TW_ROWS * 4;
    static constexpr int OWN = DSUM + TW_ROWS * 4;
    static constexpr int LOSS = OWN + (2 * TW_ROWS + 4) * 4;
    static constexpr int WDW = LOSS + TW_ROWS * 8;                   // [4 waves][520] floats: the wide term's weight-gradient partials
    static constexpr int PAR = WDW + 4 * 520 * 4;                    // small vectors (TW_PAR_*: mean, gamma * invstd, invstd per layer; head weight)
    static constexpr int SMEM = PAR + TW_PAR_FLOATS * 4;
};

// column sums over this block's rows of tile u (and of u * w, in double) by NPT = 256 / C threads per column, parts added in order
template <int C>
__device__ __forceinline__ void tw_block_sums(const float* u, const float* w, int cs, int nrows, double* part, double* out, int tid) {
    constexpr int NPT = TW_THREADS / C, RPP = TW_ROWS / NPT;
    const int j = tid % C, pt = tid / C;
    double s1 = 0.0, s2 = 0.0;
    const int r_end = min((pt + 1) * RPP, nrows);
    for (int r = pt * RPP; r < r_end; ++r) {
        const float x = u[r * cs + j];
        s1 += (double)x;
        s2 += (double)x * (double)w[r * cs + j];
    }
    part[(0 * NPT + pt) * C + j] = s1; part[(1 * NPT + pt) * C + j] = s2;
    __syncthreads();
    if (tid < C) {
        double b1 = 0.0, b2 = 0.0;
#pragma unroll
        for (int q = 0; q < NPT; ++q) { b1 += part[(0 * NPT + q) * C + tid]; b2 += part[(1 * NPT + q) * C + tid]; }
        out[tid] = b1; out[C + tid] = b2;
    }
    __syncthreads();
}

// the wide term's gradients of the rows in `own` (a wave per row, TW_WNB rows per round, their loads in flight together):
// dx (+)= dsum * w,  dw += dsum * x,  db += dsum.  A round = its rows' inputs in registers (fetched ahead: the first round's loads
// are issued before the layer-2 phase and land under it).
#define TW_WNB 12
struct TwWideRound { f32x4_t xv[TW_WNB][2]; int lr[TW_WNB]; };
__device__ __forceinline__ void tw_wide_load(TwWideRound& R, const float* wide_x, int64_t ld_wide, const int* own, const float* dsum_s, int n_own,
                                             int i0, int row0, int K4, int lane) {
    if (i0 >= n_own) {                                                   // (uniform) nothing for this wave: no list entry to read, no row to fetch
#pragma unroll
        for (int b = 0; b < TW_WNB; ++b) { R.lr[b] = 0; R.xv[b][0] = R.xv[b][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        return;
    }
#pragma unroll
    for (int b = 0; b < TW_WNB; ++b) {
        const int i = i0 + 4 * b;
        const int lr = own[i < n_own ? i : i0];
        R.lr[b] = lr;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            R.xv[b][q] = (lane + 64 * q) < K4 ? *reinterpret_cast<const f32x4_t*>(wide_x + (int64_t)(row0 + lr) * ld_wide + 4 * (lane + 64 * q)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
}
template <bool RMW>
__device__ __forceinline__ void tw_wide_apply(const TwWideRound& R, float* wide_dx, int64_t ld_wide_dx, const float* dsum_s, int row0, int n_own, int i0, int K4,
                                              const f32x4_t (&wv)[2], f32x4_t (&dwv)[2], float& dbv, int lane) {
    f32x4_t ov[RMW ? TW_WNB : 1][2];
    if constexpr (RMW) {
#pragma unroll
        for (int b = 0; b < TW_WNB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                ov[b][q] = (lane + 64 * q) < K4 ? *reinterpret_cast<const f32x4_t*>(wide_dx + (int64_t)(row0 + R.lr[b]) * ld_wide_dx + 4 * (lane + 64 * q)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int b = 0; b < TW_WNB; ++b) {
        if (i0 + 4 * b >= n_own) break;                                  // uniform
        const float ds = dsum_s[R.lr[b]];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if ((lane + 64 * q) >= K4) continue;
            dwv[q] += R.xv[b][q] * ds;
            if (wide_dx) {
                f32x4_t v = wv[q] * ds;
                if constexpr (RMW) v = ov[b][q] + v;
                *reinterpret_cast<f32x4_t*>(wide_dx + (int64_t)(row0 + R.lr[b]) * ld_wide_dx + 4 * (lane + 64 * q)) = v;
            }
        }
        dbv += ds;
    }
}

// `keep`: see tw_fwd_body — the batch statistics every workgroup of the forward holds anyway (save_mean / save_invstd in global
// memory are written by the tower's FIRST workgroup only, which the others must not wait for)
template <int NK0, int H1, int H2>
__device__ __forceinline__ void tw_bwd_body(const TW_KARG cdc_tower_args& a, unsigned char* smem, const float* keep) {
    typedef TwBwdCfg<NK0, H1, H2> Cfg;
    constexpr int H0 = Cfg::H0;
#if TW_TRACE
    const unsigned long long t_top = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = (int)a.M, n_tower = a.n_tower;
    const int G = (M + TW_ROWS - 1) / TW_ROWS;
    const int n_wg = n_tower * G;
    const int t = (int)blockIdx.x / G, jb = (int)blockIdx.x - t * G;
    const int row0 = jb * TW_ROWS, rows = min(TW_ROWS, M - row0);
    const TW_KARG cdc_tower_desc& T = a.t[t];
    int* hdr = reinterpret_cast<int*>(a.workspace);
    const bool has_wide = a.wide_x != nullptr;
    const int wide_K = has_wide ? a.wide_K : 0;
    const TwLayout L = tw_layout(n_tower, H1, H2, M, wide_K);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(a.workspace);
    int32_t* err = a.err;
    const bool writer = jb == 0;
    const bool bce = a.bce_y_i16 != nullptr || a.bce_y_f32 != nullptr;

    unsigned char* WT2S = smem + Cfg::WT2S;
    unsigned char* WT1S = smem + Cfg::WT1S;
    unsigned char* AS = smem + Cfg::AS;
    float* P = reinterpret_cast<float*>(smem + Cfg::P);
    double* part = reinterpret_cast<double*>(smem + Cfg::PART);
    double* sums = reinterpret_cast<double*>(smem + Cfg::SUMS);
    float* d_s = reinterpret_cast<float*>(smem + Cfg::DS);
    float* dsum_s = reinterpret_cast<float*>(smem + Cfg::DSUM);
    int* own_s = reinterpret_cast<int*>(smem + Cfg::OWN);            // [0..n_own): rows of this block whose wide gradient is formed here; [128] = n_own
    int* mine_s = own_s + TW_ROWS + 4;                               // [128] flags
    double* loss_s = reinterpret_cast<double*>(smem + Cfg::LOSS);
    float* wdw_s = reinterpret_cast<float*>(smem + Cfg::WDW);
    float* par_s = reinterpret_cast<float*>(smem + Cfg::PAR);
    {
        const int k = tid & 63, which = tid >> 6;
        if (which == 0 && k < H1) {
            const float inv = keep ? keep[64 + k] : T.l1.save_invstd[k];
            par_s[TW_PAR_B1 + k] = keep ? keep[k] : T.l1.save_mean[k]; par_s[TW_PAR_G1 + k] = T.l1.gamma[k] * inv; par_s[TW_PAR_BE1 + k] = inv;
        }
        if (which == 1 && k < H2) {
            const float inv = keep ? keep[192 + k] : T.l2.save_invstd[k];
            par_s[TW_PAR_B2 + k] = keep ? keep[128 + k] : T.l2.save_mean[k]; par_s[TW_PAR_G2 + k] = T.l2.gamma[k] * inv; par_s[TW_PAR_BE2 + k] = inv;
        }
        if (which == 2 && k < H2) par_s[TW_PAR_WO + k] = T.wo[k];
    }

    // the record copy of this launch; this workgroup's slots of the OTHER copy are cleared for the next launch
    const int epoch = __builtin_amdgcn_readfirstlane(tw_ld(hdr + TW_BEPOCH * TW_LINE));
    unsigned char* const rec = wsb + (epoch & 1) * L.copy;
    {
        unsigned char* const nxt = wsb + ((epoch + 1) & 1) * L.copy;
        for (int i = tid; i < 2 * H2; i += TW_THREADS) tw_st(reinterpret_cast<double*>(nxt + L.b2) + ((int64_t)jb * n_tower * H2 + t * H2) * 2 + i, 0.0);
        for (int i = tid; i < 2 * H1; i += TW_THREADS) tw_st(reinterpret_cast<double*>(nxt + L.b1) + ((int64_t)jb * n_tower * H1 + t * H1) * 2 + i, 0.0);
        if (tid <= H2) tw_st(reinterpret_cast<float*>(nxt + L.hd) + ((int64_t)jb * n_tower + t) * (H2 + 4) + tid, 0.f);
        if (has_wide)
            for (int i = tid; i < L.wd_ld; i += TW_THREADS) tw_st(reinterpret_cast<float*>(nxt + L.wd) + ((int64_t)t * G + jb) * L.wd_ld + i, 0.f);
        if (tid == 0) tw_st(reinterpret_cast<double*>(nxt + L.loss) + blockIdx.x, 0.0);
    }

    // ---- grad-input operands: W2^T and W1^T -> LDS (needed after the first exchange: they land under everything before it)
    tw_load_tile(reinterpret_cast<const __bf16*>(T.l2.wt), T.l2.ldwt, 0, H1, H1, WT2S, wave, lane);
    tw_load_tile(reinterpret_cast<const __bf16*>(T.l1.wt), T.l1.ldwt, 0, H0, H0, WT1S, wave, lane);

    const float mask_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const bool masked = a.relu != 0 || mask_scale != 1.f;
    const float invM = 1.f / (float)M;
    // this thread's layer-2 pieces (saved pre-normalisation values, activations): in flight under the logit-gradient phase
    constexpr int L2P = TW_ROWS / (TW_THREADS / (H2 / 8));
    f32x4_t qzl[L2P], qzh[L2P], qal[L2P], qah[L2P];
    {
        constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
        const int c = (tid % C8) * 8, lr0 = tid / C8;
        const float* z2 = T.l2.z; const int64_t ldz = T.l2.ldz;
        const float* a2 = T.a2; const int64_t lda = T.lda2;
#pragma unroll
        for (int i = 0; i < L2P; ++i) {
            const int lr = min(lr0 + i * RPP, rows - 1);
            qzl[i] = *reinterpret_cast<const f32x4_t*>(z2 + (int64_t)(row0 + lr) * ldz + c);
            qzh[i] = *reinterpret_cast<const f32x4_t*>(z2 + (int64_t)(row0 + lr) * ldz + c + 4);
            qal[i] = *reinterpret_cast<const f32x4_t*>(a2 + (int64_t)(row0 + lr) * lda + c);
            qah[i] = *reinterpret_cast<const f32x4_t*>(a2 + (int64_t)(row0 + lr) * lda + c + 4);
        }
    }
    TW_STAMP(16);
#if TW_TRACE
    if (blockIdx.x == TW_TRACE - 1 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.workspace) + 4096)[31] = t_top;
#endif

    // ---- (1) logit gradients of the block's rows (k_head_bwd's arithmetic: BCELoss(mean) on the row's own tower, or d_out)
    if (keep) {                                                          // cdc_tower_step: formed by the forward body's head
        if (tid < TW_ROWS) {
            const float d_t = keep[TW_KEEP_D + tid];
            d_s[tid] = d_t; dsum_s[tid] = d_t;
            loss_s[tid] = reinterpret_cast<const double*>(keep + TW_KEEP_LOSS)[tid];
            mine_s[tid] = reinterpret_cast<const int*>(keep)[TW_KEEP_MINE + tid];
        }
    } else if (tid < TW_ROWS) {
        float d_t = 0.f, dsum = 0.f;
        int mine = 0;
        double lp = 0.0;
        if (tid < rows) {
            const int64_t r = row0 + tid;
            int own = 0;
            float tgt = 0.f;
            if (bce) {
                int64_t c = a.bce_group ? a.bce_group[r] : 0;
                if (c < 0 || c >= n_tower) c = 0;
                own = (int)c;
                tgt = a.bce_y_i16 ? (float)a.bce_y_i16[r] : a.bce_y_f32[r];
            }
            if (bce) {
                // only the row's OWN tower has a gradient, and only that tower's workgroup looks at the row: it reads nothing but
                // what the forward of the SAME workgroup wrote (which is what lets cdc_tower_step run both in one launch)
                if (own == t) {
                    const float o = a.out[r * a.ld_out + t];
                    lp = (double)((tgt - 1.f) * fmaxf(log1pf(-o), -100.f) - tgt * fmaxf(logf(o), -100.f));
                    const float dout = a.bce_inv_count * (o - tgt) / fmaxf((1.f - o) * o, 1e-12f);
                    d_t = a.sigmoid ? dout * o * (1.f - o) : dout;
                    dsum = d_t;
                }
            } else {
                for (int tt = 0; tt < n_tower; ++tt) {
                    const float o = a.out[r * a.ld_out + tt];
                    const float dout = a.d_out[r * a.ld_dout + tt];
                    const float d = a.sigmoid ? dout * o * (1.f - o) : dout;
                    dsum += d;                                           // ascending tower order
                    if (tt == t) d_t = d;
                }
            }
            const int owner = bce ? own : (int)(r % n_tower);            // which tower's workgroup forms the row's wide gradient
            mine = owner == t;
        }
        d_s[tid] = d_t; dsum_s[tid] = dsum;
        loss_s[tid] = lp;
        mine_s[tid] = mine;
    }
    __syncthreads();
    if (wave == 0) {                                                     // compact list of the rows whose wide gradient is formed here
        int base = 0;
#pragma unroll
        for (int h = 0; h < TW_ROWS / 64; ++h) {
            const int flag = mine_s[h * 64 + lane];
            const unsigned long long bal = __ballot(flag != 0);
            if (flag) own_s[base + __popcll(bal & ((1ull << lane) - 1ull))] = h * 64 + lane;
            base += __popcll(bal);
        }
        if (lane == 0) own_s[TW_ROWS] = base;
    }
    __syncthreads();

    TwWideRound wr;                                                      // first round of the wide gradients' inputs: lands under phase (2)
    if (has_wide) tw_wide_load(wr, a.wide_x, a.ld_wide, own_s, dsum_s, own_s[TW_ROWS], wave, row0, wide_K >> 2, lane);
    TW_STAMP(17);
    // ---- (2) layer-2 pieces: dz2 = mask(a2) * d * wo, xhat2 -> tiles; head weight-gradient products ride on the a2 tile
    float* DZ2 = P;
    float* XH2 = P + TW_ROWS * Cfg::CS2;
    float* A2T = P + 2 * TW_ROWS * Cfg::CS2;
    {
        constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
        const int c = (tid % C8) * 8, lr0 = tid / C8;
        float wo[8], mean[8], inv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { wo[q] = par_s[TW_PAR_WO + c + q]; mean[q] = par_s[TW_PAR_B2 + c + q]; inv[q] = par_s[TW_PAR_BE2 + c + q]; }
#pragma unroll
        for (int i = 0; i < L2P; ++i) {
            const int lr = lr0 + i * RPP;
            const bool live = lr < rows;
            const float d = d_s[lr];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float zv = q < 4 ? qzl[i][q] : qzh[i][q - 4], av = live ? (q < 4 ? qal[i][q] : qah[i][q - 4]) : 0.f;
                float dz = d * wo[q];
                if (masked) dz = av > 0.f ? dz * mask_scale : 0.f;
                DZ2[lr * Cfg::CS2 + c + q] = dz;
                XH2[lr * Cfg::CS2 + c + q] = (zv - mean[q]) * inv[q];
                A2T[lr * Cfg::CS2 + c + q] = av;
            }
        }
    }
    __syncthreads();
    tw_block_sums<H2>(DZ2, XH2, Cfg::CS2, rows, part, sums, tid);
    {
        double* b2 = reinterpret_cast<double*>(rec + L.b2) + ((int64_t)jb * n_tower * H2 + t * H2) * 2;
        if (tid < H2) { tw_st(b2 + 2 * tid, tw_nz(sums[tid])); tw_st(b2 + 2 * tid + 1, tw_nz(sums[H2 + tid])); }
    }
    // head weight gradient of this block: dwo[c] = sum_r d[r] a2[r, c], dbo = sum_r d[r]  (fp64 sums, stored as fp32 partials)
    {
        constexpr int NPT = TW_THREADS / H2, RPP = TW_ROWS / NPT;
        const int j = tid % H2, pt = tid / H2;
        double s = 0.0, sb = 0.0;
        for (int r = pt * RPP; r < min((pt + 1) * RPP, rows); ++r) { s += (double)(d_s[r] * A2T[r * Cfg::CS2 + j]); sb += (double)d_s[r]; }
        part[(0 * NPT + pt) * H2 + j] = s; part[(1 * NPT + pt) * H2 + j] = sb;
        __syncthreads();
        float* hd = reinterpret_cast<float*>(rec + L.hd) + ((int64_t)jb * n_tower + t) * (H2 + 4);
        if (tid < H2) {
            double b = 0.0;
#pragma unroll
            for (int q = 0; q < NPT; ++q) b += part[(0 * NPT + q) * H2 + tid];
            tw_st(hd + tid, tw_nz((float)b));
        } else if (tid == H2) {
            double b = 0.0;
#pragma unroll
            for (int q = 0; q < NPT; ++q) b += part[(1 * NPT + q) * H2 + 0];
            tw_st(hd + H2, tw_nz((float)b));
        }
        if (bce && wave == 1) {                                          // the loss of the block's rows of this tower
            double s_ = (lane < rows ? loss_s[lane] : 0.0) + (lane + 64 < rows ? loss_s[lane + 64] : 0.0);
            s_ = wave_sum_d(s_);
            if (lane == 0) tw_st(reinterpret_cast<double*>(rec + L.loss) + blockIdx.x, tw_nz(s_));
        }
    }
    __syncthreads();
    TW_STAMP(18);

    // ---- while the other workgroups' sums arrive: the wide term's gradients for the rows this workgroup owns (k_head_bwd (3)): a wave per row
    if (has_wide) {
        const int K4 = wide_K >> 2;
        const int n_own = own_s[TW_ROWS];
        f32x4_t wv[2], dwv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wv[q] = (lane + 64 * q) < K4 ? *reinterpret_cast<const f32x4_t*>(a.wide_w + 4 * (lane + 64 * q)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            dwv[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        float dbv = 0.f;
        const bool rmw = a.wide_dx && a.accumulate_wide_dx;
        for (int i0 = wave; i0 < n_own; i0 += 4 * TW_WNB) {
            if (i0 != wave) tw_wide_load(wr, a.wide_x, a.ld_wide, own_s, dsum_s, n_own, i0, row0, K4, lane);   // (the first round is in registers)
            if (rmw) tw_wide_apply<true>(wr, a.wide_dx, a.ld_wide_dx, dsum_s, row0, n_own, i0, K4, wv, dwv, dbv, lane);
            else tw_wide_apply<false>(wr, a.wide_dx, a.ld_wide_dx, dsum_s, row0, n_own, i0, K4, wv, dwv, dbv, lane);
        }
        // rows nobody owns here still need their wide_dx written when this launch is the first writer: every row has exactly one
        // owner (its own tower's workgroup of the same block), so nothing is left out
        float* mine = wdw_s + wave * 520;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if ((lane + 64 * q) < K4) *reinterpret_cast<f32x4_t*>(mine + 4 * (lane + 64 * q)) = dwv[q];
        if (lane == 0) mine[wide_K] = dbv;
        __syncthreads();
        float* wd = reinterpret_cast<float*>(rec + L.wd) + ((int64_t)t * G + jb) * L.wd_ld;
        for (int k = 2 * tid; k <= wide_K; k += 2 * TW_THREADS) {            // 8-byte write-through stores (a 4-byte one costs as much)
            union { float f[2]; unsigned long long u; } pk;
            pk.f[0] = tw_nz(((wdw_s[k] + wdw_s[520 + k]) + wdw_s[2 * 520 + k]) + wdw_s[3 * 520 + k]);
            pk.f[1] = k + 1 <= wide_K ? tw_nz(((wdw_s[k + 1] + wdw_s[520 + k + 1]) + wdw_s[2 * 520 + k + 1]) + wdw_s[3 * 520 + k + 1]) : 0.f;
            tw_st(reinterpret_cast<unsigned long long*>(wd + k), pk.u);
        }
    }
    // layer 1's saved pre-normalisation values and hidden activations of this thread's pieces: in flight under the exchange
    constexpr int L1P = TW_ROWS / (TW_THREADS / (H1 / 8));
    f32x4_t pzl[L1P], pzh[L1P];
    bf16x8_t pm8[L1P];
    {
        constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
        const int c = (tid % C8) * 8, lr0 = tid / C8;
        const float* z1 = T.l1.z; const int64_t ldz = T.l1.ldz;
        const __bf16* a1h = reinterpret_cast<const __bf16*>(T.a1h); const int64_t lda = T.lda1h;
#pragma unroll
        for (int i = 0; i < L1P; ++i) {
            const int lr = min(lr0 + i * RPP, rows - 1);
            pzl[i] = *reinterpret_cast<const f32x4_t*>(z1 + (int64_t)(row0 + lr) * ldz + c);
            pzh[i] = *reinterpret_cast<const f32x4_t*>(z1 + (int64_t)(row0 + lr) * ldz + c + 4);
            pm8[i] = *reinterpret_cast<const bf16x8_t*>(a1h + (int64_t)(row0 + lr) * lda + c);
        }
    }
    TW_STAMP(19);
    TW_STAMP(20);
    tw_gather_sums<H2>(reinterpret_cast<const double*>(rec + L.b2), G, n_tower * H2, t * H2, part, sums, tid, hdr, err);
    if (writer) {
        if (tid < H2) {
            if (T.l2.dbeta) T.l2.dbeta[tid] = (float)sums[tid];
            if (T.l2.dgamma) T.l2.dgamma[tid] = (float)sums[H2 + tid];
        }
    }
    if (writer) {
        // the tower's head gradient: eight lanes per column take block partials l, l + 8, ... (their loads in flight together), a butterfly
        // adds the eight
        for (int k = tid >> 3; k <= H2; k += TW_THREADS / 8) {
            const int p8 = tid & 7;
            const float* hd = reinterpret_cast<const float*>(rec + L.hd) + (int64_t)t * (H2 + 4) + k;
            float s = 0.f;
            unsigned spins = 0;
            for (int b0 = p8; b0 < G; b0 += 64) {
                float v[8];
                for (;;) {                                               // (published in front of the sums gathered above: rarely a second pass)
                    bool bad = false;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        v[q] = (b0 + 8 * q) < G ? tw_ld(hd + (int64_t)(b0 + 8 * q) * n_tower * (H2 + 4)) : 0.f;
                        bad = bad || ((b0 + 8 * q) < G && tw_empty(v[q]));
                    }
                    if (!bad || !tw_spin(spins, hdr, err)) break;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) s += v[q];
            }
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            if (p8 == 0) {
                if (k < H2) { if (T.dwo) T.dwo[k] = s; }
                else if (T.dbo) T.dbo[0] = s;
            }
        }
    }
    TW_STAMP(21);
    // ---- (3) dZ2 = gamma invstd (dz - (db + xhat dg) / M) -> A operand image + global bf16 copy (grad-weight launch)
    {
        constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
        const int c8 = tid % C8, c = c8 * 8, lr0 = tid / C8;
        float k1[8], db[8], dg[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            k1[q] = par_s[TW_PAR_G2 + c + q];
            db[q] = (float)sums[c + q]; dg[q] = (float)sums[H2 + c + q];
        }
        __bf16* dzh = reinterpret_cast<__bf16*>(T.l2.dzh); const int64_t ldd = T.l2.lddzh;
#pragma unroll
        for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = k1[q] * (DZ2[lr * Cfg::CS2 + c + q] - invM * (db[q] + XH2[lr * Cfg::CS2 + c + q] * dg[q]));
            tw_put8(AS, lr, c8, v);
            if (lr < rows) {
                bf16x8_t h;
#pragma unroll
                for (int q = 0; q < 8; ++q) h[q] = (__bf16)v[q];
                *reinterpret_cast<bf16x8_t*>(dzh + (int64_t)(row0 + lr) * ldd + c) = h;
            }
        }
    }
    tw_drain();                                                          // (also: the W^T tiles have landed)
    __syncthreads();
    TW_STAMP(22);
    // ---- (4) dA1 = dZ2 W2 -> tile; dz1 = mask(a1) * dA1, xhat1 -> tiles (dz1 in place)
    float* DZ1 = P;
    float* XH1 = P + TW_ROWS * Cfg::CS1;
    {
        f32x4_t acc[2][H1 / 16];
        tw_mfma<H1 / 16, H2 / 32>(AS, WT2S, 0, acc, wave, lane);
        tw_acc_to_tile<H1 / 16>(acc, DZ1, Cfg::CS1, wave, lane);
    }
    __syncthreads();
    {
        constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
        const int c = (tid % C8) * 8, lr0 = tid / C8;
        float mean[8], inv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { mean[q] = par_s[TW_PAR_B1 + c + q]; inv[q] = par_s[TW_PAR_BE1 + c + q]; }
#pragma unroll
        for (int i = 0; i < L1P; ++i) {
            const int lr = lr0 + i * RPP;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float dz = DZ1[lr * Cfg::CS1 + c + q];
                if (masked) dz = (float)pm8[i][q] > 0.f ? dz * mask_scale : 0.f;
                if (lr >= rows) dz = 0.f;
                DZ1[lr * Cfg::CS1 + c + q] = dz;
                XH1[lr * Cfg::CS1 + c + q] = ((q < 4 ? pzl[i][q] : pzh[i][q - 4]) - mean[q]) * inv[q];
            }
        }
    }
    __syncthreads();
    tw_block_sums<H1>(DZ1, XH1, Cfg::CS1, rows, part, sums, tid);
    {
        double* b1 = reinterpret_cast<double*>(rec + L.b1) + ((int64_t)jb * n_tower * H1 + t * H1) * 2;
        if (tid < H1) { tw_st(b1 + 2 * tid, tw_nz(sums[tid])); tw_st(b1 + 2 * tid + 1, tw_nz(sums[H1 + tid])); }
    }
    TW_STAMP(23);
    TW_STAMP(24);
    tw_gather_sums<H1>(reinterpret_cast<const double*>(rec + L.b1), G, n_tower * H1, t * H1, part, sums, tid, hdr, err);
    TW_STAMP(25);
    if (writer && tid < H1) {
        if (T.l1.dbeta) T.l1.dbeta[tid] = (float)sums[tid];
        if (T.l1.dgamma) T.l1.dgamma[tid] = (float)sums[H1 + tid];
    }
    // ---- (5) dZ1 -> A operand image + global bf16 copy
    {
        constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
        const int c8 = tid % C8, c = c8 * 8, lr0 = tid / C8;
        float k1[8], db[8], dg[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            k1[q] = par_s[TW_PAR_G1 + c + q];
            db[q] = (float)sums[c + q]; dg[q] = (float)sums[H1 + c + q];
        }
        __bf16* dzh = reinterpret_cast<__bf16*>(T.l1.dzh); const int64_t ldd = T.l1.lddzh;
#pragma unroll
        for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = k1[q] * (DZ1[lr * Cfg::CS1 + c + q] - invM * (db[q] + XH1[lr * Cfg::CS1 + c + q] * dg[q]));
            tw_put8(AS, lr, c8, v);
            if (lr < rows) {
                bf16x8_t h;
#pragma unroll
                for (int q = 0; q < 8; ++q) h[q] = (__bf16)v[q];
                *reinterpret_cast<bf16x8_t*>(dzh + (int64_t)(row0 + lr) * ldd + c) = h;
            }
        }
    }
    __syncthreads();
    TW_STAMP(26);
    // ---- (6) dX = dZ1 W1 -> tile -> global
    {
        f32x4_t acc[2][H0 / 16];
        tw_mfma<H0 / 16, H1 / 32>(AS, WT1S, 0, acc, wave, lane);
        tw_acc_to_tile<H0 / 16>(acc, P, Cfg::CSX, wave, lane);
    }
    __syncthreads();
    if (T.dx) {
        constexpr int C8 = H0 / 8, RPP = TW_THREADS / C8;
        const int c = (tid % C8) * 8, lr0 = tid / C8;
        float* dx = T.dx; const int64_t ldd = T.lddx;
        const bool accx = T.accumulate_dx != 0;
#pragma unroll
        for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
            if (lr >= rows) break;
            f32x4_t lo = *reinterpret_cast<const f32x4_t*>(P + lr * Cfg::CSX + c);
            f32x4_t hi = *reinterpret_cast<const f32x4_t*>(P + lr * Cfg::CSX + c + 4);
            float* dst = dx + (int64_t)(row0 + lr) * ldd + c;
            if (accx) { lo = *reinterpret_cast<const f32x4_t*>(dst) + lo; hi = *reinterpret_cast<const f32x4_t*>(dst + 4) + hi; }
            *reinterpret_cast<f32x4_t*>(dst) = lo;
            *reinterpret_cast<f32x4_t*>(dst + 4) = hi;
        }
    }
    TW_STAMP(27);
    // ---- (7) sums over ALL workgroups (published before the last exchange): the wide term's weight gradient (a wave per
    // column: lanes take partials l, l + 64, ... in order, then a butterfly) and the loss
    if (has_wide) {
        const float* wd = reinterpret_cast<const float*>(rec + L.wd);
        for (int k = (int)blockIdx.x * 4 + wave; k <= wide_K; k += n_wg * 4) {
            float v[4];                                                  // n_wg <= 256: at most four partials per lane, in flight together
            unsigned spins = 0;
            for (;;) {
                bool bad = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = lane + 64 * q;
                    v[q] = p < n_wg ? tw_ld(wd + (int64_t)p * L.wd_ld + k) : 0.f;
                    bad = bad || (p < n_wg && tw_empty(v[q]));
                }
                if (!__any(bad) || !tw_spin(spins, hdr, err)) break;
            }
            float s = ((v[0] + v[1]) + v[2]) + v[3];
            s = wave_sum(s);
            if (lane == 0) {
                if (k < wide_K) { if (a.wide_dw) a.wide_dw[k] = s; }
                else if (a.wide_dbias) a.wide_dbias[0] = s;
            }
        }
    }
    if (bce && blockIdx.x == 0 && wave == 1 && a.bce_loss) {
        const double* lp = reinterpret_cast<const double*>(rec + L.loss);
        double v[4];
        unsigned spins = 0;
        for (;;) {
            bool bad = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = lane + 64 * q;
                v[q] = p < n_wg ? tw_ld(lp + p) : 0.0;
                bad = bad || (p < n_wg && tw_empty(v[q]));
            }
            if (!__any(bad) || !tw_spin(spins, hdr, err)) break;
        }
        double s = ((v[0] + v[1]) + v[2]) + v[3];
        s = wave_sum_d(s);
        if (lane == 0) *a.bce_loss = (float)(s * (double)a.bce_inv_count);
    }
    TW_STAMP(28);
    if (tid == 0) tw_finish(hdr, TW_BDONE, TW_BEPOCH, n_wg, epoch);
}

template <int NK0, int H1, int H2>
__global__ void __launch_bounds__(TW_THREADS) k_tower_fwd(const cdc_tower_args a_by_value) {
    CDC_PRIO_MAIN();
    (void)a_by_value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tw_fwd_body<NK0, H1, H2>(*(const TW_KARG cdc_tower_args*)__builtin_amdgcn_kernarg_segment_ptr(), smem, nullptr);
}
template <int NK0, int H1, int H2>
__global__ void __launch_bounds__(TW_THREADS) k_tower_bwd(const cdc_tower_args a_by_value) {
    CDC_PRIO_MAIN();
    (void)a_by_value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tw_bwd_body<NK0, H1, H2>(*(const TW_KARG cdc_tower_args*)__builtin_amdgcn_kernarg_segment_ptr(), smem, nullptr);
}
template <int NK0, int H1, int H2>
struct TwStepCfg {
    static constexpr int F = TwFwdCfg<NK0, H1, H2>::SMEM, B = TwBwdCfg<NK0, H1, H2>::SMEM;
    static constexpr int KEEP = ((F > B ? F : B) + 15) / 16 * 16;    // the bodies' regions overlay each other; the statistics sit behind both
    static constexpr int SMEM = KEEP + TW_KEEP_FLOATS * 4;
    static_assert(SMEM <= 150 * 1024, "one workgroup per CU");
};
// Forward and backward of a TRAINING step in one launch (cdc_tower_step): with the fused loss a workgroup's backward needs nothing
// but what the same workgroup's forward produced (a row's logit gradient comes from its OWN tower's output, formed by the tower's
// workgroup of that block), so the second body simply follows the first — one launch and one ramp less, and the backward reads
// z1 / a1 / z2 / a2 / out where its own forward has just written them.
template <int NK0, int H1, int H2>
__global__ void __launch_bounds__(TW_THREADS) k_tower_step(const cdc_tower_args a_by_value) {
    CDC_PRIO_MAIN();
    (void)a_by_value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const TW_KARG cdc_tower_args& a = *(const TW_KARG cdc_tower_args*)__builtin_amdgcn_kernarg_segment_ptr();
    float* keep = reinterpret_cast<float*>(smem + TwStepCfg<NK0, H1, H2>::KEEP);
    tw_fwd_body<NK0, H1, H2>(a, smem, keep);
    tw_drain();                                                          // this workgroup's stores of the forward have left
    __syncthreads();
    tw_bwd_body<NK0, H1, H2>(a, smem, keep);
}

// =================================================================================================
// the split form (data parallel with global-batch BatchNorm statistics): cdc_tower_dp, phases 1..6
// =================================================================================================
// A phase that ends in front of an exchange: the workgroups of tower t have published their partial records (write-through stores,
// drained, behind a workgroup barrier); the LAST one to arrive (the value its add returned says so) adds them up in the fixed order
// of tw_gather_sums and writes the tower's local sums and row count into the exchange buffer; it also puts the counter back to zero.
// All threads call.  Returns (uniformly per workgroup) whether this workgroup was the last.
template <int C>
__device__ __forceinline__ bool tw_last_sums(int* hdr, int line, int G, const double* ws, int n_parts, int total_c, int col0, double* part,
                                             double* sums, int* flag_s, double* ex, int t, int n_tower, int M, int tid) {
    if (tid == 0) {
        const int old = tw_add(hdr + line * TW_LINE, 1);
        *flag_s = old == G - 1;
        if (old == G - 1) tw_st(hdr + line * TW_LINE, 0);
    }
    __syncthreads();
    if (!*flag_s) return false;
    tw_gather_sums<C>(ws, n_parts, total_c, col0, part, sums, tid);
    if (tid < C) {
        ex[2 * (t * C + tid)] = sums[tid];
        ex[2 * (t * C + tid) + 1] = sums[C + tid];
    }
    if (tid == 0) ex[2 * n_tower * C + t] = (double)M;
    return true;
}
// global sums -> mean / invstd in LDS (forward) from an all-reduced exchange buffer
template <int C>
__device__ __forceinline__ void tw_stats_from_exchange(const double* ex, int t, int n_tower, float eps, float momentum, bool writer, float* save_mean,
                                                       float* save_invstd, float* running_mean, float* running_var, int64_t* nbt, float* col_mean,
                                                       float* col_inv, double* sums, int tid) {
    const int Ms = (int)(ex[2 * n_tower * C + t] + 0.5);
    if (tid < C) { sums[tid] = ex[2 * (t * C + tid)]; sums[C + tid] = ex[2 * (t * C + tid) + 1]; }
    __syncthreads();
    tw_finish_stats<C>(sums, Ms, eps, momentum, writer, save_mean, save_invstd, running_mean, running_var, nbt, col_mean, col_inv, tid);
}

struct TwDpCfg {                                                     // LDS of every phase (H0 <= 128, H1 = 64, H2 = 32)
    static constexpr int OPA = 0;                                    // A operand tile: up to two 64-column slabs of [128][64] bf16
    static constexpr int OPB = OPA + 2 * TW_SLAB;                    // B operand tile: up to 128 rows x 64 columns bf16 (x 2 slabs of 64 rows)
    static constexpr int CT = OPB + 128 * 128;                       // fp32 tiles: [128][132] (dX) / 2 x [128][68] / 3 x [128][36]
    static constexpr int PART = CT + TW_ROWS * 136 * 4;
    static constexpr int SUMS = PART + 2 * 8 * 64 * 8;
    static constexpr int CMEAN = SUMS + 2 * 64 * 8;
    static constexpr int CINV = CMEAN + 256;
    static constexpr int PAR = CINV + 256;
    static constexpr int DS = PAR + TW_PAR_FLOATS * 4;               // d[128], dsum[128] floats, own list + flags, loss
    static constexpr int DSUM = DS + 512;
    static constexpr int OWN = DSUM + 512;
    static constexpr int LOSS = OWN + (2 * TW_ROWS + 4) * 4;
    static constexpr int WDW = LOSS + TW_ROWS * 8;
    static constexpr int FLAG = WDW + 4 * 520 * 4;
    static constexpr int SMEM = FLAG + 16;
    static_assert(SMEM <= 150 * 1024, "one workgroup per CU with room to spare");
};

template <int NK0, int H1, int H2, int PHASE>
__global__ void __launch_bounds__(TW_THREADS) k_tower_dp(const cdc_tower_args a_by_value) {
    CDC_PRIO_MAIN();
    (void)a_by_value;
    const TW_KARG cdc_tower_args& a = *(const TW_KARG cdc_tower_args*)__builtin_amdgcn_kernarg_segment_ptr();
    typedef TwDpCfg Cfg;
    constexpr int H0 = NK0 * 64;
    constexpr int CS1 = H1 + 4, CS2 = H2 + 4, CSX = H0 + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = (int)a.M, n_tower = a.n_tower;
    const int G = (M + TW_ROWS - 1) / TW_ROWS;
    const int n_wg = n_tower * G;
    const int t = (int)blockIdx.x / G, jb = (int)blockIdx.x - t * G;
    const int row0 = jb * TW_ROWS, rows = min(TW_ROWS, M - row0);
    const TW_KARG cdc_tower_desc& T = a.t[t];
    int* hdr = reinterpret_cast<int*>(a.workspace);
    const bool has_wide = a.wide_x != nullptr;
    const int wide_K = has_wide ? a.wide_K : 0;
    const TwLayout L = tw_layout(n_tower, H1, H2, M, wide_K);
    unsigned char* wsb = reinterpret_cast<unsigned char*>(a.workspace);
    const bool writer = jb == 0;
    unsigned char* OPA = smem + Cfg::OPA;
    unsigned char* OPB = smem + Cfg::OPB;
    float* ct = reinterpret_cast<float*>(smem + Cfg::CT);
    double* part = reinterpret_cast<double*>(smem + Cfg::PART);
    double* sums = reinterpret_cast<double*>(smem + Cfg::SUMS);
    float* col_mean = reinterpret_cast<float*>(smem + Cfg::CMEAN);
    float* col_inv = reinterpret_cast<float*>(smem + Cfg::CINV);
    float* par_s = reinterpret_cast<float*>(smem + Cfg::PAR);
    float* d_s = reinterpret_cast<float*>(smem + Cfg::DS);
    float* dsum_s = reinterpret_cast<float*>(smem + Cfg::DSUM);
    int* own_s = reinterpret_cast<int*>(smem + Cfg::OWN);
    int* mine_s = own_s + TW_ROWS + 4;
    double* loss_s = reinterpret_cast<double*>(smem + Cfg::LOSS);
    float* wdw_s = reinterpret_cast<float*>(smem + Cfg::WDW);
    int* flag_s = reinterpret_cast<int*>(smem + Cfg::FLAG);
    (void)d_s; (void)dsum_s; (void)own_s; (void)mine_s; (void)loss_s; (void)wdw_s; (void)col_mean; (void)col_inv; (void)n_wg; (void)CSX;
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t thr16 = (uint32_t)(a.drop_p * 65536.f + 0.5f);
    const bool relu = a.relu != 0;
    const float drop_p = a.drop_p;
    const bool masked = relu || keep_scale != 1.f;

    if constexpr (PHASE == 1) {
        // ---- Z1 = X W1^T + b1 -> z1; local chunk sums -> exchange[0]; the wide term of ALL this block's rows of tower 0's share...
        const __bf16* xh = reinterpret_cast<const __bf16*>(T.xh) + (int64_t)row0 * T.ldxh;
#pragma unroll
        for (int s = 0; s < NK0; ++s) {
            tw_load_tile(xh, T.ldxh, s, TW_ROWS, rows, OPA + s * TW_SLAB, wave, lane);
            tw_load_tile(reinterpret_cast<const __bf16*>(T.l1.wh), T.l1.ldwh, s, H1, H1, OPB + s * H1 * 128, wave, lane);
        }
        if (tid < H1) par_s[TW_PAR_B1 + tid] = T.l1.bias[tid];
        tw_drain();
        __syncthreads();
        {
            f32x4_t acc[2][H1 / 16];
            tw_mfma<H1 / 16, NK0 * 2>(OPA, OPB, H1 * 128, acc, wave, lane);
            tw_acc_to_tile<H1 / 16>(acc, ct, CS1, wave, lane);
        }
        __syncthreads();
        {
            constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
            const int c = (tid % C8) * 8, lr0 = tid / C8;
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(par_s + TW_PAR_B1 + c), b1 = *reinterpret_cast<const f32x4_t*>(par_s + TW_PAR_B1 + c + 4);
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * CS1 + c) + b0, hi = *reinterpret_cast<const f32x4_t*>(ct + lr * CS1 + c + 4) + b1;
                *reinterpret_cast<f32x4_t*>(ct + lr * CS1 + c) = lo;
                *reinterpret_cast<f32x4_t*>(ct + lr * CS1 + c + 4) = hi;
                if (lr < rows) {
                    *reinterpret_cast<f32x4_t*>(T.l1.z + (int64_t)(row0 + lr) * T.l1.ldz + c) = lo;
                    *reinterpret_cast<f32x4_t*>(T.l1.z + (int64_t)(row0 + lr) * T.l1.ldz + c + 4) = hi;
                }
            }
        }
        __syncthreads();
        tw_fwd_chunk_sums<H1>(ct, CS1, row0, M, part, reinterpret_cast<double*>(wsb + L.st1), n_tower * H1, t * H1, wave, lane);
        // the wide term of the block's rows t, t + n_tower, ... (read by phase 3 of every tower: a later launch)
        if (has_wide) {
            const int K4 = wide_K >> 2;
            float* pub = reinterpret_cast<float*>(wsb + L.wide);
            f32x4_t wv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) wv[q] = (lane + 64 * q) < K4 ? *reinterpret_cast<const f32x4_t*>(a.wide_w + 4 * (lane + 64 * q)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            const float wb = a.wide_bias ? a.wide_bias[0] : 0.f;
            const int n_mine = (rows - t + n_tower - 1) / n_tower;
            for (int i0 = wave; i0 < n_mine; i0 += 4 * 12) {             // twelve rows per round, their loads in flight together
                f32x4_t xv[12][2];
#pragma unroll
                for (int b = 0; b < 12; ++b) {
                    const int i = i0 + 4 * b;
                    const int64_t gr = row0 + t + (int64_t)n_tower * (i < n_mine ? i : i0);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        xv[b][q] = (lane + 64 * q) < K4 ? *reinterpret_cast<const f32x4_t*>(a.wide_x + gr * a.ld_wide + 4 * (lane + 64 * q)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
                float sv[12];
#pragma unroll
                for (int b = 0; b < 12; ++b) {
                    float s_ = 0.f;
#pragma unroll
                    for (int q = 0; q < 2; ++q) s_ += (xv[b][q][0] * wv[q][0] + xv[b][q][1] * wv[q][1]) + (xv[b][q][2] * wv[q][2] + xv[b][q][3] * wv[q][3]);
                    sv[b] = s_;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                    for (int b = 0; b < 12; ++b) sv[b] += __shfl_xor(sv[b], o, 64);
#pragma unroll
                for (int b = 0; b < 12; ++b) {
                    const int i = i0 + 4 * b;
                    if (lane == 0 && i < n_mine) pub[row0 + t + n_tower * i] = sv[b] + wb;
                }
            }
        }
        (void)tw_last_sums<H1>(hdr, TW_S(0, t), G, reinterpret_cast<const double*>(wsb + L.st1), (M + 63) / 64, n_tower * H1, t * H1, part, sums,
                               flag_s, a.exchange[0], t, n_tower, M, tid);
    }

    if constexpr (PHASE == 2) {
        // ---- global statistics of layer 1 -> A1 (bf16: operand image + global copy); Z2 = A1 W2^T + b2 -> z2; local sums -> exchange[1]
        tw_load_tile(reinterpret_cast<const __bf16*>(T.l2.wh), T.l2.ldwh, 0, H2, H2, OPB, wave, lane);
        if (tid < H1) { par_s[TW_PAR_G1 + tid] = T.l1.gamma[tid]; par_s[TW_PAR_BE1 + tid] = T.l1.beta[tid]; }
        if (tid >= 64 && tid < 64 + H2) par_s[TW_PAR_B2 + tid - 64] = T.l2.bias[tid - 64];
        tw_stats_from_exchange<H1>(a.exchange[0], t, n_tower, a.eps, a.momentum, writer, T.l1.save_mean, T.l1.save_invstd, T.l1.running_mean,
                                   T.l1.running_var, T.l1.num_batches_tracked, col_mean, col_inv, sums, tid);
        {
            constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
            const int c8 = tid % C8, c = c8 * 8, lr0 = tid / C8;
            const uint32_t seed32 = drop_p > 0.f ? g2_seed32(a.seed1, a.seed_offset_dev, 64 + t) : 0u;
            __bf16* a1h = reinterpret_cast<__bf16*>(T.a1h);
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                const int lrc = min(lr, rows - 1);
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(T.l1.z + (int64_t)(row0 + lrc) * T.l1.ldz + c);
                const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(T.l1.z + (int64_t)(row0 + lrc) * T.l1.ldz + c + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                tw_bn_apply8(v, col_mean, col_inv, par_s + TW_PAR_G1, par_s + TW_PAR_BE1, c, relu, drop_p, keep_scale, thr16, seed32, row0 + lr);
                tw_put8(OPA, lr, c8, v);
                if (lr < rows) {
                    bf16x8_t h;
#pragma unroll
                    for (int q = 0; q < 8; ++q) h[q] = (__bf16)v[q];
                    *reinterpret_cast<bf16x8_t*>(a1h + (int64_t)(row0 + lr) * T.lda1h + c) = h;
                }
            }
        }
        tw_drain();
        __syncthreads();
        {
            f32x4_t acc[2][H2 / 16];
            tw_mfma<H2 / 16, H1 / 32>(OPA, OPB, H2 * 128, acc, wave, lane);
            tw_acc_to_tile<H2 / 16>(acc, ct, CS2, wave, lane);
        }
        __syncthreads();
        {
            constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
            const int c = (tid % C8) * 8, lr0 = tid / C8;
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(par_s + TW_PAR_B2 + c), b1 = *reinterpret_cast<const f32x4_t*>(par_s + TW_PAR_B2 + c + 4);
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * CS2 + c) + b0, hi = *reinterpret_cast<const f32x4_t*>(ct + lr * CS2 + c + 4) + b1;
                *reinterpret_cast<f32x4_t*>(ct + lr * CS2 + c) = lo;
                *reinterpret_cast<f32x4_t*>(ct + lr * CS2 + c + 4) = hi;
                if (lr < rows) {
                    *reinterpret_cast<f32x4_t*>(T.l2.z + (int64_t)(row0 + lr) * T.l2.ldz + c) = lo;
                    *reinterpret_cast<f32x4_t*>(T.l2.z + (int64_t)(row0 + lr) * T.l2.ldz + c + 4) = hi;
                }
            }
        }
        __syncthreads();
        tw_fwd_chunk_sums<H2>(ct, CS2, row0, M, part, reinterpret_cast<double*>(wsb + L.st2), n_tower * H2, t * H2, wave, lane);
        (void)tw_last_sums<H2>(hdr, TW_S(1, t), G, reinterpret_cast<const double*>(wsb + L.st2), (M + 63) / 64, n_tower * H2, t * H2, part, sums,
                               flag_s, a.exchange[1], t, n_tower, M, tid);
    }

    if constexpr (PHASE == 3) {
        // ---- global statistics of layer 2 -> A2 (fp32, written out); head + wide term + sigmoid -> out
        if (tid < H2) { par_s[TW_PAR_G2 + tid] = T.l2.gamma[tid]; par_s[TW_PAR_BE2 + tid] = T.l2.beta[tid]; par_s[TW_PAR_WO + tid] = T.wo[tid]; }
        if (tid == H2) par_s[TW_PAR_WO + H2] = T.bo ? T.bo[0] : 0.f;
        tw_stats_from_exchange<H2>(a.exchange[1], t, n_tower, a.eps, a.momentum, writer, T.l2.save_mean, T.l2.save_invstd, T.l2.running_mean,
                                   T.l2.running_var, T.l2.num_batches_tracked, col_mean, col_inv, sums, tid);
        {
            constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
            const int c = (tid % C8) * 8, lr0 = tid / C8;
            const uint32_t seed32 = drop_p > 0.f ? g2_seed32(a.seed2, a.seed_offset_dev, 64 + t) : 0u;
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                const int lrc = min(lr, rows - 1);
                const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(T.l2.z + (int64_t)(row0 + lrc) * T.l2.ldz + c);
                const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(T.l2.z + (int64_t)(row0 + lrc) * T.l2.ldz + c + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                tw_bn_apply8(v, col_mean, col_inv, par_s + TW_PAR_G2, par_s + TW_PAR_BE2, c, relu, drop_p, keep_scale, thr16, seed32, row0 + lr);
#pragma unroll
                for (int q = 0; q < 8; ++q) ct[lr * CS2 + c + q] = v[q];
                if (lr < rows) {
                    *reinterpret_cast<f32x4_t*>(T.a2 + (int64_t)(row0 + lr) * T.lda2 + c) = f32x4_t{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4_t*>(T.a2 + (int64_t)(row0 + lr) * T.lda2 + c + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
                }
            }
        }
        __syncthreads();
        {
            const int lr = tid >> 1, hf = tid & 1;
            const float* arow = ct + lr * CS2 + hf * (H2 / 2);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < H2 / 2; k += 4) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t*>(arow + k);
                const f32x4_t w = *reinterpret_cast<const f32x4_t*>(par_s + TW_PAR_WO + hf * (H2 / 2) + k);
                acc += v[0] * w[0]; acc += v[1] * w[1]; acc += v[2] * w[2]; acc += v[3] * w[3];
            }
            acc += __shfl_xor(acc, 1, 64);
            if (hf == 0 && lr < rows) {
                if (T.bo) acc += par_s[TW_PAR_WO + H2];
                if (has_wide) acc += reinterpret_cast<const float*>(wsb + L.wide)[row0 + lr];
                if (a.sigmoid) acc = 1.f / (1.f + expf(-acc));
                a.out[(int64_t)(row0 + lr) * a.ld_out + t] = acc;
            }
        }
    }

    const bool bce = a.bce_y_i16 != nullptr || a.bce_y_f32 != nullptr;
    if constexpr (PHASE == 4) {
        // ---- logit gradients (BCE over the GLOBAL batch: bce_inv_count), dy2 = mask(a2) * d * wo -> scratch; local sums -> exchange[2];
        //      head / wide weight-gradient partials and the wide term's input gradient (local sums: the arena all-reduce adds the ranks)
        if (tid < H2) { par_s[TW_PAR_WO + tid] = T.wo[tid]; par_s[TW_PAR_B2 + tid] = T.l2.save_mean[tid]; par_s[TW_PAR_BE2 + tid] = T.l2.save_invstd[tid]; }
        if (tid < TW_ROWS) {
            float d_t = 0.f, dsum = 0.f;
            int mine = 0;
            double lp = 0.0;
            if (tid < rows) {
                const int64_t r = row0 + tid;
                int own = 0;
                float tgt = 0.f;
                if (bce) {
                    int64_t c = a.bce_group ? a.bce_group[r] : 0;
                    if (c < 0 || c >= n_tower) c = 0;
                    own = (int)c;
                    tgt = a.bce_y_i16 ? (float)a.bce_y_i16[r] : a.bce_y_f32[r];
                }
                for (int tt = 0; tt < n_tower; ++tt) {
                    const float o = a.out[r * a.ld_out + tt];
                    float dout;
                    if (bce) {
                        if (tt == own) {
                            lp = (double)((tgt - 1.f) * fmaxf(log1pf(-o), -100.f) - tgt * fmaxf(logf(o), -100.f));
                            dout = a.bce_inv_count * (o - tgt) / fmaxf((1.f - o) * o, 1e-12f);
                        } else dout = 0.f;
                    } else dout = a.d_out[r * a.ld_dout + tt];
                    const float d = a.sigmoid ? dout * o * (1.f - o) : dout;
                    dsum += d;
                    if (tt == t) d_t = d;
                }
                mine = (bce ? own : (int)(r % n_tower)) == t;
            }
            d_s[tid] = d_t; dsum_s[tid] = dsum;
            loss_s[tid] = (t == 0) ? lp : 0.0;
            mine_s[tid] = mine;
        }
        __syncthreads();
        if (wave == 0) {
            int base = 0;
#pragma unroll
            for (int h = 0; h < TW_ROWS / 64; ++h) {
                const int flag = mine_s[h * 64 + lane];
                const unsigned long long bal = __ballot(flag != 0);
                if (flag) own_s[base + __popcll(bal & ((1ull << lane) - 1ull))] = h * 64 + lane;
                base += __popcll(bal);
            }
            if (lane == 0) own_s[TW_ROWS] = base;
        }
        __syncthreads();
        float* DZ2 = ct;
        float* XH2 = ct + TW_ROWS * CS2;
        float* A2T = ct + 2 * TW_ROWS * CS2;
        {
            constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
            const int c = (tid % C8) * 8, lr0 = tid / C8;
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                const int lrc = min(lr, rows - 1);
                const bool live = lr < rows;
                const f32x4_t zl = *reinterpret_cast<const f32x4_t*>(T.l2.z + (int64_t)(row0 + lrc) * T.l2.ldz + c);
                const f32x4_t zh = *reinterpret_cast<const f32x4_t*>(T.l2.z + (int64_t)(row0 + lrc) * T.l2.ldz + c + 4);
                const f32x4_t al = *reinterpret_cast<const f32x4_t*>(T.a2 + (int64_t)(row0 + lrc) * T.lda2 + c);
                const f32x4_t ah = *reinterpret_cast<const f32x4_t*>(T.a2 + (int64_t)(row0 + lrc) * T.lda2 + c + 4);
                const float d = d_s[lr];
                float dzv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float zv = q < 4 ? zl[q] : zh[q - 4], av = live ? (q < 4 ? al[q] : ah[q - 4]) : 0.f;
                    float dz = d * par_s[TW_PAR_WO + c + q];
                    if (masked) dz = av > 0.f ? dz * keep_scale : 0.f;
                    dzv[q] = dz;
                    DZ2[lr * CS2 + c + q] = dz;
                    XH2[lr * CS2 + c + q] = (zv - par_s[TW_PAR_B2 + c + q]) * par_s[TW_PAR_BE2 + c + q];
                    A2T[lr * CS2 + c + q] = av;
                }
                if (live) {
                    *reinterpret_cast<f32x4_t*>(T.dy2 + (int64_t)(row0 + lr) * T.lddy2 + c) = f32x4_t{dzv[0], dzv[1], dzv[2], dzv[3]};
                    *reinterpret_cast<f32x4_t*>(T.dy2 + (int64_t)(row0 + lr) * T.lddy2 + c + 4) = f32x4_t{dzv[4], dzv[5], dzv[6], dzv[7]};
                }
            }
        }
        __syncthreads();
        tw_block_sums<H2>(DZ2, XH2, CS2, rows, part, sums, tid);
        {
            double* b2 = reinterpret_cast<double*>(wsb + L.b2) + ((int64_t)jb * n_tower * H2 + t * H2) * 2;
            if (tid < H2) { tw_st(b2 + 2 * tid, sums[tid]); tw_st(b2 + 2 * tid + 1, sums[H2 + tid]); }
        }
        {
            constexpr int NPT = TW_THREADS / H2, RPP = TW_ROWS / NPT;
            const int j = tid % H2, pt = tid / H2;
            double s_ = 0.0, sb = 0.0;
            for (int r = pt * RPP; r < min((pt + 1) * RPP, rows); ++r) { s_ += (double)(d_s[r] * A2T[r * CS2 + j]); sb += (double)d_s[r]; }
            part[(0 * NPT + pt) * H2 + j] = s_; part[(1 * NPT + pt) * H2 + j] = sb;
            __syncthreads();
            float* hd = reinterpret_cast<float*>(wsb + L.hd) + ((int64_t)jb * n_tower + t) * (H2 + 4);
            if (tid < H2) {
                double b = 0.0;
#pragma unroll
                for (int q = 0; q < NPT; ++q) b += part[(0 * NPT + q) * H2 + tid];
                tw_st(hd + tid, (float)b);
            } else if (tid == H2) {
                double b = 0.0;
#pragma unroll
                for (int q = 0; q < NPT; ++q) b += part[(1 * NPT + q) * H2 + 0];
                tw_st(hd + H2, (float)b);
            }
            if (t == 0 && wave == 1) {
                double l_ = (lane < rows ? loss_s[lane] : 0.0) + (lane + 64 < rows ? loss_s[lane + 64] : 0.0);
                l_ = wave_sum_d(l_);
                if (lane == 0) reinterpret_cast<double*>(wsb + L.loss)[jb] = l_;         // read by phase 6 (a later launch)
            }
        }
        if (has_wide) {
            const int K4 = wide_K >> 2;
            const int n_own = own_s[TW_ROWS];
            f32x4_t wv[2], dwv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                wv[q] = (lane + 64 * q) < K4 ? *reinterpret_cast<const f32x4_t*>(a.wide_w + 4 * (lane + 64 * q)) : f32x4_t{0.f, 0.f, 0.f, 0.f};
                dwv[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            float dbv = 0.f;
            const bool rmw = a.wide_dx && a.accumulate_wide_dx;
            TwWideRound wr;
            for (int i0 = wave; i0 < n_own; i0 += 4 * TW_WNB) {
                tw_wide_load(wr, a.wide_x, a.ld_wide, own_s, dsum_s, n_own, i0, row0, K4, lane);
                if (rmw) tw_wide_apply<true>(wr, a.wide_dx, a.ld_wide_dx, dsum_s, row0, n_own, i0, K4, wv, dwv, dbv, lane);
                else tw_wide_apply<false>(wr, a.wide_dx, a.ld_wide_dx, dsum_s, row0, n_own, i0, K4, wv, dwv, dbv, lane);
            }
            float* mine = wdw_s + wave * 520;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if ((lane + 64 * q) < K4) *reinterpret_cast<f32x4_t*>(mine + 4 * (lane + 64 * q)) = dwv[q];
            if (lane == 0) mine[wide_K] = dbv;
            __syncthreads();
            float* wd = reinterpret_cast<float*>(wsb + L.wd) + ((int64_t)t * G + jb) * L.wd_ld;
            for (int k = tid; k <= wide_K; k += TW_THREADS)                                // plain stores: summed by phase 6, a later launch
                wd[k] = ((wdw_s[k] + wdw_s[520 + k]) + wdw_s[2 * 520 + k]) + wdw_s[3 * 520 + k];
        }
        tw_drain();
        __syncthreads();
        if (tw_last_sums<H2>(hdr, TW_S(2, t), G, reinterpret_cast<const double*>(wsb + L.b2), G, n_tower * H2, t * H2, part, sums, flag_s,
                             a.exchange[2], t, n_tower, M, tid)) {
            // the tower's LOCAL parameter gradients: BatchNorm 2 (the sums just formed) and the head (block partials in block order)
            if (tid < H2) {
                if (T.l2.dbeta) T.l2.dbeta[tid] = (float)sums[tid];
                if (T.l2.dgamma) T.l2.dgamma[tid] = (float)sums[H2 + tid];
            }
            for (int k = tid >> 3; k <= H2; k += TW_THREADS / 8) {
                const int p8 = tid & 7;
                const float* hd = reinterpret_cast<const float*>(wsb + L.hd) + (int64_t)t * (H2 + 4) + k;
                float s_ = 0.f;
                for (int b0 = p8; b0 < G; b0 += 8) s_ += tw_ld(hd + (int64_t)b0 * n_tower * (H2 + 4));
                s_ += __shfl_xor(s_, 1, 64); s_ += __shfl_xor(s_, 2, 64); s_ += __shfl_xor(s_, 4, 64);
                if (p8 == 0) {
                    if (k < H2) { if (T.dwo) T.dwo[k] = s_; }
                    else if (T.dbo) T.dbo[0] = s_;
                }
            }
        }
    }

    if constexpr (PHASE == 5) {
        // ---- global sums of layer 2 -> dZ2 (bf16: operand image + global copy); dA1 = dZ2 W2; dy1 = mask(a1) * dA1 -> scratch;
        //      local sums -> exchange[3]
        tw_load_tile(reinterpret_cast<const __bf16*>(T.l2.wt), T.l2.ldwt, 0, H1, H1, OPB, wave, lane);
        const double* ex = a.exchange[2];
        const int Ms = (int)(ex[2 * n_tower * H2 + t] + 0.5);
        const float invM = Ms > 0 ? 1.f / (float)Ms : 0.f;
        if (tid < H2) {
            const float inv = T.l2.save_invstd[tid];
            par_s[TW_PAR_B2 + tid] = T.l2.save_mean[tid]; par_s[TW_PAR_G2 + tid] = T.l2.gamma[tid] * inv; par_s[TW_PAR_BE2 + tid] = inv;
            col_mean[tid] = (float)ex[2 * (t * H2 + tid)]; col_inv[tid] = (float)ex[2 * (t * H2 + tid) + 1];     // db, dg of the global batch
        }
        if (tid >= 64 && tid < 64 + H1) { par_s[TW_PAR_B1 + tid - 64] = T.l1.save_mean[tid - 64]; par_s[TW_PAR_BE1 + tid - 64] = T.l1.save_invstd[tid - 64]; }
        __syncthreads();
        {
            constexpr int C8 = H2 / 8, RPP = TW_THREADS / C8;
            const int c8 = tid % C8, c = c8 * 8, lr0 = tid / C8;
            __bf16* dzh = reinterpret_cast<__bf16*>(T.l2.dzh);
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                const int lrc = min(lr, rows - 1);
                const f32x4_t zl = *reinterpret_cast<const f32x4_t*>(T.l2.z + (int64_t)(row0 + lrc) * T.l2.ldz + c);
                const f32x4_t zh = *reinterpret_cast<const f32x4_t*>(T.l2.z + (int64_t)(row0 + lrc) * T.l2.ldz + c + 4);
                const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(T.dy2 + (int64_t)(row0 + lrc) * T.lddy2 + c);
                const f32x4_t dh = *reinterpret_cast<const f32x4_t*>(T.dy2 + (int64_t)(row0 + lrc) * T.lddy2 + c + 4);
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float xh = ((q < 4 ? zl[q] : zh[q - 4]) - par_s[TW_PAR_B2 + c + q]) * par_s[TW_PAR_BE2 + c + q];
                    const float dz = lr < rows ? (q < 4 ? dl[q] : dh[q - 4]) : 0.f;
                    v[q] = par_s[TW_PAR_G2 + c + q] * (dz - invM * (col_mean[c + q] + xh * col_inv[c + q]));
                }
                tw_put8(OPA, lr, c8, v);
                if (lr < rows) {
                    bf16x8_t h;
#pragma unroll
                    for (int q = 0; q < 8; ++q) h[q] = (__bf16)v[q];
                    *reinterpret_cast<bf16x8_t*>(dzh + (int64_t)(row0 + lr) * T.l2.lddzh + c) = h;
                }
            }
        }
        tw_drain();
        __syncthreads();
        float* DZ1 = ct;
        float* XH1 = ct + TW_ROWS * CS1;
        {
            f32x4_t acc[2][H1 / 16];
            tw_mfma<H1 / 16, H2 / 32>(OPA, OPB, 0, acc, wave, lane);
            tw_acc_to_tile<H1 / 16>(acc, DZ1, CS1, wave, lane);
        }
        __syncthreads();
        {
            constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
            const int c = (tid % C8) * 8, lr0 = tid / C8;
            const __bf16* a1h = reinterpret_cast<const __bf16*>(T.a1h);
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                const int lrc = min(lr, rows - 1);
                const f32x4_t zl = *reinterpret_cast<const f32x4_t*>(T.l1.z + (int64_t)(row0 + lrc) * T.l1.ldz + c);
                const f32x4_t zh = *reinterpret_cast<const f32x4_t*>(T.l1.z + (int64_t)(row0 + lrc) * T.l1.ldz + c + 4);
                const bf16x8_t m8 = *reinterpret_cast<const bf16x8_t*>(a1h + (int64_t)(row0 + lrc) * T.lda1h + c);
                float dzv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float dz = DZ1[lr * CS1 + c + q];
                    if (masked) dz = (float)m8[q] > 0.f ? dz * keep_scale : 0.f;
                    if (lr >= rows) dz = 0.f;
                    dzv[q] = dz;
                    DZ1[lr * CS1 + c + q] = dz;
                    XH1[lr * CS1 + c + q] = ((q < 4 ? zl[q] : zh[q - 4]) - par_s[TW_PAR_B1 + c + q]) * par_s[TW_PAR_BE1 + c + q];
                }
                if (lr < rows) {
                    *reinterpret_cast<f32x4_t*>(T.dy1 + (int64_t)(row0 + lr) * T.lddy1 + c) = f32x4_t{dzv[0], dzv[1], dzv[2], dzv[3]};
                    *reinterpret_cast<f32x4_t*>(T.dy1 + (int64_t)(row0 + lr) * T.lddy1 + c + 4) = f32x4_t{dzv[4], dzv[5], dzv[6], dzv[7]};
                }
            }
        }
        __syncthreads();
        tw_block_sums<H1>(DZ1, XH1, CS1, rows, part, sums, tid);
        {
            double* b1 = reinterpret_cast<double*>(wsb + L.b1) + ((int64_t)jb * n_tower * H1 + t * H1) * 2;
            if (tid < H1) { tw_st(b1 + 2 * tid, sums[tid]); tw_st(b1 + 2 * tid + 1, sums[H1 + tid]); }
        }
        tw_drain();
        __syncthreads();
        if (tw_last_sums<H1>(hdr, TW_S(3, t), G, reinterpret_cast<const double*>(wsb + L.b1), G, n_tower * H1, t * H1, part, sums, flag_s,
                             a.exchange[3], t, n_tower, M, tid)) {
            if (tid < H1) {
                if (T.l1.dbeta) T.l1.dbeta[tid] = (float)sums[tid];
                if (T.l1.dgamma) T.l1.dgamma[tid] = (float)sums[H1 + tid];
            }
        }
    }

    if constexpr (PHASE == 6) {
        // ---- global sums of layer 1 -> dZ1 (bf16: operand image + global copy); dX = dZ1 W1 -> dx; the wide term's weight gradient and
        //      the loss: sums over this rank's workgroups (partials of phase 4)
        tw_load_tile(reinterpret_cast<const __bf16*>(T.l1.wt), T.l1.ldwt, 0, H0, H0, OPB, wave, lane);
        const double* ex = a.exchange[3];
        const int Ms = (int)(ex[2 * n_tower * H1 + t] + 0.5);
        const float invM = Ms > 0 ? 1.f / (float)Ms : 0.f;
        if (tid < H1) {
            const float inv = T.l1.save_invstd[tid];
            par_s[TW_PAR_B1 + tid] = T.l1.save_mean[tid]; par_s[TW_PAR_G1 + tid] = T.l1.gamma[tid] * inv; par_s[TW_PAR_BE1 + tid] = inv;
            col_mean[tid] = (float)ex[2 * (t * H1 + tid)]; col_inv[tid] = (float)ex[2 * (t * H1 + tid) + 1];
        }
        __syncthreads();
        {
            constexpr int C8 = H1 / 8, RPP = TW_THREADS / C8;
            const int c8 = tid % C8, c = c8 * 8, lr0 = tid / C8;
            __bf16* dzh = reinterpret_cast<__bf16*>(T.l1.dzh);
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                const int lrc = min(lr, rows - 1);
                const f32x4_t zl = *reinterpret_cast<const f32x4_t*>(T.l1.z + (int64_t)(row0 + lrc) * T.l1.ldz + c);
                const f32x4_t zh = *reinterpret_cast<const f32x4_t*>(T.l1.z + (int64_t)(row0 + lrc) * T.l1.ldz + c + 4);
                const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(T.dy1 + (int64_t)(row0 + lrc) * T.lddy1 + c);
                const f32x4_t dh = *reinterpret_cast<const f32x4_t*>(T.dy1 + (int64_t)(row0 + lrc) * T.lddy1 + c + 4);
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float xh = ((q < 4 ? zl[q] : zh[q - 4]) - par_s[TW_PAR_B1 + c + q]) * par_s[TW_PAR_BE1 + c + q];
                    const float dz = lr < rows ? (q < 4 ? dl[q] : dh[q - 4]) : 0.f;
                    v[q] = par_s[TW_PAR_G1 + c + q] * (dz - invM * (col_mean[c + q] + xh * col_inv[c + q]));
                }
                tw_put8(OPA, lr, c8, v);
                if (lr < rows) {
                    bf16x8_t h;
#pragma unroll
                    for (int q = 0; q < 8; ++q) h[q] = (__bf16)v[q];
                    *reinterpret_cast<bf16x8_t*>(dzh + (int64_t)(row0 + lr) * T.l1.lddzh + c) = h;
                }
            }
        }
        tw_drain();
        __syncthreads();
        {
            f32x4_t acc[2][H0 / 16];
            tw_mfma<H0 / 16, H1 / 32>(OPA, OPB, 0, acc, wave, lane);
            tw_acc_to_tile<H0 / 16>(acc, ct, CSX, wave, lane);
        }
        __syncthreads();
        if (T.dx) {
            constexpr int C8 = H0 / 8, RPP = TW_THREADS / C8;
            const int c = (tid % C8) * 8, lr0 = tid / C8;
            const bool accx = T.accumulate_dx != 0;
#pragma unroll
            for (int lr = lr0; lr < TW_ROWS; lr += RPP) {
                if (lr >= rows) break;
                f32x4_t lo = *reinterpret_cast<const f32x4_t*>(ct + lr * CSX + c), hi = *reinterpret_cast<const f32x4_t*>(ct + lr * CSX + c + 4);
                float* dst = T.dx + (int64_t)(row0 + lr) * T.lddx + c;
                if (accx) { lo = *reinterpret_cast<const f32x4_t*>(dst) + lo; hi = *reinterpret_cast<const f32x4_t*>(dst + 4) + hi; }
                *reinterpret_cast<f32x4_t*>(dst) = lo;
                *reinterpret_cast<f32x4_t*>(dst + 4) = hi;
            }
        }
        if (has_wide) {
            const float* wd = reinterpret_cast<const float*>(wsb + L.wd);
            for (int k = (int)blockIdx.x * 4 + wave; k <= wide_K; k += n_wg * 4) {
                float s_ = 0.f;
                for (int p = lane; p < n_wg; p += 64) s_ += wd[(int64_t)p * L.wd_ld + k];
                s_ = wave_sum(s_);
                if (lane == 0) {
                    if (k < wide_K) { if (a.wide_dw) a.wide_dw[k] = s_; }
                    else if (a.wide_dbias) a.wide_dbias[0] = s_;
                }
            }
        }
        if (bce && blockIdx.x == 0 && wave == 1 && a.bce_loss) {
            const double* lp = reinterpret_cast<const double*>(wsb + L.loss);
            double s_ = 0.0;
            for (int b = lane; b < G; b += 64) s_ += lp[b];
            s_ = wave_sum_d(s_);
            if (lane == 0) *a.bce_loss = (float)(s_ * (double)a.bce_inv_count);
        }
    }
}

// =================================================================================================
// host
// =================================================================================================
static int tower_check(const cdc_tower_args* a, const char* who, bool bwd, int min_rows = 2) {
    CDC_CHECK_ARG(a && a->n_tower > 0 && a->n_tower <= CDC_TOWER_MAX, CDC_E_BADARG, "%s: bad tower count", who);
    CDC_CHECK_ARG((a->H0 == 64 || a->H0 == 128) && a->H1 == 64 && a->H2 == 32, CDC_E_BADARG,
                  "%s: instantiated for H0 in {64,128}, H1 = 64, H2 = 32 (got %d, %d, %d)", who, a->H0, a->H1, a->H2);
    CDC_CHECK_ARG(a->M >= min_rows, CDC_E_BADARG, "%s: needs at least two rows (a batch of one skips BatchNorm: use the unfused launches)", who);
    const int64_t G = cdc_ceil_div(a->M, TW_ROWS);
    CDC_CHECK_ARG(a->n_tower * G <= 256, CDC_E_TOOBIG, "%s: %lld workgroups cannot all be resident (one per CU)", who, (long long)(a->n_tower * G));
    CDC_CHECK_ARG(a->drop_p >= 0.f && a->drop_p < 1.f, CDC_E_BADARG, "%s: dropout p out of range", who);
    CDC_CHECK_ARG(a->workspace && (((uintptr_t)a->workspace) & 127) == 0 && a->out && a->ld_out >= a->n_tower, CDC_E_BADARG, "%s: workspace / out", who);
    auto al16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    if (a->wide_x) {
        CDC_CHECK_ARG(a->wide_w && a->wide_K > 0 && a->wide_K <= 512 && a->wide_K % 4 == 0 && a->ld_wide % 4 == 0 && al16(a->wide_x) && al16(a->wide_w),
                      CDC_E_BADARG, "%s: wide term malformed (K <= 512, K %% 4 == 0, 16-byte aligned rows)", who);
        if (bwd && a->wide_dx) CDC_CHECK_ARG(al16(a->wide_dx) && a->ld_wide_dx % 4 == 0, CDC_E_ALIGN, "%s: wide_dx alignment", who);
    }
    for (int t = 0; t < a->n_tower; ++t) {
        const cdc_tower_desc& T = a->t[t];
        CDC_CHECK_ARG(T.xh && al16(T.xh) && T.ldxh % 8 == 0 && T.ldxh >= a->H0, CDC_E_BADARG, "%s: tower %d input", who, t);
        const cdc_tower_layer* Ls[2] = {&T.l1, &T.l2};
        const int Ns[2] = {a->H1, a->H2}, Ks[2] = {a->H0, a->H1};
        for (int l = 0; l < 2; ++l) {
            const cdc_tower_layer& Y = *Ls[l];
            CDC_CHECK_ARG(Y.bias && Y.z && al16(Y.z) && Y.ldz % 4 == 0 && Y.ldz >= Ns[l] && Y.gamma && Y.beta && Y.save_mean && Y.save_invstd,
                          CDC_E_BADARG, "%s: tower %d layer %d", who, t, l + 1);
            if (!bwd) CDC_CHECK_ARG(Y.wh && al16(Y.wh) && Y.ldwh % 8 == 0 && Y.ldwh >= (Ks[l] + 63) / 64 * 64, CDC_E_BADARG, "%s: tower %d layer %d weight copy", who, t, l + 1);
            else CDC_CHECK_ARG(Y.wt && al16(Y.wt) && Y.ldwt % 8 == 0 && Y.ldwt >= 64 && Y.dzh && al16(Y.dzh) && Y.lddzh % 8 == 0 && Y.lddzh >= Ns[l],
                               CDC_E_BADARG, "%s: tower %d layer %d transposed weight copy / dzh", who, t, l + 1);
        }
        CDC_CHECK_ARG(T.a1h && al16(T.a1h) && T.lda1h % 8 == 0 && T.a2 && al16(T.a2) && T.lda2 % 4 == 0 && T.wo, CDC_E_BADARG, "%s: tower %d activations / head", who, t);
        if (bwd && T.dx) CDC_CHECK_ARG(al16(T.dx) && T.lddx % 4 == 0 && T.lddx >= a->H0, CDC_E_ALIGN, "%s: tower %d dx", who, t);
    }
    return 0;
}

extern "C" int64_t cdc_tower_workspace_bytes(const cdc_tower_args* a) {
    if (!a || a->n_tower <= 0 || a->n_tower > CDC_TOWER_MAX || a->M <= 0) return -1;
    return tw_layout(a->n_tower, a->H1, a->H2, a->M, a->wide_x ? a->wide_K : 0).total;
}

template <typename K>
static void tower_attr(K kern, int bytes) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }

extern "C" int cdc_tower_fwd(const cdc_tower_args* a, void* stream) {
    const int rc = tower_check(a, "tower_fwd", false);
    if (rc) return rc;
    const unsigned grid = (unsigned)(a->n_tower * cdc_ceil_div(a->M, TW_ROWS));
    static bool attr_done = false;
    if (!attr_done) {
        tower_attr(k_tower_fwd<1, 64, 32>, TwFwdCfg<1, 64, 32>::SMEM);
        tower_attr(k_tower_fwd<2, 64, 32>, TwFwdCfg<2, 64, 32>::SMEM);
        attr_done = true;
    }
    constexpr int lds1 = TwFwdCfg<1, 64, 32>::SMEM, lds2 = TwFwdCfg<2, 64, 32>::SMEM;
    if (a->H0 == 64) hipLaunchKernelGGL((k_tower_fwd<1, 64, 32>), dim3(grid), dim3(TW_THREADS), lds1, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((k_tower_fwd<2, 64, 32>), dim3(grid), dim3(TW_THREADS), lds2, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("tower_fwd");
    return 0;
}

extern "C" int cdc_tower_bwd(const cdc_tower_args* a, void* stream) {
    const int rc = tower_check(a, "tower_bwd", true);
    if (rc) return rc;
    const bool bce = a->bce_y_i16 || a->bce_y_f32;
    CDC_CHECK_ARG(bce || a->d_out, CDC_E_BADARG, "tower_bwd: needs the output gradient or the fused loss");
    CDC_CHECK_ARG(!bce || (a->sigmoid && a->bce_loss && a->bce_inv_count > 0.f), CDC_E_BADARG, "tower_bwd: the fused BCE needs sigmoid outputs and a loss pointer");
    const unsigned grid = (unsigned)(a->n_tower * cdc_ceil_div(a->M, TW_ROWS));
    static bool attr_done = false;
    if (!attr_done) {
        tower_attr(k_tower_bwd<1, 64, 32>, TwBwdCfg<1, 64, 32>::SMEM);
        tower_attr(k_tower_bwd<2, 64, 32>, TwBwdCfg<2, 64, 32>::SMEM);
        attr_done = true;
    }
    constexpr int lds1 = TwBwdCfg<1, 64, 32>::SMEM, lds2 = TwBwdCfg<2, 64, 32>::SMEM;
    if (a->H0 == 64) hipLaunchKernelGGL((k_tower_bwd<1, 64, 32>), dim3(grid), dim3(TW_THREADS), lds1, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((k_tower_bwd<2, 64, 32>), dim3(grid), dim3(TW_THREADS), lds2, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("tower_bwd");
    return 0;
}

// Both directions of a training step in one launch: the fused-loss form of cdc_tower_bwd's arguments (which include everything
// cdc_tower_fwd reads).  Same results as cdc_tower_fwd followed by cdc_tower_bwd, bit for bit.
extern "C" int cdc_tower_step(const cdc_tower_args* a, void* stream) {
    int rc = tower_check(a, "tower_step", false);
    if (rc) return rc;
    rc = tower_check(a, "tower_step", true);
    if (rc) return rc;
    const bool bce = a->bce_y_i16 || a->bce_y_f32;
    CDC_CHECK_ARG(bce && a->sigmoid && a->bce_loss && a->bce_inv_count > 0.f, CDC_E_BADARG,
                  "tower_step: needs the fused BCE (labels, sigmoid outputs, a loss pointer): with an output gradient from outside the two directions are two launches");
    const unsigned grid = (unsigned)(a->n_tower * cdc_ceil_div(a->M, TW_ROWS));
    static bool attr_done = false;
    if (!attr_done) {
        tower_attr(k_tower_step<1, 64, 32>, TwStepCfg<1, 64, 32>::SMEM);
        tower_attr(k_tower_step<2, 64, 32>, TwStepCfg<2, 64, 32>::SMEM);
        attr_done = true;
    }
    constexpr int lds1 = TwStepCfg<1, 64, 32>::SMEM, lds2 = TwStepCfg<2, 64, 32>::SMEM;
    if (a->H0 == 64) hipLaunchKernelGGL((k_tower_step<1, 64, 32>), dim3(grid), dim3(TW_THREADS), lds1, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((k_tower_step<2, 64, 32>), dim3(grid), dim3(TW_THREADS), lds2, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("tower_step");
    return 0;
}

template <int NK0, int PHASE>
static void tower_dp_launch(const cdc_tower_args* a, unsigned grid, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) { tower_attr(k_tower_dp<NK0, 64, 32, PHASE>, TwDpCfg::SMEM); attr_done = true; }
    constexpr int lds = TwDpCfg::SMEM;
    hipLaunchKernelGGL((k_tower_dp<NK0, 64, 32, PHASE>), dim3(grid), dim3(TW_THREADS), lds, st, *a);
}
extern "C" int cdc_tower_dp(const cdc_tower_args* a, int32_t phase, void* stream) {
    CDC_CHECK_ARG(phase >= 1 && phase <= 6, CDC_E_BADARG, "tower_dp: phase must be 1..6");
    const int rc = tower_check(a, "tower_dp", phase >= 4, 1);        // (the GLOBAL batch decides about BatchNorm; a rank may hold one row)
    if (rc) return rc;
    for (int e = 0; e < 4; ++e) CDC_CHECK_ARG(a->exchange[e] && (((uintptr_t)a->exchange[e]) & 7) == 0, CDC_E_BADARG, "tower_dp: exchange buffer %d", e);
    if (phase >= 4) {
        const bool bce = a->bce_y_i16 || a->bce_y_f32;
        CDC_CHECK_ARG(bce || a->d_out, CDC_E_BADARG, "tower_dp: needs the output gradient or the fused loss");
        CDC_CHECK_ARG(!bce || (a->sigmoid && a->bce_loss && a->bce_inv_count > 0.f), CDC_E_BADARG, "tower_dp: the fused BCE needs sigmoid outputs and a loss pointer");
        for (int t = 0; t < a->n_tower; ++t)
            CDC_CHECK_ARG(a->t[t].dy2 && a->t[t].dy1 && (((uintptr_t)a->t[t].dy2 | (uintptr_t)a->t[t].dy1) & 15) == 0 && a->t[t].lddy2 % 4 == 0 &&
                              a->t[t].lddy1 % 4 == 0, CDC_E_BADARG, "tower_dp: tower %d gradient scratch", t);
    }
    const unsigned grid = (unsigned)(a->n_tower * cdc_ceil_div(a->M, TW_ROWS));
    hipStream_t st = (hipStream_t)stream;
#define TW_DP(P) case P: if (a->H0 == 64) tower_dp_launch<1, P>(a, grid, st); else tower_dp_launch<2, P>(a, grid, st); break;
    switch (phase) { TW_DP(1) TW_DP(2) TW_DP(3) TW_DP(4) TW_DP(5) TW_DP(6) }
#undef TW_DP
    CDC_LAUNCH_CHECK("tower_dp");
    return 0;
}
