// cgc.hip — the boundary between two extraction levels of PLE as one launch per direction (include/cdcmdr.h: cdc_cgc_mid_*).
//
// Reference: CGC.forward called level after level (model/ple.py:54-57,96-125): level k's softmax gates pool its experts, the
// pooled vectors are the inputs of level k+1's experts (one nn.Linear + ReLU + dropout each, model/layer.py:185-191) and gates
// (model/ple.py:89-94), whose outputs are pooled again.  Unfused that is pool -> grouped contraction -> pool: three launches
// whose intermediates ([B, n_gate*H1] pooled vectors, [B, n_exp*H2] expert tiles) make a round trip through memory between
// latency-bound launches.  Here a workgroup owns 16 batch rows and walks the whole boundary: the pooled vectors are formed once
// into LDS as bf16 (the MFMA A operand — the same rounding the shadow gets), the level-k+1 experts are 16 x 16 MFMA tiles whose
// B fragments come straight from the L2-resident bf16 weights, and the second pooling reads the tiles out of LDS.  The
// backward runs the same chain in reverse (pool backward -> grad-input contraction -> pool backward) and writes only what the
// grad-weight launches and the next grad-input launch read: bf16 dZ of both expert levels and the gate-logit gradients.
//
// Shape of the kernels (what three slower versions taught, profiles/round3/README.md):
//   * 1024 threads per workgroup: in the forward's pooling phases wave w IS batch row w (lane = column: two columns of H1 = 128, one
//     of H2 = 64), in the MFMA phases wave w takes tiles w, w+16, ...  A 256-thread version ran one wave per SIMD and every LDS /
//     DPP / scalar-load latency of its dependent chains lay bare (30 / 47 us); four waves per SIMD cover each other.
//   * nothing is unrolled over (gate, expert): a fully unrolled version was 50-85 KB of straight-line code that is executed
//     once per workgroup, i.e. fetched cold.  Loops are rolled; what they index with run-time values sits in LDS, put there by
//     direct-to-LDS loads issued at kernel entry (no registers, so the issuing loop is rolled as well).
//   * every global load that depends on nothing is issued at entry (expert rows, B fragments + bias of the wave's tiles); no
//     load sits between a value's first use and the previous barrier.
//   * the argument block is copied from the kernarg segment to LDS at entry, every thread one 8-byte piece (round 4): what the entry
//     reads of it — counts, then the descriptors they index, then what those point to — was a chain of dependent reads of memory
//     that is cold at every launch, 1-2 us each and most of a 9-10 us entry; now ONE such read.  (Indexing the by-value struct with
//     run-time values had made the compiler keep a private copy of all 2.5 KB of it in scratch; round 3 read it through the
//     kernarg segment pointer.)
//   * round 4, after phase stamps (tools/mid_trace.py, profiles/round4/README.md section 11): with 16 waves on the CU a row-wise
//     phase is bound by instruction ISSUE as soon as it spends a whole wave on one row — the backward's two pool phases (a 64-lane
//     reduction per (expert, gate) pair and row: 7.4 us each) now give a row to the 16 lanes of a DPP row, a lane four (two times
//     four) columns, a wave four rows x every fourth expert; gate softmax and gate-logit gradients take a lane per TERM instead of a
//     lane per gate (sums in index order out of LDS: the same bits); tables are built from one descriptor read per lane.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define MID_BM 16
#define MID_THREADS 1024
#define MID_WAVES (MID_THREADS / 64)
#define MID_G CDC_MID_MAX_GATE
#define MID_E CDC_MID_MAX_EXPERT
#define MID_KARG __attribute__((address_space(4)))   /* the kernarg segment: constant address space */
#define MID_JPW 3                       /* forward: MFMA tiles per wave whose B fragments are fetched at kernel entry */
#define MID_JPW2 2                      /* backward: the same for the grad-input tiles */
#define MID_XW 2                        /* backward: experts whose row pieces a lane holds in registers at once (a wave takes every fourth expert) */
#ifndef MID_TRACE
#define MID_TRACE 0                     /* probe builds only (tools/mid_trace.py): thread 0 of workgroup 0 stamps the 100 MHz wall clock at phase boundaries */
#endif
#if MID_TRACE
__device__ unsigned long long g_mid_stamps[32];
#define MID_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mid_stamps[(i)] = wall_clock64(); } while (0)
extern "C" int cdc_debug_mid_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mid_stamps), sizeof(unsigned long long) * 32); }
#else
#define MID_STAMP(i) do { } while (0)
#endif
#define MID_MAXSTEP 6                   /* K steps of 32 per grad-input tile: 2 per expert reading the source + 1 per gate */

template <int H1, int H2>
struct MidCfg {
    // row strides of the images the MFMA A fragments are read from with ds_read_b128: 32 bytes modulo 64 — the one class of
    // strides that is conflict-free for that instruction's lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...:
    // MI355X_MICROARCH.md, LDS); H + 8 elements (16 bytes modulo 64) measured 35 % extra LDS cycles (profiles/round3)
    static constexpr int PH_LD = H1 + 16;                           // bf16 row stride of the pooled level-k vectors in LDS
    static constexpr int E2_LD = H2 + 4;                            // fp32 row stride of the level-k+1 expert tiles
    static constexpr int DZ_LD = H2 + 16;                           // bf16 row stride of dZ (level k+1) in LDS
    static constexpr int DP_LD = H1 + 4;                            // fp32 row stride of d(pooled level-k) in LDS
    static constexpr int DL_LD = 32 + 16;                           // bf16 row stride of the gate-logit gradients (one K step of 32)
    static constexpr int NT1 = H1 / 16, NT2 = H2 / 16;
    static_assert(H1 == 128 && H2 == 64, "lane = column mapping: two columns of H1, one of H2");
};

__device__ __forceinline__ void mid_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void mid_glds4(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
// sum over the 16 lanes of a DPP row, result in every lane of the row: four cross-lane adds in the VALU (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror) — the pool backwards give a batch row to 16 lanes
__device__ __forceinline__ float mid_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// sel lists and position tables in LDS: sel[g][j] (bytes) and pos[g][e] = index of expert e in gate g's ascending sel list or -1.
// Thread (g, j) fetches ONE entry of gate g's list from the argument block (one round trip for the whole table: a loop over the list
// in every thread was a dependent chain of them, several microseconds at the head of both kernels); the 16 threads of a gate sit in
// one wave, whose LDS instructions execute in order, so thread (g, e) finds the complete list in LDS for its look-up.
struct MidTabReg { int ns, sv; };
template <typename GatePtr>
__device__ __forceinline__ MidTabReg mid_tables_fetch(GatePtr gates, int n_gate, int tid) {            // the round trip (issue early)
    MidTabReg t = {0, 0};
    if (tid < MID_G * MID_E) {
        const int g = tid / MID_E, e = tid % MID_E;
        if (g < n_gate) {
            t.ns = gates[g].n_sel;
            t.sv = gates[g].sel[e];                                    // (slots >= n_sel of the argument block are readable; masked below)
        }
    }
    return t;
}
__device__ __forceinline__ void mid_tables_store(signed char (*pos)[MID_E], unsigned char (*sel)[CDC_MAX_SEL], const MidTabReg& t, int tid) {
    if (tid < MID_G * MID_E) {
        const int g = tid / MID_E, e = tid % MID_E;
        sel[g][e] = (unsigned char)(e < t.ns ? t.sv : 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int p = -1;
#pragma unroll
        for (int q = 0; q < MID_E; ++q)
            if (q < t.ns && sel[g][q] == e) p = q;
        pos[g][e] = (signed char)p;
    }
}
static_assert(MID_E == CDC_MAX_SEL, "mid_tables_fetch / _store fill both tables with one index");
static_assert(2 * MID_E + MID_G <= 64, "k_cgc_mid_bwd: one lane of a wave per grad-input K-step candidate");

// softmax over n_sel <= 16 values exactly as cdc_gate_pool_fwd computes it (max-subtract, expf, sum in index order, one divide),
// with ONE gate's logits spread over the 16 lanes of a DPP row (lane j holds logit j): the maximum is order-free,
// every lane takes one expf instead of one lane sixteen, and the sum is formed in index order out of the LDS slot the probabilities
// go to anyway (by every lane of the row: no broadcast) — the bits of mid_softmax.  active: the row's gate exists (its slot may be
// written); valid: j < n_sel.  Returns the lane's probability.
__device__ __forceinline__ float mid_softmax_row16(float l, bool active, bool valid, float* slot, int j) {
    const float v = valid ? l : -INFINITY;
    float mx = v;
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0xB1, 0xF, 0xF, true)));
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x4E, 0xF, 0xF, true)));
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x141, 0xF, 0xF, true)));
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x140, 0xF, 0xF, true)));
    float p = valid ? expf(v - mx) : 0.f;
    if (active) slot[j] = p;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float sum = 0.f;
    if (active) {
#pragma unroll
        for (int q = 0; q < CDC_MAX_SEL; q += 4) {
            const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(slot + q);
            sum += t4[0]; sum += t4[1]; sum += t4[2]; sum += t4[3];
        }
    }
    const float inv = 1.f / sum;
    p *= inv;
    __builtin_amdgcn_wave_barrier();                                   // (every lane has read the slot before it is overwritten)
    if (active) slot[j] = p;
    return p;
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <int H1, int H2>
__global__ void __launch_bounds__(MID_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) k_cgc_mid_fwd(const cdc_cgc_mid_fwd_args a_by_value) {
    CDC_PRIO_MAIN();
    (void)a_by_value;
    MID_STAMP(0);
    // the argument block -> LDS, all of it at once (every thread one 8-byte piece): what the entry reads of it — counts, then the
    // descriptors they index, then what those point to — was a chain of dependent reads of memory that is cold at every launch (1-2 us
    // each); now ONE such read, and the chain runs out of LDS
    __shared__ __attribute__((aligned(16))) cdc_cgc_mid_fwd_args a_s;
    {
        static_assert(sizeof(cdc_cgc_mid_fwd_args) % 8 == 0, "copied in 8-byte pieces");
        const MID_KARG unsigned long long* src = (const MID_KARG unsigned long long*)__builtin_amdgcn_kernarg_segment_ptr();
        for (int i = threadIdx.x; i < (int)(sizeof(cdc_cgc_mid_fwd_args) / 8); i += MID_THREADS) reinterpret_cast<unsigned long long*>(&a_s)[i] = src[i];
        __syncthreads();
    }
    const cdc_cgc_mid_fwd_args& a = a_s;
    typedef MidCfg<H1, H2> Cfg;
    constexpr int KS = H1 / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ signed char pos1[MID_G][MID_E], pos2[MID_G][MID_E];
    __shared__ __attribute__((aligned(16))) unsigned char sel1[MID_G][CDC_MAX_SEL], sel2[MID_G][CDC_MAX_SEL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // uniform: descriptor reads indexed by it are scalar loads
    const int64_t row0 = (int64_t)blockIdx.x * MID_BM;
    const int64_t B = ((int64_t)__builtin_amdgcn_readfirstlane((int)(a.B >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)a.B);
    const int ng1 = __builtin_amdgcn_readfirstlane(a.n_gate1), ng2 = __builtin_amdgcn_readfirstlane(a.n_gate2),
              ne1 = __builtin_amdgcn_readfirstlane(a.n_exp1), ne2 = __builtin_amdgcn_readfirstlane(a.n_exp2);
    // LDS: p1 [16][ng1][16] f32 | p2 [16][ng2][16] f32 | pooled [ng1][16][PH_LD] bf16 | U = { xs [ne1][16][H1] f32 (phase A) ,
    //      x2 [ne2][16][E2_LD] f32 (phases B..D) }
    float* p1 = reinterpret_cast<float*>(smem);
    float* p2 = p1 + MID_BM * ng1 * 16;
    __bf16* ph = reinterpret_cast<__bf16*>(p2 + MID_BM * ng2 * 16);
    float* xs = reinterpret_cast<float*>(ph + (size_t)ng1 * MID_BM * Cfg::PH_LD);
    float* x2 = xs;
    const int n_tile_jobs = ne2 * Cfg::NT2, n_jobs = n_tile_jobs + ng2;
    const int frow = lane & 15, fk = (lane >> 4) * 8;
    const int r = wave;                                                // row-wise phases: wave = batch row, lane = column
    const int64_t row = row0 + r;
    const bool live = row < B;
    const int64_t last_row = B - 1;

    const int32_t sp_val = (a.drop_p > 0.f && a.seed_offset_dev) ? *a.seed_offset_dev : 0;     // (consumed behind the entry's other loads)
    // ---- (0) the level-k gate logits of this wave's row, lane g = gate g: they head the longest dependent chain of the entry
    //      (descriptor -> logits -> softmax -> pooling), so they are issued first; the descriptors come by scalar loads (a lane-indexed
    //      read of the argument block is a vector-memory round trip of its own)
    // EVERY read of the argument block the entry needs goes out before the first of them is used (the block lives where the launch put
    // it — host-visible memory for a plain launch: a read is a microsecond or two, and reads issued one behind the other's result were
    // most of the entry's 9 us): the gates' descriptors — all MID_G slots, unconditionally: the slots exist whatever n_gate1 is —, the
    // descriptors of this wave's first tiles, the table entries (a lane-indexed read)
    // gate probabilities: lane (q, j) = (lane >> 4, lane & 15) takes logit j of gate q (pass 0) and of gate q + 4 (pass 1)
    const int gq = lane >> 4, gj = lane & 15;
    int ns_g1[2] = {0, 0}, ns_g2[2] = {0, 0};
    float* pr_g1[2] = {nullptr, nullptr};
    float* pr_g2[2] = {nullptr, nullptr};
    const float* lg_g1[2] = {nullptr, nullptr};
#pragma unroll
    for (int g = 0; g < MID_G; ++g) {
        const int ns = a.g1[g].n_sel, ns2 = a.g2[g].n_sel;
        const float* lgp = a.g1[g].logits + row * a.g1[g].ld_logits;
        float* prp = a.g1[g].probs + row * ns;
        float* prp2 = a.g2[g].probs + row * ns2;
        if (gq == (g & 3)) { ns_g1[g >> 2] = ns; lg_g1[g >> 2] = lgp; pr_g1[g >> 2] = prp; ns_g2[g >> 2] = ns2; pr_g2[g >> 2] = prp2; }
    }
    struct JobDesc { const __bf16* W; int64_t ldw; const float* bp; int n_rows, nt; bool on; };
    auto job_desc = [&](int i) __attribute__((always_inline)) {
        const int job = wave + MID_WAVES * i;
        const bool on = job < n_jobs, is_gate = job >= n_tile_jobs;
        const int e = (on && !is_gate) ? job / Cfg::NT2 : 0, nt = (on && !is_gate) ? job % Cfg::NT2 : 0;
        const int t = (on && is_gate) ? job - n_tile_jobs : 0;
        JobDesc d;
        d.W = reinterpret_cast<const __bf16*>(is_gate ? a.g2[t].w : a.e2[e].w);
        d.ldw = is_gate ? a.g2[t].ldw : a.e2[e].ldw;
        d.bp = is_gate ? a.g2[t].bias : a.e2[e].bias;
        d.n_rows = is_gate ? a.g2[t].n_sel : H2;                       // weight rows (= output columns) that exist
        d.nt = nt; d.on = on;
        return d;
    };
    JobDesc jd[MID_JPW];
#pragma unroll
    for (int i = 0; i < MID_JPW; ++i) jd[i] = job_desc(i);
    const MidTabReg tab1 = mid_tables_fetch(a.g1, ng1, tid), tab2 = mid_tables_fetch(a.g2, ng2, tid);
    float lgt[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) lgt[ps] = (live && gq + 4 * ps < ng1 && gj < ns_g1[ps]) ? lg_g1[ps][gj] : 0.f;
    // ---- entry: every load that depends on nothing.  (1) the level-k expert rows of the block -> LDS, 1 KiB (two rows of one
    //      expert) per wave instruction; rows past the batch re-read the last row (their results are never stored)
    {
        constexpr int CH_PER_E = MID_BM * H1 * 4 / 1024;
        const float* ex1 = a.ex1;
        const int64_t ld_ex1 = a.ld_ex1;
        for (int k = wave; k < ne1 * CH_PER_E; k += MID_WAVES) {
            const int e = k / CH_PER_E, rr = (k % CH_PER_E) * (256 / H1) + lane / (H1 / 4);
            int64_t grow = row0 + rr;
            grow = grow < B ? grow : last_row;
            mid_glds16(ex1 + grow * ld_ex1 + (int64_t)e * H1 + (lane % (H1 / 4)) * 4, reinterpret_cast<unsigned char*>(xs) + (size_t)k * 1024);
        }
    }
    // (2) the B fragments (and the bias of the lane's output column) of this wave's tiles: job = wave + 16 * i -> expert tile (e, nt) or
    //     gate t.  The descriptors of a job are scalar loads; those of the first MID_JPW jobs are gathered before any of the fragment
    //     loads is issued, so that they travel together (job by job they were a chain of round trips: 2.4 us of the entry)
    auto load_wd = [&](bf16x8_t (&w)[KS], float& bias, const JobDesc& d) __attribute__((always_inline)) {
        const int wrow = d.nt * 16 + frow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            w[ks] = (d.on && wrow < d.n_rows) ? *reinterpret_cast<const bf16x8_t*>(d.W + (int64_t)wrow * d.ldw + ks * 32 + fk) : (bf16x8_t)(__bf16)0.f;
        bias = (d.on && d.bp && wrow < d.n_rows) ? d.bp[wrow] : 0.f;
    };
    auto load_w = [&](bf16x8_t (&w)[KS], float& bias, int i) __attribute__((always_inline)) { load_wd(w, bias, job_desc(i)); };
    bf16x8_t wq[MID_JPW][KS];
    float bq[MID_JPW];
#pragma unroll
    for (int i = 0; i < MID_JPW; ++i) load_wd(wq[i], bq[i], jd[i]);
    uint32_t seed_base = 0u;                                           // g2_seed32 without its per-call read of the step counter
    const float drop_p = a.drop_p;
    const int relu = a.relu;
    if (drop_p > 0.f) {
        const uint64_t seed = a.seed;
        seed_base = (uint32_t)seed ^ (uint32_t)(seed >> 32) * 0x9E3779B1U;
        if (a.seed_offset_dev) seed_base ^= (uint32_t)sp_val * 0x85EBCA77U;
    }
    mid_tables_store(pos1, sel1, tab1, tid);
    mid_tables_store(pos2, sel2, tab2, tid);
    // ---- A1: level-k gate probabilities of row r: the 16 lanes of a DPP row per gate
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        if (4 * ps >= ng1) continue;                                   // uniform
        const int g = gq + 4 * ps;
        const bool active = g < ng1, valid = active && gj < ns_g1[ps];
        const float pv = mid_softmax_row16(lgt[ps], active, valid, p1 + (r * ng1 + (active ? g : 0)) * 16, gj);
        if (valid && live) pr_g1[ps][gj] = pv;
    }
    MID_STAMP(1);
    __syncthreads();                                                   // (waits for the direct-to-LDS loads as well)
    MID_STAMP(2);
    // ---- A2: pooled level-k vectors; sel ascending = the summation order of the reference.  A gate's list comes out of LDS in one
    //      16-byte read and its terms in groups of four whose reads are in flight together (one at a time, index and operand were a chain
    //      of two LDS round trips per term)
    {
        const int c = lane * 2;
        for (int g = 0; g < ng1; ++g) {
            const int ns = a.g1[g].n_sel;
            __bf16* gp = reinterpret_cast<__bf16*>(a.g1[g].pooled_h);
            const int64_t ldp = a.g1[g].ld_pooled_h;
            const uint4 sv4 = *reinterpret_cast<const uint4*>(&sel1[g][0]);
            const uint32_t sw[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)sv4.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)sv4.y),
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)sv4.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)sv4.w)};
            f32x2_t acc = {0.f, 0.f};
#pragma unroll
            for (int j0 = 0; j0 < CDC_MAX_SEL; j0 += 4) {
                if (j0 >= ns) break;                                   // uniform
                const f32x4_t pw = *reinterpret_cast<const f32x4_t*>(p1 + (r * ng1 + g) * 16 + j0);     // (slots >= n_sel hold 0: their terms add +0)
                f32x2_t xv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = (int)((sw[j0 >> 2] >> (8 * q)) & 0xffu);                               // (0 beyond n_sel: a readable row)
                    xv[q] = *reinterpret_cast<const f32x2_t*>(xs + ((size_t)e * MID_BM + r) * H1 + c);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (j0 + q < ns) acc += pw[q] * xv[q];             // uniform
            }
            const bf16x2_t h = {(__bf16)acc[0], (__bf16)acc[1]};
            *reinterpret_cast<bf16x2_t*>(ph + ((size_t)g * MID_BM + r) * Cfg::PH_LD + c) = h;
            if (live && gp) *reinterpret_cast<bf16x2_t*>(gp + row * ldp + c) = h;
        }
    }
    __syncthreads();                                                   // pooled vectors complete; xs is dead from here (x2 takes its place)
    MID_STAMP(3);
    // ---- B: level-k+1 experts (16 x 16 tiles, K = H1) and gate logits
    {
        const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        const uint32_t thr16 = (uint32_t)(drop_p * 65536.f + 0.5f);
        auto tile = [&](const bf16x8_t (&w)[KS], float bv, int i) __attribute__((always_inline)) {
            const int job = wave + MID_WAVES * i;
            if (job >= n_jobs) return;
            const bool is_gate = job >= n_tile_jobs;
            const int e = is_gate ? 0 : job / Cfg::NT2, nt = is_gate ? 0 : job % Cfg::NT2;
            const int t = is_gate ? job - n_tile_jobs : 0;
            const int src = is_gate ? a.g2[t].src : a.e2[e].src;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(ph + ((size_t)src * MID_BM + frow) * Cfg::PH_LD + ks * 32 + fk);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, w[ks], acc, 0, 0, 0);
            }
            const int col = nt * 16 + (lane & 15);
            if (is_gate) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rl = (lane >> 4) * 4 + r4;
                    p2[(rl * ng2 + t) * 16 + col] = acc[r4] + bv;                 // logits; columns >= n_sel are ignored by the softmax
                }
            } else {
                const uint32_t seed32 = g2_hash32(seed_base + (uint32_t)a.e2[e].stream_id * 0xC2B2AE3DU);     // = g2_seed32(seed, step, stream_id)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rl = (lane >> 4) * 4 + r4;
                    float x = acc[r4] + bv;
                    if (relu) x = fmaxf(x, 0.f);
                    if (drop_p > 0.f) {
                        const uint32_t h = g2_drop_bits(seed32, (int)(row0 + rl), col >> 1);
                        const uint32_t bits = (col & 1) ? (h >> 16) : (h & 0xFFFFu);
                        x = bits < thr16 ? 0.f : x * keep_scale;
                    }
                    x2[((size_t)e * MID_BM + rl) * Cfg::E2_LD + col] = x;
                }
            }
        };
#pragma unroll
        for (int i = 0; i < MID_JPW; ++i) tile(wq[i], bq[i], i);
        for (int i = MID_JPW; wave + MID_WAVES * i < n_jobs; ++i) {      // more tiles than the entry fetch covers (n_exp2 > 11)
            load_w(wq[0], bq[0], i);
            tile(wq[0], bq[0], i);
        }
    }
    __syncthreads();
    MID_STAMP(4);
    // ---- D1: level-k+1 gate probabilities of row r, in place: the 16 lanes of a DPP row per gate
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        if (4 * ps >= ng2) continue;                                   // uniform
        const int t = gq + 4 * ps;
        const bool active = t < ng2, valid = active && gj < ns_g2[ps];
        float* slot = p2 + (r * ng2 + (active ? t : 0)) * 16;
        const float pv = mid_softmax_row16(valid ? slot[gj] : 0.f, active, valid, slot, gj);
        if (valid && live) pr_g2[ps][gj] = pv;
    }
    __syncthreads();
    MID_STAMP(5);
    // ---- D2: second pooling out of LDS (lane = column); the expert tiles go to memory for the backward
    {
        const int c2 = lane;
        if (live) {
            float* ex2 = a.ex2 + row * a.ld_ex2 + c2;
            for (int e = 0; e < ne2; ++e) ex2[(int64_t)e * H2] = x2[((size_t)e * MID_BM + r) * Cfg::E2_LD + c2];
        }
        for (int g = 0; g < ng2; ++g) {
            const int ns = a.g2[g].n_sel;
            float* out = a.g2[g].out;
            __bf16* outh = reinterpret_cast<__bf16*>(a.g2[g].out_h);
            const int64_t ldo = a.g2[g].ld_out, ldoh = a.g2[g].ld_out_h;
            const uint4 sv4 = *reinterpret_cast<const uint4*>(&sel2[g][0]);
            const uint32_t sw[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)sv4.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)sv4.y),
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)sv4.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)sv4.w)};
            float acc = 0.f;
#pragma unroll
            for (int j0 = 0; j0 < CDC_MAX_SEL; j0 += 4) {
                if (j0 >= ns) break;                                   // uniform
                const f32x4_t pw = *reinterpret_cast<const f32x4_t*>(p2 + (r * ng2 + g) * 16 + j0);
                float xv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[q] = x2[((size_t)((sw[j0 >> 2] >> (8 * q)) & 0xffu) * MID_BM + r) * Cfg::E2_LD + c2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (j0 + q < ns) acc += pw[q] * xv[q];             // uniform
            }
            if (live) {
                if (out) out[row * ldo + c2] = acc;
                if (outh) outh[row * ldoh + c2] = (__bf16)acc;
            }
        }
    }
    MID_STAMP(6);
}

template <int H1, int H2>
static size_t mid_fwd_lds(int ng1, int ng2, int ne1, int ne2) {
    typedef MidCfg<H1, H2> Cfg;
    const size_t xs = (size_t)ne1 * MID_BM * H1 * 4, x2 = (size_t)ne2 * MID_BM * Cfg::E2_LD * 4;
    return (size_t)MID_BM * ng1 * 16 * 4 + (size_t)MID_BM * ng2 * 16 * 4 + (size_t)ng1 * MID_BM * Cfg::PH_LD * 2 + (xs > x2 ? xs : x2);
}

static int mid_check_sel(const int32_t* sel, int n_sel, int n_expert) {
    if (n_sel <= 0 || n_sel > CDC_MAX_SEL) return 0;
    for (int j = 0; j < n_sel; ++j) {
        if (sel[j] < 0 || sel[j] >= n_expert) return 0;
        if (j && sel[j] <= sel[j - 1]) return 0;
    }
    return 1;
}

extern "C" int cdc_cgc_mid_fwd(const cdc_cgc_mid_fwd_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->B >= 0 && a->ex1 && a->ex2, CDC_E_BADARG, "cgc_mid_fwd: null pointer");
    CDC_CHECK_ARG(a->H1 == 128 && a->H2 == 64, CDC_E_BADARG, "cgc_mid_fwd: built for expert widths (128, 64), got (%d, %d)", a->H1, a->H2);
    CDC_CHECK_ARG(a->n_exp1 > 0 && a->n_exp1 <= MID_E && a->n_exp2 > 0 && a->n_exp2 <= MID_E && a->n_gate1 > 0 && a->n_gate1 <= MID_G &&
                      a->n_gate2 > 0 && a->n_gate2 <= MID_G, CDC_E_BADARG, "cgc_mid_fwd: bad counts");
    CDC_CHECK_ARG(a->drop_p >= 0.f && a->drop_p < 1.f, CDC_E_BADARG, "cgc_mid_fwd: dropout p out of range");
    CDC_CHECK_ARG((((uintptr_t)a->ex1 | (uintptr_t)a->ex2) & 15) == 0 && a->ld_ex1 % 4 == 0 && a->ld_ex2 % 4 == 0 &&
                      a->ld_ex1 >= (int64_t)a->n_exp1 * a->H1 && a->ld_ex2 >= (int64_t)a->n_exp2 * a->H2, CDC_E_ALIGN,
                  "cgc_mid_fwd: expert buffers must be 16-byte aligned and wide enough");
    for (int g = 0; g < a->n_gate1; ++g) {
        const cdc_mid_gate1& G = a->g1[g];
        CDC_CHECK_ARG(G.logits && G.probs && mid_check_sel(G.sel, G.n_sel, a->n_exp1), CDC_E_BADARG, "cgc_mid_fwd: level-k gate %d malformed (sel ascending)", g);
        CDC_CHECK_ARG(!G.pooled_h || ((((uintptr_t)G.pooled_h) & 15) == 0 && G.ld_pooled_h % 8 == 0 && G.ld_pooled_h >= a->H1), CDC_E_ALIGN,
                      "cgc_mid_fwd: level-k gate %d: bf16 output must be 16-byte aligned", g);
    }
    for (int e = 0; e < a->n_exp2; ++e) {
        const cdc_mid_expert2& E = a->e2[e];
        CDC_CHECK_ARG(E.w && E.src >= 0 && E.src < a->n_gate1 && E.ldw >= a->H1, CDC_E_BADARG, "cgc_mid_fwd: expert %d malformed", e);
        CDC_CHECK_ARG((((uintptr_t)E.w) & 15) == 0 && E.ldw % 8 == 0, CDC_E_ALIGN, "cgc_mid_fwd: expert %d: weight must be 16-byte aligned", e);
    }
    for (int g = 0; g < a->n_gate2; ++g) {
        const cdc_mid_gate2& G = a->g2[g];
        CDC_CHECK_ARG(G.w && G.probs && (G.out || G.out_h) && G.src >= 0 && G.src < a->n_gate1 && G.ldw >= a->H1 &&
                          mid_check_sel(G.sel, G.n_sel, a->n_exp2), CDC_E_BADARG, "cgc_mid_fwd: level-k+1 gate %d malformed (sel ascending)", g);
        CDC_CHECK_ARG((((uintptr_t)G.w) & 15) == 0 && G.ldw % 8 == 0 && (!G.out || ((((uintptr_t)G.out) & 15) == 0 && G.ld_out % 4 == 0)) &&
                          (!G.out_h || ((((uintptr_t)G.out_h) & 7) == 0 && G.ld_out_h % 4 == 0)), CDC_E_ALIGN,
                      "cgc_mid_fwd: level-k+1 gate %d: misaligned operand", g);
    }
    if (a->B == 0) return 0;
    const size_t lds = mid_fwd_lds<128, 64>(a->n_gate1, a->n_gate2, a->n_exp1, a->n_exp2);
    CDC_CHECK_ARG(lds <= 150 * 1024, CDC_E_TOOBIG, "cgc_mid_fwd: %zu bytes of LDS needed", lds);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_cgc_mid_fwd<128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_cgc_mid_fwd<128, 64>), dim3((unsigned)cdc_ceil_div(a->B, MID_BM)), dim3(MID_THREADS), lds, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("cgc_mid_fwd");
    return 0;
}

// softmax backward of ONE gate over the 16 lanes of a DPP row (lane j takes term j): dot = sum_k p_k dp_k in index order (formed by
// every lane of the row out of the two LDS slots), d_logit_j = p_j (dp_j - dot).  Slots >= n_sel of dp are never written: masked.
__device__ __forceinline__ float mid_dlogit_row16(const float* pp, const float* dd, int ns, int j) {
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < CDC_MAX_SEL; q += 4) {
        if (q >= ns) break;                                            // (uniform per row; lanes of inactive rows pass ns = 0)
        const f32x4_t p4 = *reinterpret_cast<const f32x4_t*>(pp + q), d4 = *reinterpret_cast<const f32x4_t*>(dd + q);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (q + k < ns) dot += p4[k] * d4[k];
    }
    return j < ns ? pp[j] * (dd[j] - dot) : 0.f;
}

// =====================================================================================================================
// backward
// =====================================================================================================================
struct MidStep { const __bf16* wt; int32_t ldwt; int32_t a_off; int32_t a_ld; int32_t pad_; };

template <int H1, int H2>
__global__ void __launch_bounds__(MID_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) k_cgc_mid_bwd(const cdc_cgc_mid_bwd_args a_by_value) {
    CDC_PRIO_MAIN();
    (void)a_by_value;
    const MID_KARG cdc_cgc_mid_bwd_args& a = *(const MID_KARG cdc_cgc_mid_bwd_args*)__builtin_amdgcn_kernarg_segment_ptr();
    typedef MidCfg<H1, H2> Cfg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ signed char pos1[MID_G][MID_E], pos2[MID_G][MID_E];
    __shared__ unsigned char sel1[MID_G][CDC_MAX_SEL], sel2[MID_G][CDC_MAX_SEL];
    __shared__ MidStep steps[MID_G][MID_MAXSTEP];
    __shared__ int n_steps[MID_G];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * MID_BM;
    const int64_t B = a.B;
    const int ng1 = a.n_gate1, ng2 = a.n_gate2, ne1 = a.n_exp1, ne2 = a.n_exp2;
    // LDS: p1, dp1 [ng1][16][16] f32 | p2, dp2 [ng2][16][16] f32 | V = { dO2 [ng2][16][H2] f32 (phase 1), dP1 [ng1][16][DP_LD] f32
    //      (phases 2, 3) } | dz2 [ne2][16][DZ_LD] bf16 | dl2 [ng2][16][DL_LD] bf16
    float* p1 = reinterpret_cast<float*>(smem);
    float* dp1 = p1 + ng1 * MID_BM * 16;
    float* p2 = dp1 + ng1 * MID_BM * 16;
    float* dp2 = p2 + ng2 * MID_BM * 16;
    float* dP1 = dp2 + ng2 * MID_BM * 16;
    float* dO2 = dP1;
    const size_t v_floats = (size_t)ng1 * MID_BM * Cfg::DP_LD > (size_t)ng2 * MID_BM * H2 ? (size_t)ng1 * MID_BM * Cfg::DP_LD : (size_t)ng2 * MID_BM * H2;
    __bf16* dz2 = reinterpret_cast<__bf16*>(dP1 + v_floats);
    __bf16* dl2 = dz2 + (size_t)ne2 * MID_BM * Cfg::DZ_LD;
    const int dl2_off = ne2 * MID_BM * Cfg::DZ_LD;
    const int frow = lane & 15, fk = (lane >> 4) * 8;
    const int r = wave;                                                // row-wise phases: wave = batch row, lane = column
    const int64_t row = row0 + r;
    const bool live = row < B;
    const int64_t last_row = B - 1;

    MID_STAMP(16);
    const MidTabReg tab1 = mid_tables_fetch(a.g1, ng1, tid), tab2 = mid_tables_fetch(a.g2, ng2, tid);    // (stored into LDS behind the entry's loads)
    // ---- entry: direct-to-LDS loads of the gradients of the level-k+1 pooled outputs (1 KiB = four rows per wave instruction)
    //      and of both levels' probabilities (256 B = four rows of one gate per wave instruction; slots j >= n_sel hold a
    //      duplicate and are never read); rows past the batch re-read the last row (nothing of theirs is stored)
    {
        constexpr int CH = MID_BM * H2 * 4 / 1024;
        for (int k = wave; k < ng2 * CH; k += MID_WAVES) {
            const int t = k / CH, rr = (k % CH) * (256 / H2) + lane / (H2 / 4);
            int64_t grow = row0 + rr;
            grow = grow < B ? grow : last_row;
            mid_glds16(a.g2[t].d_out + grow * a.g2[t].ld_dout + (lane % (H2 / 4)) * 4, reinterpret_cast<unsigned char*>(dO2) + (size_t)k * 1024);
        }
        for (int k = wave; k < (ng1 + ng2) * 4; k += MID_WAVES) {
            const bool lvl2 = k >= ng1 * 4;
            const int g = (lvl2 ? k - ng1 * 4 : k) / 4, rr = (k % 4) * 4 + lane / 16;
            const int ns = lvl2 ? a.g2[g].n_sel : a.g1[g].n_sel;
            const float* pr = lvl2 ? a.g2[g].probs : a.g1[g].probs;
            int64_t grow = row0 + rr;
            grow = grow < B ? grow : last_row;
            const int j = (lane & 15) < ns ? (lane & 15) : ns - 1;
            mid_glds4(pr + grow * ns + j, reinterpret_cast<unsigned char*>(lvl2 ? p2 : p1) + ((size_t)g * MID_BM + (k % 4) * 4) * 64);
        }
    }
    // the pool backwards (phases 1 and 3): a batch row belongs to the 16 lanes of a DPP row — wave w takes rows 4 (w & 3) .. + 3 and
    // the experts e = (w >> 2), (w >> 2) + 4, ...; a lane holds four columns of level k+1 (two times four of level k) of its row, so
    // that a gate-gradient dot product is four (eight) multiply-adds and four in-row DPP adds for FOUR rows at once.  (Round 3 gave
    // a row to a whole wave, a column to a lane: a 64-lane tree with two LDS-crossbar steps per (expert, gate) pair and row — with 16
    // waves on the CU the two phases were bound by instruction ISSUE, 7.4 us each; profiles/round4/README.md section 11.)
    // MID_XW experts per wave in registers at once (all of them when a level has at most 4 MID_XW experts: then no load of theirs is
    // issued after this point)
    const int rq = 4 * (wave & 3) + (lane >> 4), l16 = lane & 15, eg = wave >> 2;
    const int64_t rowq = row0 + rq;
    const bool liveq = rowq < B;
    f32x4_t x2q[MID_XW], x1q[MID_XW][2];
    const float* ex2p = a.ex2 + rowq * a.ld_ex2 + 4 * l16;
    const float* ex1p = a.ex1 + rowq * a.ld_ex1 + 4 * l16;
    auto load_x2 = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < MID_XW; ++u) {
            const int e = eg + 4 * (MID_XW * c + u);
            x2q[u] = (liveq && e < ne2) ? *reinterpret_cast<const f32x4_t*>(ex2p + (int64_t)e * H2) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto load_x1 = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < MID_XW; ++u) {
            const int e = eg + 4 * (MID_XW * c + u);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                x1q[u][h] = (liveq && e < ne1) ? *reinterpret_cast<const f32x4_t*>(ex1p + (int64_t)e * H1 + 64 * h) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };
    load_x2(0);
    load_x1(0);
    // the K steps of every source's grad-input tiles: experts reading it (ascending), then gates — the segment order of the
    // unfused grad-input launch
    if (wave == 0) {
        // lane i is candidate i of every source's list — i < 2 MID_E: K step i & 1 of expert i >> 1, then the gates — and fetches its
        // own descriptor once (one round trip for the whole table; a loop over the experts in one thread per source was a chain of
        // dependent scalar loads at the head of the kernel); its place in source s's list is the number of matching candidates below it
        const bool is_e = lane < 2 * MID_E, is_g = !is_e && lane < 2 * MID_E + MID_G;
        const int ce = lane >> 1, ct = lane - 2 * MID_E;
        int c_src = -1, c_ld = 0, c_off = 0, c_ald = 0;
        const __bf16* c_wt = nullptr;
        if (is_e && ce < ne2) {
            c_src = a.e2[ce].src; c_ld = (int32_t)a.e2[ce].ldwt;
            c_wt = reinterpret_cast<const __bf16*>(a.e2[ce].wt) + (lane & 1) * 32;
            c_off = ce * MID_BM * Cfg::DZ_LD + (lane & 1) * 32; c_ald = Cfg::DZ_LD;
        } else if (is_g && ct < ng2) {
            c_src = a.g2[ct].src; c_ld = (int32_t)a.g2[ct].ldwt;
            c_wt = reinterpret_cast<const __bf16*>(a.g2[ct].wt);
            c_off = dl2_off + ct * MID_BM * Cfg::DL_LD; c_ald = Cfg::DL_LD;
        }
        for (int sidx = 0; sidx < ng1; ++sidx) {
            const unsigned long long m = __ballot(c_src == sidx);
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            if (c_src == sidx && rank < MID_MAXSTEP) steps[sidx][rank] = MidStep{c_wt, c_ld, c_off, c_ald, 0};
            if (lane == 0) n_steps[sidx] = min((int)__popcll(m), MID_MAXSTEP);
        }
    }
    mid_tables_store(pos1, sel1, tab1, tid);
    mid_tables_store(pos2, sel2, tab2, tid);
    const int mask1 = __builtin_amdgcn_readfirstlane(a.mask1), mask2 = __builtin_amdgcn_readfirstlane(a.mask2);
    const float scale1 = a.scale1, scale2 = a.scale2;
    MID_STAMP(17);
    __syncthreads();                                                   // tables, dO2 and the probabilities are in LDS
    MID_STAMP(18);
    // B fragments of this wave's first grad-input tiles: job = wave + 16 * i -> (source s, column tile nt)
    const int n_jobs = ng1 * Cfg::NT1;
    auto load_w = [&](bf16x8_t (&w)[MID_MAXSTEP], int i) __attribute__((always_inline)) {
        const int job = wave + MID_WAVES * i;
        const bool on = job < n_jobs;
        const int s = on ? job / Cfg::NT1 : 0, nt = on ? job % Cfg::NT1 : 0;
        const int ns = on ? n_steps[s] : 0;
        const int wrow = nt * 16 + frow;
#pragma unroll
        for (int st = 0; st < MID_MAXSTEP; ++st)
            w[st] = st < ns ? *reinterpret_cast<const bf16x8_t*>(steps[s][st].wt + (int64_t)wrow * steps[s][st].ldwt + fk) : (bf16x8_t)(__bf16)0.f;
    };
    // ---- 1: pool backward of level k+1.  dp_tj = <dOut_t, expert_sel(t,j)> over the row; dExpert_e = sum over the gates t that
    //         select e, in gate order, of p_tj * dOut_t, then the activation mask
    {
        __bf16* dzg = reinterpret_cast<__bf16*>(a.dz2_h) + rowq * a.ld_dz2_h + 4 * l16;
        for (int c = 0; eg + 4 * MID_XW * c < ne2; ++c) {
            if (c > 0) load_x2(c);
#pragma unroll
            for (int u = 0; u < MID_XW; ++u) {
                const int e = eg + 4 * (MID_XW * c + u);
                if (e >= ne2) continue;                                         // uniform
                const f32x4_t x = x2q[u];
                f32x4_t d = {0.f, 0.f, 0.f, 0.f};
                for (int t = 0; t < ng2; ++t) {
                    const int j = __builtin_amdgcn_readfirstlane((int)pos2[t][e]);
                    if (j < 0) continue;                                        // uniform
                    const f32x4_t dO = *reinterpret_cast<const f32x4_t*>(dO2 + ((size_t)t * MID_BM + rq) * H2 + 4 * l16);
                    const float part = mid_row16_sum((dO[0] * x[0] + dO[1] * x[1]) + (dO[2] * x[2] + dO[3] * x[3]));
                    if (l16 == 0) dp2[(t * MID_BM + rq) * 16 + j] = part;
                    d += p2[(t * MID_BM + rq) * 16 + j] * dO;
                }
                if (mask2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = x[k] > 0.f ? d[k] * scale2 : 0.f;
                }
                const bf16x4_t h = {(__bf16)d[0], (__bf16)d[1], (__bf16)d[2], (__bf16)d[3]};
                *reinterpret_cast<bf16x4_t*>(dz2 + ((size_t)e * MID_BM + rq) * Cfg::DZ_LD + 4 * l16) = h;
                if (liveq) *reinterpret_cast<bf16x4_t*>(dzg + (int64_t)e * H2) = h;
            }
        }
    }
    // the B fragments of this wave's first grad-input tiles: fetched here, behind the first pool backward (whose side-by-side wave
    // sums need the registers), in flight under the gate-logit phase
    bf16x8_t wq[MID_JPW2][MID_MAXSTEP];
#pragma unroll
    for (int i = 0; i < MID_JPW2; ++i) load_w(wq[i], i);
    MID_STAMP(19);
    __syncthreads();
    MID_STAMP(20);
    // ---- 1b: d_logit_tj = p_tj * (dp_tj - sum_k p_tk dp_tk) of row r: the 16 lanes of a DPP row per gate (lane = term)
    {
        const int gq = lane >> 4, gj = lane & 15;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            if (4 * ps >= ng2) continue;                               // uniform
            const int t = gq + 4 * ps;
            const bool active = t < ng2;
            const int tc = active ? t : 0;
            const int ns = active ? a.g2[tc].n_sel : 0;
            const float dl = mid_dlogit_row16(p2 + (tc * MID_BM + r) * 16, dp2 + (tc * MID_BM + r) * 16, ns, gj);
            if (active) {
                __bf16* lrow = dl2 + ((size_t)t * MID_BM + r) * Cfg::DL_LD;
                lrow[gj] = (__bf16)dl;                                 // (0 beyond n_sel)
                lrow[16 + gj] = (__bf16)0.f;
                if (live && gj < ns) {
                    a.g2[tc].d_logits[row * a.g2[tc].ld_dlogits + gj] = dl;
                    if (a.g2[tc].d_logits_h) reinterpret_cast<__bf16*>(a.g2[tc].d_logits_h)[row * a.g2[tc].ld_dlogits_h + gj] = (__bf16)dl;
                }
            }
        }
    }
    __syncthreads();                                                   // dz2, dl2 complete; dO2 is dead (dP1 takes its place)
    MID_STAMP(21);
    // ---- 2: d(pooled level-k output s) [16, H1] = sum over the experts e reading s of dZ_e . W_e  +  the gates reading s
    {
        auto tile = [&](const bf16x8_t (&w)[MID_MAXSTEP], int i) __attribute__((always_inline)) {
            const int job = wave + MID_WAVES * i;
            if (job >= n_jobs) return;
            const int s = job / Cfg::NT1, nt = job % Cfg::NT1;
            const int ns = n_steps[s];
            bf16x8_t af[MID_MAXSTEP];
#pragma unroll
            for (int st = 0; st < MID_MAXSTEP; ++st) {                 // past the source's steps: w[st] is zero, any finite A fragment will do
                const int sc = st < ns ? st : ns - 1;
                af[st] = *reinterpret_cast<const bf16x8_t*>(dz2 + steps[s][sc].a_off + frow * steps[s][sc].a_ld + fk);
            }
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < MID_MAXSTEP; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[st], w[st], acc, 0, 0, 0);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                dP1[((size_t)s * MID_BM + (lane >> 4) * 4 + r4) * Cfg::DP_LD + nt * 16 + (lane & 15)] = acc[r4];
        };
#pragma unroll
        for (int i = 0; i < MID_JPW2; ++i) tile(wq[i], i);
        for (int i = MID_JPW2; wave + MID_WAVES * i < n_jobs; ++i) {
            load_w(wq[0], i);
            tile(wq[0], i);
        }
    }
    __syncthreads();
    MID_STAMP(22);
    // ---- 3: pool backward of level k (a lane: columns 4 l .. 4 l + 3 and 64 + 4 l .. of its row)
    {
        __bf16* dzg = reinterpret_cast<__bf16*>(a.dz1_h) + rowq * a.ld_dz1_h + 4 * l16;
        for (int c = 0; eg + 4 * MID_XW * c < ne1; ++c) {
            if (c > 0) load_x1(c);
#pragma unroll
            for (int u = 0; u < MID_XW; ++u) {
                const int e = eg + 4 * (MID_XW * c + u);
                if (e >= ne1) continue;                                         // uniform
                f32x4_t d[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                for (int g = 0; g < ng1; ++g) {
                    const int j = __builtin_amdgcn_readfirstlane((int)pos1[g][e]);
                    if (j < 0) continue;                                        // uniform
                    const float* src = dP1 + ((size_t)g * MID_BM + rq) * Cfg::DP_LD + 4 * l16;
                    const f32x4_t dA = *reinterpret_cast<const f32x4_t*>(src), dB = *reinterpret_cast<const f32x4_t*>(src + 64);
                    const f32x4_t xa = x1q[u][0], xb = x1q[u][1];
                    const float part = mid_row16_sum(((dA[0] * xa[0] + dA[1] * xa[1]) + (dA[2] * xa[2] + dA[3] * xa[3])) +
                                                     ((dB[0] * xb[0] + dB[1] * xb[1]) + (dB[2] * xb[2] + dB[3] * xb[3])));
                    if (l16 == 0) dp1[(g * MID_BM + rq) * 16 + j] = part;
                    const float pw = p1[(g * MID_BM + rq) * 16 + j];
                    d[0] += pw * dA;
                    d[1] += pw * dB;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4_t dv = d[h];
                    if (mask1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) dv[k] = x1q[u][h][k] > 0.f ? dv[k] * scale1 : 0.f;
                    }
                    if (liveq) *reinterpret_cast<bf16x4_t*>(dzg + (int64_t)e * H1 + 64 * h) = bf16x4_t{(__bf16)dv[0], (__bf16)dv[1], (__bf16)dv[2], (__bf16)dv[3]};
                }
            }
        }
    }
    MID_STAMP(23);
    __syncthreads();
    MID_STAMP(24);
    {
        const int gq = lane >> 4, gj = lane & 15;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            if (4 * ps >= ng1) continue;                               // uniform
            const int g = gq + 4 * ps;
            const bool active = g < ng1;
            const int gc = active ? g : 0;
            const int ns = active ? a.g1[gc].n_sel : 0;
            const float dl = mid_dlogit_row16(p1 + (gc * MID_BM + r) * 16, dp1 + (gc * MID_BM + r) * 16, ns, gj);
            if (active && live && gj < ns) {
                a.g1[gc].d_logits[row * a.g1[gc].ld_dlogits + gj] = dl;
                if (a.g1[gc].d_logits_h) reinterpret_cast<__bf16*>(a.g1[gc].d_logits_h)[row * a.g1[gc].ld_dlogits_h + gj] = (__bf16)dl;
            }
        }
    }
    MID_STAMP(25);
}

template <int H1, int H2>
static size_t mid_bwd_lds(int ng1, int ng2, int ne2) {
    typedef MidCfg<H1, H2> Cfg;
    const size_t dp = (size_t)ng1 * MID_BM * Cfg::DP_LD * 4, dO = (size_t)ng2 * MID_BM * H2 * 4;
    return (size_t)2 * MID_BM * ng1 * 16 * 4 + (size_t)2 * MID_BM * ng2 * 16 * 4 + (dp > dO ? dp : dO) + (size_t)ne2 * MID_BM * Cfg::DZ_LD * 2 +
           (size_t)ng2 * MID_BM * Cfg::DL_LD * 2;
}

extern "C" int cdc_cgc_mid_bwd(const cdc_cgc_mid_bwd_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->B >= 0 && a->ex1 && a->ex2 && a->dz1_h && a->dz2_h, CDC_E_BADARG, "cgc_mid_bwd: null pointer");
    CDC_CHECK_ARG(a->H1 == 128 && a->H2 == 64, CDC_E_BADARG, "cgc_mid_bwd: built for expert widths (128, 64), got (%d, %d)", a->H1, a->H2);
    CDC_CHECK_ARG(a->n_exp1 > 0 && a->n_exp1 <= MID_E && a->n_exp2 > 0 && a->n_exp2 <= MID_E && a->n_gate1 > 0 && a->n_gate1 <= MID_G &&
                      a->n_gate2 > 0 && a->n_gate2 <= MID_G, CDC_E_BADARG, "cgc_mid_bwd: bad counts");
    CDC_CHECK_ARG((((uintptr_t)a->ex1 | (uintptr_t)a->ex2 | (uintptr_t)a->dz1_h) & 15) == 0 && (((uintptr_t)a->dz2_h) & 7) == 0 &&
                      a->ld_ex1 % 4 == 0 && a->ld_ex2 % 4 == 0 && a->ld_dz1_h % 8 == 0 && a->ld_dz2_h % 4 == 0, CDC_E_ALIGN,
                  "cgc_mid_bwd: misaligned expert buffers");
    for (int g = 0; g < a->n_gate1; ++g)
        CDC_CHECK_ARG(a->g1[g].probs && a->g1[g].d_logits && mid_check_sel(a->g1[g].sel, a->g1[g].n_sel, a->n_exp1), CDC_E_BADARG,
                      "cgc_mid_bwd: level-k gate %d malformed", g);
    for (int e = 0; e < a->n_exp2; ++e)
        CDC_CHECK_ARG(a->e2[e].wt && a->e2[e].src >= 0 && a->e2[e].src < a->n_gate1 && a->e2[e].ldwt >= a->H2 && a->e2[e].ldwt % 8 == 0 &&
                          (((uintptr_t)a->e2[e].wt) & 15) == 0, CDC_E_BADARG, "cgc_mid_bwd: expert %d malformed", e);
    for (int g = 0; g < a->n_gate2; ++g) {
        const cdc_mid_bgate2& G = a->g2[g];
        CDC_CHECK_ARG(G.d_out && G.probs && G.d_logits && G.wt && G.src >= 0 && G.src < a->n_gate1 && G.ldwt >= 32 && G.ldwt % 8 == 0 &&
                          (((uintptr_t)G.wt) & 15) == 0 && (((uintptr_t)G.d_out) & 15) == 0 && G.ld_dout % 4 == 0 &&
                          mid_check_sel(G.sel, G.n_sel, a->n_exp2), CDC_E_BADARG, "cgc_mid_bwd: level-k+1 gate %d malformed", g);
    }
    for (int s = 0; s < a->n_gate1; ++s) {
        int n = 0;
        for (int e = 0; e < a->n_exp2; ++e) n += a->e2[e].src == s ? a->H2 / 32 : 0;
        for (int g = 0; g < a->n_gate2; ++g) n += a->g2[g].src == s ? 1 : 0;
        CDC_CHECK_ARG(n <= MID_MAXSTEP, CDC_E_TOOBIG, "cgc_mid_bwd: level-k output %d feeds %d K-steps of 32 (limit %d: two experts and a gate, or three experts)",
                      s, n, MID_MAXSTEP);
    }
    if (a->B == 0) return 0;
    const size_t lds = mid_bwd_lds<128, 64>(a->n_gate1, a->n_gate2, a->n_exp2);
    CDC_CHECK_ARG(lds <= 150 * 1024, CDC_E_TOOBIG, "cgc_mid_bwd: %zu bytes of LDS needed", lds);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_cgc_mid_bwd<128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_cgc_mid_bwd<128, 64>), dim3((unsigned)cdc_ceil_div(a->B, MID_BM)), dim3(MID_THREADS), lds, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("cgc_mid_bwd");
    return 0;
}

// Does the fused boundary fit?  1 = both launches' LDS needs are within the 150 KB they may ask for, 0 = not (the caller keeps the
// three launches per direction: plan.CGCMid.match), < 0 = counts outside the supported range.
extern "C" int cdc_cgc_mid_fits(int32_t n_exp1, int32_t n_gate1, int32_t n_exp2, int32_t n_gate2) {
    if (n_exp1 <= 0 || n_exp1 > MID_E || n_exp2 <= 0 || n_exp2 > MID_E || n_gate1 <= 0 || n_gate1 > MID_G || n_gate2 <= 0 || n_gate2 > MID_G)
        return CDC_E_BADARG;
    return mid_fwd_lds<128, 64>(n_gate1, n_gate2, n_exp1, n_exp2) <= 150 * 1024 && mid_bwd_lds<128, 64>(n_gate1, n_gate2, n_exp2) <= 150 * 1024 ? 1 : 0;
}
