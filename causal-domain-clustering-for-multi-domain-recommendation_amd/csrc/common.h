// common.h — shared host/device helpers for the gfx950 kernels behind include/cdcmdr.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "cdcmdr.h"

#define CDC_WAVE 64

void cdc_set_error(const char* fmt, ...);

#define CDC_CHECK_ARG(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            cdc_set_error(__VA_ARGS__);           \
            return (code);                        \
        }                                         \
    } while (0)

// call after every launch: reports launch-configuration errors without synchronising
#define CDC_LAUNCH_CHECK(name)                                                     \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            cdc_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                       \
        }                                                                          \
    } while (0)

// gemm2.hip: grad-weight from bf16 shadows (called by cdc_glinear_bwd_w)
int g2_launch_bwd_w(const cdc_lin_bwdw_args* a, int64_t slab_stride, hipStream_t st);
bool g2_bwd_w_uses_small_tiles(const cdc_lin_bwdw_args* a);
int g2_launch_bwd_w_dual(const cdc_lin_bwdw_args* wide, const cdc_lin_bwdw_args* narrow, const cdc_lin_bwdw_args* tabs_dev, int64_t slab_wide,
                         int64_t slab_narrow, hipStream_t st);

static inline int64_t cdc_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
// Wave issue priority (s_setprio, 0..3; waves start at 0).  The launches of the forward/backward chain raise theirs, so that a
// VALU-only background launch sharing the CUs (the lazy table's replay slice, which stays at 0) only takes the issue cycles
// the chain's waves leave idle while they wait on L2 / LDS / MFMA results.
#define CDC_PRIO_MAIN() __builtin_amdgcn_s_setprio(2)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based dropout stream: a 64-bit mix of (seed, element index) -> uniform in [0,1).
// The backward pass never regenerates it: a dropped or relu-clamped unit has output exactly 0,
// so d_in = (out > 0) ? d_out * 1/(1-p) : 0 recovers the mask from the saved output.
__device__ __forceinline__ float cdc_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// The cheap dropout stream (32-bit "lowbias" mix, 16 bits per element: keep iff bits >= round(p * 65536)); ~25 issue cycles per
// hash against ~150 for the 64-bit mix of cdc_uniform.  Never regenerated in backward: the mask is read off the saved output.
__device__ __forceinline__ uint32_t g2_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t g2_seed32(uint64_t seed, const int32_t* step_dev, int stream_id) {
    uint32_t s = (uint32_t)seed ^ (uint32_t)(seed >> 32) * 0x9E3779B1U;
    if (step_dev) s ^= (uint32_t)(*step_dev) * 0x85EBCA77U;
    return g2_hash32(s + (uint32_t)stream_id * 0xC2B2AE3DU);
}
__device__ __forceinline__ uint32_t g2_drop_bits(uint32_t seed32, int row, int colpair) {      // two 16-bit uniforms: columns 2*colpair, 2*colpair+1
    return g2_hash32(seed32 + (uint32_t)row * 0x9E3779B1U + (uint32_t)colpair * 0x85EBCA77U);
}


// One element of torch's CPU Adam (torch/optim/adam.py _single_tensor_adam, as run.py:720-721
// configures it) with the L2 term of model/layer.py:96-112 folded into the gradient:
//   g  = g_in + l2_twice*w          (autograd: grad of sum(l2*w^2))
//   g  = g + wd*w                   (grad.add(param, alpha=wd))
//   m  = lerp(m, g, 1-beta1)        (exp_avg.lerp_)
//   v  = fma((1-beta2)*g, g, v*beta2) (mul_ then addcmul_)
//   w -= step_size * m / (sqrt(v)/bc2_sqrt + eps)
struct AdamConsts {
    float lerp_w, beta2, omb2, eps, wd, l2_twice;
};
__device__ __forceinline__ void adam_elem(float& w, float& m, float& v, float g_in, const AdamConsts& c,
                                          float step_size, float bc2_sqrt) {
    // every rounding is pinned (no compiler contraction): the dense, touched-row and lazy-replay
    // kernels must produce identical bits for identical inputs.
    float g = __fadd_rn(g_in, __fmul_rn(c.l2_twice, w));
    g = fmaf(w, c.wd, g);                                   // ATen add(alpha): vec fmadd
    m = fmaf(c.lerp_w, __fsub_rn(g, m), m);                 // ATen lerp, |weight| < 0.5: fmadd
    v = __fmul_rn(v, c.beta2);
    v = fmaf(__fmul_rn(c.omb2, g), g, v);                   // addcmul: fmadd(value*t1, t2, self) — the rounding pattern that
                                                            // reproduces torch CPU Adam bit for bit (oracle/adam_elem_ref.c)
    float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), c.eps);
    w = __fadd_rn(w, __fdiv_rn(__fmul_rn(-step_size, m), denom));   // addcdiv: self + (value*t1)/t2
}
// The same L2-only step (g_in = 0) with the hardware reciprocal / square root (1 ulp each) instead of the IEEE
// sequences: ~15 instead of ~90 instructions per element-step.  Used only for the lazy replay of untouched rows.
__device__ __forceinline__ void adam_elem_fast(float& w, float& m, float& v, const AdamConsts& c, float step_size, float inv_bc2) {
    float g = __fmul_rn(c.l2_twice, w);
    g = fmaf(w, c.wd, g);
    m = fmaf(c.lerp_w, __fsub_rn(g, m), m);
    v = fmaf(__fmul_rn(c.omb2, g), g, __fmul_rn(v, c.beta2));
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), inv_bc2, c.eps);
    w = fmaf(__fmul_rn(-step_size, m), __builtin_amdgcn_rcpf(denom), w);
}
// Two elements per lane with the packed fp32 instructions of gfx950 (v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per
// issue slot).  Operation for operation the sequence of adam_elem_fast, so the bits are the same; only the square root and
// the reciprocal remain scalar (quarter rate) — they are 8 of the ~13 issue cycles an element-step now costs (18 unpacked).
typedef float cdc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void adam_elem_fast_pk(cdc_f2& w, cdc_f2& m, cdc_f2& v, const AdamConsts& c, float step_size, float inv_bc2) {
    cdc_f2 g = w * c.l2_twice;
    g = __builtin_elementwise_fma(w, (cdc_f2){c.wd, c.wd}, g);
    m = __builtin_elementwise_fma((cdc_f2){c.lerp_w, c.lerp_w}, g - m, m);
    v = __builtin_elementwise_fma(g * c.omb2, g, v * c.beta2);
    cdc_f2 sq;
    sq.x = __builtin_amdgcn_sqrtf(v.x);
    sq.y = __builtin_amdgcn_sqrtf(v.y);
    const cdc_f2 denom = __builtin_elementwise_fma(sq, (cdc_f2){inv_bc2, inv_bc2}, (cdc_f2){c.eps, c.eps});
    cdc_f2 r;
    r.x = __builtin_amdgcn_rcpf(denom.x);
    r.y = __builtin_amdgcn_rcpf(denom.y);
    w = __builtin_elementwise_fma(m * (-step_size), r, w);
}
// replays steps from+1 .. to of the L2-only recurrence for N elements of one row in lockstep (N independent dependency
// chains per thread hide the ~10-deep latency chain of one element-step).  The per-step scalars come from the host table
// while the bias corrections still move (t < n_scalars-1, ~1700 steps) and are constants afterwards.  When every lane of
// the wave replays the same steps (rows not looked up since the last whole-table catch-up) the step index is kept in
// SGPRs, so the table reads are scalar loads.
template <bool FAST, int N>
__device__ __forceinline__ void adam_replay(float (&w)[N], float (&m)[N], float (&v)[N], int from, int to, const AdamConsts& c,
                                            const cdc_adam_hp& hp) {
    const int last_i = hp.n_scalars - 1;
    const float ss_conv = hp.step_scalars[2 * last_i];
    const float bc_conv = FAST ? hp.inv_bc2[last_i] : hp.step_scalars[2 * last_i + 1];
    const int from0 = __builtin_amdgcn_readfirstlane(from);
    const bool uniform = __all(from == from0);
    for (int s = (uniform ? from0 : from) + 1; s <= to; ++s) {      // uniform: s, ss, bc live in SGPRs
        float ss = ss_conv, bc = bc_conv;
        if (s < last_i) {
            ss = hp.step_scalars[2 * s];
            bc = FAST ? hp.inv_bc2[s] : hp.step_scalars[2 * s + 1];
        }
        if constexpr (FAST && N % 2 == 0) {
#pragma unroll
            for (int k = 0; k < N; k += 2) {
                cdc_f2 w2 = {w[k], w[k + 1]}, m2 = {m[k], m[k + 1]}, v2 = {v[k], v[k + 1]};
                adam_elem_fast_pk(w2, m2, v2, c, ss, bc);
                w[k] = w2.x; w[k + 1] = w2.y; m[k] = m2.x; m[k + 1] = m2.y; v[k] = v2.x; v[k + 1] = v2.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (FAST) adam_elem_fast(w[k], m[k], v[k], c, ss, bc);
                else adam_elem(w[k], m[k], v[k], 0.f, c, ss, bc);
            }
        }
    }
}
// The same for a whole wave whose lanes may start at different steps: ONE uniform loop from the earliest `from` of the wave
// (step index and step scalars stay in SGPRs, fetched by scalar loads) with the update predicated on s > from — lanes that
// start later simply sit out the first rounds.  Every lane of the wave must call this (from >= to: nothing to do): the
// minimum is taken with shuffles.  Against adam_replay's per-lane loop this saves two vector loads and their address
// arithmetic per replayed step whenever a wave is not uniform — which is nearly always (a quarter of the rows of a flush
// slice were looked up since its last flush).
// The fast replay in SCALED state: with c = 2*l2 + wd the L2-only step is g = c*w, so M = m / ((1-beta1) c) and
// V = v / ((1-beta2) c^2) obey  M <- beta1*M + w,  V <- beta2*V + w^2,  w <- w - A_t * M / (sqrt(V) + E_t)  with the per-step
// scalars A_t = step_size_t * (1-beta1) c * bc2_t / sqrt((1-beta2) c^2),  E_t = eps * bc2_t / sqrt((1-beta2) c^2)  (formed in double on
// the host; the table holds C1_t = -1/A_t and C2_t = -E_t/A_t, see adam_scaled_step_pk).  Six packed fp32 operations + sqrt + rcp per element pair and step instead of nine: the replay is
// VALU-issue bound (profiles/round2), so the slice launch shrinks with the instruction count.  Same mathematics as
// adam_elem_fast_pk; roundings differ in the last bit per step (the fast replay is 1-ulp arithmetic already).
__device__ __forceinline__ void adam_scaled_step_pk(cdc_f2& w, cdc_f2& M, cdc_f2& V, float beta1, float beta2, float C1, float C2) {
    // round 3: the step scalars travel as C1 = -1/A_t and C2 = -E_t/A_t, so that -A_t / (sqrt(V) + E_t) = 1 / fma(sqrt(V), C1, C2):
    // the scaling by A_t rides in the FMA that formed the denominator anyway — one multiply less per element and step
    // (5 plain + 2 transcendental instead of 6 + 2; the replay is bound by its instruction count)
    M = __builtin_elementwise_fma((cdc_f2){beta1, beta1}, M, w);
    V = __builtin_elementwise_fma((cdc_f2){beta2, beta2}, V, w * w);
    cdc_f2 sq;
    sq.x = __builtin_amdgcn_sqrtf(V.x);
    sq.y = __builtin_amdgcn_sqrtf(V.y);
    const cdc_f2 d = __builtin_elementwise_fma(sq, (cdc_f2){C1, C1}, (cdc_f2){C2, C2});
    cdc_f2 r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    w = __builtin_elementwise_fma(M, r, w);
}
// (Tried: ONE v_rcp_f32 for four denominators, 1/(d0 d1 d2 d3) times the complementary products — 5 instead of 8 transcendentals
// per four element-steps for 5 more multiplies.  9.7 % SLOWER (107.5 vs 98 us per slice launch): a transcendental costs about
// what a packed-fp32 instruction costs here, the replay is bound by the instruction COUNT, 10 per element pair and step.)
template <int N>
__device__ __forceinline__ void adam_replay_wave_scaled(float (&w)[N], float (&m)[N], float (&v)[N], int from, int to, const cdc_adam_hp& hp) {
    static_assert(N % 2 == 0, "pairs of elements");
    const int last_i = hp.n_scalars - 1;
    const float A_conv = hp.replay_tab[2 * last_i], E_conv = hp.replay_tab[2 * last_i + 1];
    const float beta1 = 1.f - hp.lerp_w;
    int fmin = from < to ? from : to;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(fmin, o, 64);
        fmin = other < fmin ? other : fmin;
    }
    fmin = __builtin_amdgcn_readfirstlane(fmin);
    if (fmin >= to) return;                                           // wave-uniform: nothing to replay
    const float ik1 = hp.ik1, ik2 = hp.ik2;
    cdc_f2 W[N / 2], M[N / 2], V[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
        W[k] = (cdc_f2){w[2 * k], w[2 * k + 1]};
        M[k] = (cdc_f2){m[2 * k], m[2 * k + 1]} * ik1;
        V[k] = (cdc_f2){v[2 * k], v[2 * k + 1]} * ik2;
    }
    for (int s = fmin + 1; s <= to; ++s) {                            // s, A, E in SGPRs (scalar loads of the host table)
        float A = A_conv, E = E_conv;
        if (s < last_i) { A = hp.replay_tab[2 * s]; E = hp.replay_tab[2 * s + 1]; }
        if (s > from) {
#pragma unroll
            for (int k = 0; k < N / 2; ++k) adam_scaled_step_pk(W[k], M[k], V[k], beta1, hp.beta2, A, E);
        }
    }
    if (from < to) {
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            // k + k_lo = 1 / ik to 2^-48: out and in are inverses of each other, a segment leaves no factor (1 + 6e-8) on the state
            const cdc_f2 mo = __builtin_elementwise_fma(M[k], (cdc_f2){hp.k1, hp.k1}, M[k] * hp.k1_lo);
            const cdc_f2 vo = __builtin_elementwise_fma(V[k], (cdc_f2){hp.k2, hp.k2}, V[k] * hp.k2_lo);
            w[2 * k] = W[k].x; w[2 * k + 1] = W[k].y;
            m[2 * k] = mo.x; m[2 * k + 1] = mo.y;
            v[2 * k] = vo.x; v[2 * k + 1] = vo.y;
        }
    }
}

template <bool FAST, int N>
__device__ __forceinline__ void adam_replay_wave(float (&w)[N], float (&m)[N], float (&v)[N], int from, int to, const AdamConsts& c,
                                                 const cdc_adam_hp& hp) {
    if constexpr (FAST && N % 2 == 0) {
        if (hp.replay_tab && hp.k1 > 0.f && hp.k2 > 0.f) {            // uniform: the scaled form needs a decay term (c > 0)
            adam_replay_wave_scaled<N>(w, m, v, from, to, hp);
            return;
        }
    }
    const int last_i = hp.n_scalars - 1;
    const float ss_conv = hp.step_scalars[2 * last_i];
    const float bc_conv = FAST ? hp.inv_bc2[last_i] : hp.step_scalars[2 * last_i + 1];
    int fmin = from < to ? from : to;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(fmin, o, 64);
        fmin = other < fmin ? other : fmin;
    }
    fmin = __builtin_amdgcn_readfirstlane(fmin);
    for (int s = fmin + 1; s <= to; ++s) {
        float ss = ss_conv, bc = bc_conv;
        if (s < last_i) {
            ss = hp.step_scalars[2 * s];
            bc = FAST ? hp.inv_bc2[s] : hp.step_scalars[2 * s + 1];
        }
        if (s > from) {
            if constexpr (FAST && N % 2 == 0) {
#pragma unroll
                for (int k = 0; k < N; k += 2) {
                    cdc_f2 w2 = {w[k], w[k + 1]}, m2 = {m[k], m[k + 1]}, v2 = {v[k], v[k + 1]};
                    adam_elem_fast_pk(w2, m2, v2, c, ss, bc);
                    w[k] = w2.x; w[k + 1] = w2.y; m[k] = m2.x; m[k + 1] = m2.y; v[k] = v2.x; v[k + 1] = v2.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    if (FAST) adam_elem_fast(w[k], m[k], v[k], c, ss, bc);
                    else adam_elem(w[k], m[k], v[k], 0.f, c, ss, bc);
                }
            }
        }
    }
}

template <bool FAST>
__device__ __forceinline__ void adam_replay(float& w, float& m, float& v, int from, int to, const AdamConsts& c,
                                            const cdc_adam_hp& hp) {
    float wa[1] = {w}, ma[1] = {m}, va[1] = {v};
    adam_replay<FAST, 1>(wa, ma, va, from, to, c, hp);
    w = wa[0]; m = ma[0]; v = va[0];
}
__device__ __forceinline__ AdamConsts make_consts(const cdc_adam_hp& hp) {
    AdamConsts c;
    c.lerp_w = hp.lerp_w; c.beta2 = hp.beta2; c.omb2 = hp.one_minus_beta2;
    c.eps = hp.eps; c.wd = hp.weight_decay; c.l2_twice = hp.l2_twice;
    return c;
}
__device__ __forceinline__ void step_scalars_at(const float* tab, int n, int t, float& step_size, float& bc2s) {
    int i = t < n ? t : n - 1;
    if (i < 0) i = 0;
    step_size = tab[2 * i];
    bc2s = tab[2 * i + 1];
}

// Which group owns work item `id`?  Walking the descriptors in the argument block one group at a time costs a scalar cache
// miss per group before the first load of the tile can be issued; here lane g reads group g's item count (one vector load
// of the argument block for the whole search), a wave scan turns the counts into offsets and a ballot names the owner.
// Every lane of the wave must call this.  Returns the group (wave-uniform) or -1; local = id - first item of the group;
// if PREFIX: *prefix_out = sum of extra(g') over the groups before the owner.
template <bool PREFIX, typename Count, typename Extra>
__device__ __forceinline__ int find_group(int n_groups, int id, Count count, Extra extra, int& local, int64_t* prefix_out) {
    const int lane = threadIdx.x & 63;
    const int c = lane < n_groups ? count(lane) : 0;
    int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    const int exc = inc - c;
    const unsigned long long owner = __ballot(id >= exc && id < inc);
    if (owner == 0ull) return -1;
    const int g = __builtin_amdgcn_readfirstlane(__ffsll((long long)owner) - 1);
    local = id - __builtin_amdgcn_readlane(exc, g);
    if constexpr (PREFIX) {
        const int64_t x = lane < n_groups ? extra(lane) : 0;
        int64_t xi = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t t = __shfl_up(xi, off, 64);
            if (lane >= off) xi += t;
        }
        const int64_t xe = xi - x;
        const int lo = __builtin_amdgcn_readlane((int)(uint32_t)xe, g), hi = __builtin_amdgcn_readlane((int)(xe >> 32), g);
        *prefix_out = ((int64_t)hi << 32) | (uint32_t)lo;
    }
    return g;
}

