// adam_dense.h — the dense parameters' Adam step on one chunk of one tensor (run.py:720-721 + the L2 term of model/layer.py:96-112),
// shared by the multi-tensor launches (csrc/rowops.hip cdc_adam_multi / cdc_adam_multi_table) and the launch that updates the step's
// table rows and the dense parameters together (csrc/embedding.hip cdc_embed_segsum_lazy_update_dense).  256 threads per workgroup.
#pragma once
#include "common.h"

#define ADAM_CHUNK CDC_ADAM_CHUNK
#define ADAM_THREADS 256
typedef float adam_f4 __attribute__((ext_vector_type(4)));
struct AdamHdr {                        // what a chunk needs of cdc_adam_args besides its tensor
    float lerp_w, beta2, one_minus_beta2, eps, weight_decay, grad_scale;
    const float* step_scalars; int32_t n_scalars;
    const int32_t* step_dev;
    double* reg_sum; const double* reg_seed;
};

// elements [begin, min(begin + SPAN, T.n)) of tensor T.  With so few workgroups the launch lives on memory-level parallelism, so
// every thread issues ALL its loads of a span (SPAN / 256 elements x 4 arrays, as 16-byte loads when the tensors allow) before it
// computes.  gradient: the tensor T.g, or the sum of the split-K slabs of the grad-weight launch that left its reduction to us
// (same adds in the same order as k_bwd_w_reduce: 0 + slab 0 + slab 1 + ...)
template <int SPAN>
__device__ __forceinline__ void adam_span(const AdamHdr& a, const cdc_adam_tensor& T, const int64_t begin, const AdamConsts& c,
                                          const float step_size, const float bc2s, double& sq) {
    const int64_t end = min(begin + SPAN, T.n);
    const int n_slabs = T.n_slabs;
    const float* const slabs = T.slabs;
    const int64_t sstride = T.slab_stride;
    const bool vec = ((((uintptr_t)T.w | (uintptr_t)T.m | (uintptr_t)T.v | (uintptr_t)(n_slabs > 0 ? slabs : T.g)) & 15) == 0) &&
                     (n_slabs <= 0 || sstride % 4 == 0);
    if (vec && begin + SPAN <= T.n) {
        constexpr int R = SPAN / (4 * ADAM_THREADS);
        adam_f4 w[R], m[R], v[R], g[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int64_t i = begin + ((int64_t)q * ADAM_THREADS + threadIdx.x) * 4;
            w[q] = *reinterpret_cast<const adam_f4*>(T.w + i);
            m[q] = *reinterpret_cast<const adam_f4*>(T.m + i);
            v[q] = *reinterpret_cast<const adam_f4*>(T.v + i);
            g[q] = (T.g && n_slabs <= 0) ? *reinterpret_cast<const adam_f4*>(T.g + i) : adam_f4{0.f, 0.f, 0.f, 0.f};
        }
        // slabs in rounds of eight (four), every load of a round in flight before the first add (a round per slab was one memory latency
        // per slab: 70 us for the two launches of C2 instead of 16); the adds stay in slab order
        constexpr int SB = SPAN == ADAM_CHUNK ? 8 : 4;               // (the two-pass form keeps its register count at four waves per SIMD)
        for (int s0 = 0; s0 < n_slabs; s0 += SB) {
            adam_f4 t[SB][R];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                if (s0 + j < n_slabs) {                              // uniform
                    const float* sp = slabs + (int64_t)(s0 + j) * sstride + begin + (int64_t)threadIdx.x * 4;
#pragma unroll
                    for (int q = 0; q < R; ++q) t[j][q] = *reinterpret_cast<const adam_f4*>(sp + (int64_t)q * ADAM_THREADS * 4);
                }
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                if (s0 + j < n_slabs) {
#pragma unroll
                    for (int q = 0; q < R; ++q) g[q] += t[j][q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int64_t i = begin + ((int64_t)q * ADAM_THREADS + threadIdx.x) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float we = w[q][e], me = m[q][e], ve = v[q][e], ge = g[q][e];
                if (a.grad_scale != 1.f) ge *= a.grad_scale;
                sq += (double)(we * we);
                adam_elem(we, me, ve, ge, c, step_size, bc2s);
                w[q][e] = we; m[q][e] = me; v[q][e] = ve;
            }
            *reinterpret_cast<adam_f4*>(T.w + i) = w[q];
            *reinterpret_cast<adam_f4*>(T.m + i) = m[q];
            *reinterpret_cast<adam_f4*>(T.v + i) = v[q];
        }
    } else {
        for (int64_t i0 = begin + threadIdx.x; i0 < end; i0 += 4 * ADAM_THREADS) {
            float w[4], m[4], v[4], g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t i = i0 + (int64_t)q * ADAM_THREADS;
                const bool ok = i < end;
                w[q] = ok ? T.w[i] : 0.f; m[q] = ok ? T.m[i] : 0.f; v[q] = ok ? T.v[i] : 0.f;
                g[q] = (ok && T.g && n_slabs <= 0) ? T.g[i] : 0.f;
            }
            for (int s0 = 0; s0 < n_slabs; s0 += 8) {                // rounds of eight slabs, loads first (see the vector path)
                float t[8][4];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (s0 + j < n_slabs) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int64_t i = i0 + (int64_t)q * ADAM_THREADS;
                            t[j][q] = i < end ? slabs[(int64_t)(s0 + j) * sstride + i] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (s0 + j < n_slabs) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) g[q] += t[j][q];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t i = i0 + (int64_t)q * ADAM_THREADS;
                if (i >= end) continue;
                if (a.grad_scale != 1.f) g[q] *= a.grad_scale;
                sq += (double)(w[q] * w[q]);
                adam_elem(w[q], m[q], v[q], g[q], c, step_size, bc2s);
                T.w[i] = w[q]; T.m[i] = m[q]; T.v[i] = v[q];
            }
        }
    }
}

// chunk `chunk` (ADAM_CHUNK elements) of tensor T in passes of SPAN elements (a pass keeps SPAN / 256 elements x 4 arrays (+ eight
// slabs) per thread in registers: SPAN = ADAM_CHUNK is one pass at two waves per SIMD, SPAN = ADAM_CHUNK / 2 fits four).  One
// workgroup covers a chunk so that the launch ends in a few hundred (not thousands of) same-address double atomics for the
// regularisation sum.  first_wg: the launch's first Adam workgroup (adds the cached table term, cdc_adam_args.reg_seed).
template <int SPAN>
__device__ __forceinline__ void adam_chunk(const AdamHdr& a, const cdc_adam_tensor& T, const int chunk, const bool first_wg) {
    static_assert(ADAM_CHUNK % SPAN == 0 && SPAN % (4 * ADAM_THREADS) == 0, "passes of whole 16-byte lanes");
    AdamConsts c;
    c.lerp_w = a.lerp_w; c.beta2 = a.beta2; c.omb2 = a.one_minus_beta2; c.eps = a.eps; c.wd = a.weight_decay;
    c.l2_twice = 2.f * T.l2;
    float step_size, bc2s;
    step_scalars_at(a.step_scalars, a.n_scalars, *a.step_dev, step_size, bc2s);
    const int64_t begin = (int64_t)chunk * ADAM_CHUNK;
    double sq = 0.0;
#pragma unroll 1
    for (int64_t b = begin; b < begin + ADAM_CHUNK && b < T.n; b += SPAN) adam_span<SPAN>(a, T, b, c, step_size, bc2s, sq);
    if (a.reg_sum && a.reg_seed && first_wg && threadIdx.x == 0) atomicAdd(a.reg_sum, *a.reg_seed);
    if (a.reg_sum && T.l2 != 0.f) {
        __shared__ double part[ADAM_THREADS / 64];
        sq = wave_sum_d(sq);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(a.reg_sum, (double)T.l2 * (part[0] + part[1] + part[2] + part[3]));
    }
}
