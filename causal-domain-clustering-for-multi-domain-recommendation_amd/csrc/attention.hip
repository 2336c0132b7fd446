// attention.hip — the per-sample self-attention over the field tokens of the reference's attention branch
// (model/layer.py:58-84: BaseModel.build_atten / atten_forward, nn.MultiheadAttention(A, H) applied to the [F, B, A]
// token tensor, i.e. every sample attends over its own F field tokens).
//
// The projections around it (token embedding D->A, in_proj A->3A, out_proj A->A, residual D->A, final F*A->1) are grouped
// linears / row dots on [B*F, .] buffers; this file holds what torch.nn.functional.multi_head_attention_forward does in
// between: q*dh^-1/2, scores = q k^T, softmax over the keys, dropout on the probabilities (training), probs @ v — and its
// backward.  Sequences are short (F = 26 field tokens, dh = 32): one wave per (sample, head), lane = query token; K, V (and
// in the backward Q, dO, the probabilities and dS) are staged in LDS.  fp32 throughout.
#include "common.h"

#define ATTN_WAVES 4
#define ATTN_THREADS (ATTN_WAVES * 64)

// dropout on the attention probabilities: the same counter-based stream as the linears (common.h), keyed by the flat index
// of the probability; the backward regenerates the decisions instead of storing a mask
__device__ __forceinline__ float attn_keep(uint64_t seed, uint64_t idx, float drop_p, float keep_scale) {
    return cdc_uniform(seed, idx) < drop_p ? 0.f : keep_scale;
}

typedef float attn_f4 __attribute__((ext_vector_type(4)));
// row operations on an LDS row of DH floats (DH % 4 == 0) through 16-byte reads: broadcast rows cost a quarter of the LDS
// issue slots of scalar reads
template <int DH>
__device__ __forceinline__ float lds_dot(const float (&r)[DH], const float* row) {
    const attn_f4* r4 = reinterpret_cast<const attn_f4*>(row);
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH / 4; ++d) {
        const attn_f4 x = r4[d];
        s += r[4 * d] * x[0]; s += r[4 * d + 1] * x[1]; s += r[4 * d + 2] * x[2]; s += r[4 * d + 3] * x[3];
    }
    return s;
}
template <int DH>
__device__ __forceinline__ void lds_axpy(float (&acc)[DH], float a, const float* row) {
    const attn_f4* r4 = reinterpret_cast<const attn_f4*>(row);
#pragma unroll
    for (int d = 0; d < DH / 4; ++d) {
        const attn_f4 x = r4[d];
        acc[4 * d] += a * x[0]; acc[4 * d + 1] += a * x[1]; acc[4 * d + 2] += a * x[2]; acc[4 * d + 3] += a * x[3];
    }
}

template <int DH>
__global__ void __launch_bounds__(ATTN_THREADS) k_attn_fwd(const float* __restrict__ qkv, int64_t ld, float* __restrict__ out, int64_t ldo,
                                                           float* __restrict__ probs, int64_t B, int32_t F, int32_t A, int32_t H,
                                                           float drop_p, uint64_t seed, const int32_t* __restrict__ seed_offset_dev) {
    extern __shared__ __attribute__((aligned(16))) float attn_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pair = (int64_t)blockIdx.x * ATTN_WAVES + wave;          // (sample, head)
    float* Ks = attn_smem + (int64_t)wave * (2 * F * DH + F * (F + 1));
    float* Vs = Ks + F * DH;
    float* Ps = Vs + F * DH;                                                // [F][F+1]
    const bool live = pair < B * H;
    const int64_t b = live ? pair / H : 0;
    const int h = live ? (int)(pair % H) : 0;
    const float* base = qkv + b * F * ld + h * DH;                          // token f of the sample: row b*F + f
    for (int i = lane; i < F * DH; i += 64) {
        const int f = i / DH, d = i % DH;
        Ks[i] = base[(int64_t)f * ld + A + d];
        Vs[i] = base[(int64_t)f * ld + 2 * A + d];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!live || lane >= F) return;
    if (drop_p > 0.f && seed_offset_dev) seed += (uint64_t)(uint32_t)(*seed_offset_dev) * 0xD1342543DE82EF95ull;
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const float scale = rsqrtf((float)DH);
    float q[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) q[d] = base[(int64_t)lane * ld + d] * scale;       // torch scales q before the product
    float* prow = Ps + lane * (F + 1);
    float mx = -INFINITY;
    for (int j = 0; j < F; ++j) {
        const float s = lds_dot<DH>(q, Ks + j * DH);
        prow[j] = s;
        mx = fmaxf(mx, s);
    }
    float sum = 0.f;
    for (int j = 0; j < F; ++j) {
        const float e = expf(prow[j] - mx);
        prow[j] = e;
        sum += e;
    }
    const float inv = 1.f / sum;
    float o[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = 0.f;
    float* pg = probs ? probs + ((pair * F + lane) * (int64_t)F) : nullptr;
    for (int j = 0; j < F; ++j) {
        const float p = prow[j] * inv;
        if (pg) pg[j] = p;                                                   // the softmax output (before dropout), for the backward
        float pd = p;
        if (drop_p > 0.f) pd *= attn_keep(seed, (uint64_t)((pair * F + lane) * (int64_t)F + j), drop_p, keep_scale);
        lds_axpy<DH>(o, pd, Vs + j * DH);
    }
    float* dst = out + (b * F + lane) * ldo + h * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) dst[d] = o[d];
}

template <int DH>
__global__ void __launch_bounds__(ATTN_THREADS) k_attn_bwd(const float* __restrict__ qkv, int64_t ld, const float* __restrict__ probs,
                                                           const float* __restrict__ dout, int64_t lddo, float* __restrict__ dqkv,
                                                           int64_t lddq, int64_t B, int32_t F, int32_t A, int32_t H, float drop_p,
                                                           uint64_t seed, const int32_t* __restrict__ seed_offset_dev) {
    extern __shared__ __attribute__((aligned(16))) float attn_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pair = (int64_t)blockIdx.x * ATTN_WAVES + wave;
    // phase 1 (per query row) reads K and V, phase 2 (per key row) reads Q and dO: the two pairs share one LDS region, which
    // keeps a wave at 12 KB (F = 26, dh = 32) and thirteen waves resident per CU instead of eight
    float* Ks = attn_smem + (int64_t)wave * (2 * F * DH + 2 * F * (F + 1));
    float* Vs = Ks + F * DH;
    float* Qs = Ks;                                                         // phase 2 aliases
    float* Os = Vs;                                                         // dO
    float* Pd = Vs + F * DH;                                                // dropped probabilities  [F][F+1]
    float* Ds = Pd + F * (F + 1);                                           // dS                     [F][F+1]
    const bool live = pair < B * H;
    const int64_t b = live ? pair / H : 0;
    const int h = live ? (int)(pair % H) : 0;
    const float* base = qkv + b * F * ld + h * DH;
    const float scale = rsqrtf((float)DH);
    for (int i = lane; i < F * DH; i += 64) {
        const int f = i / DH, d = i % DH;
        Ks[i] = base[(int64_t)f * ld + A + d];
        Vs[i] = base[(int64_t)f * ld + 2 * A + d];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (drop_p > 0.f && seed_offset_dev) seed += (uint64_t)(uint32_t)(*seed_offset_dev) * 0xD1342543DE82EF95ull;
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    if (live && lane < F) {
        // query row `lane`: dP = dO v^T (through the dropout mask), dS = P * (dP - sum_j dP P)
        const float* pg = probs + ((pair * F + lane) * (int64_t)F);
        float orow[DH];                                                    // own dO row in registers: reading it from LDS at a
#pragma unroll                                                             // stride of DH floats would put every lane on two banks
        for (int d = 0; d < DH; ++d) orow[d] = dout[(b * F + lane) * lddo + h * DH + d];
        float dot = 0.f;
        for (int j = 0; j < F; ++j) {
            float dp = lds_dot<DH>(orow, Vs + j * DH);
            const float p = pg[j];
            float keep = 1.f;
            if (drop_p > 0.f) keep = attn_keep(seed, (uint64_t)((pair * F + lane) * (int64_t)F + j), drop_p, keep_scale);
            Pd[lane * (F + 1) + j] = p * keep;
            dp *= keep;
            Ds[lane * (F + 1) + j] = dp;
            dot += dp * p;
        }
        float dq[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) dq[d] = 0.f;
        for (int j = 0; j < F; ++j) {
            const float ds = pg[j] * (Ds[lane * (F + 1) + j] - dot);
            Ds[lane * (F + 1) + j] = ds;
            lds_axpy<DH>(dq, ds, Ks + j * DH);
        }
        float* dst = dqkv + (b * F + lane) * lddq + h * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) dst[d] = dq[d] * scale;                 // q was scaled before the product
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < F * DH; i += 64) {                               // K and V are done: their space takes Q and dO
        const int f = i / DH, d = i % DH;
        Qs[i] = base[(int64_t)f * ld + d] * scale;
        Os[i] = dout[(b * F + f) * lddo + h * DH + d];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (live && lane < F) {
        // key/value row `lane`: dk = sum_i dS[i][lane] * (q_i * scale), dv = sum_i Pd[i][lane] * dO_i   (i ascending)
        float dk[DH], dv[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        for (int i = 0; i < F; ++i) {
            const float ds = Ds[i * (F + 1) + lane], pd = Pd[i * (F + 1) + lane];
            lds_axpy<DH>(dk, ds, Qs + i * DH);
            lds_axpy<DH>(dv, pd, Os + i * DH);
        }
        float* dst = dqkv + (b * F + lane) * lddq + h * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) { dst[A + d] = dk[d]; dst[2 * A + d] = dv[d]; }
    }
}

static bool attn_args_ok(int64_t B, int32_t F, int32_t A, int32_t H, float drop_p) {
    if (B < 0 || F <= 0 || F > 64 || A <= 0 || H <= 0 || A % H != 0 || drop_p < 0.f || drop_p >= 1.f) return false;
    const int dh = A / H;
    return dh == 4 || dh == 8 || dh == 16 || dh == 32 || dh == 64;
}

#define ATTN_DISPATCH(KERNEL, LDS_FLOATS, ...)                                                                               \
    do {                                                                                                                      \
        const int dh__ = A / H;                                                                                               \
        const size_t lds__ = (size_t)ATTN_WAVES * (LDS_FLOATS) * sizeof(float);                                               \
        const dim3 grid__((unsigned)cdc_ceil_div(B * H, ATTN_WAVES));                                                         \
        CDC_CHECK_ARG(lds__ <= 160 * 1024, CDC_E_TOOBIG, "attention: F=%d dh=%d needs %zu bytes of LDS", F, dh__, lds__);     \
        if (lds__ > 64 * 1024) {                                                                                              \
            const void* fn__ = dh__ == 4 ? (const void*)KERNEL<4> : dh__ == 8 ? (const void*)KERNEL<8> :                      \
                               dh__ == 16 ? (const void*)KERNEL<16> : dh__ == 32 ? (const void*)KERNEL<32> : (const void*)KERNEL<64>; \
            hipError_t e__ = hipFuncSetAttribute(fn__, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);               \
            if (e__ != hipSuccess) { cdc_set_error("attention: cannot raise the LDS limit: %s", hipGetErrorString(e__)); return (int)e__; } \
        }                                                                                                                     \
        switch (dh__) {                                                                                                       \
            case 4:  hipLaunchKernelGGL(KERNEL<4>,  grid__, dim3(ATTN_THREADS), lds__, (hipStream_t)stream, __VA_ARGS__); break; \
            case 8:  hipLaunchKernelGGL(KERNEL<8>,  grid__, dim3(ATTN_THREADS), lds__, (hipStream_t)stream, __VA_ARGS__); break; \
            case 16: hipLaunchKernelGGL(KERNEL<16>, grid__, dim3(ATTN_THREADS), lds__, (hipStream_t)stream, __VA_ARGS__); break; \
            case 32: hipLaunchKernelGGL(KERNEL<32>, grid__, dim3(ATTN_THREADS), lds__, (hipStream_t)stream, __VA_ARGS__); break; \
            default: hipLaunchKernelGGL(KERNEL<64>, grid__, dim3(ATTN_THREADS), lds__, (hipStream_t)stream, __VA_ARGS__); break; \
        }                                                                                                                     \
    } while (0)

extern "C" int cdc_attn_fwd(const float* qkv, int64_t ld, float* out, int64_t ldo, float* probs, int64_t B, int32_t F, int32_t A,
                            int32_t H, float drop_p, uint64_t seed, const int32_t* seed_offset_dev, void* stream) {
    CDC_CHECK_ARG(qkv && out && attn_args_ok(B, F, A, H, drop_p) && ld >= 3 * (int64_t)A && ldo >= A, CDC_E_BADARG,
                  "attn_fwd: bad argument (F <= 64, head dim in {4,8,16,32,64})");
    if (B == 0) return 0;
    ATTN_DISPATCH(k_attn_fwd, 2 * F * (A / H) + F * (F + 1), qkv, ld, out, ldo, probs, B, F, A, H, drop_p, seed, seed_offset_dev);
    CDC_LAUNCH_CHECK("attn_fwd");
    return 0;
}

extern "C" int cdc_attn_bwd(const float* qkv, int64_t ld, const float* probs, const float* dout, int64_t lddo, float* dqkv,
                            int64_t lddq, int64_t B, int32_t F, int32_t A, int32_t H, float drop_p, uint64_t seed,
                            const int32_t* seed_offset_dev, void* stream) {
    CDC_CHECK_ARG(qkv && probs && dout && dqkv && attn_args_ok(B, F, A, H, drop_p) && ld >= 3 * (int64_t)A && lddo >= A &&
                      lddq >= 3 * (int64_t)A, CDC_E_BADARG, "attn_bwd: bad argument (F <= 64, head dim in {4,8,16,32,64})");
    if (B == 0) return 0;
    ATTN_DISPATCH(k_attn_bwd, 2 * F * (A / H) + 2 * F * (F + 1), qkv, ld, probs, dout, lddo, dqkv, lddq, B, F, A, H, drop_p, seed,
                  seed_offset_dev);
    CDC_LAUNCH_CHECK("attn_bwd");
    return 0;
}

// out = relu(a + b) over [rows, cols] (model/layer.py:80-82: `cross_term += V_res; F.relu(cross_term)`), and its backward:
// d = (out > 0) ? dout : 0 goes to both addends
__global__ void __launch_bounds__(256) k_add_relu_fwd(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                      float* __restrict__ out, int64_t ldo, int64_t rows, int32_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        out[r * ldo + c] = fmaxf(a[r * lda + c] + b[r * ldb + c], 0.f);
    }
}
__global__ void __launch_bounds__(256) k_add_relu_bwd(const float* __restrict__ out, int64_t ldo, const float* __restrict__ dout,
                                                      int64_t lddo, float* __restrict__ da, int64_t ldda, int32_t acc_a,
                                                      float* __restrict__ db, int64_t lddb, int32_t acc_b, int64_t rows, int32_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const float d = out[r * ldo + c] > 0.f ? dout[r * lddo + c] : 0.f;
        float* pa = da + r * ldda + c;
        float* pb = db + r * lddb + c;
        *pa = acc_a ? *pa + d : d;
        *pb = acc_b ? *pb + d : d;
    }
}
extern "C" int cdc_add_relu_fwd(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int64_t rows,
                                int32_t cols, void* stream) {
    CDC_CHECK_ARG(a && b && out && rows >= 0 && cols > 0 && lda >= cols && ldb >= cols && ldo >= cols, CDC_E_BADARG, "add_relu_fwd: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, 256), 8192);
    hipLaunchKernelGGL(k_add_relu_fwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out, ldo, rows, cols);
    CDC_LAUNCH_CHECK("add_relu_fwd");
    return 0;
}
extern "C" int cdc_add_relu_bwd(const float* out, int64_t ldo, const float* dout, int64_t lddo, float* da, int64_t ldda, int32_t acc_a,
                                float* db, int64_t lddb, int32_t acc_b, int64_t rows, int32_t cols, void* stream) {
    CDC_CHECK_ARG(out && dout && da && db && rows >= 0 && cols > 0 && ldo >= cols && lddo >= cols && ldda >= cols && lddb >= cols,
                  CDC_E_BADARG, "add_relu_bwd: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, 256), 8192);
    hipLaunchKernelGGL(k_add_relu_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, ldo, dout, lddo, da, ldda, acc_a, db, lddb,
                       acc_b, rows, cols);
    CDC_LAUNCH_CHECK("add_relu_bwd");
    return 0;
}
