// misc.hip — error plumbing and the small byte-moving helpers of the step driver.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void cdc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int cdc_abi_version(void) { return CDC_ABI_VERSION; }
extern "C" const char* cdc_last_error(void) { return g_err; }

__global__ void k_step_increment(int32_t* step) { *step += 1; }
extern "C" int cdc_step_increment(int32_t* step_dev, void* stream) {
    CDC_CHECK_ARG(step_dev, CDC_E_BADARG, "step_increment: null pointer");
    hipLaunchKernelGGL(k_step_increment, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    CDC_LAUNCH_CHECK("step_increment");
    return 0;
}

__global__ void k_begin_step(int32_t* step, double* acc, int32_t n) {
    if (threadIdx.x == 0) *step += 1;
    if ((int)threadIdx.x < n) acc[threadIdx.x] = 0.0;
}
extern "C" int cdc_begin_step(int32_t* step_dev, double* accumulators, int32_t n_acc, void* stream) {
    CDC_CHECK_ARG(step_dev && n_acc >= 0 && n_acc <= 64 && (n_acc == 0 || accumulators), CDC_E_BADARG, "begin_step: bad argument");
    hipLaunchKernelGGL(k_begin_step, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, accumulators, n_acc);
    CDC_LAUNCH_CHECK("begin_step");
    return 0;
}

template <typename T>
__global__ void __launch_bounds__(256) k_fill(T* p, T value, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = value;
}
extern "C" int cdc_fill_f32(float* p, float value, int64_t n, void* stream) {
    CDC_CHECK_ARG(p && n >= 0, CDC_E_BADARG, "fill_f32: bad argument");
    if (n == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(n, 256), 2048);
    hipLaunchKernelGGL(k_fill<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, value, n);
    CDC_LAUNCH_CHECK("fill_f32");
    return 0;
}
extern "C" int cdc_fill_f64(double* p, double value, int64_t n, void* stream) {
    CDC_CHECK_ARG(p && n >= 0, CDC_E_BADARG, "fill_f64: bad argument");
    if (n == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(n, 256), 2048);
    hipLaunchKernelGGL(k_fill<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, value, n);
    CDC_LAUNCH_CHECK("fill_f64");
    return 0;
}

// STAR: effective weight = W_domain ⊙ W_shared for every domain in one launch (model/star.py:90-93,100)
__global__ void __launch_bounds__(256) k_mul_bcast(const float* __restrict__ a, const float* __restrict__ b,
                                                   float* __restrict__ out, int64_t na, int64_t nb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] * b[i % nb];
}
extern "C" int cdc_mul_bcast(const float* a, const float* b, float* out, int64_t na, int64_t nb, void* stream) {
    CDC_CHECK_ARG(a && b && out && na > 0 && nb > 0 && na % nb == 0, CDC_E_BADARG, "mul_bcast: bad argument");
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(na, 256), 2048);
    hipLaunchKernelGGL(k_mul_bcast, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, na, nb);
    CDC_LAUNCH_CHECK("mul_bcast");
    return 0;
}
// da[i] = d_out[i]*b[i%nb];  db[j] = sum_r d_out[r*nb+j]*a[r*nb+j]  (fixed order over r: deterministic)
__global__ void __launch_bounds__(256) k_mul_bcast_bwd(const float* __restrict__ d_out, const float* __restrict__ a,
                                                       const float* __restrict__ b, float* __restrict__ da,
                                                       float* __restrict__ db, int64_t na, int64_t nb) {
    const int64_t reps = na / nb;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nb; j += (int64_t)gridDim.x * blockDim.x) {
        const float bj = b[j];
        float acc = 0.f;
        for (int64_t r = 0; r < reps; ++r) {
            const float g = d_out[r * nb + j];
            if (da) da[r * nb + j] = g * bj;
            acc += g * a[r * nb + j];
        }
        if (db) db[j] = acc;
    }
}
extern "C" int cdc_mul_bcast_bwd(const float* d_out, const float* a, const float* b, float* da, float* db, int64_t na,
                                 int64_t nb, void* stream) {
    CDC_CHECK_ARG(d_out && a && b && na > 0 && nb > 0 && na % nb == 0, CDC_E_BADARG, "mul_bcast_bwd: bad argument");
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(nb, 256), 2048);
    hipLaunchKernelGGL(k_mul_bcast_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_out, a, b, da, db, na, nb);
    CDC_LAUNCH_CHECK("mul_bcast_bwd");
    return 0;
}

// Stable partition of the batch by group id (model/star.py:84-86: boolean masks in ascending group
// order keep the original row order inside a group).  One workgroup; B is a minibatch.
__global__ void __launch_bounds__(1024) k_group_partition(const int64_t* __restrict__ group, int32_t* __restrict__ row_offsets,
                                                          int32_t* __restrict__ order, int32_t B, int32_t n_group) {
    extern __shared__ int32_t cnt[];   // [n_group + 1] then per-thread scratch is not needed
    const int tid = threadIdx.x;
    for (int g = tid; g <= n_group; g += blockDim.x) cnt[g] = 0;
    __syncthreads();
    for (int i = tid; i < B; i += blockDim.x) {
        const int64_t g = group[i];
        if (g >= 0 && g < n_group) atomicAdd(&cnt[g], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int g = 0; g < n_group; ++g) { const int c = cnt[g]; cnt[g] = run; run += c; }
        cnt[n_group] = run;
    }
    __syncthreads();
    for (int g = tid; g <= n_group; g += blockDim.x) row_offsets[g] = cnt[g];
    // stable placement: wave-ordered ranking. Each group is ranked by one wave scanning the batch in order.
    const int wave = tid >> 6, lane = tid & 63, n_wave = blockDim.x >> 6;
    for (int g = wave; g < n_group; g += n_wave) {
        int base = cnt[g];
        for (int i0 = 0; i0 < B; i0 += 64) {
            const int i = i0 + lane;
            const bool hit = (i < B) && (group[i] == g);
            const unsigned long long mask = __ballot(hit);
            if (hit) order[base + __popcll(mask & ((1ull << lane) - 1ull))] = i;
            base += __popcll(mask);
        }
    }
}
extern "C" int cdc_group_partition(const int64_t* group, int32_t* row_offsets, int32_t* order, int64_t B, int32_t n_group,
                                   void* stream) {
    CDC_CHECK_ARG(group && row_offsets && order && B > 0 && n_group > 0 && n_group <= 8192 && B < (1ll << 31), CDC_E_BADARG,
                  "group_partition: bad argument");
    hipLaunchKernelGGL(k_group_partition, dim3(1), dim3(1024), (n_group + 1) * sizeof(int32_t), (hipStream_t)stream, group,
                       row_offsets, order, (int32_t)B, n_group);
    CDC_LAUNCH_CHECK("group_partition");
    return 0;
}

__global__ void __launch_bounds__(256) k_rows_permute(const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ order,
                                                      float* __restrict__ out, int64_t ld_out, int64_t B, int32_t C,
                                                      int32_t inverse) {
    const int64_t total = B * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pos = i / C;
        const int c = (int)(i - pos * C);
        const int64_t src = order[pos];
        if (inverse) out[src * ld_out + c] = in[pos * ld_in + c];
        else out[pos * ld_out + c] = in[src * ld_in + c];
    }
}
extern "C" int cdc_rows_permute(const float* in, int64_t ld_in, const int32_t* order, float* out, int64_t ld_out, int64_t B,
                                int32_t C, int32_t inverse, void* stream) {
    CDC_CHECK_ARG(in && order && out && B > 0 && C > 0, CDC_E_BADARG, "rows_permute: bad argument");
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(B * C, 256), 4096);
    hipLaunchKernelGGL(k_rows_permute, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ld_in, order, out, ld_out, B, C, inverse);
    CDC_LAUNCH_CHECK("rows_permute");
    return 0;
}

__global__ void __launch_bounds__(256) k_add_inplace(float* __restrict__ dst, int64_t ld_dst, const float* __restrict__ src,
                                                     int64_t ld_src, int64_t rows, int32_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ld_dst + c] += src[r * ld_src + c];
    }
}
extern "C" int cdc_add_inplace(float* dst, int64_t ld_dst, const float* src, int64_t ld_src, int64_t rows, int32_t cols,
                               void* stream) {
    CDC_CHECK_ARG(dst && src && rows >= 0 && cols > 0, CDC_E_BADARG, "add_inplace: bad argument");
    if (rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(rows * cols, 256), 4096);
    hipLaunchKernelGGL(k_add_inplace, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, ld_dst, src, ld_src, rows, cols);
    CDC_LAUNCH_CHECK("add_inplace");
    return 0;
}

// dst (=|+=) src_0 + src_1 + ... (added in list order): the fan-in of one gradient from several producers in ONE launch
__global__ void __launch_bounds__(256) k_add_n(const cdc_add_n_args a) {
    const int64_t total = a.rows * a.cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / a.cols;
        const int c = (int)(i - r * a.cols);
        float* d = a.dst + r * a.ld_dst + c;
        float v[CDC_MAX_GROUPS];
#pragma unroll
        for (int k = 0; k < CDC_MAX_GROUPS; ++k)
            if (k < a.n) v[k] = a.src[k][r * a.ld_src[k] + c];
        float s = a.accumulate ? *d : 0.f;
#pragma unroll
        for (int k = 0; k < CDC_MAX_GROUPS; ++k)
            if (k < a.n) s = (k == 0 && !a.accumulate) ? v[0] : s + v[k];
        *d = s;
    }
}
extern "C" int cdc_add_n(const cdc_add_n_args* a, void* stream) {
    CDC_CHECK_ARG(a && a->dst && a->n > 0 && a->n <= CDC_MAX_GROUPS && a->rows >= 0 && a->cols > 0, CDC_E_BADARG, "add_n: bad argument");
    for (int k = 0; k < a->n; ++k) CDC_CHECK_ARG(a->src[k], CDC_E_BADARG, "add_n: source %d is NULL", k);
    if (a->rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(a->rows * a->cols, 256), 4096);
    hipLaunchKernelGGL(k_add_n, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    CDC_LAUNCH_CHECK("add_n");
    return 0;
}


// one launch instead of three device-to-device copies: a batch (ids int32 [B,F], labels int16 [B], tower index int64 [B] or
// NULL) into the buffers a replayed launch sequence reads (run.py:476-479 hands the step exactly these three tensors)
__global__ void __launch_bounds__(256) k_stage_batch(const int32_t* __restrict__ ids, const int16_t* __restrict__ y,
                                                     const int64_t* __restrict__ group, int32_t* __restrict__ ids_dst,
                                                     int16_t* __restrict__ y_dst, int64_t* __restrict__ group_dst, int64_t B, int32_t F,
                                                     const int32_t* __restrict__ field_dims, int32_t* __restrict__ alias_flag) {
    const int64_t n = B * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = ids[i];
        ids_dst[i] = id;
        if (field_dims && (id < 0 || id >= field_dims[i % F])) atomicMax(alias_flag, (int32_t)(i < 0x7ffffffe ? i + 1 : 0x7fffffff));
        if (i < B) {
            y_dst[i] = y[i];
            if (group) group_dst[i] = group[i];
        }
    }
}
extern "C" int cdc_stage_batch(const int32_t* ids, const int16_t* y, const int64_t* group, int32_t* ids_dst, int16_t* y_dst,
                               int64_t* group_dst, int64_t B, int32_t F, const int32_t* field_dims, int32_t* alias_flag, void* stream) {
    CDC_CHECK_ARG(ids && y && ids_dst && y_dst && B > 0 && F > 0 && (!group || group_dst) && (!field_dims || alias_flag), CDC_E_BADARG,
                  "stage_batch: bad argument");
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(B * F, 256), 2048);
    hipLaunchKernelGGL(k_stage_batch, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, y, group, ids_dst, y_dst, group_dst, B, F,
                       field_dims, alias_flag);
    CDC_LAUNCH_CHECK("stage_batch");
    return 0;
}
// the same for a step whose sort was (or is being) done a step ahead: this batch as above, the NEXT batch's ids into the buffer the
// look-ahead sort reads, and cdc_begin_step's work (the sort's first launch, which does it otherwise, is no longer first)
__global__ void __launch_bounds__(256) k_stage_batch_next(const int32_t* __restrict__ ids, const int16_t* __restrict__ y,
                                                          const int64_t* __restrict__ group, int32_t* __restrict__ ids_dst,
                                                          int16_t* __restrict__ y_dst, int64_t* __restrict__ group_dst, int64_t B, int32_t F,
                                                          const int32_t* __restrict__ next_ids, int32_t* __restrict__ next_dst,
                                                          int32_t* step_dev, double* acc, int32_t n_acc,
                                                          const int32_t* __restrict__ field_dims, int32_t* __restrict__ alias_flag) {
    if (blockIdx.x == 0 && step_dev) {
        if (threadIdx.x == 0) *step_dev += 1;
        if ((int)threadIdx.x < n_acc) acc[threadIdx.x] = 0.0;
    }
    const int64_t n = B * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = ids[i];
        ids_dst[i] = id;
        if (field_dims && (id < 0 || id >= field_dims[i % F])) atomicMax(alias_flag, (int32_t)(i < 0x7ffffffe ? i + 1 : 0x7fffffff));
        if (next_ids) next_dst[i] = next_ids[i];
        if (i < B) {
            y_dst[i] = y[i];
            if (group) group_dst[i] = group[i];
        }
    }
}
extern "C" int cdc_stage_batch_next(const int32_t* ids, const int16_t* y, const int64_t* group, int32_t* ids_dst, int16_t* y_dst,
                                    int64_t* group_dst, int64_t B, int32_t F, const int32_t* next_ids, int32_t* next_dst,
                                    int32_t* step_dev, double* accumulators, int32_t n_acc, const int32_t* field_dims, int32_t* alias_flag,
                                    void* stream) {
    CDC_CHECK_ARG(ids && y && ids_dst && y_dst && B > 0 && F > 0 && (!group || group_dst) && (!next_ids || next_dst) &&
                      (!field_dims || alias_flag), CDC_E_BADARG, "stage_batch_next: bad argument");
    CDC_CHECK_ARG(n_acc >= 0 && n_acc <= 64 && (n_acc == 0 || (accumulators && step_dev)), CDC_E_BADARG, "stage_batch_next: bad accumulators");
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(B * F, 256), 2048);
    hipLaunchKernelGGL(k_stage_batch_next, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, y, group, ids_dst, y_dst, group_dst, B, F,
                       next_ids, next_dst, step_dev, accumulators, n_acc, field_dims, alias_flag);
    CDC_LAUNCH_CHECK("stage_batch_next");
    return 0;
}
