// metrics.hip — evaluation metrics of the reference's test loop on the device.
//
// Stands in for run.py:684-711 (Run.test / evaluate_multi_domain): sklearn.metrics.roc_auc_score and log_loss over the
// whole evaluation set and per domain (pandas groupby).  The reference moves every batch's predictions to the host and
// computes there; here predictions stay on the GPU and one call returns all figures.
//
//   AUC  = Mann-Whitney U with mid-ranks (ties share the average rank) — what roc_auc_score's trapezoid over the ROC
//          curve equals.  Rank sums are accumulated as INTEGERS (twice the mid-rank), so the result does not depend on
//          the order of accumulation: (S2/2 - P(P+1)/2) / (P*N) is formed once, in double.
//   loss = -mean(log(y ? p : 1-p)) with sklearn 1.7's arithmetic: 1-p and the clip to [eps, 1-eps] in float32 (the dtype
//          of the predictions), logarithm and mean in double; one workgroup per domain sums its contiguous, sorted
//          segment in a fixed order.
//
// Every row is keyed twice — (its domain, score) and (pseudo-domain n_domain = "all rows", score) — and ONE radix sort
// (rocPRIM) of the 2n keys lays out every domain's rows, and the whole set, as contiguous score-ordered segments.
#include <algorithm>
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>

#define MET_THREADS 256
#define MET_LOSS_THREADS 1024

__device__ __forceinline__ uint32_t score_key(float p) {        // monotone float -> uint32 (-0.0 folded onto +0.0: a tie)
    if (p == 0.f) p = 0.f;
    const uint32_t u = __float_as_uint(p);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__global__ void __launch_bounds__(MET_THREADS) k_metric_keys(const float* __restrict__ pred, const int16_t* __restrict__ label,
                                                             const int32_t* __restrict__ domain, int64_t ld_domain, int64_t n,
                                                             int32_t n_domain, uint64_t* __restrict__ keys,
                                                             uint8_t* __restrict__ vals, int32_t* __restrict__ err) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = pred[i];
        int32_t d = domain ? domain[i * ld_domain] : 0;
        const int16_t y = label[i];
        if (err && (p != p || d < 0 || d >= n_domain || (y != 0 && y != 1))) {
            atomicMax(err, (int32_t)(i < 0x7ffffffe ? i + 1 : 0x7fffffff));
            d = d < 0 ? 0 : (d >= n_domain ? n_domain - 1 : d);
        }
        const uint32_t sk = score_key(p);
        keys[i] = ((uint64_t)(uint32_t)d << 32) | sk;
        keys[n + i] = ((uint64_t)(uint32_t)n_domain << 32) | sk;
        vals[i] = vals[n + i] = (uint8_t)(y != 0);
    }
}

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* a, int64_t lo, int64_t hi, uint64_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int64_t upper_bound_u64(const uint64_t* a, int64_t lo, int64_t hi, uint64_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// start[d] = first sorted position of (pseudo-)domain d, d in [0, n_domain + 1]; start[n_domain + 1] = 2n
__global__ void k_metric_starts(const uint64_t* __restrict__ keys, int64_t n2, int32_t n_seg, int64_t* __restrict__ start,
                                unsigned long long* __restrict__ s2, unsigned long long* __restrict__ npos) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d <= n_seg) start[d] = d == n_seg ? n2 : lower_bound_u64(keys, 0, n2, (uint64_t)(uint32_t)d << 32);
    if (d < n_seg) { s2[d] = 0ull; npos[d] = 0ull; }
}

// every positive row adds twice its mid-rank inside its segment: (first + last position of its score) + 2, 0-based -> 1-based
__global__ void __launch_bounds__(MET_THREADS) k_metric_ranks(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ vals,
                                                              int64_t n2, const int64_t* __restrict__ start,
                                                              unsigned long long* __restrict__ s2,
                                                              unsigned long long* __restrict__ npos) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        if (!vals[i]) continue;
        const uint64_t key = keys[i];
        const int d = (int)(key >> 32);
        const int64_t s0 = start[d], s1 = start[d + 1];
        const int64_t first = lower_bound_u64(keys, s0, i + 1, key);
        const int64_t last = upper_bound_u64(keys, i, s1, key) - 1;
        atomicAdd(&s2[d], (unsigned long long)((first - s0) + (last - s0) + 2));
        atomicAdd(&npos[d], 1ull);
    }
}

// one workgroup per segment: sum of the clipped log-loss terms in a fixed order (strided per thread, then a tree)
__global__ void __launch_bounds__(MET_LOSS_THREADS) k_metric_loss(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ vals,
                                                                  const int64_t* __restrict__ start, double* __restrict__ loss_sum) {
    __shared__ double part[MET_LOSS_THREADS];
    const int d = blockIdx.x, tid = threadIdx.x;
    const int64_t s0 = start[d], s1 = start[d + 1];
    // sklearn forms [1-p, p] and clips it in the predictions' dtype (float32); only log and mean are double
    const float eps = 1.1920928955078125e-07f, hi = 1.0f - eps;            // numpy.finfo(float32).eps
    double acc = 0.0;
    for (int64_t i = s0 + tid; i < s1; i += MET_LOSS_THREADS) {
        const float p = key_score((uint32_t)keys[i]);
        float c = vals[i] ? p : __fsub_rn(1.0f, p);
        c = c < eps ? eps : (c > hi ? hi : c);
        acc -= log((double)c);
    }
    part[tid] = acc;
    __syncthreads();
    for (int off = MET_LOSS_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) part[tid] += part[tid + off];
        __syncthreads();
    }
    if (tid == 0) loss_sum[d] = part[0];
}

__global__ void k_metric_final(const int64_t* __restrict__ start, const unsigned long long* __restrict__ s2,
                               const unsigned long long* __restrict__ npos, const double* __restrict__ loss_sum, int32_t n_seg,
                               double* __restrict__ out, int64_t* __restrict__ counts) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_seg) return;
    const int64_t rows = start[d + 1] - start[d];
    const int64_t P = (int64_t)npos[d], N = rows - P;
    counts[d] = rows;
    counts[n_seg + d] = P;
    double auc = __longlong_as_double(0x7ff8000000000000ll), loss = auc;        // NaN: run.py:699-704's ValueError branch
    if (rows > 0 && P > 0 && N > 0) {
        const double u = ((double)s2[d] - (double)P * (double)(P + 1)) * 0.5;   // both terms exact integers < 2^53
        auc = u / ((double)P * (double)N);
        loss = loss_sum[d] / (double)rows;
    }
    out[d] = auc;
    out[n_seg + d] = loss;
}

static int64_t align_up(int64_t v) { return (v + 255) & ~(int64_t)255; }

struct MetricLayout {
    int64_t keys_in, keys_out, vals_in, vals_out, start, s2, npos, loss, temp, temp_bytes, total;
};
static int metric_layout(int64_t n, int32_t n_domain, MetricLayout* L) {
    const int64_t n2 = 2 * n, seg = n_domain + 1;
    size_t temp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, temp_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint8_t*)nullptr,
                                             (uint8_t*)nullptr, (size_t)n2, 0, 64, (hipStream_t)0, false);
    if (e != hipSuccess) { cdc_set_error("eval_metrics: rocprim size query failed: %s", hipGetErrorString(e)); return (int)e; }
    int64_t off = 0;
    L->keys_in = off;  off += align_up(n2 * 8);
    L->keys_out = off; off += align_up(n2 * 8);
    L->vals_in = off;  off += align_up(n2);
    L->vals_out = off; off += align_up(n2);
    L->start = off;    off += align_up((seg + 1) * 8);
    L->s2 = off;       off += align_up(seg * 8);
    L->npos = off;     off += align_up(seg * 8);
    L->loss = off;     off += align_up(seg * 8);
    L->temp = off;     off += align_up((int64_t)temp_bytes);
    L->temp_bytes = (int64_t)temp_bytes;
    L->total = off;
    return 0;
}

extern "C" int64_t cdc_eval_workspace_bytes(int64_t n, int32_t n_domain) {
    if (n <= 0 || n_domain <= 0) return 0;
    MetricLayout L;
    if (metric_layout(n, n_domain, &L) != 0) return -1;
    return L.total;
}

extern "C" int cdc_eval_metrics(const float* pred, const int16_t* label, const int32_t* domain, int64_t ld_domain, int64_t n,
                                int32_t n_domain, double* out, int64_t* counts, int32_t* err_flag, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    CDC_CHECK_ARG(pred && label && out && counts && workspace, CDC_E_BADARG, "eval_metrics: null pointer");
    CDC_CHECK_ARG(n > 0 && n < (1ll << 31) && n_domain > 0 && n_domain < (1 << 20) && (domain || n_domain == 1) && ld_domain >= 0,
                  CDC_E_BADARG, "eval_metrics: bad sizes n=%ld n_domain=%d", (long)n, n_domain);
    MetricLayout L;
    int rc = metric_layout(n, n_domain, &L);
    if (rc != 0) return rc;
    CDC_CHECK_ARG(workspace_bytes >= L.total, CDC_E_BADARG, "eval_metrics: workspace %ld < %ld bytes", (long)workspace_bytes, (long)L.total);
    CDC_CHECK_ARG((((uintptr_t)workspace) & 255) == 0, CDC_E_BADARG, "eval_metrics: workspace must be 256-byte aligned");
    char* base = (char*)workspace;
    uint64_t* keys_in = (uint64_t*)(base + L.keys_in);
    uint64_t* keys_out = (uint64_t*)(base + L.keys_out);
    uint8_t* vals_in = (uint8_t*)(base + L.vals_in);
    uint8_t* vals_out = (uint8_t*)(base + L.vals_out);
    int64_t* start = (int64_t*)(base + L.start);
    unsigned long long* s2 = (unsigned long long*)(base + L.s2);
    unsigned long long* npos = (unsigned long long*)(base + L.npos);
    double* loss = (double*)(base + L.loss);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n2 = 2 * n;
    const int seg = n_domain + 1;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(n, MET_THREADS), 4096);
    hipLaunchKernelGGL(k_metric_keys, dim3(blocks), dim3(MET_THREADS), 0, st, pred, label, domain, ld_domain, n, n_domain, keys_in,
                       vals_in, err_flag);
    CDC_LAUNCH_CHECK("eval_metrics(keys)");
    size_t temp_bytes = (size_t)L.temp_bytes;
    // the sort only has to look at the bits a key can have: 32 score bits + the bits of n_domain
    int end_bit = 33;
    while (end_bit < 64 && ((uint64_t)n_domain >> (end_bit - 32)) != 0) ++end_bit;
    hipError_t e = rocprim::radix_sort_pairs(base + L.temp, temp_bytes, (const uint64_t*)keys_in, keys_out, (const uint8_t*)vals_in,
                                             vals_out, (size_t)n2, 0, end_bit, st, false);
    if (e != hipSuccess) { cdc_set_error("eval_metrics: radix sort failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(k_metric_starts, dim3((int)cdc_ceil_div(seg + 1, 64)), dim3(64), 0, st, keys_out, n2, seg, start, s2, npos);
    CDC_LAUNCH_CHECK("eval_metrics(starts)");
    blocks = (int)std::min<int64_t>(cdc_ceil_div(n2, MET_THREADS), 8192);
    hipLaunchKernelGGL(k_metric_ranks, dim3(blocks), dim3(MET_THREADS), 0, st, keys_out, vals_out, n2, start, s2, npos);
    CDC_LAUNCH_CHECK("eval_metrics(ranks)");
    hipLaunchKernelGGL(k_metric_loss, dim3(seg), dim3(MET_LOSS_THREADS), 0, st, keys_out, vals_out, start, loss);
    CDC_LAUNCH_CHECK("eval_metrics(loss)");
    hipLaunchKernelGGL(k_metric_final, dim3((int)cdc_ceil_div(seg, 64)), dim3(64), 0, st, start, s2, npos, loss, seg, out, counts);
    CDC_LAUNCH_CHECK("eval_metrics(final)");
    return 0;
}
