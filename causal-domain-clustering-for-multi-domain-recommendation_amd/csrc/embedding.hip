// embedding.hip — sparse id -> embedding row gather, per-field sort/dedupe, and the table optimiser
// (exact dense-Adam semantics of the reference, in dense and lazy-exact forms).
//
// Reference behaviour restated here:
//   model/layer.py:147-157  FeaturesEmbedding.forward  (offset add in int32, row lookup, flatten)
//   model/layer.py:31,96-112 + run.py:489,720-721       whole-table L2 + dense torch.optim.Adam
// HBM-bound byte work: no MFMA here; coalesced 16-B lanes, LDS only for the per-field sort.
#include "common.h"
#include "adam_dense.h"

// ------------------------------------------------------------------------------------------------
// gather
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_gather_fwd(const int32_t* __restrict__ ids, const int32_t* __restrict__ offsets,
                                                    const float* __restrict__ table, float* __restrict__ out,
                                                    __bf16* __restrict__ out_h, int64_t ld_out_h,
                                                    int32_t* __restrict__ idx_out, int32_t* __restrict__ err_flag,
                                                    int64_t n_pos, int32_t F, int32_t D, int64_t R) {
    CDC_PRIO_MAIN();
    const int chunks = D / VEC;                       // chunks of VEC floats per row
    const int64_t total = n_pos * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pos = i / chunks;
        const int c = (int)(i - pos * chunks);
        const int f = (int)(pos % F);
        // model/layer.py:152 — the sum is formed in x's dtype (int32): wraps like torch's int32 add
        const int32_t row = (int32_t)((uint32_t)ids[pos] + (uint32_t)offsets[f]);
        const bool ok = row >= 0 && (int64_t)row < R;
        if (c == 0 && idx_out) idx_out[pos] = ok ? row : -1;    // -1: skipped by every kernel that walks row lists
        if (!ok && c == 0 && err_flag) atomicMax(err_flag, (int32_t)(pos < 0x7ffffffe ? pos + 1 : 0x7fffffff));
        if (VEC == 4) {
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) val = *reinterpret_cast<const float4*>(table + (int64_t)row * D + c * 4);
            *reinterpret_cast<float4*>(out + pos * D + c * 4) = val;
            if (out_h) {                                      // bf16 shadow [B, ld_out_h]: position (b, f) at column f*D
                typedef __bf16 h4_t __attribute__((ext_vector_type(4)));
                h4_t h = {(__bf16)val.x, (__bf16)val.y, (__bf16)val.z, (__bf16)val.w};
                *reinterpret_cast<h4_t*>(out_h + (pos / F) * ld_out_h + (int64_t)f * D + c * 4) = h;
            }
        } else {
            const float v = ok ? table[(int64_t)row * D + c] : 0.f;
            out[pos * D + c] = v;
            if (out_h) out_h[(pos / F) * ld_out_h + (int64_t)f * D + c] = (__bf16)v;
        }
    }
}

extern "C" int cdc_embed_gather_fwd(const int32_t* ids, const int32_t* offsets, const float* table, float* out,
                                    int32_t* idx_out, int32_t* err_flag, int64_t B, int32_t F, int32_t D, int64_t R,
                                    void* stream) {
    return cdc_embed_gather_fwd_h(ids, offsets, table, out, nullptr, 0, idx_out, err_flag, B, F, D, R, stream);
}
extern "C" int cdc_embed_gather_fwd_h(const int32_t* ids, const int32_t* offsets, const float* table, float* out, void* out_h_,
                                      int64_t ld_out_h, int32_t* idx_out, int32_t* err_flag, int64_t B, int32_t F, int32_t D, int64_t R,
                                      void* stream) {
    __bf16* out_h = reinterpret_cast<__bf16*>(out_h_);
    CDC_CHECK_ARG(ids && offsets && table && out, CDC_E_BADARG, "embed_gather_fwd: null pointer");
    CDC_CHECK_ARG(!out_h || (ld_out_h >= (int64_t)F * D && (((uintptr_t)out_h) & 7) == 0 && ld_out_h % 4 == 0), CDC_E_BADARG,
                  "embed_gather_fwd: malformed bf16 shadow");
    CDC_CHECK_ARG(B >= 0 && F > 0 && D > 0 && R > 0, CDC_E_BADARG, "embed_gather_fwd: bad sizes B=%ld F=%d D=%d R=%ld",
                  (long)B, F, D, (long)R);
    if (B == 0) return 0;
    const int64_t n_pos = B * F;
    const bool vec = (D % 4 == 0) && (((uintptr_t)table | (uintptr_t)out) % 16 == 0);
    const int64_t total = n_pos * (vec ? D / 4 : D);
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(total, 256), 256 * 16);
    if (vec)
        hipLaunchKernelGGL(k_gather_fwd<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, offsets, table, out, out_h, ld_out_h,
                           idx_out, err_flag, n_pos, F, D, R);
    else
        hipLaunchKernelGGL(k_gather_fwd<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, offsets, table, out, out_h, ld_out_h,
                           idx_out, err_flag, n_pos, F, D, R);
    CDC_LAUNCH_CHECK("embed_gather_fwd");
    return 0;
}

// row indices only (the lazy table optimiser needs them BEFORE the gather to bring the rows up to date)
__global__ void __launch_bounds__(256) k_embed_index(const int32_t* __restrict__ ids, const int32_t* __restrict__ offsets,
                                                     int32_t* __restrict__ idx_out, int32_t* __restrict__ err_flag, int64_t n_pos,
                                                     int32_t F, int64_t R) {
    for (int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pos < n_pos; pos += (int64_t)gridDim.x * blockDim.x) {
        const int32_t row = (int32_t)((uint32_t)ids[pos] + (uint32_t)offsets[pos % F]);
        const bool ok = row >= 0 && (int64_t)row < R;
        if (!ok && err_flag) atomicMax(err_flag, (int32_t)(pos < 0x7ffffffe ? pos + 1 : 0x7fffffff));
        idx_out[pos] = ok ? row : -1;
    }
}
extern "C" int cdc_embed_index(const int32_t* ids, const int32_t* offsets, int32_t* idx_out, int32_t* err_flag, int64_t B, int32_t F,
                               int64_t R, void* stream) {
    CDC_CHECK_ARG(ids && offsets && idx_out && B > 0 && F > 0 && R > 0, CDC_E_BADARG, "embed_index: bad argument");
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(B * F, 256), 4096);
    hipLaunchKernelGGL(k_embed_index, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ids, offsets, idx_out, err_flag, B * F, F, R);
    CDC_LAUNCH_CHECK("embed_index");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-field sort + dedupe: one workgroup per field, bitonic sort of (row<<32 | b) in LDS
// ------------------------------------------------------------------------------------------------
#define SORT_THREADS 1024
#define SORT_CHUNK CDC_SORT_MAX_B        /* keys one workgroup sorts in LDS (16384 x 8 B = 128 KB) */

// bitonic sort of n_pad 64-bit keys in LDS (n_pad a power of two >= SORT_THREADS).
// A compare-exchange stage with distance j only moves keys inside aligned blocks of 2j keys, and those are handled by j
// consecutive (virtual) threads: for j <= 32 every block lives inside one wavefront, whose LDS operations execute in issue
// order — such stages need no workgroup barrier, only a wavefront-scope fence.  Of the 78 stages of a 4096-key sort 27
// keep their barrier (those with j >= 64, and the last stage before one).
__device__ __forceinline__ void lds_bitonic_sort(uint64_t* keys, int n_pad, int tid) {
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n_pad >> 1); t += SORT_THREADS) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j cleared
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            const int next_j = j > 1 ? (j >> 1) : k;                     // distance of the stage that reads these keys next
            if (j >= 64 || next_j >= 64) __syncthreads();
            else {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
}

// block-wide exclusive scan of one int per thread (SORT_THREADS threads): shuffles inside a wave, the 16 wave totals through LDS
__device__ __forceinline__ int block_exclusive_scan(int v, int32_t* wave_tot, int tid, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < SORT_THREADS / 64; ++w) {
        const int t = wave_tot[w];
        base += (w < wave) ? t : 0;
        all += t;
    }
    __syncthreads();                                                     // wave_tot may be reused by the caller
    total = all;
    return base + inc - v;
}

// head flags + block-wide exclusive scan over n sorted keys (LDS or global): unique rows, segment starts, permutation
__device__ __forceinline__ void dedupe_sorted(const uint64_t* keys, int n, int32_t* scan, int32_t* __restrict__ urow,
                                              int32_t* __restrict__ sst, int32_t* __restrict__ prm, int32_t* __restrict__ cnt_out,
                                              int tid) {
    const int per = (n + SORT_THREADS - 1) / SORT_THREADS;
    const int begin = tid * per;
    int local = 0;
    for (int i = begin; i < begin + per && i < n; ++i) {
        const bool head = (i == 0) || ((keys[i] >> 32) != (keys[i - 1] >> 32));
        local += head ? 1 : 0;
    }
    int total;
    int u = block_exclusive_scan(local, scan, tid, total);
    for (int i = begin; i < begin + per && i < n; ++i) {
        const uint64_t k = keys[i];
        const bool head = (i == 0) || ((k >> 32) != (keys[i - 1] >> 32));
        if (head) {
            urow[u] = (int32_t)(uint32_t)(k >> 32);
            sst[u] = i;
            ++u;
        }
        prm[i] = (int32_t)(uint32_t)k;
    }
    if (tid == 0) {
        sst[total] = n;
        *cnt_out = total;
    }
}

// What the sort kernels read: row indices [B,F], or (offsets != NULL) raw ids [B,F] that become rows by adding the field
// offsets (out-of-range -> -1, as cdc_embed_index writes them).  step != NULL: the launch also does cdc_begin_step's work
// (first kernel of a training step: ++*step, clear the accumulators) — two launches less per step.
struct SortSrc {
    const int32_t* idx;
    const int32_t* offsets;
    int64_t R;
    int32_t* step;
    double* acc;
    int32_t n_acc;
    int32_t* err_flag;      // offsets != NULL: 1 + flat position of an out-of-range id (atomicMax), as cdc_embed_index reports it
};
__device__ __forceinline__ uint32_t sort_row(const SortSrc& s, int64_t i, int f, int F) {
    const int32_t v = s.idx[i * F + f];
    if (!s.offsets) return (uint32_t)v;
    const int32_t row = (int32_t)((uint32_t)v + (uint32_t)s.offsets[f]);
    if (row >= 0 && (int64_t)row < s.R) return (uint32_t)row;
    if (s.err_flag) {
        const int64_t pos = i * F + f;
        atomicMax(s.err_flag, (int32_t)(pos < 0x7ffffffe ? pos + 1 : 0x7fffffff));
    }
    return 0xffffffffu;
}
__device__ __forceinline__ void sort_begin_step(const SortSrc& s, int tid) {
    if (s.step && blockIdx.x == 0 && blockIdx.y == 0) {
        if (tid == 0) *s.step += 1;
        if (tid < s.n_acc) s.acc[tid] = 0.0;
    }
}

// B <= SORT_CHUNK: sort + dedupe in one workgroup per field
__global__ void __launch_bounds__(SORT_THREADS) k_sort_dedupe(const SortSrc src, int32_t* __restrict__ uniq_row,
                                                              int32_t* __restrict__ seg_start, int32_t* __restrict__ perm,
                                                              int32_t* __restrict__ uniq_cnt, int32_t B, int32_t F,
                                                              int32_t n_pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);              // n_pad entries
    int32_t* scan = reinterpret_cast<int32_t*>(keys + n_pad);            // SORT_THREADS entries
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_pad; i += SORT_THREADS) {
        uint64_t k = ~0ull;                                               // padding sorts last
        if (i < B) k = ((uint64_t)sort_row(src, i, f, F) << 32) | (uint32_t)i;
        keys[i] = k;
    }
    sort_begin_step(src, tid);
    __syncthreads();
    lds_bitonic_sort(keys, n_pad, tid);
    dedupe_sorted(keys, B, scan, uniq_row + (int64_t)f * B, seg_start + (int64_t)f * (B + 1), perm + (int64_t)f * B, uniq_cnt + f, tid);
}

// Chunked path (a scratch buffer is given and B > 1024): one workgroup's LDS moves 16 bytes per key and stage, and a
// 4096-key sort has 78 stages — a single workgroup per field is bound by its CU's LDS bandwidth (45 us at B = 4096, 26 CUs
// busy).  So (1) chunks of `chunk` rows are sorted by separate workgroups (fewer stages, more CUs) and written out,
// (2) the sorted runs are merged by rank — a key's final position is the sum over the runs of the number of smaller keys;
// keys are unique, the batch row is part of them — and (3) dedupe runs from global memory.
__global__ void __launch_bounds__(SORT_THREADS) k_sort_chunk(const SortSrc src, uint64_t* __restrict__ runs, int32_t B,
                                                             int32_t F, int32_t chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);
    const int f = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    const int r0 = c * chunk;
    const int n = min(chunk, B - r0);
    for (int i = tid; i < chunk; i += SORT_THREADS) {
        uint64_t k = ~0ull;
        if (i < n) k = ((uint64_t)sort_row(src, r0 + i, f, F) << 32) | (uint32_t)(r0 + i);
        keys[i] = k;
    }
    sort_begin_step(src, tid);
    __syncthreads();
    lds_bitonic_sort(keys, chunk, tid);
    uint64_t* out = runs + (int64_t)f * B + r0;
    for (int i = tid; i < n; i += SORT_THREADS) out[i] = keys[i];
}
// P (a power of two >= n_runs, <= 16) neighbouring lanes share one key: lane q counts the smaller keys of run q with its own
// binary search (the searches of a key run side by side instead of one after the other), a butterfly adds the counts
__device__ __forceinline__ int group_sum(int v, int P) {
    for (int off = 1; off < P; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__global__ void __launch_bounds__(256) k_merge_runs(const uint64_t* __restrict__ runs, uint64_t* __restrict__ merged, int32_t B,
                                                    int32_t F, int32_t chunk, int32_t n_runs, int32_t P) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = t / P;
    const int q = (int)(t - item * P);
    const bool valid = item < (int64_t)F * B;
    const int f = valid ? (int)(item / B) : 0;
    const int i = valid ? (int)(item - (int64_t)f * B) : 0;
    const uint64_t* a = runs + (int64_t)f * B;
    const uint64_t key = a[i];
    const int r = i / chunk;
    int lo = 0;
    if (valid && q < n_runs) {
        if (q == r) lo = i - r * chunk;
        else {
            const uint64_t* other = a + (int64_t)q * chunk;
            int hi = min(chunk, B - q * chunk);                    // number of keys of run q that are smaller
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (other[mid] < key) lo = mid + 1; else hi = mid;
            }
        }
    }
    const int pos = group_sum(lo, P);
    if (valid && q == 0) merged[(int64_t)f * B + pos] = key;
}
// The same rank merge with the field's keys staged in LDS: one workgroup per (field, run) loads all runs of its field once
// (the probes of the binary searches then cost an LDS access instead of an L2 round trip each) and places its run's keys.
// IDX: the keys are the uint32 rows of an id batch [B,F] made of ascending runs (ties between runs go to the earlier run);
// otherwise they are the unique 64-bit (row, position) keys the chunk sort wrote.
template <typename K, bool IDX>
__global__ void __launch_bounds__(SORT_THREADS) k_merge_lds(const void* __restrict__ src, uint64_t* __restrict__ merged, int32_t B,
                                                            int32_t F, int32_t n_runs, int32_t run_len, int32_t P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    K* keys = reinterpret_cast<K*>(smem_raw);
    const int f = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
    if (IDX) {
        const int32_t* idx = static_cast<const int32_t*>(src);
        for (int i = tid; i < B; i += SORT_THREADS) keys[i] = (K)(uint32_t)idx[(int64_t)i * F + f];
    } else {
        const uint64_t* a = static_cast<const uint64_t*>(src) + (int64_t)f * B;
        for (int i = tid; i < B; i += SORT_THREADS) keys[i] = (K)a[i];
    }
    __syncthreads();
    const int base = r * run_len;
    const int n_r = min(run_len, B - base);
    const int rounds = (n_r * P + SORT_THREADS - 1) / SORT_THREADS;      // uniform: every lane takes part in the butterflies
    for (int it = 0; it < rounds; ++it) {
        const int t = it * SORT_THREADS + tid;
        const int item = t / P, q = t - item * P;
        const bool valid = item < n_r;
        const K key = keys[valid ? base + item : base];
        int lo = 0;
        if (valid && q < n_runs) {
            if (q == r) lo = item;
            else {
                const K* other = keys + q * run_len;
                int hi = min(run_len, B - q * run_len);
                if (IDX && q < r) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (other[mid] <= key) lo = mid + 1; else hi = mid; } }
                else              { while (lo < hi) { const int mid = (lo + hi) >> 1; if (other[mid] <  key) lo = mid + 1; else hi = mid; } }
            }
        }
        const int pos = group_sum(lo, P);
        if (valid && q == 0)
            merged[(int64_t)f * B + pos] = IDX ? (((uint64_t)key << 32) | (uint32_t)(base + item)) : (uint64_t)key;
    }
}
#define MERGE_LDS_MAX (144 * 1024)
__global__ void __launch_bounds__(SORT_THREADS) k_dedupe_merged(const uint64_t* __restrict__ merged, int32_t* __restrict__ uniq_row,
                                                                int32_t* __restrict__ seg_start, int32_t* __restrict__ perm,
                                                                int32_t* __restrict__ uniq_cnt, int32_t B, int32_t F) {
    __shared__ int32_t scan[SORT_THREADS];
    const int f = blockIdx.x;
    dedupe_sorted(merged + (int64_t)f * B, B, scan, uniq_row + (int64_t)f * B, seg_start + (int64_t)f * (B + 1), perm + (int64_t)f * B,
                  uniq_cnt + f, threadIdx.x);
}

static int sort_dedupe_launch(const SortSrc& src, int32_t* uniq_row, int32_t* seg_start, int32_t* perm, int32_t* uniq_cnt,
                              uint64_t* scratch, int64_t B, int32_t F, void* stream) {
    const int32_t* idx = src.idx;
    CDC_CHECK_ARG(idx && uniq_row && seg_start && perm && uniq_cnt, CDC_E_BADARG, "embed_sort_dedupe: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0, CDC_E_BADARG, "embed_sort_dedupe: bad sizes");
    CDC_CHECK_ARG(B <= CDC_SORT_MAX_ROWS, CDC_E_TOOBIG, "embed_sort_dedupe: B=%ld exceeds %d", (long)B, CDC_SORT_MAX_ROWS);
    CDC_CHECK_ARG(scratch || B <= SORT_CHUNK, CDC_E_BADARG, "embed_sort_dedupe: B > %d needs a scratch buffer of 2*F*B uint64", SORT_CHUNK);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_sort_dedupe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_sort_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_merge_lds<uint64_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { cdc_set_error("embed_sort_dedupe: cannot raise LDS limit: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (B <= 1024 || (!scratch && B <= SORT_CHUNK)) {
        int n_pad = SORT_THREADS;   // at least one key per thread keeps the chunking simple
        while (n_pad < B) n_pad <<= 1;
        const size_t lds = (size_t)n_pad * 8 + SORT_THREADS * 4;
        hipLaunchKernelGGL(k_sort_dedupe, dim3(F), dim3(SORT_THREADS), lds, st, src, uniq_row, seg_start, perm, uniq_cnt, (int32_t)B, F, n_pad);
        CDC_LAUNCH_CHECK("embed_sort_dedupe");
        return 0;
    }
    int chunk = 512;                                        // about eight runs (more workgroups, fewer stages each)
    while (chunk < SORT_CHUNK && (int64_t)chunk * 8 < B) chunk <<= 1;
    const int n_runs = (int)cdc_ceil_div(B, chunk);
    uint64_t* runs = scratch;
    uint64_t* merged = scratch + (int64_t)F * B;
    hipLaunchKernelGGL(k_sort_chunk, dim3(F, n_runs), dim3(SORT_THREADS), (size_t)chunk * 8, st, src, runs, (int32_t)B, F, chunk);
    CDC_LAUNCH_CHECK("embed_sort_chunk");
    int P = 1;
    while (P < n_runs) P <<= 1;
    CDC_CHECK_ARG(P <= 64, CDC_E_TOOBIG, "embed_sort_dedupe: %d runs", n_runs);
    if ((size_t)B * 8 <= MERGE_LDS_MAX) {
        hipLaunchKernelGGL((k_merge_lds<uint64_t, false>), dim3(F, n_runs), dim3(SORT_THREADS), (size_t)B * 8, st, runs, merged, (int32_t)B, F,
                           n_runs, chunk, P);
    } else {
        const int64_t blocks = cdc_ceil_div((int64_t)F * B * P, 256);
        hipLaunchKernelGGL(k_merge_runs, dim3((unsigned)blocks), dim3(256), 0, st, runs, merged, (int32_t)B, F, chunk, n_runs, P);
    }
    CDC_LAUNCH_CHECK("embed_merge_runs");
    hipLaunchKernelGGL(k_dedupe_merged, dim3(F), dim3(SORT_THREADS), 0, st, merged, uniq_row, seg_start, perm, uniq_cnt, (int32_t)B, F);
    CDC_LAUNCH_CHECK("embed_dedupe_merged");
    return 0;
}

extern "C" int cdc_embed_sort_dedupe(const int32_t* idx, int32_t* uniq_row, int32_t* seg_start, int32_t* perm,
                                     int32_t* uniq_cnt, uint64_t* scratch, int64_t B, int32_t F, void* stream) {
    const SortSrc src = {idx, nullptr, 0, nullptr, nullptr, 0, nullptr};
    return sort_dedupe_launch(src, uniq_row, seg_start, perm, uniq_cnt, scratch, B, F, stream);
}
extern "C" int cdc_embed_sort_dedupe_ids(const int32_t* ids, const int32_t* offsets, int64_t R, int32_t* step_dev, double* accumulators,
                                         int32_t n_acc, int32_t* err_flag, int32_t* uniq_row, int32_t* seg_start, int32_t* perm,
                                         int32_t* uniq_cnt, uint64_t* scratch, int64_t B, int32_t F, void* stream) {
    CDC_CHECK_ARG(offsets && R > 0 && n_acc >= 0 && n_acc <= 64 && (n_acc == 0 || accumulators) && (step_dev || n_acc == 0), CDC_E_BADARG,
                  "embed_sort_dedupe_ids: bad argument");
    const SortSrc src = {ids, offsets, R, step_dev, accumulators, n_acc, err_flag};
    return sort_dedupe_launch(src, uniq_row, seg_start, perm, uniq_cnt, scratch, B, F, stream);
}

// Same result as cdc_embed_sort_dedupe for a batch that already consists of n_runs runs of run_len rows, each ascending
// per field (as unsigned: -1 padding last) — the row lists an owner receives from the ranks.  No sort: every key's final
// position is the sum over the runs of the number of smaller keys (binary searches), then the usual dedupe.
__global__ void __launch_bounds__(256) k_merge_n(const int32_t* __restrict__ idx, uint64_t* __restrict__ merged, int32_t B, int32_t F,
                                                 int32_t n_runs, int32_t run_len, int32_t P) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t item = t / P;
    const int q = (int)(t - item * P);
    const bool valid = item < (int64_t)F * B;
    const int f = valid ? (int)(item / B) : 0;
    const int i = valid ? (int)(item - (int64_t)f * B) : 0;
    const uint32_t row = (uint32_t)idx[(int64_t)i * F + f];
    const int r = i / run_len;
    int lo = 0;
    if (valid && q < n_runs) {
        if (q == r) lo = i - r * run_len;
        else {
            const int32_t* col = idx + (int64_t)q * run_len * F + f;
            int hi = run_len;
            if (q < r) { while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)col[(int64_t)mid * F] <= row) lo = mid + 1; else hi = mid; } }
            else       { while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)col[(int64_t)mid * F] <  row) lo = mid + 1; else hi = mid; } }
        }
    }
    const int pos = group_sum(lo, P);
    if (valid && q == 0) merged[(int64_t)f * B + pos] = ((uint64_t)row << 32) | (uint32_t)i;
}
extern "C" int cdc_embed_merge_dedupe(const int32_t* idx, int32_t* uniq_row, int32_t* seg_start, int32_t* perm, int32_t* uniq_cnt,
                                      uint64_t* scratch, int64_t B, int32_t F, int32_t n_runs, void* stream) {
    CDC_CHECK_ARG(idx && uniq_row && seg_start && perm && uniq_cnt && scratch, CDC_E_BADARG, "embed_merge_dedupe: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && n_runs > 0 && B % n_runs == 0, CDC_E_BADARG, "embed_merge_dedupe: bad sizes");
    CDC_CHECK_ARG(B <= CDC_SORT_MAX_ROWS, CDC_E_TOOBIG, "embed_merge_dedupe: B=%ld exceeds %d", (long)B, CDC_SORT_MAX_ROWS);
    hipStream_t st = (hipStream_t)stream;
    int P = 1;
    while (P < n_runs) P <<= 1;
    CDC_CHECK_ARG(P <= 64, CDC_E_TOOBIG, "embed_merge_dedupe: %d runs", n_runs);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_merge_lds<uint32_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { cdc_set_error("embed_merge_dedupe: cannot raise LDS limit: %s", hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    if ((size_t)B * 4 <= MERGE_LDS_MAX) {
        hipLaunchKernelGGL((k_merge_lds<uint32_t, true>), dim3(F, n_runs), dim3(SORT_THREADS), (size_t)B * 4, st, idx, scratch, (int32_t)B, F,
                           n_runs, (int32_t)(B / n_runs), P);
    } else {
        const int64_t blocks = cdc_ceil_div((int64_t)F * B * P, 256);
        hipLaunchKernelGGL(k_merge_n, dim3((unsigned)blocks), dim3(256), 0, st, idx, scratch, (int32_t)B, F, n_runs, (int32_t)(B / n_runs), P);
    }
    CDC_LAUNCH_CHECK("embed_merge_n");
    hipLaunchKernelGGL(k_dedupe_merged, dim3(F), dim3(SORT_THREADS), 0, st, scratch, uniq_row, seg_start, perm, uniq_cnt, (int32_t)B, F);
    CDC_LAUNCH_CHECK("embed_dedupe_merged");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-row gradient: rowgrad[f, j, :] = sum over the segment of unique row j of d_out[b, f*D:(f+1)*D]
// A batch has ~F*B unique rows and nearly all of their segments hold one or two entries; a launch with a wave per row
// spends its time starting 100 K waves that each wait on three dependent loads.  So:
//  (1) k_segment_sum_direct: one THREAD per (unique row, 16-byte chunk) adds segments of fewer than SEG_DIRECT entries
//      straight from d_out through perm, all of a segment's loads in flight together;
//  (2) k_segment_sum_long: per field a few workgroups list the longer segments and give each a wave (sub-lanes split
//      segments of >= SEG_SPLIT entries into 64/D contiguous parts) or, from SEG_BLOCK entries on (a domain column: three
//      rows of a thousand entries each), the whole workgroup (256/D parts).
// Sums run in ascending batch order inside a segment / part, parts are combined in part order: segments shorter than
// SEG_SPLIT equal aten::embedding_dense_backward's CPU result to the last bit.
// ------------------------------------------------------------------------------------------------
#define SEG_SPLIT 64
#define SEG_DIRECT 8
#define SEG_BLOCK 512
#define SEG_LONG_BLOCKS 64
#define SEG_LIST_CAP (CDC_SORT_MAX_ROWS / SEG_LONG_BLOCKS)

// What happens to a finished per-row gradient: stored to rowgrad, or (w != NULL) consumed on the spot by the lazy table's
// Adam step t for that row — cdc_embed_lazy_update's arithmetic without the round trip through rowgrad and its launch.
struct SegSink {
    float* rowgrad;
    float* w; float* m; float* v;
    int32_t* last;
    const int32_t* uniq_row;
    AdamConsts c;
    float step_size, bc2s;
    int32_t t;
};
template <int VEC>
__device__ __forceinline__ void seg_finish(const SegSink& k, int64_t slot, int D, int d0, const float (&acc)[VEC]) {
    if (!k.w) {
        float* dst = k.rowgrad + slot * D + d0;
        if (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
        else dst[0] = acc[0];
        return;
    }
    const int64_t row = k.uniq_row[slot];
    if (row < 0) return;
    const int64_t e0 = row * D + d0;
    float wv[VEC], mv[VEC], vv[VEC];
    if (VEC == 4) {
        const float4 a4 = *reinterpret_cast<const float4*>(k.w + e0), b4 = *reinterpret_cast<const float4*>(k.m + e0),
                     c4 = *reinterpret_cast<const float4*>(k.v + e0);
        wv[0] = a4.x; wv[1 % VEC] = a4.y; wv[2 % VEC] = a4.z; wv[3 % VEC] = a4.w;
        mv[0] = b4.x; mv[1 % VEC] = b4.y; mv[2 % VEC] = b4.z; mv[3 % VEC] = b4.w;
        vv[0] = c4.x; vv[1 % VEC] = c4.y; vv[2 % VEC] = c4.z; vv[3 % VEC] = c4.w;
    } else {
        wv[0] = k.w[e0]; mv[0] = k.m[e0]; vv[0] = k.v[e0];
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) adam_elem(wv[q], mv[q], vv[q], acc[q], k.c, k.step_size, k.bc2s);
    if (VEC == 4) {
        *reinterpret_cast<float4*>(k.w + e0) = make_float4(wv[0], wv[1 % VEC], wv[2 % VEC], wv[3 % VEC]);
        *reinterpret_cast<float4*>(k.m + e0) = make_float4(mv[0], mv[1 % VEC], mv[2 % VEC], mv[3 % VEC]);
        *reinterpret_cast<float4*>(k.v + e0) = make_float4(vv[0], vv[1 % VEC], vv[2 % VEC], vv[3 % VEC]);
    } else {
        k.w[e0] = wv[0]; k.m[e0] = mv[0]; k.v[e0] = vv[0];
    }
    if (d0 == 0) k.last[row] = k.t;
}
__device__ __forceinline__ SegSink seg_sink_store(float* rowgrad) {
    SegSink k = {};
    k.rowgrad = rowgrad;
    return k;
}

// One thread per (unique row, 16-byte chunk): the segment's gradient rows are added straight from d_out through perm in
// segment order, eight entries' loads in flight per round.  max_len > 0: rows whose segment holds max_len entries or more
// are left to k_segment_sum_long.  uniq_row != NULL: rows < 0 (the -1 padding of an owner's received lists) are skipped.
template <int VEC>
__device__ __forceinline__ void seg_direct_body(int bid, int nblocks, const float* __restrict__ d_out, const int32_t* __restrict__ seg_start,
                                                const int32_t* __restrict__ perm, const int32_t* __restrict__ uniq_cnt,
                                                const int32_t* __restrict__ uniq_row, const SegSink& sink,
                                                int32_t B, int32_t F, int32_t D, int32_t max_len) {
    const int chunks = D / VEC;
    const int64_t total = (int64_t)F * B * chunks;
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < total; i += (int64_t)nblocks * blockDim.x) {
        const int c = (int)(i % chunks);
        const int64_t slot = i / chunks;
        const int f = (int)(slot / B);
        const int j = (int)(slot - (int64_t)f * B);
        if (j >= uniq_cnt[f]) continue;
        if (uniq_row && uniq_row[slot] < 0) continue;
        const int32_t* sst = seg_start + (int64_t)f * (B + 1);
        const int32_t* prm = perm + (int64_t)f * B;
        const int k0 = sst[j], k1 = sst[j + 1];
        if (max_len > 0 && k1 - k0 >= max_len) continue;
        float acc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
        for (int k = k0; k < k1; k += 8) {
            int p[8];
            float v[8][VEC];
#pragma unroll
            for (int q = 0; q < 8; ++q) p[q] = (k + q < k1) ? prm[k + q] : -1;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float* src = d_out + ((int64_t)(p[q] >= 0 ? p[q] : 0) * F + f) * D + c * VEC;
                if (p[q] >= 0) {
                    if (VEC == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(src);
                        v[q][0] = t.x; v[q][1] = t.y; v[q][2] = t.z; v[q][3] = t.w;
                    } else {
                        v[q][0] = src[0];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (p[q] >= 0) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = __fadd_rn(acc[e], v[q][e]);
                }
        }
        seg_finish<VEC>(sink, slot, D, c * VEC, acc);
    }
}
template <int VEC>
__global__ void __launch_bounds__(256) k_segment_sum_direct(const float* __restrict__ d_out, const int32_t* __restrict__ seg_start,
                                                            const int32_t* __restrict__ perm, const int32_t* __restrict__ uniq_cnt,
                                                            const int32_t* __restrict__ uniq_row, float* __restrict__ rowgrad,
                                                            int32_t B, int32_t F, int32_t D, int32_t max_len) {
    seg_direct_body<VEC>(blockIdx.x, gridDim.x, d_out, seg_start, perm, uniq_cnt, uniq_row, seg_sink_store(rowgrad), B, F, D, max_len);
}
extern "C" int cdc_embed_segment_sum_direct(const float* d_out, const int32_t* seg_start, const int32_t* perm,
                                            const int32_t* uniq_cnt, const int32_t* uniq_row, float* rowgrad, int64_t B, int32_t F,
                                            int32_t D, void* stream) {
    CDC_CHECK_ARG(d_out && seg_start && perm && uniq_cnt && rowgrad, CDC_E_BADARG, "embed_segment_sum_direct: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0 && B <= CDC_SORT_MAX_ROWS, CDC_E_BADARG, "embed_segment_sum_direct: bad sizes");
    const bool vec = (D % 4 == 0) && (((uintptr_t)d_out | (uintptr_t)rowgrad) % 16 == 0);
    int blocks = (int)std::min<int64_t>(cdc_ceil_div((int64_t)F * B * (vec ? D / 4 : D), 256), 8192);
    if (vec)
        hipLaunchKernelGGL(k_segment_sum_direct<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_out, seg_start, perm, uniq_cnt, uniq_row,
                           rowgrad, (int32_t)B, F, D, 0);
    else
        hipLaunchKernelGGL(k_segment_sum_direct<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_out, seg_start, perm, uniq_cnt, uniq_row,
                           rowgrad, (int32_t)B, F, D, 0);
    CDC_LAUNCH_CHECK("embed_segment_sum_direct");
    return 0;
}

// sum of d_out rows perm[k], k in [begin, end), column d of field f — ascending k, BATCH loads in flight
template <int BATCH = 16>
__device__ __forceinline__ float seg_serial_sum(const float* __restrict__ d_out, const int32_t* __restrict__ prm, int begin, int end,
                                                int F, int D, int f, int d) {
    float acc = 0.f;
    for (int k = begin; k < end; k += BATCH) {
        int p[BATCH];
        float x[BATCH];
#pragma unroll
        for (int q = 0; q < BATCH; ++q) p[q] = (k + q < end) ? prm[k + q] : -1;
#pragma unroll
        for (int q = 0; q < BATCH; ++q) x[q] = p[q] >= 0 ? d_out[((int64_t)p[q] * F + f) * D + d] : 0.f;
#pragma unroll
        for (int q = 0; q < BATCH; ++q)
            if (p[q] >= 0) acc = __fadd_rn(acc, x[q]);
    }
    return acc;
}
template <int VEC>
__device__ __forceinline__ void seg_long_body(int f, int by, const float* __restrict__ d_out, const int32_t* __restrict__ seg_start,
                                              const int32_t* __restrict__ perm, const int32_t* __restrict__ uniq_cnt,
                                              const SegSink& sink, int32_t B, int32_t F, int32_t D, int32_t subs,
                                              int32_t min_len) {
    __shared__ int32_t list[SEG_LIST_CAP];
    __shared__ int32_t n_list;
    __shared__ __attribute__((aligned(16))) float parts[256 * VEC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = uniq_cnt[f];
    const int32_t* sst = seg_start + (int64_t)f * (B + 1);
    const int32_t* prm = perm + (int64_t)f * B;
    if (tid == 0) n_list = 0;
    __syncthreads();
    for (int j = by + SEG_LONG_BLOCKS * tid; j < n; j += SEG_LONG_BLOCKS * 256)
        if (sst[j + 1] - sst[j] >= min_len) list[atomicAdd(&n_list, 1)] = j;       // order is irrelevant: rows are independent
    __syncthreads();
    const int count = n_list;
    if (count == 0) return;
    const bool block_rows = subs > 1;                                  // whole-workgroup treatment needs D | 64
    // ---- a wave per listed row -----------------------------------------------------------------
    for (int i = wave; i < count; i += 4) {
        const int j = list[i];
        const int k0 = sst[j], len = sst[j + 1] - k0;
        if (block_rows && len >= SEG_BLOCK) continue;
        const int64_t slot = (int64_t)f * B + j;
        if (subs <= 1) {                                               // general D: lanes stride over d, serial ascending sum
            for (int d = lane; d < D; d += 64) {
                const float one[1] = {seg_serial_sum(d_out, prm + k0, 0, len, F, D, f, d)};
                seg_finish<1>(sink, slot, D, d, one);
            }
            continue;
        }
        const int sub = lane / D, d = lane - sub * D;
        const bool split = len >= SEG_SPLIT;
        int begin = 0, end = (sub == 0) ? len : 0;
        if (split) {
            const int q = (len + subs - 1) / subs;
            begin = min(sub * q, len);
            end = min(begin + q, len);
        }
        float acc = seg_serial_sum(d_out, prm + k0, begin, end, F, D, f, d);
        if (split) {
            float total = __shfl(acc, d, 64);                          // part 0, then the others in order
            for (int s2 = 1; s2 < subs; ++s2) total = __fadd_rn(total, __shfl(acc, s2 * D + d, 64));
            acc = total;
        }
        if (sub == 0) {
            const float one[1] = {acc};
            seg_finish<1>(sink, slot, D, d, one);
        }
    }
    if (!block_rows) return;
    // ---- the workgroup per very long row: a thread owns VEC consecutive columns, 256 / (D / VEC) parts ----------
    const int chunks = D / VEC;
    const int P = 256 / chunks;
    const int part = tid / chunks, d = (tid - part * chunks) * VEC;
    for (int i = 0; i < count; ++i) {
        const int j = list[i];
        const int k0 = sst[j], len = sst[j + 1] - k0;
        if (len < SEG_BLOCK) continue;                                 // uniform over the workgroup
        const int q = (len + P - 1) / P;
        const int begin = min(part * q, len), end = min(begin + q, len);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        if (part < P) {
            constexpr int NB = VEC == 4 ? 16 : 32;                     // loads in flight per round (register budget)
            for (int k = begin; k < end; k += NB) {
                int pr[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) pr[u] = (k + u < end) ? prm[k0 + k + u] : -1;
                float x[NB][VEC];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (pr[u] >= 0) {
                        const float* src = d_out + ((int64_t)pr[u] * F + f) * D + d;
                        if (VEC == 4) {
                            const float4 t4 = *reinterpret_cast<const float4*>(src);
                            x[u][0] = t4.x; x[u][1 % VEC] = t4.y; x[u][2 % VEC] = t4.z; x[u][3 % VEC] = t4.w;
                        } else {
                            x[u][0] = src[0];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (pr[u] >= 0) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[e] = __fadd_rn(acc[e], x[u][e]);
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) parts[tid * VEC + e] = acc[e];
        __syncthreads();
        if (tid < chunks) {
            float total[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) total[e] = parts[tid * VEC + e];
            for (int p2 = 1; p2 < P; ++p2)
#pragma unroll
                for (int e = 0; e < VEC; ++e) total[e] = __fadd_rn(total[e], parts[(p2 * chunks + tid) * VEC + e]);
            seg_finish<VEC>(sink, (int64_t)f * B + j, D, tid * VEC, total);
        }
        __syncthreads();
    }
}

// one launch for both: the first F * SEG_LONG_BLOCKS workgroups take the long segments (they start first and the few of
// them that find work run longest), the others the short ones
template <int VEC>
__global__ void __launch_bounds__(256) k_segment_sum(const float* __restrict__ d_out, const int32_t* __restrict__ seg_start,
                                                     const int32_t* __restrict__ perm, const int32_t* __restrict__ uniq_cnt,
                                                     SegSink sink, cdc_adam_hp hp, const int32_t* __restrict__ step_dev, int32_t B,
                                                     int32_t F, int32_t D, int32_t subs) {
    if (sink.w) {                                                      // the rows' Adam step t happens right here
        sink.c = make_consts(hp);
        sink.t = *step_dev;
        step_scalars_at(hp.step_scalars, hp.n_scalars, sink.t, sink.step_size, sink.bc2s);
    }
    const int n_long = F * SEG_LONG_BLOCKS;
    if ((int)blockIdx.x < n_long)
        seg_long_body<VEC>(blockIdx.x / SEG_LONG_BLOCKS, blockIdx.x % SEG_LONG_BLOCKS, d_out, seg_start, perm, uniq_cnt, sink, B, F, D,
                           subs, SEG_DIRECT);
    else
        seg_direct_body<VEC>(blockIdx.x - n_long, gridDim.x - n_long, d_out, seg_start, perm, uniq_cnt, nullptr, sink, B, F, D, SEG_DIRECT);
}
// ... and the dense parameters' Adam step in the same launch (cdc_embed_segsum_lazy_update_dense): the two updates that end a training
// step touch disjoint memory (the step's table rows with their gradient sums; the dense parameters with theirs) and each alone is a
// launch of a few hundred to two thousand short-lived workgroups living on memory latency.  Workgroups [n_long, n_long + n_dense)
// take one ADAM_CHUNK of one dense tensor each (csrc/adam_dense.h, in two passes so that the launch keeps four waves per SIMD for
// the row gathers), the others are k_segment_sum's.
template <int VEC>
__global__ void __launch_bounds__(256, 4) k_segment_sum_dense(const float* __restrict__ d_out, const int32_t* __restrict__ seg_start,
                                                           const int32_t* __restrict__ perm, const int32_t* __restrict__ uniq_cnt,
                                                           SegSink sink, cdc_adam_hp hp, const int32_t* __restrict__ step_dev, int32_t B,
                                                           int32_t F, int32_t D, int32_t subs, const AdamHdr h,
                                                           const cdc_adam_tensor* __restrict__ tab, const int32_t* __restrict__ wg_tensor,
                                                           const int32_t* __restrict__ wg_chunk, int32_t n_dense, int32_t short_only) {
    const int n_long = short_only ? 0 : F * SEG_LONG_BLOCKS;          // short_only: an owner's merged row lists (k_segment_sum_short_sink's case)
    const int b = (int)blockIdx.x;
    if (b >= n_long && b < n_long + n_dense) {
        const int ti = __builtin_amdgcn_readfirstlane(wg_tensor[b - n_long]);
        const int chunk = __builtin_amdgcn_readfirstlane(wg_chunk[b - n_long]);
        const cdc_adam_tensor T = tab[ti];
        adam_chunk<ADAM_CHUNK / 2>(h, T, chunk, b == n_long);
        return;
    }
    sink.c = make_consts(hp);
    sink.t = *step_dev;
    step_scalars_at(hp.step_scalars, hp.n_scalars, sink.t, sink.step_size, sink.bc2s);
    if (b < n_long)
        seg_long_body<VEC>(b / SEG_LONG_BLOCKS, b % SEG_LONG_BLOCKS, d_out, seg_start, perm, uniq_cnt, sink, B, F, D, subs, SEG_DIRECT);
    else
        seg_direct_body<VEC>(b - n_long - n_dense, (int)gridDim.x - n_long - n_dense, d_out, seg_start, perm, uniq_cnt,
                             short_only ? sink.uniq_row : nullptr, sink, B, F, D, short_only ? 0 : SEG_DIRECT);
}
// short segments only (an owner's merged row lists), same sink
template <int VEC>
__global__ void __launch_bounds__(256) k_segment_sum_short_sink(const float* __restrict__ d_out, const int32_t* __restrict__ seg_start,
                                                                const int32_t* __restrict__ perm, const int32_t* __restrict__ uniq_cnt,
                                                                SegSink sink, cdc_adam_hp hp, const int32_t* __restrict__ step_dev,
                                                                int32_t B, int32_t F, int32_t D) {
    sink.c = make_consts(hp);
    sink.t = *step_dev;
    step_scalars_at(hp.step_scalars, hp.n_scalars, sink.t, sink.step_size, sink.bc2s);
    seg_direct_body<VEC>(blockIdx.x, gridDim.x, d_out, seg_start, perm, uniq_cnt, sink.uniq_row, sink, B, F, D, 0);
}

extern "C" int cdc_embed_segment_sum(const float* d_out, const int32_t* seg_start, const int32_t* perm,
                                     const int32_t* uniq_cnt, float* sorted_scratch, float* rowgrad, int64_t B, int32_t F,
                                     int32_t D, void* stream) {
    (void)sorted_scratch;                                              // kept in the signature; no sorted copy is made any more
    CDC_CHECK_ARG(d_out && seg_start && perm && uniq_cnt && rowgrad, CDC_E_BADARG, "embed_segment_sum: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0 && B <= CDC_SORT_MAX_ROWS, CDC_E_BADARG, "embed_segment_sum: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (D % 4 == 0) && (((uintptr_t)d_out | (uintptr_t)rowgrad) % 16 == 0);
    const int blocks = (int)std::min<int64_t>(cdc_ceil_div((int64_t)F * B * (vec ? D / 4 : D), 256), 8192) + F * SEG_LONG_BLOCKS;
    const int subs = (D <= 64 && 64 % D == 0) ? 64 / D : 1;
    SegSink sink = {};
    sink.rowgrad = rowgrad;
    const cdc_adam_hp no_hp = {};
    if (vec) hipLaunchKernelGGL(k_segment_sum<4>, dim3(blocks), dim3(256), 0, st, d_out, seg_start, perm, uniq_cnt, sink, no_hp, nullptr, (int32_t)B, F, D, subs);
    else     hipLaunchKernelGGL(k_segment_sum<1>, dim3(blocks), dim3(256), 0, st, d_out, seg_start, perm, uniq_cnt, sink, no_hp, nullptr, (int32_t)B, F, D, subs);
    CDC_LAUNCH_CHECK("embed_segment_sum");
    return 0;
}

// cdc_embed_segment_sum + cdc_embed_lazy_update in one launch: a row's summed gradient is used for its Adam step t where it is
// formed.  short_only: every segment is short (an owner's merged lists; rows < 0 skipped) — the direct kernel alone.
extern "C" int cdc_embed_segsum_lazy_update(const float* d_out, const int32_t* seg_start, const int32_t* perm, const int32_t* uniq_cnt,
                                            const int32_t* uniq_row, float* w, float* m, float* v, int32_t* last, cdc_adam_hp hp,
                                            const int32_t* step_dev, int64_t B, int32_t F, int32_t D, int32_t short_only, void* stream) {
    CDC_CHECK_ARG(d_out && seg_start && perm && uniq_cnt && uniq_row && w && m && v && last && step_dev && hp.step_scalars &&
                      hp.n_scalars > 0, CDC_E_BADARG, "embed_segsum_lazy_update: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0 && B <= CDC_SORT_MAX_ROWS, CDC_E_BADARG, "embed_segsum_lazy_update: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (D % 4 == 0) && ((((uintptr_t)d_out | (uintptr_t)w | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    SegSink sink = {};
    sink.w = w; sink.m = m; sink.v = v; sink.last = last; sink.uniq_row = uniq_row;
    const int direct_blocks = (int)std::min<int64_t>(cdc_ceil_div((int64_t)F * B * (vec ? D / 4 : D), 256), 8192);
    if (short_only) {
        if (vec) hipLaunchKernelGGL(k_segment_sum_short_sink<4>, dim3(direct_blocks), dim3(256), 0, st, d_out, seg_start, perm, uniq_cnt, sink, hp, step_dev, (int32_t)B, F, D);
        else     hipLaunchKernelGGL(k_segment_sum_short_sink<1>, dim3(direct_blocks), dim3(256), 0, st, d_out, seg_start, perm, uniq_cnt, sink, hp, step_dev, (int32_t)B, F, D);
    } else {
        const int blocks = direct_blocks + F * SEG_LONG_BLOCKS;
        const int subs = (D <= 64 && 64 % D == 0) ? 64 / D : 1;
        if (vec) hipLaunchKernelGGL(k_segment_sum<4>, dim3(blocks), dim3(256), 0, st, d_out, seg_start, perm, uniq_cnt, sink, hp, step_dev, (int32_t)B, F, D, subs);
        else     hipLaunchKernelGGL(k_segment_sum<1>, dim3(blocks), dim3(256), 0, st, d_out, seg_start, perm, uniq_cnt, sink, hp, step_dev, (int32_t)B, F, D, subs);
    }
    CDC_LAUNCH_CHECK("embed_segsum_lazy_update");
    return 0;
}

// cdc_embed_segsum_lazy_update (all segment lengths) and cdc_adam_multi_table in ONE launch: see k_segment_sum_dense
extern "C" int cdc_embed_segsum_lazy_update_dense(const float* d_out, const int32_t* seg_start, const int32_t* perm, const int32_t* uniq_cnt,
                                                  const int32_t* uniq_row, float* w, float* m, float* v, int32_t* last, cdc_adam_hp hp,
                                                  const int32_t* step_dev, int64_t B, int32_t F, int32_t D, int32_t short_only,
                                                  const cdc_adam_args* dense, const cdc_adam_tensor* tensors_dev, const int32_t* wg_tensor_dev,
                                                  const int32_t* wg_chunk_dev, int32_t n_dense_workgroups, void* stream) {
    CDC_CHECK_ARG(d_out && seg_start && perm && uniq_cnt && uniq_row && w && m && v && last && step_dev && hp.step_scalars &&
                      hp.n_scalars > 0, CDC_E_BADARG, "embed_segsum_lazy_update_dense: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0 && B <= CDC_SORT_MAX_ROWS, CDC_E_BADARG, "embed_segsum_lazy_update_dense: bad sizes");
    CDC_CHECK_ARG(dense && tensors_dev && wg_tensor_dev && wg_chunk_dev && n_dense_workgroups > 0 && n_dense_workgroups < (1 << 24) &&
                      dense->step_dev && dense->step_scalars && dense->n_scalars > 0, CDC_E_BADARG,
                  "embed_segsum_lazy_update_dense: the dense parameters' descriptor table (as for cdc_adam_multi_table) is missing");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (D % 4 == 0) && ((((uintptr_t)d_out | (uintptr_t)w | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    SegSink sink = {};
    sink.w = w; sink.m = m; sink.v = v; sink.last = last; sink.uniq_row = uniq_row;
    const int direct_blocks = (int)std::min<int64_t>(cdc_ceil_div((int64_t)F * B * (vec ? D / 4 : D), 256), 8192);
    const int blocks = direct_blocks + (short_only ? 0 : F * SEG_LONG_BLOCKS) + n_dense_workgroups;
    const int subs = (D <= 64 && 64 % D == 0) ? 64 / D : 1;
    const AdamHdr h = {dense->lerp_w, dense->beta2, dense->one_minus_beta2, dense->eps, dense->weight_decay, dense->grad_scale,
                       dense->step_scalars, dense->n_scalars, dense->step_dev, dense->reg_sum, dense->reg_seed};
    if (vec) hipLaunchKernelGGL(k_segment_sum_dense<4>, dim3(blocks), dim3(256), 0, st, d_out, seg_start, perm, uniq_cnt, sink, hp, step_dev, (int32_t)B, F, D, subs,
                                h, tensors_dev, wg_tensor_dev, wg_chunk_dev, n_dense_workgroups, short_only);
    else     hipLaunchKernelGGL(k_segment_sum_dense<1>, dim3(blocks), dim3(256), 0, st, d_out, seg_start, perm, uniq_cnt, sink, hp, step_dev, (int32_t)B, F, D, subs,
                                h, tensors_dev, wg_tensor_dev, wg_chunk_dev, n_dense_workgroups, short_only);
    CDC_LAUNCH_CHECK("embed_segsum_lazy_update_dense");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// dense gradient (drop-in path: feeds torch.optim.Adam like aten::embedding_dense_backward)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_grad_dense(const float* __restrict__ rowgrad, const int32_t* __restrict__ uniq_row,
                                                    const int32_t* __restrict__ uniq_cnt, float* __restrict__ grad,
                                                    int32_t B, int32_t F, int32_t D, int64_t R) {
    const int64_t total = (int64_t)F * B * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int64_t slot = i / D;
        const int f = (int)(slot / B);
        const int j = (int)(slot - (int64_t)f * B);
        if (j >= uniq_cnt[f]) continue;
        const int32_t row = uniq_row[slot];
        if (row < 0 || (int64_t)row >= R) continue;
        grad[(int64_t)row * D + d] += rowgrad[i];
    }
}

extern "C" int cdc_embed_grad_dense(const float* rowgrad, const int32_t* uniq_row, const int32_t* uniq_cnt, float* grad,
                                    int64_t B, int32_t F, int32_t D, int64_t R, void* stream) {
    CDC_CHECK_ARG(rowgrad && uniq_row && uniq_cnt && grad, CDC_E_BADARG, "embed_grad_dense: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0 && B <= CDC_SORT_MAX_ROWS, CDC_E_BADARG, "embed_grad_dense: bad sizes");
    const int64_t total = (int64_t)F * B * D;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(total, 256), 256 * 16);
    hipLaunchKernelGGL(k_grad_dense, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowgrad, uniq_row, uniq_cnt, grad,
                       (int32_t)B, F, D, R);
    CDC_LAUNCH_CHECK("embed_grad_dense");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// table optimiser, dense form
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_adam_touched(const float* __restrict__ rowgrad, const int32_t* __restrict__ uniq_row,
                                                      const int32_t* __restrict__ uniq_cnt, const float* __restrict__ w,
                                                      const float* __restrict__ m, const float* __restrict__ v,
                                                      float* __restrict__ side, cdc_adam_hp hp,
                                                      const int32_t* __restrict__ step_dev, int32_t B, int32_t F, int32_t D) {
    const AdamConsts c = make_consts(hp);
    float step_size, bc2s;
    step_scalars_at(hp.step_scalars, hp.n_scalars, *step_dev, step_size, bc2s);
    const int64_t total = (int64_t)F * B * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int64_t slot = i / D;
        const int f = (int)(slot / B);
        const int j = (int)(slot - (int64_t)f * B);
        if (j >= uniq_cnt[f]) continue;
        const int64_t row = uniq_row[slot];
        if (row < 0) continue;
        float wv = w[row * D + d], mv = m[row * D + d], vv = v[row * D + d];
        adam_elem(wv, mv, vv, rowgrad[i], c, step_size, bc2s);
        float* s = side + slot * 3 * D;
        s[d] = wv; s[D + d] = mv; s[2 * D + d] = vv;
    }
}

extern "C" int cdc_embed_adam_touched(const float* rowgrad, const int32_t* uniq_row, const int32_t* uniq_cnt, const float* w,
                                      const float* m, const float* v, float* side, cdc_adam_hp hp, const int32_t* step_dev,
                                      int64_t B, int32_t F, int32_t D, void* stream) {
    CDC_CHECK_ARG(rowgrad && uniq_row && uniq_cnt && w && m && v && side && step_dev && hp.step_scalars, CDC_E_BADARG,
                  "embed_adam_touched: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0 && B <= CDC_SORT_MAX_ROWS && hp.n_scalars > 0, CDC_E_BADARG, "embed_adam_touched: bad sizes");
    const int64_t total = (int64_t)F * B * D;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(total, 256), 256 * 16);
    hipLaunchKernelGGL(k_adam_touched, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowgrad, uniq_row, uniq_cnt, w, m, v, side,
                       hp, step_dev, (int32_t)B, F, D);
    CDC_LAUNCH_CHECK("embed_adam_touched");
    return 0;
}

// One streaming pass over the whole table: 12 B read + 12 B written per element.
__global__ void __launch_bounds__(256) k_adam_dense_pass(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                         int64_t n_vec4, int64_t n_elems, cdc_adam_hp hp,
                                                         const int32_t* __restrict__ step_dev, double* __restrict__ reg_sum) {
    const AdamConsts c = make_consts(hp);
    float step_size, bc2s;
    step_scalars_at(hp.step_scalars, hp.n_scalars, *step_dev, step_size, bc2s);
    double sq = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec4; i += stride) {
        float4 wv = reinterpret_cast<float4*>(w)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        sq += (double)(wv.x * wv.x + wv.y * wv.y) + (double)(wv.z * wv.z + wv.w * wv.w);
        adam_elem(wv.x, mv.x, vv.x, 0.f, c, step_size, bc2s);
        adam_elem(wv.y, mv.y, vv.y, 0.f, c, step_size, bc2s);
        adam_elem(wv.z, mv.z, vv.z, 0.f, c, step_size, bc2s);
        adam_elem(wv.w, mv.w, vv.w, 0.f, c, step_size, bc2s);
        reinterpret_cast<float4*>(w)[i] = wv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // scalar tail (n_elems not a multiple of 4)
    for (int64_t i = n_vec4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += stride) {
        float wv = w[i], mv = m[i], vv = v[i];
        sq += (double)(wv * wv);
        adam_elem(wv, mv, vv, 0.f, c, step_size, bc2s);
        w[i] = wv; m[i] = mv; v[i] = vv;
    }
    if (reg_sum) {
        __shared__ double part[4];
        sq = wave_sum_d(sq);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(reg_sum, part[0] + part[1] + part[2] + part[3]);
    }
}

extern "C" int cdc_embed_adam_dense_pass(float* w, float* m, float* v, int64_t n_elems, cdc_adam_hp hp,
                                         const int32_t* step_dev, double* reg_sum, void* stream) {
    CDC_CHECK_ARG(w && m && v && step_dev && hp.step_scalars && hp.n_scalars > 0, CDC_E_BADARG, "embed_adam_dense_pass: null pointer");
    CDC_CHECK_ARG(n_elems > 0, CDC_E_BADARG, "embed_adam_dense_pass: bad size");
    CDC_CHECK_ARG((((uintptr_t)w | (uintptr_t)m | (uintptr_t)v) % 16) == 0, CDC_E_ALIGN, "embed_adam_dense_pass: buffers must be 16-byte aligned");
    const int64_t n_vec4 = n_elems / 4;
    int blocks = (int)std::min<int64_t>(std::max<int64_t>(cdc_ceil_div(n_vec4, 256), 1), 256 * 8);
    hipLaunchKernelGGL(k_adam_dense_pass, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, m, v, n_vec4, n_elems, hp,
                       step_dev, reg_sum);
    CDC_LAUNCH_CHECK("embed_adam_dense_pass");
    return 0;
}

__global__ void __launch_bounds__(256) k_adam_patch(const float* __restrict__ side, const int32_t* __restrict__ uniq_row,
                                                    const int32_t* __restrict__ uniq_cnt, float* __restrict__ w,
                                                    float* __restrict__ m, float* __restrict__ v, int32_t B, int32_t F,
                                                    int32_t D) {
    const int64_t total = (int64_t)F * B * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int64_t slot = i / D;
        const int f = (int)(slot / B);
        const int j = (int)(slot - (int64_t)f * B);
        if (j >= uniq_cnt[f]) continue;
        const int64_t row = uniq_row[(int64_t)f * B + j];
        if (row < 0) continue;
        const float* s = side + slot * 3 * D;
        w[row * D + d] = s[d];
        m[row * D + d] = s[D + d];
        v[row * D + d] = s[2 * D + d];
    }
}

extern "C" int cdc_embed_adam_patch(const float* side, const int32_t* uniq_row, const int32_t* uniq_cnt, float* w,
                                    float* m, float* v, int64_t B, int32_t F, int32_t D, void* stream) {
    CDC_CHECK_ARG(side && uniq_row && uniq_cnt && w && m && v, CDC_E_BADARG, "embed_adam_patch: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0, CDC_E_BADARG, "embed_adam_patch: bad sizes");
    const int64_t total = (int64_t)F * B * D;
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(total, 256), 256 * 16);
    hipLaunchKernelGGL(k_adam_patch, dim3(blocks), dim3(256), 0, (hipStream_t)stream, side, uniq_row, uniq_cnt, w, m, v,
                       (int32_t)B, F, D);
    CDC_LAUNCH_CHECK("embed_adam_patch");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// table optimiser, lazy-exact form: a row's (w,m,v) are valid for step last[row]; the untouched-row
// recurrence (g = 2*l2*w + wd*w) is replayed on demand.  Every element-step is computed exactly once
// with the same adam_elem as the dense pass, so both forms give identical bits.
// ------------------------------------------------------------------------------------------------
template <bool FAST, int VEC>
__global__ void __launch_bounds__(256) k_lazy_catchup(const int32_t* __restrict__ uniq_row, const int32_t* __restrict__ uniq_cnt,
                                                      float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                      int32_t* __restrict__ last, cdc_adam_hp hp,
                                                      const int32_t* __restrict__ step_dev, int32_t B, int32_t F, int32_t D,
                                                      int32_t mark) {
    const AdamConsts c = make_consts(hp);
    const int target = *step_dev - 1;
    const int chunks = D / VEC;
    const int64_t total = (int64_t)F * B * chunks;
    // uniform trip count and no early exits: every lane of a wave reaches the replay (adam_replay_wave takes a wave minimum)
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < total; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        bool act = i < total;
        int d = 0, from = target;
        int64_t row = -1;
        if (act) {
            d = (int)(i % chunks) * VEC;
            const int64_t slot = i / chunks;
            const int f = (int)(slot / B);
            const int j = (int)(slot - (int64_t)f * B);
            act = j < uniq_cnt[f];
            if (act) row = uniq_row[(int64_t)f * B + j];
            act = act && row >= 0;                               // < 0: padding entry of an exchanged row list
        }
        if (act) from = last[row];
        act = act && from < target;
        if (!act) from = target;
        if (!__any(act)) continue;                               // wave-uniform
        const int64_t e0 = act ? row * D + d : 0;
        float wv[VEC], mv[VEC], vv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) { wv[k] = 0.f; mv[k] = 0.f; vv[k] = 1.f; }
        if (act) {
            if (VEC == 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(w + e0), b4 = *reinterpret_cast<const float4*>(m + e0),
                             c4 = *reinterpret_cast<const float4*>(v + e0);
                wv[0] = a4.x; wv[1] = a4.y; wv[2] = a4.z; wv[3] = a4.w;
                mv[0] = b4.x; mv[1] = b4.y; mv[2] = b4.z; mv[3] = b4.w;
                vv[0] = c4.x; vv[1] = c4.y; vv[2] = c4.z; vv[3] = c4.w;
            } else {
                wv[0] = w[e0]; mv[0] = m[e0]; vv[0] = v[e0];
            }
        }
        adam_replay_wave<FAST, VEC>(wv, mv, vv, from, target, c, hp);
        if (!act) continue;
        if (VEC == 4) {
            *reinterpret_cast<float4*>(w + e0) = make_float4(wv[0], wv[1], wv[2], wv[3]);
            *reinterpret_cast<float4*>(m + e0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
            *reinterpret_cast<float4*>(v + e0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            w[e0] = wv[0]; m[e0] = mv[0]; v[e0] = vv[0];
        }
        // every lane of the row has read last[row] above.  mark: the row's `chunks` lanes sit in ONE wave (chunks divides 64),
        // whose load of last[row] has completed for all of them before this later store issues — lane 0 of the row
        // advances it here; otherwise k_lazy_mark does it in a separate launch.
        if (mark && d == 0) last[row] = target;
    }
}
__global__ void __launch_bounds__(256) k_lazy_mark(const int32_t* __restrict__ uniq_row, const int32_t* __restrict__ uniq_cnt,
                                                   int32_t* __restrict__ last, const int32_t* __restrict__ step_dev, int32_t B,
                                                   int32_t F) {
    const int target = *step_dev - 1;
    const int64_t total = (int64_t)F * B;
    for (int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; slot < total; slot += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(slot / B);
        const int j = (int)(slot - (int64_t)f * B);
        if (j >= uniq_cnt[f]) continue;
        const int64_t row = uniq_row[slot];
        if (row < 0) continue;
        if (last[row] < target) last[row] = target;
    }
}

extern "C" int cdc_embed_lazy_catchup(const int32_t* uniq_row, const int32_t* uniq_cnt, float* w, float* m, float* v,
                                      int32_t* last, cdc_adam_hp hp, const int32_t* step_dev, double* reg_ring,
                                      int32_t ring_len, int64_t B, int32_t F, int32_t D, void* stream) {
    (void)reg_ring; (void)ring_len;
    CDC_CHECK_ARG(uniq_row && uniq_cnt && w && m && v && last && step_dev && hp.step_scalars && hp.n_scalars > 0, CDC_E_BADARG,
                  "embed_lazy_catchup: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0, CDC_E_BADARG, "embed_lazy_catchup: bad sizes");
    const bool vec = (D % 4 == 0) && ((((uintptr_t)w | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const int64_t total = (int64_t)F * B * (vec ? D / 4 : D);
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(total, 256), 256 * 16);
    CDC_CHECK_ARG(!hp.fast_replay || hp.inv_bc2, CDC_E_BADARG, "embed_lazy_catchup: fast_replay needs the inv_bc2 table");
    const int32_t chunks = vec ? D / 4 : D;
    const int32_t mark = (64 % chunks == 0) ? 1 : 0;
#define CDC_CATCHUP(FASTV, VECV)                                                                                                 \
    hipLaunchKernelGGL((k_lazy_catchup<FASTV, VECV>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, uniq_row, uniq_cnt, w, m, \
                       v, last, hp, step_dev, (int32_t)B, F, D, mark)
    if (hp.fast_replay) { if (vec) CDC_CATCHUP(true, 4); else CDC_CATCHUP(true, 1); }
    else                { if (vec) CDC_CATCHUP(false, 4); else CDC_CATCHUP(false, 1); }
#undef CDC_CATCHUP
    CDC_LAUNCH_CHECK("embed_lazy_catchup");
    if (!mark) {
        int blocks2 = (int)std::min<int64_t>(cdc_ceil_div((int64_t)F * B, 256), 256 * 16);
        hipLaunchKernelGGL(k_lazy_mark, dim3(blocks2), dim3(256), 0, (hipStream_t)stream, uniq_row, uniq_cnt, last, step_dev,
                           (int32_t)B, F);
        CDC_LAUNCH_CHECK("embed_lazy_mark");
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// catch-up + gather in ONE pass over the batch's table rows (model/layer.py:147-157 on the lazy table): the group of lanes that
// brings a unique row up to date holds the current row in registers — it writes the row back to the table AND to every batch
// position that looks it up (fp32 embeddings + their bf16 shadow), so the forward never re-reads the table.  Rows looked up
// a few times (<= 4) are written by their own lanes; a row looked up more often (a domain column: three rows, ~B/3 positions
// each; the head of a Zipf distribution) is handed to the whole wave: its values are broadcast lane to lane and the 64 lanes
// write its positions side by side — the hot row is read from HBM once per step, however many samples carry it.
// Ids outside the table (row -1 in the sorted lists) yield zero rows, like cdc_embed_gather_fwd.
// ------------------------------------------------------------------------------------------------
#define CG_HOT 32                       /* a row looked up by more than this many samples of the batch is a HOT row */
template <bool FAST>
__global__ void __launch_bounds__(256) k_lazy_catchup_gather(const int32_t* __restrict__ uniq_row, const int32_t* __restrict__ uniq_cnt,
                                                             const int32_t* __restrict__ seg_start, const int32_t* __restrict__ perm,
                                                             float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                             int32_t* __restrict__ last, cdc_adam_hp hp,
                                                             const int32_t* __restrict__ step_dev, float* __restrict__ out,
                                                             __bf16* __restrict__ out_h, int64_t ld_out_h, int32_t B, int32_t F, int32_t D,
                                                             int32_t n_short_blocks, int32_t hot_cap) {
    typedef __bf16 h4_t __attribute__((ext_vector_type(4)));
    const AdamConsts c = make_consts(hp);
    const int target = *step_dev - 1;
    const int chunks = D / 4;                                           // lanes per row; divides 64 (checked by the launcher)
    const int lane = threadIdx.x & 63;
    const int64_t total = (int64_t)F * B * chunks;
    const int64_t ld_out = (int64_t)F * D;
    auto put = [&](int b, int f, int ch, float x0, float x1, float x2, float x3) {
        *reinterpret_cast<float4*>(out + (int64_t)b * ld_out + (int64_t)f * D + ch * 4) = make_float4(x0, x1, x2, x3);
        if (out_h) {
            h4_t h = {(__bf16)x0, (__bf16)x1, (__bf16)x2, (__bf16)x3};
            *reinterpret_cast<h4_t*>(out_h + (int64_t)b * ld_out_h + (int64_t)f * D + ch * 4) = h;
        }
    };
    if ((int)blockIdx.x >= n_short_blocks) {
        // ---- HOT rows of one field (a domain column: three rows with ~B/3 positions each; the head of a Zipf distribution): the
        //      workgroup lists them, brings all of them up to date in one pass (a thread per (row, 16-byte chunk)), keeps the
        //      current rows in LDS and writes every position that looks one of them up with all 256 threads — a hot row is read
        //      from HBM once per step however many samples carry it.  The other workgroups skip these rows.
        extern __shared__ __attribute__((aligned(16))) unsigned char cg_smem[];
        __shared__ int n_hot;
        // hot_cap = B / CG_HOT + 1 rounded up to 4: no more than that many rows can each hold more than CG_HOT of a field's B lookups
        int32_t* hot_j = reinterpret_cast<int32_t*>(cg_smem);                      // [hot_cap] unique-row slots of the hot rows
        float4* stage = reinterpret_cast<float4*>(hot_j + hot_cap);                // [hot_cap][chunks] the rows as the forward sees them
        const int f = (int)blockIdx.x - n_short_blocks;
        const int cnt = uniq_cnt[f];
        const int32_t* ss = seg_start + (int64_t)f * (B + 1);
        if (threadIdx.x == 0) n_hot = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += blockDim.x)
            if (ss[j + 1] - ss[j] > CG_HOT) {
                const int slot = atomicAdd(&n_hot, 1);                             // (the order of the list changes no result)
                if (slot < hot_cap) hot_j[slot] = j;
            }
        __syncthreads();
        const int nh = n_hot < hot_cap ? n_hot : hot_cap;                          // (n_hot <= B / CG_HOT < hot_cap by counting)
        const int n_item = nh * chunks;
        // uniform trip count: every lane of a wave reaches the replay
        for (int base = 0; base < n_item; base += blockDim.x) {
            const int i = base + threadIdx.x;
            const bool has = i < n_item;
            const int hs = has ? i / chunks : 0, ch = has ? i % chunks : 0;
            const int64_t row = has ? uniq_row[(int64_t)f * B + hot_j[hs]] : -1;
            const bool real = has && row >= 0;
            int from = real ? last[row] : target;
            const bool act = real && from < target;
            if (!act) from = target;
            const int64_t e0 = real ? row * D + ch * 4 : 0;
            float wv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {1.f, 1.f, 1.f, 1.f};
            if (real) { const float4 a4 = *reinterpret_cast<const float4*>(w + e0); wv[0] = a4.x; wv[1] = a4.y; wv[2] = a4.z; wv[3] = a4.w; }
            if (act) {
                const float4 b4 = *reinterpret_cast<const float4*>(m + e0), c4 = *reinterpret_cast<const float4*>(v + e0);
                mv[0] = b4.x; mv[1] = b4.y; mv[2] = b4.z; mv[3] = b4.w;
                vv[0] = c4.x; vv[1] = c4.y; vv[2] = c4.z; vv[3] = c4.w;
            }
            adam_replay_wave<FAST, 4>(wv, mv, vv, from, target, c, hp);
            if (act) {
                *reinterpret_cast<float4*>(w + e0) = make_float4(wv[0], wv[1], wv[2], wv[3]);
                *reinterpret_cast<float4*>(m + e0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
                *reinterpret_cast<float4*>(v + e0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                if (ch == 0) last[row] = target;
            }
            if (has) stage[i] = make_float4(wv[0], wv[1], wv[2], wv[3]);            // (ids outside the table: a zero row)
        }
        __syncthreads();
        for (int hs = 0; hs < nh; ++hs) {
            const int j = hot_j[hs];
            const int p0 = ss[j], n = ss[j + 1] - p0;
            for (int q = threadIdx.x; q < n * chunks; q += blockDim.x) {
                const int b = perm[(int64_t)f * B + p0 + q / chunks], ch = q % chunks;
                const float4 x = stage[hs * chunks + ch];
                put(b, f, ch, x.x, x.y, x.z, x.w);
            }
        }
        return;
    }
    // uniform trip count and no early exits: every lane of a wave reaches the replay and the wave-cooperative part
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < total; base += (int64_t)n_short_blocks * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        bool has = i < total;                                           // this lane belongs to a unique row of the batch
        int ch = 0, f = 0, j = 0, from = target;
        int64_t row = -1;
        int p0 = 0, n = 0;
        if (has) {
            ch = (int)(i % chunks);
            const int64_t slot = i / chunks;
            f = (int)(slot / B);
            j = (int)(slot - (int64_t)f * B);
            has = j < uniq_cnt[f];
        }
        if (has) {
            row = uniq_row[(int64_t)f * B + j];
            p0 = seg_start[(int64_t)f * (B + 1) + j];
            n = seg_start[(int64_t)f * (B + 1) + j + 1] - p0;
            has = n <= CG_HOT;                                          // hot rows belong to their field's workgroup (above)
        }
        const bool real = has && row >= 0;                              // < 0: ids outside the table (zero rows)
        if (real) from = last[row];
        const bool act = real && from < target;
        if (!act) from = target;
        if (!__any(has)) continue;                                      // wave-uniform
        const int64_t e0 = real ? row * D + ch * 4 : 0;
        float wv[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {1.f, 1.f, 1.f, 1.f};
        if (real) {
            const float4 a4 = *reinterpret_cast<const float4*>(w + e0);
            wv[0] = a4.x; wv[1] = a4.y; wv[2] = a4.z; wv[3] = a4.w;
        }
        if (act) {
            const float4 b4 = *reinterpret_cast<const float4*>(m + e0), c4 = *reinterpret_cast<const float4*>(v + e0);
            mv[0] = b4.x; mv[1] = b4.y; mv[2] = b4.z; mv[3] = b4.w;
            vv[0] = c4.x; vv[1] = c4.y; vv[2] = c4.z; vv[3] = c4.w;
        }
        // short segments: the positions' batch rows, fetched before the replay (their latency hides under it)
        int bq[4] = {0, 0, 0, 0};
        const bool inline_seg = has && n <= 4;
        if (inline_seg) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < n) bq[q] = perm[(int64_t)f * B + p0 + q];
        }
        adam_replay_wave<FAST, 4>(wv, mv, vv, from, target, c, hp);
        if (act) {
            *reinterpret_cast<float4*>(w + e0) = make_float4(wv[0], wv[1], wv[2], wv[3]);
            *reinterpret_cast<float4*>(m + e0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
            *reinterpret_cast<float4*>(v + e0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            if (ch == 0) last[row] = target;                            // the row's lanes sit in one wave and have all read last[row]
        }
        if (inline_seg) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < n) put(bq[q], f, ch, wv[0], wv[1], wv[2], wv[3]);
        }
        // longer segments: one at a time, the whole wave writes the row's positions
        unsigned long long todo = __ballot(has && n > 4 && ch == 0);
        while (todo) {
            const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);      // lane of the row's chunk 0
            todo &= todo - 1;
            const int sf = __builtin_amdgcn_readlane(f, L), sp0 = __builtin_amdgcn_readlane(p0, L), sn = __builtin_amdgcn_readlane(n, L);
            const int items = sn * chunks;
            // lane l serves chunk (l % chunks) of positions l / chunks, + 64 / chunks, ...: its four values are fixed for the whole
            // segment (one shuffle each); the positions' batch rows are fetched eight at a time (a dependent load per position
            // made a domain column's ~B/3 positions the longest chain of the launch)
            const int pc = lane % chunks, ppl = lane / chunks, pstep = 64 / chunks;
            const float x0 = __shfl(wv[0], L + pc, 64), x1 = __shfl(wv[1], L + pc, 64);
            const float x2 = __shfl(wv[2], L + pc, 64), x3 = __shfl(wv[3], L + pc, 64);
            (void)items;
            for (int pb = ppl; pb < sn; pb += 8 * pstep) {
                int bb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int pp = pb + q * pstep;
                    bb[q] = pp < sn ? perm[(int64_t)sf * B + sp0 + pp] : -1;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (bb[q] >= 0) put(bb[q], sf, pc, x0, x1, x2, x3);
            }
        }
    }
}

extern "C" int cdc_embed_lazy_catchup_gather(const int32_t* uniq_row, const int32_t* uniq_cnt, const int32_t* seg_start, const int32_t* perm,
                                             float* w, float* m, float* v, int32_t* last, cdc_adam_hp hp, const int32_t* step_dev,
                                             float* out, void* out_h, int64_t ld_out_h, int64_t B, int32_t F, int32_t D, void* stream) {
    CDC_CHECK_ARG(uniq_row && uniq_cnt && seg_start && perm && w && m && v && last && step_dev && out && hp.step_scalars && hp.n_scalars > 0,
                  CDC_E_BADARG, "embed_lazy_catchup_gather: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0 && D % 4 == 0 && 64 % (D / 4) == 0, CDC_E_BADARG,
                  "embed_lazy_catchup_gather: emb_dim must be 4, 8, 16, 32, 64, 128 or 256 (a row's 16-byte lanes share one wave)");
    CDC_CHECK_ARG(((((uintptr_t)w | (uintptr_t)m | (uintptr_t)v | (uintptr_t)out) & 15) == 0), CDC_E_ALIGN,
                  "embed_lazy_catchup_gather: table and output must be 16-byte aligned");
    CDC_CHECK_ARG(!out_h || (ld_out_h >= (int64_t)F * D && ld_out_h % 4 == 0 && (((uintptr_t)out_h) & 7) == 0), CDC_E_BADARG,
                  "embed_lazy_catchup_gather: malformed bf16 shadow");
    CDC_CHECK_ARG(!hp.fast_replay || hp.inv_bc2, CDC_E_BADARG, "embed_lazy_catchup_gather: fast_replay needs the inv_bc2 table");
    const int32_t hot_cap = (int32_t)((B / CG_HOT + 1 + 3) / 4 * 4);
    const int64_t total = (int64_t)F * B * (D / 4);
    const int blocks = (int)std::min<int64_t>(cdc_ceil_div(total, 256), 256 * 16);
    // the F hot-row workgroups go FIRST in dispatch order is not needed: they are short (a few hundred positions per thread at
    // most) and run beside the others; they sit behind the short blocks in the grid
    const size_t lds = (size_t)hot_cap * 4 + (size_t)hot_cap * (D / 4) * 16;
    CDC_CHECK_ARG(lds <= 64 * 1024, CDC_E_TOOBIG, "embed_lazy_catchup_gather: batch too large for the hot-row staging area");
    if (hp.fast_replay)
        hipLaunchKernelGGL((k_lazy_catchup_gather<true>), dim3(blocks + F), dim3(256), lds, (hipStream_t)stream, uniq_row, uniq_cnt, seg_start, perm,
                           w, m, v, last, hp, step_dev, out, reinterpret_cast<__bf16*>(out_h), ld_out_h, (int32_t)B, F, D, blocks, hot_cap);
    else
        hipLaunchKernelGGL((k_lazy_catchup_gather<false>), dim3(blocks + F), dim3(256), lds, (hipStream_t)stream, uniq_row, uniq_cnt, seg_start, perm,
                           w, m, v, last, hp, step_dev, out, reinterpret_cast<__bf16*>(out_h), ld_out_h, (int32_t)B, F, D, blocks, hot_cap);
    CDC_LAUNCH_CHECK("embed_lazy_catchup_gather");
    return 0;
}

// step t for the batch's rows: rows are at t-1 after catchup (last[] still holds the older value,
// which is ignored here); writes last[row] = t.
template <int VEC>
__global__ void __launch_bounds__(256) k_lazy_update(const float* __restrict__ rowgrad, const int32_t* __restrict__ uniq_row,
                                                     const int32_t* __restrict__ uniq_cnt, float* __restrict__ w,
                                                     float* __restrict__ m, float* __restrict__ v, int32_t* __restrict__ last,
                                                     cdc_adam_hp hp, const int32_t* __restrict__ step_dev, int32_t B,
                                                     int32_t F, int32_t D) {
    const AdamConsts c = make_consts(hp);
    const int t = *step_dev;
    float step_size, bc2s;
    step_scalars_at(hp.step_scalars, hp.n_scalars, t, step_size, bc2s);
    const int chunks = D / VEC;
    const int64_t total = (int64_t)F * B * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % chunks) * VEC;
        const int64_t slot = i / chunks;
        const int f = (int)(slot / B);
        const int j = (int)(slot - (int64_t)f * B);
        if (j >= uniq_cnt[f]) continue;
        const int64_t row = uniq_row[slot];
        if (row < 0) continue;
        const int64_t e0 = row * D + d;
        float wv[VEC], mv[VEC], vv[VEC], gv[VEC];
        if (VEC == 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(w + e0), b4 = *reinterpret_cast<const float4*>(m + e0),
                         c4 = *reinterpret_cast<const float4*>(v + e0), g4 = *reinterpret_cast<const float4*>(rowgrad + slot * D + d);
            wv[0] = a4.x; wv[1] = a4.y; wv[2] = a4.z; wv[3] = a4.w;
            mv[0] = b4.x; mv[1] = b4.y; mv[2] = b4.z; mv[3] = b4.w;
            vv[0] = c4.x; vv[1] = c4.y; vv[2] = c4.z; vv[3] = c4.w;
            gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        } else {
            wv[0] = w[e0]; mv[0] = m[e0]; vv[0] = v[e0]; gv[0] = rowgrad[slot * D + d];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) adam_elem(wv[k], mv[k], vv[k], gv[k], c, step_size, bc2s);
        if (VEC == 4) {
            *reinterpret_cast<float4*>(w + e0) = make_float4(wv[0], wv[1], wv[2], wv[3]);
            *reinterpret_cast<float4*>(m + e0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
            *reinterpret_cast<float4*>(v + e0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {
            w[e0] = wv[0]; m[e0] = mv[0]; v[e0] = vv[0];
        }
        if (d == 0) last[row] = t;
    }
}

extern "C" int cdc_embed_lazy_update(const float* rowgrad, const int32_t* uniq_row, const int32_t* uniq_cnt, float* w, float* m,
                                     float* v, int32_t* last, cdc_adam_hp hp, const int32_t* step_dev, double* reg_ring,
                                     int32_t ring_len, int64_t B, int32_t F, int32_t D, void* stream) {
    (void)reg_ring; (void)ring_len;
    CDC_CHECK_ARG(rowgrad && uniq_row && uniq_cnt && w && m && v && last && step_dev && hp.step_scalars && hp.n_scalars > 0,
                  CDC_E_BADARG, "embed_lazy_update: null pointer");
    CDC_CHECK_ARG(B > 0 && F > 0 && D > 0, CDC_E_BADARG, "embed_lazy_update: bad sizes");
    const bool vec = (D % 4 == 0) && ((((uintptr_t)w | (uintptr_t)m | (uintptr_t)v | (uintptr_t)rowgrad) & 15) == 0);
    const int64_t total = (int64_t)F * B * (vec ? D / 4 : D);
    int blocks = (int)std::min<int64_t>(cdc_ceil_div(total, 256), 256 * 16);
    if (vec)
        hipLaunchKernelGGL(k_lazy_update<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowgrad, uniq_row, uniq_cnt, w, m, v, last, hp,
                           step_dev, (int32_t)B, F, D);
    else
        hipLaunchKernelGGL(k_lazy_update<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowgrad, uniq_row, uniq_cnt, w, m, v, last, hp,
                           step_dev, (int32_t)B, F, D);
    CDC_LAUNCH_CHECK("embed_lazy_update");
    return 0;
}

// rows -> step target = *step_dev + step_bias: all of them (period <= 1) or the slice (target mod period) of the table.
// Rows not looked up since their last flush share one `last`, so whole waves replay the same steps.
template <bool FAST, int VEC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_lazy_flush(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                    int32_t* __restrict__ last, int64_t R, int32_t D, cdc_adam_hp hp,
                                                    const int32_t* __restrict__ step_dev, int32_t step_bias, int32_t period,
                                                    int32_t own_mod, int32_t own_rem, int32_t mark, int64_t win_lo, int64_t win_n) {
    const int target = *step_dev + step_bias;
    // period > 1: this call handles one slice of the table, slice (target mod period) — every row is brought up to date
    // once per `period` steps, a 1/period share of the work in every step instead of a burst
    int64_t row_lo = 0, row_hi = R;
    if (period > 1) {
        const int64_t rps = (R + period - 1) / period;
        row_lo = (int64_t)(((target % period) + period) % period) * rps;
        row_hi = row_lo + rps < R ? row_lo + rps : R;
        if (row_lo >= row_hi) return;
    }
    // the launcher's row window [win_lo, win_lo + win_n): a launch's work items are counted in 32 bits (below), so a table of more
    // than ~2^31 items (rows x 16-byte chunks) is flushed window by window
    if (win_lo > row_lo) row_lo = win_lo;
    if (win_lo + win_n < row_hi) row_hi = win_lo + win_n;
    if (row_lo >= row_hi) return;
    const AdamConsts c = make_consts(hp);
    // row-sharded table: only the rows this rank owns (row % own_mod == own_rem) are visited
    const int64_t stride = own_mod > 1 ? own_mod : 1;
    int64_t first = row_lo;
    if (stride > 1) first = row_lo + ((own_rem - row_lo % stride) + stride) % stride;
    if (first >= row_hi) return;
    const int64_t n_rows = (row_hi - first + stride - 1) / stride;
    const int chunks = D / VEC;
    const int64_t total = n_rows * chunks;
    // software pipeline: the loads of the NEXT item are in flight under the replay of this one (the background form of this
    // launch runs with one or two waves per SIMD, too few to hide a ~2 us load behind other waves).  Only the item's VALUES and
    // its start step are carried across the iteration; its row and element offset are recomputed from the index where they
    // are needed (register budget: 64 VGPRs, so that two background waves sit beside four waves of a 96-VGPR kernel on a SIMD)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int32_t itotal = (int32_t)total;                              // (items of a launch + its grid stride < 2^31: lazy_flush_launch's windows)
    auto row_of = [&](int32_t i) -> int64_t { return first + (int64_t)(i / chunks) * stride; };
    auto fetch = [&](int32_t i, int& from, float (&wv)[VEC], float (&mv)[VEC], float (&vv)[VEC]) __attribute__((always_inline)) {
        const bool in = i < itotal;
        const int64_t row = in ? row_of(i) : first;
        from = in ? last[row] : target;
        const bool act = in && from < target;
        if (!act) from = target;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { wv[k] = 0.f; mv[k] = 0.f; vv[k] = 1.f; }
        if (act) {
            const int64_t e0 = row * D + (i % chunks) * VEC;
            if (VEC == 4) {
                // streamed once per flush period: non-temporal, so that the slice does not push the contractions' operands out of L2
                const f4v a4 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(w + e0)),
                          b4 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(m + e0)),
                          c4 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(v + e0));
#pragma unroll
                for (int k = 0; k < 4; ++k) { wv[k] = a4[k]; mv[k] = b4[k]; vv[k] = c4[k]; }
            } else {
                wv[0] = w[e0]; mv[0] = m[e0]; vv[0] = v[e0];
            }
        }
    };
    const int32_t gstride = (int32_t)(gridDim.x * blockDim.x);
    int cfrom, nfrom;
    float cw[VEC], cm[VEC], cv[VEC], nw[VEC], nm[VEC], nv[VEC];
    int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    fetch(i, cfrom, cw, cm, cv);
    for (int32_t base = (int32_t)(blockIdx.x * blockDim.x); base < itotal; base += gstride, i += gstride) {
        fetch(base + gstride < itotal ? i + gstride : itotal, nfrom, nw, nm, nv);     // (past the end: an inactive item, no loads)
        const bool act = cfrom < target;
        if (__any(act)) {                                        // wave-uniform
            adam_replay_wave<FAST, VEC>(cw, cm, cv, cfrom, target, c, hp);
            if (act) {
                const int64_t row = row_of(i);
                const int64_t e0 = row * D + (i % chunks) * VEC;
                if (VEC == 4) {
                    __builtin_nontemporal_store((f4v){cw[0], cw[1], cw[2], cw[3]}, reinterpret_cast<f4v*>(w + e0));
                    __builtin_nontemporal_store((f4v){cm[0], cm[1], cm[2], cm[3]}, reinterpret_cast<f4v*>(m + e0));
                    __builtin_nontemporal_store((f4v){cv[0], cv[1], cv[2], cv[3]}, reinterpret_cast<f4v*>(v + e0));
                } else {
                    w[e0] = cw[0]; m[e0] = cm[0]; v[e0] = cv[0];
                }
                if (mark && (i % chunks) == 0) last[row] = target;   // the row's lanes share one wave (see k_lazy_catchup)
            }
        }
        cfrom = nfrom;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { cw[k] = nw[k]; cm[k] = nm[k]; cv[k] = nv[k]; }
    }
}
__global__ void __launch_bounds__(256) k_lazy_set_last(int32_t* __restrict__ last, int64_t R, const int32_t* __restrict__ step_dev,
                                                       int32_t step_bias, int32_t period, int32_t own_mod, int32_t own_rem) {
    const int target = *step_dev + step_bias;
    int64_t row_lo = 0, row_hi = R;
    if (period > 1) {
        const int64_t rps = (R + period - 1) / period;
        row_lo = (int64_t)(((target % period) + period) % period) * rps;
        row_hi = row_lo + rps < R ? row_lo + rps : R;
    }
    for (int64_t i = row_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_hi; i += (int64_t)gridDim.x * blockDim.x) {
        if (own_mod > 1 && (i % own_mod) != own_rem) continue;
        if (last[i] < target) last[i] = target;
    }
}

static int lazy_flush_launch(float* w, float* m, float* v, int32_t* last, int64_t R, int32_t D, cdc_adam_hp hp,
                             const int32_t* step_dev, int32_t step_bias, int32_t period, int32_t own_mod, int32_t own_rem,
                             int32_t waves_per_simd, void* stream) {
    CDC_CHECK_ARG(w && m && v && last && step_dev && hp.step_scalars && hp.n_scalars > 0, CDC_E_BADARG, "embed_lazy_flush: null pointer");
    CDC_CHECK_ARG(R > 0 && D > 0 && period >= 0 && own_mod >= 0 && (own_mod <= 1 || (own_rem >= 0 && own_rem < own_mod)), CDC_E_BADARG,
                  "embed_lazy_flush: bad sizes");
    CDC_CHECK_ARG(!hp.fast_replay || hp.inv_bc2, CDC_E_BADARG, "embed_lazy_flush: fast_replay needs the inv_bc2 table");
    const bool vec = (D % 4 == 0) && ((((uintptr_t)w | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    int64_t rows_call = period > 1 ? cdc_ceil_div(R, period) : R;
    if (own_mod > 1) rows_call = cdc_ceil_div(rows_call, own_mod) + 1;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdc_ceil_div(rows_call * D / (vec ? 4 : 1), 256), 256 * 16));
    if (waves_per_simd > 0) {                                    // background form: a workgroup of 256 threads = one wave per SIMD of its CU
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0, v_ = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v_, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v_ <= 0) v_ = 256;
            n_cu = v_;
        }
        const int64_t cap = (int64_t)waves_per_simd * n_cu;
        if (blocks > cap) blocks = (int)cap;
    }
    hipStream_t st = (hipStream_t)stream;
    const int32_t chunks = vec ? D / 4 : D;
    const int32_t mark = (64 % chunks == 0) ? 1 : 0;             // last[] advanced inside the kernel; else by k_lazy_set_last
    // the kernel counts a launch's work items (rows x chunks) and its grid-stride index in 32 bits (its 64-VGPR budget): rows per
    // launch are capped so that items + one grid stride stay below 2^31.  A whole-table flush (period <= 1: flush_table,
    // state_dict, evaluation) of a larger table goes out as several launches over consecutive row windows; a slice launch
    // (its rows are chosen on the device from the step counter) that large is refused.
    const int64_t max_items = ((int64_t)1 << 31) - 1 - (int64_t)blocks * 256 - 256;
    const int64_t win_rows = std::max<int64_t>(1, max_items / chunks);
    CDC_CHECK_ARG(period <= 1 || rows_call <= win_rows, CDC_E_TOOBIG,
                  "embed_lazy_flush: a slice of %lld rows x %d chunks exceeds the 2^31 work items of one launch (raise flush_every)",
                  (long long)rows_call, (int)chunks);
    for (int64_t lo = 0; lo < R; lo += (period > 1 ? R : win_rows)) {
        const int64_t n = period > 1 ? R : std::min<int64_t>(win_rows, R - lo);
        if (hp.fast_replay) {
            if (vec) hipLaunchKernelGGL((k_lazy_flush<true, 4>), dim3(blocks), dim3(256), 0, st, w, m, v, last, R, D, hp, step_dev, step_bias, period, own_mod, own_rem, mark, lo, n);
            else     hipLaunchKernelGGL((k_lazy_flush<true, 1>), dim3(blocks), dim3(256), 0, st, w, m, v, last, R, D, hp, step_dev, step_bias, period, own_mod, own_rem, mark, lo, n);
        } else {
            if (vec) hipLaunchKernelGGL((k_lazy_flush<false, 4>), dim3(blocks), dim3(256), 0, st, w, m, v, last, R, D, hp, step_dev, step_bias, period, own_mod, own_rem, mark, lo, n);
            else     hipLaunchKernelGGL((k_lazy_flush<false, 1>), dim3(blocks), dim3(256), 0, st, w, m, v, last, R, D, hp, step_dev, step_bias, period, own_mod, own_rem, mark, lo, n);
        }
    }
    CDC_LAUNCH_CHECK("embed_lazy_flush");
    if (!mark) {
        int blocks2 = (int)std::max<int64_t>(1, std::min<int64_t>(cdc_ceil_div(period > 1 ? cdc_ceil_div(R, period) : R, 256), 256 * 16));
        hipLaunchKernelGGL(k_lazy_set_last, dim3(blocks2), dim3(256), 0, (hipStream_t)stream, last, R, step_dev, step_bias, period, own_mod, own_rem);
        CDC_LAUNCH_CHECK("embed_lazy_set_last");
    }
    return 0;
}
extern "C" int cdc_embed_lazy_flush(float* w, float* m, float* v, int32_t* last, int64_t R, int32_t D, cdc_adam_hp hp,
                                    const int32_t* step_dev, int32_t step_bias, int32_t period, int32_t own_mod, int32_t own_rem,
                                    void* stream) {
    return lazy_flush_launch(w, m, v, last, R, D, hp, step_dev, step_bias, period, own_mod, own_rem, 0, stream);
}
extern "C" int cdc_embed_lazy_flush_bg(float* w, float* m, float* v, int32_t* last, int64_t R, int32_t D, cdc_adam_hp hp,
                                       const int32_t* step_dev, int32_t step_bias, int32_t period, int32_t own_mod, int32_t own_rem,
                                       int32_t waves_per_simd, void* stream) {
    return lazy_flush_launch(w, m, v, last, R, D, hp, step_dev, step_bias, period, own_mod, own_rem, waves_per_simd, stream);
}


// ------------------------------------------------------------------------------------------------
// Row-sharded table under data parallelism (row r is owned by rank r % n_rank).  The per-field unique rows of the local
// batch are bucketed by owner into fixed-capacity send lists [n_rank][cap][F] (row id, -1 = padding); slots are assigned
// in ascending row order (the unique rows are already sorted), so the layout is deterministic.  After the all-to-all an
// owner treats the received lists as a batch of n_rank*cap "rows" per field and reuses the sort / catch-up / segment-sum /
// update kernels above unchanged.
// ------------------------------------------------------------------------------------------------
#define SHARD_MAX_RANKS 16
// one workgroup per (field, owner): flags the field's unique rows that belong to the owner, ranks them with a block scan
// (ascending row order = slot order), writes the owner's send list (padding included) and the rows' slot numbers
__global__ void __launch_bounds__(SORT_THREADS) k_shard_bucket(const int32_t* __restrict__ uniq_row, const int32_t* __restrict__ uniq_cnt,
                                                               int32_t* __restrict__ send_ids, int32_t* __restrict__ slot_of,
                                                               int32_t* __restrict__ overflow, int32_t B, int32_t F, int32_t n_rank,
                                                               int32_t cap) {
    __shared__ int32_t wave_tot[SORT_THREADS / 64];
    const int f = blockIdx.x, o = blockIdx.y, tid = threadIdx.x;
    const int n = uniq_cnt[f];
    const int per = (B + SORT_THREADS - 1) / SORT_THREADS;
    const int begin = tid * per;
    const int32_t* ur = uniq_row + (int64_t)f * B;
    int local = 0;
    for (int j = begin; j < begin + per && j < n; ++j) {
        const int32_t row = ur[j];
        local += (row >= 0 && row % n_rank == o) ? 1 : 0;
    }
    int total;
    int slot = block_exclusive_scan(local, wave_tot, tid, total);
    for (int j = begin; j < begin + per && j < n; ++j) {
        const int32_t row = ur[j];
        if (row < 0) {
            if (o == 0) slot_of[(int64_t)f * B + j] = -1;
            continue;
        }
        if (row % n_rank != o) continue;
        int s_ = slot++;
        if (s_ < cap) send_ids[((int64_t)o * cap + s_) * F + f] = row;
        else { atomicMax(overflow, s_ + 1); s_ = -1; }
        slot_of[(int64_t)f * B + j] = s_;
    }
    for (int s_ = min(total, cap) + tid; s_ < cap; s_ += SORT_THREADS) send_ids[((int64_t)o * cap + s_) * F + f] = -1;
}
extern "C" int cdc_shard_bucket(const int32_t* uniq_row, const int32_t* uniq_cnt, int32_t* send_ids, int32_t* slot_of,
                                int32_t* overflow, int64_t B, int32_t F, int32_t n_rank, int32_t cap, void* stream) {
    CDC_CHECK_ARG(uniq_row && uniq_cnt && send_ids && slot_of && overflow && B > 0 && F > 0 && cap > 0 && n_rank > 0 &&
                      n_rank <= SHARD_MAX_RANKS && B <= CDC_SORT_MAX_ROWS, CDC_E_BADARG, "shard_bucket: bad argument");
    hipLaunchKernelGGL(k_shard_bucket, dim3(F, n_rank), dim3(SORT_THREADS), 0, (hipStream_t)stream, uniq_row, uniq_cnt, send_ids, slot_of, overflow,
                       (int32_t)B, F, n_rank, cap);
    CDC_LAUNCH_CHECK("shard_bucket");
    return 0;
}

// requester, forward: rows_recv [n_rank (owner)][cap][F][D] -> out[b, f*D:(f+1)*D] for every batch position of every unique row
template <int VEC>
__global__ void __launch_bounds__(256) k_shard_expand(const float* __restrict__ rows_recv, const int32_t* __restrict__ uniq_row,
                                                      const int32_t* __restrict__ uniq_cnt, const int32_t* __restrict__ seg_start,
                                                      const int32_t* __restrict__ perm, const int32_t* __restrict__ slot_of,
                                                      float* __restrict__ out, int32_t B, int32_t F, int32_t D, int32_t n_rank,
                                                      int32_t cap) {
    // one thread per (sorted batch position, chunk of the row): the unique row a position belongs to is found by a
    // binary search in the field's segment starts (a loop over a segment would serialise the few-distinct-values fields)
    const int chunks = D / VEC;
    const int64_t total = (int64_t)F * B * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int64_t fk = i / chunks;
        const int f = (int)(fk / B);
        const int k = (int)(fk - (int64_t)f * B);
        const int32_t* sst = seg_start + (int64_t)f * (B + 1);
        int lo = 0, hi = uniq_cnt[f];                              // largest j with sst[j] <= k
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sst[mid] <= k) lo = mid; else hi = mid;
        }
        const int64_t slotj = (int64_t)f * B + lo;
        const int32_t row = uniq_row[slotj];
        const int s_ = slot_of[slotj];
        const bool ok = row >= 0 && s_ >= 0;
        const int64_t src = ok ? ((((int64_t)(row % n_rank)) * cap + s_) * F + f) * D + c * VEC : 0;
        const int64_t dst = ((int64_t)perm[fk] * F + f) * D + c * VEC;
        if (VEC == 4) {
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) val = *reinterpret_cast<const float4*>(rows_recv + src);
            *reinterpret_cast<float4*>(out + dst) = val;
        } else {
            out[dst] = ok ? rows_recv[src] : 0.f;
        }
    }
}
extern "C" int cdc_shard_expand(const float* rows_recv, const int32_t* uniq_row, const int32_t* uniq_cnt, const int32_t* seg_start,
                                const int32_t* perm, const int32_t* slot_of, float* out, int64_t B, int32_t F, int32_t D,
                                int32_t n_rank, int32_t cap, void* stream) {
    CDC_CHECK_ARG(rows_recv && uniq_row && uniq_cnt && seg_start && perm && slot_of && out && B > 0 && F > 0 && D > 0 && cap > 0 &&
                      n_rank > 0, CDC_E_BADARG, "shard_expand: bad argument");
    const bool vec = (D % 4 == 0) && (((uintptr_t)rows_recv | (uintptr_t)out) % 16 == 0);
    int blocks = (int)std::min<int64_t>(cdc_ceil_div((int64_t)F * B * (vec ? D / 4 : D), 256), 8192);
    if (vec)
        hipLaunchKernelGGL(k_shard_expand<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows_recv, uniq_row, uniq_cnt, seg_start,
                           perm, slot_of, out, (int32_t)B, F, D, n_rank, cap);
    else
        hipLaunchKernelGGL(k_shard_expand<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows_recv, uniq_row, uniq_cnt, seg_start,
                           perm, slot_of, out, (int32_t)B, F, D, n_rank, cap);
    CDC_LAUNCH_CHECK("shard_expand");
    return 0;
}

// requester, backward: rowgrad [F][B][D] (per unique row) -> send_grads [n_rank (owner)][cap][F][D]
template <int VEC>
__global__ void __launch_bounds__(256) k_shard_pack(const float* __restrict__ rowgrad, const int32_t* __restrict__ uniq_row,
                                                    const int32_t* __restrict__ uniq_cnt, const int32_t* __restrict__ slot_of,
                                                    float* __restrict__ send, int32_t B, int32_t F, int32_t D, int32_t n_rank,
                                                    int32_t cap) {
    const int chunks = D / VEC;
    const int64_t total = (int64_t)F * B * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % chunks);
        const int64_t slotj = i / chunks;
        const int f = (int)(slotj / B);
        const int j = (int)(slotj - (int64_t)f * B);
        if (j >= uniq_cnt[f]) continue;
        const int32_t row = uniq_row[slotj];
        const int s_ = slot_of[slotj];
        if (row < 0 || s_ < 0) continue;
        float* dst = send + ((((int64_t)(row % n_rank)) * cap + s_) * F + f) * D + c * VEC;
        const float* src = rowgrad + slotj * D + c * VEC;
        if (VEC == 4) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        else dst[0] = src[0];
    }
}
extern "C" int cdc_shard_pack(const float* rowgrad, const int32_t* uniq_row, const int32_t* uniq_cnt, const int32_t* slot_of,
                              float* send, int64_t B, int32_t F, int32_t D, int32_t n_rank, int32_t cap, void* stream) {
    CDC_CHECK_ARG(rowgrad && uniq_row && uniq_cnt && slot_of && send && B > 0 && F > 0 && D > 0 && cap > 0 && n_rank > 0, CDC_E_BADARG,
                  "shard_pack: bad argument");
    const bool vec = (D % 4 == 0) && (((uintptr_t)rowgrad | (uintptr_t)send) % 16 == 0);
    int blocks = (int)std::min<int64_t>(cdc_ceil_div((int64_t)F * B * (vec ? D / 4 : D), 256), 8192);
    if (vec)
        hipLaunchKernelGGL(k_shard_pack<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowgrad, uniq_row, uniq_cnt, slot_of, send,
                           (int32_t)B, F, D, n_rank, cap);
    else
        hipLaunchKernelGGL(k_shard_pack<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rowgrad, uniq_row, uniq_cnt, slot_of, send,
                           (int32_t)B, F, D, n_rank, cap);
    CDC_LAUNCH_CHECK("shard_pack");
    return 0;
}
