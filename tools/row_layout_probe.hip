// Development probe (not part of the library): what would rows stored as [w | m | v] (192 bytes) buy the kernels that touch the batch's
// rows?  ~100 K sorted rows of a 26 M-row table (26 fields x ~3 900 distinct rows, the C2 step's catch-up / row update), D = 16:
// read w, m, v of every row, change them, write them back — from three arrays of 64-byte rows (today) and from one array of
// 192-byte rows.  Build + run: hipcc --offload-arch=gfx950 -O3 tools/row_layout_probe.hip -o tools/_build/row_layout_probe && it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) k_split(const int* __restrict__ rows, int n, float* __restrict__ w, float* __restrict__ m, float* __restrict__ v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 4) return;
    const long e = (long)rows[i >> 2] * 16 + (i & 3) * 4;
    float4 a = *reinterpret_cast<float4*>(w + e), b = *reinterpret_cast<float4*>(m + e), c = *reinterpret_cast<float4*>(v + e);
    a.x += b.x * 1e-3f; b.y += c.y * 1e-3f; c.z += a.z * 1e-3f;
    *reinterpret_cast<float4*>(w + e) = a; *reinterpret_cast<float4*>(m + e) = b; *reinterpret_cast<float4*>(v + e) = c;
}
__global__ void __launch_bounds__(256) k_inter(const int* __restrict__ rows, int n, float* __restrict__ t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 4) return;
    float* p = t + (long)rows[i >> 2] * 48 + (i & 3) * 4;
    float4 a = *reinterpret_cast<float4*>(p), b = *reinterpret_cast<float4*>(p + 16), c = *reinterpret_cast<float4*>(p + 32);
    a.x += b.x * 1e-3f; b.y += c.y * 1e-3f; c.z += a.z * 1e-3f;
    *reinterpret_cast<float4*>(p) = a; *reinterpret_cast<float4*>(p + 16) = b; *reinterpret_cast<float4*>(p + 32) = c;
}

int main() {
    const long R = 26000000;
    const int F = 26, PER = 3900, SETS = 8;
    float *w, *m, *v, *t;
    hipMalloc(&w, R * 64); hipMalloc(&m, R * 64); hipMalloc(&v, R * 64); hipMalloc(&t, R * 192);
    hipMemset(w, 0, R * 64); hipMemset(m, 0, R * 64); hipMemset(v, 0, R * 64); hipMemset(t, 0, R * 192);
    std::vector<int*> sets;
    int n = 0;
    srand(1);
    for (int s = 0; s < SETS; ++s) {
        std::vector<int> rows;
        for (int f = 0; f < F; ++f) {
            std::vector<int> r;
            for (int j = 0; j < PER; ++j) r.push_back(f * 1000000 + (int)(((long)rand() * 32768 + rand()) % 1000000));
            std::sort(r.begin(), r.end());
            r.erase(std::unique(r.begin(), r.end()), r.end());
            rows.insert(rows.end(), r.begin(), r.end());
        }
        n = (int)rows.size();
        int* d; hipMalloc(&d, rows.size() * 4);
        hipMemcpy(d, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
        sets.push_back(d);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 8; ++rep) {
            if (which == 0) hipLaunchKernelGGL(k_split, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, sets[rep % SETS], n, w, m, v);
            else hipLaunchKernelGGL(k_inter, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, sets[rep % SETS], n, t);
        }
        hipDeviceSynchronize();
        const int reps = 64;
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < reps; ++rep) {
            if (which == 0) hipLaunchKernelGGL(k_split, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, sets[rep % SETS], n, w, m, v);
            else hipLaunchKernelGGL(k_inter, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, sets[rep % SETS], n, t);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %d rows, %.2f us per pass (read + write of w, m, v: %.1f MB useful -> %.0f GB/s)\n", which ? "one array of 192-byte rows [w|m|v]" : "three arrays of 64-byte rows      ",
               n, ms * 1e3 / reps, n * 384.0 / 1e6, n * 384.0 / (ms * 1e-3 / reps) / 1e9);
    }
    return 0;
}
