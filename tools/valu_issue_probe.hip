// valu_issue_probe.hip — how many SIMD cycles does a wave instruction occupy on gfx950, with 1, 2, 4 or 8 waves per SIMD?
//
// The lazy table's replay slice (csrc/embedding.hip k_lazy_flush, csrc/common.h adam_scaled_step_pk) is bound by the ISSUE of its
// vector instructions: per element pair and step 5 packed fp32 operations (v_pk_fma_f32 / v_pk_mul_f32) and two each of
// v_sqrt_f32 and v_rcp_f32.  bench.py prices that against `valu_cycles_per_64_element_steps`; this probe measures the prices:
// independent chains of ONE instruction kind, timed with the shader clock inside the wave, W waves per SIMD on every CU.
//
//     hipcc --offload-arch=gfx950 -O3 tools/valu_issue_probe.hip -o tools/_build/valu_issue_probe && tools/_build/valu_issue_probe
//
// Output: per instruction kind and W: cycles per instruction as ONE wave sees it, and SIMD cycles per wave instruction
// (= the former / W): the issue cost when the SIMD is kept busy.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAINS 16
#define ROUNDS 4096

template <int OP>
__global__ void __launch_bounds__(256) k_probe(unsigned long long* out, float seed) {
    float a[CHAINS];
    f2 p[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + (float)i + (float)threadIdx.x * 1e-3f; p[i] = (f2){a[i], a[i] + 0.5f}; }
    const float b = 1.0000001f, c = 1e-7f;
    const f2 b2 = {b, b}, c2 = {c, c};
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
            if (OP == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i] + p[i].x + p[i].y;
    if ((threadIdx.x & 63) == 0) {
        const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * wave] = t1 - t0;
        out[2 * wave + 1] = w1 - w0;
    }
    if (s == 123.456f) out[0] = 0;
}

template <int OP>
static void run(const char* name, int n_cu) {
    for (int W : {1, 2, 4, 8}) {
        const int blocks = n_cu * W;
        unsigned long long* d;
        hipMalloc(&d, sizeof(unsigned long long) * 2 * blocks * 4);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks * 4);
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ns;
        for (int i = 0; i < blocks * 4; ++i) { cyc.push_back((double)h[2 * i]); ns.push_back((double)h[2 * i + 1] * 10.0); }
        std::sort(cyc.begin(), cyc.end());
        std::sort(ns.begin(), ns.end());
        const double med = cyc[cyc.size() / 2], medns = ns[ns.size() / 2];
        const double per_wave = med / (double)(ROUNDS * CHAINS);
        // the whole launch by the host's events: wave instructions per SIMD and nanosecond (includes the launch's ~5 us of fixed cost)
        const double per_simd_ns = (double)ms * 1e6 / ((double)ROUNDS * CHAINS * W);
        printf("%-14s W=%d  per wave: %6.2f clock ticks, %6.2f ns per instruction | per SIMD: %5.2f ticks, %5.3f ns per wave instruction (in-wave clocks) ; "
               "%5.3f ns by the launch's event time | s_memtime ticks at %.0f MHz\n",
               name, W, per_wave, medns / (double)(ROUNDS * CHAINS), per_wave / W, medns / (double)(ROUNDS * CHAINS) / W, per_simd_ns, med / medns * 1e3);
        hipFree(d);
    }
}

int main() {
    int dev = 0, n_cu = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    printf("# %d CUs; %d independent chains per wave, %d rounds; W = workgroups of 4 waves per CU = waves per SIMD\n", n_cu, CHAINS, ROUNDS);
    run<0>("v_fma_f32", n_cu);
    run<4>("v_mul_f32", n_cu);
    run<1>("v_pk_fma_f32", n_cu);
    run<5>("v_pk_mul_f32", n_cu);
    run<2>("v_sqrt_f32", n_cu);
    run<3>("v_rcp_f32", n_cu);
    return 0;
}
