// Development probe (not part of the library): times cdc_glinear_fwd on the C2 level-1 shape (8 groups sharing one
// [4096,416] input, N=256 each) with parts of the main loop compiled away (-DCDC_GEMM_PROBE=bits), to see which part
// the launch is waiting for.  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude
//   -Icausal-…_amd/csrc -DCDC_GEMM_PROBE=N tools/gemm_probe.hip causal-…_amd/csrc/misc.hip -o gemm_probe_N
#include "gemm.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    const int M = 4096, K = argc > 1 ? atoi(argv[1]) : 416, N = argc > 2 ? atoi(argv[2]) : 256, G = argc > 3 ? atoi(argv[3]) : 8;
    const float drop = argc > 4 ? atof(argv[4]) : 0.2f;
    float *x, *w, *b, *y;
    hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&w, (size_t)G * N * K * 4); hipMalloc(&b, (size_t)G * N * 4);
    hipMalloc(&y, (size_t)M * G * N * 4);
    std::vector<float> h((size_t)M * K);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hw((size_t)G * N * K);
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, (size_t)G * N * 4);
    cdc_lin_fwd_args a = {};
    a.n_groups = G; a.relu = 1; a.drop_p = drop; a.seed = 1234;
    for (int g = 0; g < G; ++g) {
        a.g[g].x = x; a.g[g].ldx = K; a.g[g].w = w + (size_t)g * N * K; a.g[g].ldw = K; a.g[g].bias = b + g * N;
        a.g[g].y = y + g * N; a.g[g].ldy = (int64_t)G * N; a.g[g].M = M; a.g[g].N = N; a.g[g].K = K; a.g[g].act_cols = N;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) cdc_glinear_fwd(&a, CDC_PREC_BF16, nullptr);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) cdc_glinear_fwd(&a, CDC_PREC_BF16, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * M * K * (double)N * G;
    printf("probe %d  M %d K %d N %d x%d drop %.2f: %.2f us/launch  %.1f TFLOP/s\n", CDC_GEMM_PROBE, M, K, N, G, drop, ms * 1e3 / reps,
           fl / (ms * 1e-3 / reps) / 1e12);
    return 0;
}
