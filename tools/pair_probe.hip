// Development probe (not part of the library): times cdc_expert_pair_fwd on the C2 shape (8 experts 416 -> 256 -> 128 on 4096 rows,
// riders 4,4,4,8) with parts of the kernel compiled away (-DPAIR_PROBE=bits, -DPAIR_NSTAGE=n, see csrc/pair.hip).
// Build + run: tools/build_pair_probes.sh (binaries in tools/_build/, git-ignored, travel with gpurun).
#include "pair.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>

static __bf16* dev_bf16(size_t n, float scale) {
    std::vector<__bf16> h(n);
    for (auto& v : h) v = (__bf16)((rand() % 2001 - 1000) * scale);
    __bf16* d;
    hipMalloc(&d, n * 2);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096;
    const float drop = argc > 2 ? atof(argv[2]) : 0.2f;
    const int riders = argc > 3 ? atoi(argv[3]) : 1;
    const int G = 8, K = 416, Kp = 448, H1 = 256, H2 = 128;
    const int gates[4] = {4, 4, 4, 8};
    cdc_expert_pair_args a = {};
    a.n_expert = G; a.M = M; a.K1r = Kp; a.H1 = H1; a.H2 = H2; a.relu = 1; a.drop_p = drop; a.seed1 = 1234; a.seed2 = 99;
    __bf16* x = dev_bf16((size_t)M * Kp, 1e-3f);
    __bf16* h; hipMalloc(&h, (size_t)M * G * H1 * 2);
    float* y; hipMalloc(&y, (size_t)M * G * H2 * 4);
    float* gy; hipMalloc(&gy, (size_t)M * 32 * 4);
    float* bias; hipMalloc(&bias, 4096 * 4); hipMemset(bias, 0, 4096 * 4);
    double flops = 0;
    int c0 = 0;
    for (int g = 0; g < G; ++g) {
        cdc_pair_expert& E = a.e[g];
        E.x = x; E.ldx = Kp;
        E.w1 = dev_bf16((size_t)H1 * Kp, 1e-4f); E.ldw1 = Kp; E.b1 = bias;
        E.w2 = dev_bf16((size_t)H2 * H1, 1e-4f); E.ldw2 = H1; E.b2 = bias;
        E.h = h + g * H1; E.ldh = G * H1;
        E.y = y + g * H2; E.ldy = G * H2;
        E.stream1 = E.stream2 = g;
        flops += 2.0 * M * (H1 * K + H2 * H1);
        if (riders && g < 4) {
            E.ws = dev_bf16((size_t)gates[g] * Kp, 1e-4f); E.ldws = Kp; E.bs = bias; E.ys = gy + c0; E.ldys = 32; E.ns = gates[g];
            c0 += gates[g];
            flops += 2.0 * M * gates[g] * K;
        }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) {
        int rc = cdc_expert_pair_fwd(&a, nullptr);
        if (rc) { printf("rc %d: %s\n", rc, cdc_last_error()); return 1; }
    }
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) cdc_expert_pair_fwd(&a, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("probe %2d nstage %d M %5d drop %.1f riders %d: %7.2f us/launch  %6.1f TFLOP/s\n", PAIR_PROBE, PAIR_NSTAGE, M, drop, riders,
           ms * 1e3 / reps, flops / (ms * 1e-3 / reps) / 1e12);
    return 0;
}
