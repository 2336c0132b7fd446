// Development probe (not part of the library): times cdc_gemm_bf16_nt on the C2 shapes with parts of the kernel compiled away
// (-DG2_PROBE=bits, see csrc/gemm2.hip), to see what a launch is waiting for.  Build (tools/build_probes.sh):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Icausal-..._amd/csrc -DG2_PROBE=N
//         tools/gemm2_probe.hip causal-..._amd/csrc/misc.hip -o tools/_build/gemm2_probe_N
// Run:  gemm2_probe_N <case> [write_f32=1] [write_bf16=1] [dropout=0.2]
//   case fwd1: level-1 experts + gates of PLE-3 (x [4096,416] -> 8 x 256 relu/dropout columns + gates 4,4,4,8)
//        fwd2: second expert layer (8 groups [4096,256] -> 128)
//        bwdx2: grad-input of fwd2 (8 outputs [4096,256], K-reduction 128, mask)
//        bwdx1: grad-input of fwd1 (ONE output [4096,416], 12 segments, reduction 2048+20)
#include "gemm2.hip"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static __bf16* dev_bf16(size_t n, float scale) {
    std::vector<__bf16> h(n);
    for (auto& v : h) v = (__bf16)((rand() % 2001 - 1000) * scale);
    __bf16* d;
    hipMalloc(&d, n * 2);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    const char* which = argc > 1 ? argv[1] : "fwd1";
    const int wf = argc > 2 ? atoi(argv[2]) : 1, wh = argc > 3 ? atoi(argv[3]) : 1;
    const float drop = argc > 4 ? atof(argv[4]) : 0.2f;
    const int tile_cfg = argc > 5 ? atoi(argv[5]) : 0;
    const int M = 4096;
    cdc_g2_args a = {};
    double flops = 0;
    a.seed = 1234;
    a.tile_cfg = tile_cfg;
    if (!strcmp(which, "fwd1") || !strcmp(which, "fwd2") || !strcmp(which, "fwd1x")) {
        const bool l1 = strcmp(which, "fwd2") != 0;
        const int K = l1 ? 416 : 256, Kp = (K + 63) / 64 * 64, G = 8, N = l1 ? 256 : 128;
        const int gates[4] = {4, 4, 4, 8};
        const int ldx = l1 ? Kp + 64 : G * K + 64;
        __bf16* x = dev_bf16((size_t)M * ldx, 1e-3f);
        float* y; hipMalloc(&y, (size_t)M * (G * N + 64) * 4);
        __bf16* yh; hipMalloc(&yh, (size_t)M * (G * N + 64) * 2);
        float* bias; hipMalloc(&bias, 4096 * 4); hipMemset(bias, 0, 4096 * 4);
        a.mode = 0; a.relu = 1; a.drop_p = drop;
        int o = 0;
        for (int g = 0; g < G; ++g, ++o) {
            a.o[o].y = wf ? y + g * N : nullptr; a.o[o].ldy = G * N + 64;
            a.o[o].yh = wh ? (void*)(yh + g * N) : nullptr; a.o[o].ldyh = G * N + 64;
            if (!wf && !wh) a.o[o].y = y + g * N;
            a.o[o].bias = bias + g * N; a.o[o].M = M; a.o[o].N = N; a.o[o].act_cols = N; a.o[o].stream_id = o;
            a.s[o].a = l1 ? x : x + g * K; a.s[o].lda = ldx; a.s[o].b = dev_bf16((size_t)N * Kp, 1e-4f); a.s[o].ldb = Kp; a.s[o].Kr = Kp; a.s[o].out = o;
            flops += 2.0 * M * N * K;
        }
        if (!strcmp(which, "fwd1")) {
            float* gy; hipMalloc(&gy, (size_t)M * 32 * 4);
            int c0 = 0;
            for (int g = 0; g < 4; ++g, ++o) {
                a.o[o].y = gy + c0; a.o[o].ldy = 32; a.o[o].bias = bias; a.o[o].M = M; a.o[o].N = gates[g]; a.o[o].stream_id = o;
                a.s[o].a = x; a.s[o].lda = ldx; a.s[o].b = dev_bf16((size_t)gates[g] * Kp, 1e-4f); a.s[o].ldb = Kp; a.s[o].Kr = Kp; a.s[o].out = o;
                c0 += gates[g];
                flops += 2.0 * M * gates[g] * K;
            }
        }
        a.n_out = a.n_seg = o;
    } else if (!strcmp(which, "bwdx2")) {
        const int G = 8, N = 128, K = 256, Np = 128;
        __bf16* dz = dev_bf16((size_t)M * (G * N + 64), 1e-3f);
        float* dx; hipMalloc(&dx, (size_t)M * (G * K + 64) * 4);
        __bf16* dxh; hipMalloc(&dxh, (size_t)M * (G * K + 64) * 2);
        float* mk; hipMalloc(&mk, (size_t)M * (G * K + 64) * 4); hipMemset(mk, 0x3f, (size_t)M * (G * K + 64) * 4);
        a.mode = 1; a.mask_scale = 1.25f;
        for (int g = 0; g < G; ++g) {
            a.o[g].y = wf ? dx + g * K : nullptr; a.o[g].ldy = G * K + 64; a.o[g].yh = wh ? (void*)(dxh + g * K) : nullptr; a.o[g].ldyh = G * K + 64;
            if (!wf && !wh) a.o[g].y = dx + g * K;
            a.o[g].mask = mk + g * K; a.o[g].ldmask = G * K + 64; a.o[g].M = M; a.o[g].N = K; a.o[g].act_cols = K;
            a.s[g].a = dz + g * N; a.s[g].lda = G * N + 64; a.s[g].b = dev_bf16((size_t)K * Np, 1e-4f); a.s[g].ldb = Np; a.s[g].Kr = Np; a.s[g].out = g;
            flops += 2.0 * M * N * K;
        }
        a.n_out = a.n_seg = G;
    } else {
        const int G = 8, N = 256, K = 416;
        const int gates[4] = {4, 4, 4, 8};
        __bf16* dz = dev_bf16((size_t)M * (G * N + 64), 1e-3f);
        __bf16* dg = dev_bf16((size_t)M * (4 * 64 + 64), 1e-3f);
        float* dx; hipMalloc(&dx, (size_t)M * 448 * 4);
        a.mode = 1; a.mask_scale = 1.f;
        a.o[0].y = dx; a.o[0].ldy = 416; a.o[0].M = M; a.o[0].N = K;
        int s = 0;
        for (int g = 0; g < G; ++g, ++s) {
            a.s[s].a = dz + g * N; a.s[s].lda = G * N + 64; a.s[s].b = dev_bf16((size_t)K * N, 1e-4f); a.s[s].ldb = N; a.s[s].Kr = N; a.s[s].out = 0;
            flops += 2.0 * M * N * K;
        }
        for (int g = 0; g < 4; ++g, ++s) {
            a.s[s].a = dg + g * 64; a.s[s].lda = 4 * 64 + 64; a.s[s].b = dev_bf16((size_t)K * 64, 1e-4f); a.s[s].ldb = 64; a.s[s].Kr = 64; a.s[s].out = 0;
            flops += 2.0 * M * gates[g] * K;
        }
        a.n_out = 1; a.n_seg = s;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) {
        int rc = cdc_gemm_bf16_nt(&a, nullptr);
        if (rc) { printf("rc %d: %s\n", rc, cdc_last_error()); return 1; }
    }
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) cdc_gemm_bf16_nt(&a, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("probe %2d %-6s cfg %2d f32 %d bf16 %d drop %.1f: %7.2f us/launch  %6.1f TFLOP/s\n", G2_PROBE, which, tile_cfg, wf, wh, drop, ms * 1e3 / reps,
           flops / (ms * 1e-3 / reps) / 1e12);
    return 0;
}
