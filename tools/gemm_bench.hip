// Stand-alone micro-benchmark + refcheck of the split-bf16 GEMM (tools only; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRGN_GEMM_TOOLS -I regennet_amd/csrc tools/gemm_bench.hip regennet_amd/csrc/rgn_gemm_x3.hip -o tools/bin/gemm_bench
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace rgn;
#ifdef RGN_GEMM_PROF
namespace rgn { void gemm_prof_read(long long* out); }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    int variant = argc > 1 ? atoi(argv[1]) : 0;
    int iters = argc > 2 ? atoi(argv[2]) : 20;
    const bool x3 = getenv("BF16") == nullptr;   // BF16=1: the plain-bf16 phase kernels (hi planes only)
    struct Shape { int M, N, K; const char* name; };
    std::vector<Shape> shapes = {{15360, 1536, 512, "qkv"}, {15360, 512, 512, "out_proj"}, {15360, 1024, 512, "ffn1"},
                                 {15360, 512, 1024, "ffn2"}, {15360, 336, 512, "pose_out"}, {15360, 512, 336, "pose_in"},
                                 {150, 200, 72, "ragged"}, {60, 512, 512, "b1_out"}, {60, 1024, 512, "b1_ffn1"}, {60, 512, 1024, "b1_ffn2"}, {600, 512, 512, "b10_out"}, {15360, 512, 32, "fixed_k32"}, {15360, 512, 64, "fixed_k64"}, {15360, 1536, 32, "fixedq_k32"}};
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    double tot_us = 0, tot_fl = 0;
    for (auto sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, Kp = (K + 31) / 32 * 32;
        std::vector<float> A((size_t)M * K), W((size_t)N * K), bias(N);
        const bool zero = getenv("ZERO") != nullptr;
        for (auto& v : A) v = zero ? 0.f : U(rng);
        for (auto& v : W) v = zero ? 0.f : U(rng) * 0.1f;
        for (auto& v : bias) v = U(rng);
        std::vector<uint16_t> Ah((size_t)M * Kp, 0), Al((size_t)M * Kp, 0), Wh((size_t)N * Kp, 0), Wl((size_t)N * Kp, 0);
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) { float v = A[(size_t)m * K + k]; uint16_t h = f2bf(v); size_t o = ((size_t)(k / 32) * M + m) * 32 + k % 32; Ah[o] = h; Al[o] = f2bf(v - bf2f(h)); }
        for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) { float v = W[(size_t)n * K + k]; uint16_t h = f2bf(v); size_t o = ((size_t)(k / 32) * N + n) * 32 + k % 32; Wh[o] = h; Wl[o] = f2bf(v - bf2f(h)); }
        __bf16 *dAh, *dAl, *dWh, *dWl; float *dC, *dB;
        CK(hipMalloc(&dAh, Ah.size() * 2)); CK(hipMalloc(&dAl, Al.size() * 2)); CK(hipMalloc(&dWh, Wh.size() * 2)); CK(hipMalloc(&dWl, Wl.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dB, N * 4));
        CK(hipMemcpy(dAh, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dAl, Al.data(), Al.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dWh, Wh.data(), Wh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dWl, Wl.data(), Wl.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, bias.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
        CK(configure_gemm_x3());
        GemmX3Args g{};
        g.Ahi = dAh; g.Alo = dAl; g.a_rows = M; g.Whi = dWh; g.Wlo = dWl; g.bias = dB; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.Kp = Kp;
        __bf16 *dCh = nullptr, *dCl = nullptr;
        const bool planes_out = getenv("PLANES") != nullptr;
        if (planes_out) { CK(hipMalloc(&dCh, (size_t)M * ((N + 31) / 32 * 32) * 2)); CK(hipMalloc(&dCl, (size_t)M * ((N + 31) / 32 * 32) * 2)); g.Chi = dCh; g.Clo = dCl; g.c_rows = M; g.act = 1; }
        CK(launch_gemm_x3(g, x3, variant, nullptr));
        CK(hipDeviceSynchronize());
        std::vector<float> C((size_t)M * N);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        std::vector<uint16_t> Ph, Pl;
        if (planes_out) { Ph.resize((size_t)M * ((N + 31) / 32 * 32)); Pl.resize(Ph.size()); CK(hipMemcpy(Ph.data(), dCh, Ph.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(Pl.data(), dCl, Pl.size() * 2, hipMemcpyDeviceToHost)); }
        double maxerr = 0, maxref = 0;
        std::mt19937 pick(7);
        const int nchk = (M * (long)N < 100000) ? M * N : 4000;
        for (int c = 0; c < nchk; ++c) {
            int m, n;
            if (nchk == M * N) { m = c / N; n = c % N; } else { m = pick() % M; n = pick() % N; if (c < 64) { m = M - 1 - (c % 8); n = N - 1 - (c / 8); } }
            double ref = bias[n];
            for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * W[(size_t)n * K + k];
            if (planes_out) { ref = 0.5 * ref * (1.0 + erf(ref * 0.7071067811865476)); size_t o = ((size_t)(n / 32) * M + m) * 32 + n % 32; double got = (double)bf2f(Ph[o]) + (double)bf2f(Pl[o]); maxerr = fmax(maxerr, fabs(ref - got)); maxerr = fmax(maxerr, fabs(ref - C[(size_t)m * N + n])); }
            else maxerr = fmax(maxerr, fabs(ref - C[(size_t)m * N + n]));
            maxref = fmax(maxref, fabs(ref));
        }
        if (getenv("PLANES_ONLY") && planes_out) g.C = nullptr;   // time the in-model linear1 form (planes out only)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) CK(launch_gemm_x3(g, x3, variant, nullptr));
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) CK(launch_gemm_x3(g, x3, variant, nullptr));
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / iters, fl = 2.0 * M * N * K;
        printf("%-9s M=%5d N=%4d K=%4d  %8.1f us  %7.1f TF(alg) %7.1f TF(raw bf16)  maxerr %.2e (ref max %.2f)\n", sh.name, M, N, K, us, fl / us * 1e-6, (x3 ? 3 : 1) * fl / us * 1e-6, maxerr, maxref);
        if (M > 1000 && N >= 512 && K >= 512) { tot_us += us; tot_fl += fl; }
#ifdef RGN_GEMM_PROF
        if (!strcmp(sh.name, getenv("PROF_SHAPE") ? getenv("PROF_SHAPE") : "ffn1")) {
            std::vector<long long> pr(1024);
            gemm_prof_read(pr.data());
            for (int w = 0; w < 2; ++w) {
                printf("  wave %s: per k-step cycles [t1-t0 | t2-t1 | t3-t2 | t4-t3 | t5-t4 | total]\n", w ? "last" : "0");
                const long long* q = pr.data() + w * 512;
                for (int kt = 0; kt < Kp / 32; ++kt) {
                    const long long* t = q + kt * 6;
                    const long long nxt = kt + 1 < Kp / 32 ? t[6] : t[5];
                    printf("   kt %2d: %5lld %5lld %5lld %5lld %5lld | %5lld\n", kt, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], nxt - t[0]);
                }
            }
        }
#endif
        hipFree(dAh); hipFree(dAl); hipFree(dWh); hipFree(dWl); hipFree(dC); hipFree(dB);
    }
    printf("layer GEMMs: %.1f us, %.1f TF(alg)\n", tot_us, tot_fl / tot_us * 1e-6);
    return 0;
}
