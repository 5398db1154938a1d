// Stand-alone micro-benchmark + refcheck of the row-complete plain-bf16 GEMM (tools only; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I regennet_amd/csrc tools/rowgemm_bench.hip regennet_amd/csrc/rgn_rowgemm.hip -o tools/bin/rowgemm_bench
//   (-DRGN_RG_PROF=<workgroup> adds per-phase cycle stamps of that workgroup)
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace rgn;
#ifdef RGN_RG_PROF
namespace rgn { void rg_prof_read(long long* out); }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    struct Shape { int M, N, K; bool ln, two; const char* name; };
    std::vector<Shape> shapes = {{15360, 512, 512, true, true, "out+LN1+LN2"}, {15360, 1024, 512, false, false, "ffn1+gelu"},
                                 {15360, 512, 1024, true, false, "ffn2+LN3"}, {3840, 512, 512, true, true, "out+LN x1/4"},
                                 {3840, 1024, 512, false, false, "ffn1 x1/4"}, {3840, 512, 1024, true, false, "ffn2+LN x1/4"},
                                 {100, 512, 512, true, true, "ragged LN"}, {100, 1024, 512, false, false, "ragged act"}};
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    CK(configure_rowgemm());
    for (auto sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, Tq = 60;
        std::vector<float> A((size_t)M * K), W((size_t)N * K), bias(N), ga(N), ba(N), gb(N), bb(N), R((size_t)M * N), sv(N), pv((size_t)(M / Tq + 1) * N);
        for (auto& v : A) v = bf2f(f2bf(U(rng)));
        for (auto& v : W) v = bf2f(f2bf(U(rng) * 0.1f));
        for (auto& v : bias) v = U(rng);
        for (auto& v : ga) v = 1.f + 0.1f * U(rng);
        for (auto& v : ba) v = 0.1f * U(rng);
        for (auto& v : gb) v = 1.f + 0.1f * U(rng);
        for (auto& v : bb) v = 0.1f * U(rng);
        for (auto& v : R) v = U(rng);
        for (auto& v : sv) v = U(rng);
        for (auto& v : pv) v = U(rng);
        std::vector<uint16_t> Ah((size_t)M * K), Wh((size_t)N * K), Rh((size_t)M * ((N + 31) / 32 * 32)), Rl((size_t)M * ((N + 31) / 32 * 32));
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) Ah[((size_t)(k / 32) * M + m) * 32 + k % 32] = f2bf(A[(size_t)m * K + k]);
        for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) { const size_t kt = k / 32, ks = (k % 32) / 16, lane = 32 * ((k % 16) / 8) + n % 32; Wh[(((kt * (N / 32) + n / 32) * 2 + ks) * 64 + lane) * 8 + k % 8] = f2bf(W[(size_t)n * K + k]); }
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { float v = R[(size_t)m * N + n]; uint16_t h = f2bf(v); size_t o = ((size_t)(n / 32) * M + m) * 32 + n % 32; Rh[o] = h; Rl[o] = f2bf(v - bf2f(h)); }
        __bf16 *dA, *dW, *dRh, *dRl, *dCh, *dCl; float *dB, *dga, *dba, *dgb, *dbb, *dsv, *dpv, *dC; int* dstep;
        const size_t Np = (size_t)(N + 31) / 32 * 32;
        CK(hipMalloc(&dA, Ah.size() * 2)); CK(hipMalloc(&dW, Wh.size() * 2)); CK(hipMalloc(&dRh, (size_t)M * Np * 2)); CK(hipMalloc(&dRl, (size_t)M * Np * 2));
        CK(hipMalloc(&dCh, (size_t)M * Np * 2)); CK(hipMalloc(&dCl, (size_t)M * Np * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMalloc(&dB, N * 4)); CK(hipMalloc(&dga, N * 4)); CK(hipMalloc(&dba, N * 4)); CK(hipMalloc(&dgb, N * 4)); CK(hipMalloc(&dbb, N * 4));
        CK(hipMalloc(&dsv, N * 4)); CK(hipMalloc(&dpv, pv.size() * 4)); CK(hipMalloc(&dstep, 4)); CK(hipMemset(dstep, 0, 4));
        CK(hipMemcpy(dA, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, Wh.data(), Wh.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, bias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dga, ga.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dba, ba.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgb, gb.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbb, bb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsv, sv.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dpv, pv.data(), pv.size() * 4, hipMemcpyHostToDevice));
        RowGemmArgs g{};
        g.A = dA; g.a_rows = M; g.W = dW; g.bias = dB; g.M = M; g.N = N; g.Kp = K;
        auto reset = [&] { if (sh.ln) { CK(hipMemcpy(dRh, Rh.data(), Rh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dRl, Rl.data(), Rl.size() * 2, hipMemcpyHostToDevice)); } };
        if (sh.ln) {
            g.Rhi = dRh; g.Rlo = getenv("HI_ONLY") ? nullptr : dRl; g.r_rows = M; g.Ohi = dRh; g.Olo = getenv("HI_ONLY") ? nullptr : dRl; g.o_rows = M;
            g.ga = dga; g.ba = dba; g.gb = sh.two ? dgb : nullptr; g.bb = sh.two ? dbb : nullptr;
            g.pervec = sh.two ? dpv : nullptr; g.ldper = N; g.stepvec = sh.two ? dsv : nullptr; g.ldstep = N; g.d_step = dstep; g.Tq = Tq;
        } else {
            g.act = 1; g.Chi = dCh; g.Clo = getenv("HI_ONLY") ? nullptr : dCl; g.c_rows = M;
        }
        reset();
        CK(launch_rowgemm(g, sh.ln, nullptr));
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> Oh((size_t)M * Np), Ol((size_t)M * Np);
        CK(hipMemcpy(Oh.data(), sh.ln ? dRh : dCh, Oh.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(Ol.data(), sh.ln ? dRl : dCl, Ol.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0;
        std::mt19937 pick(7);
        for (int c = 0; c < 48; ++c) {
            const int m = c < 8 ? M - 1 - c : (int)(pick() % M);
            std::vector<double> t(N);
            for (int n = 0; n < N; ++n) { double r = bias[n]; for (int k = 0; k < K; ++k) r += (double)A[(size_t)m * K + k] * W[(size_t)n * K + k]; t[n] = r; }
            if (sh.ln) {
                auto ln = [&](const std::vector<float>& gam, const std::vector<float>& bet) { double mu = 0, q = 0; for (double v : t) mu += v; mu /= N; for (double v : t) q += (v - mu) * (v - mu); const double rs = 1.0 / sqrt(q / N + 1e-5); for (int n = 0; n < N; ++n) t[n] = (t[n] - mu) * rs * gam[n] + bet[n]; };
                for (int n = 0; n < N; ++n) t[n] += R[(size_t)m * N + n];
                ln(ga, ba);
                if (sh.two) { for (int n = 0; n < N; ++n) t[n] += sv[n] + pv[(size_t)(m / Tq) * N + n]; ln(gb, bb); }
            } else if (g.act == 1) for (int n = 0; n < N; ++n) t[n] = 0.5 * t[n] * (1.0 + erf(t[n] * 0.7071067811865476));
            for (int n = 0; n < N; ++n) { const size_t o = ((size_t)(n / 32) * M + m) * 32 + n % 32; maxerr = fmax(maxerr, fabs(t[n] - ((double)bf2f(Oh[o]) + ((sh.ln || !getenv("HI_ONLY")) ? (double)bf2f(Ol[o]) : 0.0)))); }
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) CK(launch_rowgemm(g, sh.ln, nullptr));
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) CK(launch_rowgemm(g, sh.ln, nullptr));
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / iters, fl = 2.0 * M * N * K;
        printf("%-14s M=%5d N=%4d K=%4d  %8.1f us  %7.1f TF  maxerr %.2e\n", sh.name, M, N, K, us, fl / us * 1e-6, maxerr);
#ifdef RGN_RG_PROF
        std::vector<long long> pr(8 * 16);
        rg_prof_read(pr.data());
        for (int w : {0, 7}) {
            const long long* t = pr.data() + w * 16;
            printf("   wave %d cycles: issue A+W %lld | A landed %lld | barrier %lld | k-loop %lld | T5-T4 (LN: barrier+park | ACT: gelu+image) %lld | T6-T5 (vectors+barrier | barrier) %lld | T7-T6 (row phase | copy-out) %lld | drain %lld | total %lld\n", w,
                   t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6], t[8] - t[7], t[8] - t[0]);
        }
#endif
        hipFree(dA); hipFree(dW); hipFree(dRh); hipFree(dRl); hipFree(dCh); hipFree(dCl); hipFree(dC); hipFree(dB); hipFree(dga); hipFree(dba); hipFree(dgb); hipFree(dbb); hipFree(dsv); hipFree(dpv); hipFree(dstep);
    }
    return 0;
}
