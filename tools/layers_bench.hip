// Stand-alone check + micro-benchmark of the one-kernel decoder stack (rgn_layers.hip) against the kernel-per-stage chain it replaces
// (k_qkv_attn_rs + k_mlp per layer) on the same random bf16 inputs (tools only; the parity tests proper are tests/test_hip_parity.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I regennet_amd/csrc tools/layers_bench.hip regennet_amd/csrc/rgn_layers.hip \
//         regennet_amd/csrc/rgn_qkv_attn.hip regennet_amd/csrc/rgn_qkv_attn_long.hip regennet_amd/csrc/rgn_mlp2.hip -o tools/bin/layers_bench
//   layers_bench [Bm] [Tq] [L] [iters] [steps]     steps > 0: also time k_layers<true> over that many complete sampler steps (synthetic schedule)
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace rgn;
#ifdef RGN_LY_STAMPS
namespace rgn { void ly_stamps_read(long long* out); }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int Bm = argc > 1 ? atoi(argv[1]) : 256, Tq = argc > 2 ? atoi(argv[2]) : 60, L = argc > 3 ? atoi(argv[3]) : 8, iters = argc > 4 ? atoi(argv[4]) : 20;
    const int d = 512, ff = 1024, H = 4, M = Bm * Tq;
    std::mt19937 rng(1);
    std::normal_distribution<float> N01(0.f, 1.f);
    auto up = [&](const void* h, size_t bytes) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemcpy(p, h, bytes, hipMemcpyHostToDevice)); return p; };
    auto wgt = [&](int N, int K, float gain) {   // fragment order [K / 32][N / 32][2][64][8] (rgn_pack.cpp pack_linear)
        std::vector<uint16_t> fr((size_t)N * K);
        const size_t nb = N / 32;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const size_t kt = k / 32, ks = (k % 32) / 16, lane = 32 * ((k % 16) / 8) + n % 32;
                fr[(((kt * nb + n / 32) * 2 + ks) * 64 + lane) * 8 + k % 8] = f2bf(gain / std::sqrt((float)K) * N01(rng));
            }
        return (__bf16*)up(fr.data(), fr.size() * 2);
    };
    auto vecf = [&](size_t n, float mean, float s) { std::vector<float> v(n); for (auto& x : v) x = mean + s * N01(rng); return (float*)up(v.data(), n * 4); };
    std::vector<uint16_t> h0((size_t)M * d);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < d; ++k) h0[((size_t)(k / 32) * M + m) * 32 + k % 32] = f2bf(N01(rng));
    __bf16* hA = (__bf16*)up(h0.data(), h0.size() * 2);   // chain
    __bf16* hB = (__bf16*)up(h0.data(), h0.size() * 2);   // fused
    __bf16* att; CK(hipMalloc(&att, (size_t)M * d * 2));
    LayersArgs ga{};
    ga.rows = M; ga.Bm = Bm; ga.Tq = Tq; ga.L = L;
    for (int l = 0; l < L; ++l) {
        LayerWts& t = ga.lw[l];
        t.Wqkv = wgt(3 * d, d, 1.f); t.Wo = wgt(d, d, 1.f); t.W1 = wgt(ff, d, 1.2f); t.W2 = wgt(d, ff, 1.2f);
        t.bqkv = vecf(3 * d, 0.f, 0.05f); t.bo = vecf(d, 0.f, 0.05f); t.bf1 = vecf(ff, 0.f, 0.05f); t.bf2 = vecf(d, 0.f, 0.05f);
        t.g1 = vecf(d, 1.f, .1f); t.b1 = vecf(d, 0.f, .1f); t.g2 = vecf(d, 1.f, .1f); t.b2 = vecf(d, 0.f, .1f); t.g3 = vecf(d, 1.f, .1f); t.b3 = vecf(d, 0.f, .1f);
    }
    if (getenv("SAMEW")) for (int l = 1; l < L; ++l) ga.lw[l] = ga.lw[0];   // every layer streams the SAME 4 MB: what the L2-cold first touch of a layer's weights costs
    const int Ld = L * d;
    ga.pervec = vecf((size_t)Bm * Ld, 0.f, 0.5f); ga.ldper = Ld; ga.stepvec = vecf((size_t)4 * Ld, 0.f, 0.5f); ga.ldstep = Ld;
    int* ds; CK(hipMalloc(&ds, 4)); { int one = 1; CK(hipMemcpy(ds, &one, 4, hipMemcpyHostToDevice)); } ga.d_step = ds;
    ga.qscale = 1.0f / std::sqrt(128.f);
    CK(configure_layers()); CK(configure_qkv_attn()); CK(configure_mlp());
    auto chain = [&](__bf16* h) {
        for (int l = 0; l < L; ++l) {
            const LayerWts& t = ga.lw[l];
            QkvAttnArgs q{};
            q.Ahi = h; q.a_rows = M; q.Wfr = t.Wqkv; q.bias = t.bqkv; q.out.hi = att; q.out.lo = nullptr; q.out.rows = M;
            q.Bm = Bm; q.Kp = d; q.d = d; q.H = H; q.Tq = Tq; q.qscale = ga.qscale; q.Bm_eval = Bm;
            CK(launch_qkv_attn(q, false, nullptr));
            MlpArgs m{};
            m.att = att; m.h = h; m.out = h; m.rows = M; m.M = M; m.Wo = t.Wo; m.W1 = t.W1; m.W2 = t.W2; m.bo = t.bo; m.bf1 = t.bf1; m.bf2 = t.bf2;
            m.g1 = t.g1; m.b1 = t.b1; m.g2 = t.g2; m.b2 = t.b2; m.g3 = t.g3; m.b3 = t.b3;
            m.pervec = ga.pervec + (size_t)l * d; m.ldper = Ld; m.stepvec = ga.stepvec + (size_t)l * d; m.ldstep = Ld; m.d_step = ds; m.Tq = Tq;
            CK(launch_mlp(m, nullptr));
        }
    };
    auto fused = [&](__bf16* h) { ga.h = h; ga.out = h; CK(launch_layers(ga, nullptr)); };
    chain(hA); fused(hB);
    CK(hipDeviceSynchronize());
    {
        std::vector<uint16_t> a((size_t)M * d), b((size_t)M * d);
        CK(hipMemcpy(a.data(), hA, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), hB, b.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0, sum = 0, ref = 0; size_t nan = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            const double x = bf2f(a[i]), y = bf2f(b[i]);
            if (!(y == y) || !(x == x)) { ++nan; continue; }
            worst = std::max(worst, std::fabs(x - y)); sum += std::fabs(x - y); ref += std::fabs(x);
        }
        printf("fused vs chain after %d layers (Bm=%d, Tq=%d): max abs diff %.3e, mean abs diff %.3e (mean |x| %.3f), NaN %zu  %s\n", L, Bm, Tq, worst, sum / a.size(), ref / a.size(), nan,
               (nan == 0 && sum / a.size() < 2e-2 * std::max(1, L)) ? "OK" : "MISMATCH");
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) chain(hA);
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_c = 1e3 * ms / iters;
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) fused(hB);
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_f = 1e3 * ms / iters;
        const double fl = (2.0 * M * (4.0 * d * d + 2.0 * d * ff) + 4.0 * Bm * H * Tq * Tq * 128.0) * L;
        printf("  %d layers: chain %.1f us (%.1f per layer, %.0f TF)   fused %.1f us (%.1f per layer, %.0f TF)\n", L, us_c, us_c / L, fl / us_c * 1e-6, us_f, us_f / L, fl / us_f * 1e-6);
    }
    const int nsteps = argc > 5 ? atoi(argv[5]) : 0;
    if (nsteps > 0) {   // whole sampler steps in one launch: stack + step boundary per sample (synthetic DDPM coefficients, Philox noise)
        const int F = 336, nb_out = 11;
        auto frag = [&](int NB, int KB, float gain) {   // a fragment-ordered plane [KB][NB][2][64][8] of random bf16
            std::vector<uint16_t> fr((size_t)KB * NB * 1024);
            for (auto& x : fr) x = f2bf(gain * N01(rng));
            return (__bf16*)up(fr.data(), fr.size() * 2);
        };
        ga.steps = nsteps;
        ga.Wout = frag(nb_out, 16, 1.f / std::sqrt(512.f)); ga.bout = vecf(F, 0.f, 0.05f); ga.F = F; ga.nb_out = nb_out;
        ga.Wx = frag(16, 11, 1.f / std::sqrt(336.f));
        { std::vector<uint16_t> c0((size_t)M * 512); for (auto& x : c0) x = f2bf(0.5f * N01(rng)); ga.c0 = (__bf16*)up(c0.data(), c0.size() * 2); }
        std::vector<StepCoef> tab(1024);
        for (auto& k : tab) { k.c1 = 0.05f; k.c2 = 0.94f; k.sig_ddpm = 0.05f; k.sr = 1.1f; k.srm1 = 0.45f; k.ca = 0.9f; k.cb = 0.4f; k.sig_ddim = 0.f; k.t_model = 0; }
        ga.tab = (StepCoef*)up(tab.data(), tab.size() * sizeof(StepCoef));
        float* x; CK(hipMalloc(&x, (size_t)Bm * F * Tq * 4));
        { std::vector<float> xh((size_t)Bm * F * Tq); for (auto& v : xh) v = N01(rng); CK(hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice)); }
        SampleParams sp{}; sp.x = x; sp.seed = 7; sp.first_index = 999; sp.sampler = 0;
        ga.sp = (SampleParams*)up(&sp, sizeof(sp));
        int* dsw; CK(hipMalloc(&dsw, 64)); ga.d_stepw = dsw; ga.d_step = dsw;
        ga.B = Bm; ga.s0 = 0; ga.h = hB; ga.out = hB;
        for (int rep = 0; rep < 3; ++rep) {
            int z[16] = {0}; z[0] = 999; CK(hipMemcpy(dsw, z, 64, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, nullptr));
            CK(launch_layers(ga, nullptr));
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            int after; CK(hipMemcpy(&after, dsw, 4, hipMemcpyDeviceToHost));
            printf("  %d steps in one launch: %.1f us per step (loop index 999 -> %d)\n", nsteps, 1e3 * ms / nsteps, after);
        }
        ga.steps = 0;
    }
#ifdef RGN_LY_STAMPS
    {
        std::vector<long long> st(1024 * 16); ly_stamps_read(st.data());
        const int nwg = std::min(1024, Bm);
        double ph[12] = {0};
        for (int b = 0; b < nwg; ++b) for (int i = 1; i < 12; ++i) ph[i] += (double)(st[b * 16 + i] - st[b * 16 + i - 1]);
        printf("  layer %d, mean cycles (wave 0): in_proj(0) %.0f | attention(0) %.0f | barrier %.0f | in_proj(1) %.0f | attention(1) %.0f | barrier %.0f | att image + barrier %.0f | out_proj %.0f | res+LN1+LN2+image %.0f | ffn %.0f | res+LN3+image %.0f | total %.0f\n",
               RGN_LY_STAMPS, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg, ph[4] / nwg, ph[5] / nwg, ph[6] / nwg, ph[7] / nwg, ph[8] / nwg, ph[9] / nwg, ph[10] / nwg, ph[11] / nwg,
               [&] { double t = 0; for (int b = 0; b < nwg; ++b) t += (double)(st[b * 16 + 11] - st[b * 16]); return t / nwg; }());
        if (nsteps == 0) {   // round 0's S phase in detail (stamps 12-15 sit between stamps 1 and 2)
            double d[5] = {0, 0, 0, 0, 0};
            for (int b = 0; b < nwg; ++b) {
                const long long* t = &st[b * 16];
                d[0] += (double)(t[12] - t[1]); d[1] += (double)(t[13] - t[12]); d[2] += (double)(t[14] - t[13]); d[3] += (double)(t[15] - t[14]); d[4] += (double)(t[2] - t[15]);
            }
            printf("  attention(0) in detail (wave 0): operands + S^T partials + writes %.0f | wait for every wave's partials %.0f | sums + softmax + p writes %.0f | wait %.0f | p reads + PV + pack %.0f\n",
                   d[0] / nwg, d[1] / nwg, d[2] / nwg, d[3] / nwg, d[4] / nwg);
        }
        if (nsteps > 1) {
            double sp4[4] = {0, 0, 0, 0};
            for (int b = 0; b < nwg; ++b) for (int i = 0; i < 3; ++i) sp4[i] += (double)(st[b * 16 + 13 + i] - st[b * 16 + 12 + i]);
            printf("  step boundary of step 1, mean cycles (wave 0): output projection + x0 tile %.0f | sampler update (Philox, x) %.0f | input embedding + image %.0f | total %.0f\n",
                   sp4[0] / nwg, sp4[1] / nwg, sp4[2] / nwg, (sp4[0] + sp4[1] + sp4[2]) / nwg);
        }
    }
#endif
    return 0;
}
