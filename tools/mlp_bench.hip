// Stand-alone micro-benchmark + reference check of the row-persistent layer-tail kernel (tools only; the parity tests proper are
// tests/test_hip_parity.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DRGN_M2_STAMPS] -I regennet_amd/csrc tools/mlp_bench.hip regennet_amd/csrc/rgn_mlp2.hip -o tools/bin/mlp_bench      (REGENNET_MLP_ROWS=32: the two-workgroups-per-CU form)
//   mlp_bench [M] [iters] [check]      check = 1: compare the first and the last 64-row tile with an fp64 host evaluation of the
//                                      same bf16 inputs (the kernel rounds h', the GELU'd hidden tile and its output to bf16)
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace rgn;
#ifdef RGN_M2_STAMPS
namespace rgn { void m2_stamps_read(long long* out); }
#include <algorithm>
#include <map>
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 15360, iters = argc > 2 ? atoi(argv[2]) : 20, check = argc > 3 ? atoi(argv[3]) : 0;
    const int d = 512, ff = 1024, Tq = 60;
    std::mt19937 rng(1);
    std::normal_distribution<float> N01(0.f, 1.f);
    auto up = [&](const void* h, size_t bytes) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemcpy(p, h, bytes, hipMemcpyHostToDevice)); return p; };
    // activations: natural [M][d] values (bf16-rounded), stored K32-blocked [d / 32][M][32]
    auto act = [&](std::vector<float>& nat, float s) {
        nat.resize((size_t)M * d);
        std::vector<uint16_t> pl((size_t)M * d);
        for (int m = 0; m < M; ++m)
            for (int k = 0; k < d; ++k) {
                const uint16_t b = f2bf(s * N01(rng));
                nat[(size_t)m * d + k] = bf2f(b);
                pl[((size_t)(k / 32) * M + m) * 32 + k % 32] = b;
            }
        return (__bf16*)up(pl.data(), pl.size() * 2);
    };
    // weights: natural [N][K] values (bf16-rounded), stored in fragment order [K / 32][N / 32][2][64][8] (rgn_pack.cpp pack_linear)
    auto wgt = [&](std::vector<float>& nat, int N, int K, float gain) {
        nat.resize((size_t)N * K);
        std::vector<uint16_t> fr((size_t)N * K);
        const size_t nb = N / 32;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const uint16_t b = f2bf(gain / std::sqrt((float)K) * N01(rng));
                nat[(size_t)n * K + k] = bf2f(b);
                const size_t kt = k / 32, ks = (k % 32) / 16, lane = 32 * ((k % 16) / 8) + n % 32;
                fr[(((kt * nb + n / 32) * 2 + ks) * 64 + lane) * 8 + k % 8] = b;
            }
        return (__bf16*)up(fr.data(), fr.size() * 2);
    };
    auto vecf = [&](std::vector<float>& v, size_t n, float mean, float s) { v.resize(n); for (auto& x : v) x = mean + s * N01(rng); return (float*)up(v.data(), n * 4); };
    std::vector<float> att, h, Wo, W1, W2, bo, bf1, bf2, g1, b1, g2, b2, g3, b3, per, stepv;
    MlpArgs g{};
    g.att = act(att, 1.f); g.h = act(h, 1.f); g.rows = M; g.M = M;
    { void* p; CK(hipMalloc(&p, (size_t)M * d * 2)); CK(hipMemset(p, 0, (size_t)M * d * 2)); g.out = (__bf16*)p; }
    g.Wo = wgt(Wo, d, d, 1.f); g.W1 = wgt(W1, ff, d, 1.2f); g.W2 = wgt(W2, d, ff, 1.2f);
    g.bo = vecf(bo, d, 0.f, 0.05f); g.bf1 = vecf(bf1, ff, 0.f, 0.05f); g.bf2 = vecf(bf2, d, 0.f, 0.05f);
    g.g1 = vecf(g1, d, 1.f, .1f); g.b1 = vecf(b1, d, 0.f, .1f); g.g2 = vecf(g2, d, 1.f, .1f); g.b2 = vecf(b2, d, 0.f, .1f);
    g.g3 = vecf(g3, d, 1.f, .1f); g.b3 = vecf(b3, d, 0.f, .1f);
    const int nsamp = (M + Tq - 1) / Tq;
    g.pervec = vecf(per, (size_t)nsamp * d, 0.f, 0.5f); g.ldper = d; g.stepvec = vecf(stepv, d, 0.f, 0.5f); g.ldstep = d; g.Tq = Tq;
    int* ds; CK(hipMalloc(&ds, 4)); CK(hipMemset(ds, 0, 4)); g.d_step = ds;
    CK(configure_mlp());
    for (int i = 0; i < 3; ++i) CK(launch_mlp(g, nullptr));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) CK(launch_mlp(g, nullptr));
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters, fl = 2.0 * M * (d * d + 2.0 * d * ff);
    printf("k_mlp M=%d: %.1f us  %.1f TF\n", M, us, fl / us * 1e-6);
#ifdef RGN_M2_STAMPS
    {   // phase stamps of wave 0 of every workgroup, grouped by the CU it ran on (the last launch)
        std::vector<long long> st(1024 * 12);
        m2_stamps_read(st.data());
        const int trows = (getenv("REGENNET_MLP_ROWS") && atoi(getenv("REGENNET_MLP_ROWS")) == 32) ? 32 : 64;
        const int nwg = std::min(1024, (M + trows - 1) / trows);
        std::map<long long, std::vector<int>> by_cu;
        for (int b = 0; b < nwg; ++b) {
            const unsigned hw = (unsigned)st[b * 12 + 10];
            by_cu[(st[b * 12 + 11] << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)].push_back(b);
        }
        long long t00 = st[0];
        for (int b = 0; b < nwg; ++b) t00 = std::min(t00, st[b * 12]);
        int shown = 0, pairs = 0, singles = 0, more = 0;
        double ph[5] = {0, 0, 0, 0, 0}, life = 0, ovl = 0;
        for (auto& kv : by_cu) {
            auto& v = kv.second;
            if (v.size() == 1) ++singles; else if (v.size() == 2) ++pairs; else ++more;
            for (int b : v) {
                for (int i = 0; i < 5; ++i) ph[i] += (double)(st[b * 12 + i + 1] - st[b * 12 + i]);
                life += (double)(st[b * 12 + 5] - st[b * 12]);
            }
            if (v.size() == 2) {   // cycles during which both are inside an MFMA loop phase (stamps 1-2 and 3-4) at once
                auto inter = [&](int a0, int a1, int b0, int b1) { return (double)std::max(0LL, std::min(st[v[0] * 12 + a1], st[v[1] * 12 + b1]) - std::max(st[v[0] * 12 + a0], st[v[1] * 12 + b0])); };
                ovl += inter(1, 2, 1, 2) + inter(1, 2, 3, 4) + inter(3, 4, 1, 2) + inter(3, 4, 3, 4);
            }
            if (v.size() == 2 && shown < 6) {
                ++shown;
                printf("  CU %05llx:", kv.first);
                for (int b : v) {
                    printf("  wg %3d [", b);
                    for (int i = 0; i < 6; ++i) printf("%s%lld", i ? " " : "", st[b * 12 + i] - st[v[0] * 12]);
                    printf("]");
                }
                printf("\n");
            }
        }
        printf("  %d workgroups on %zu CUs: %d CUs with two, %d with one, %d with more\n", nwg, by_cu.size(), pairs, singles, more);
        double pro[2] = {0, 0};
        for (int b = 0; b < nwg; ++b) { pro[0] += (double)(st[b * 12 + 6] - st[b * 12]); pro[1] += (double)(st[b * 12 + 7] - st[b * 12]); }
        printf("  prologue (mean cycles from the start): DMA + first fragments issued %.0f | att tile landed (this wave) %.0f\n", pro[0] / nwg, pro[1] / nwg);
        printf("  mean cycles per workgroup (wave 0): tile wait %.0f | out_proj %.0f | res+LN1+LN2+image %.0f | ffn %.0f | res+LN3+store %.0f | lifetime %.0f; both-in-a-loop overlap per pair %.0f\n",
               ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg, ph[4] / nwg, life / nwg, pairs ? ovl / pairs : 0.0);
    }
#endif
    if (check) {
        std::vector<uint16_t> out((size_t)M * d);
        CK(hipMemcpy(out.data(), g.out, out.size() * 2, hipMemcpyDeviceToHost));
        auto ln = [&](std::vector<double>& x, const std::vector<float>& ga, const std::vector<float>* be) {
            double mu = 0, var = 0;
            for (double v : x) mu += v;
            mu /= x.size();
            for (double v : x) var += (v - mu) * (v - mu);
            const double r = 1.0 / std::sqrt(var / x.size() + 1e-5);
            for (size_t i = 0; i < x.size(); ++i) x[i] = (x[i] - mu) * r * ga[i] + (be ? (*be)[i] : 0.0);
        };
        double worst = 0, sum = 0; size_t cnt = 0;
        std::vector<int> rows;
        for (int m = 0; m < 64 && m < M; ++m) rows.push_back(m);
        for (int m = (M - 1) / 64 * 64; m < M; ++m) if (m >= 64) rows.push_back(m);
        if (M > 4096) for (int m = 2048 + 37; m < 2048 + 37 + 64; ++m) rows.push_back(m);
        for (int m : rows) {
            std::vector<double> x(d), hp(d), hid(ff), y(d);
            for (int n = 0; n < d; ++n) {
                double a = bo[n] + h[(size_t)m * d + n];
                for (int k = 0; k < d; ++k) a += (double)att[(size_t)m * d + k] * Wo[(size_t)n * d + k];
                x[n] = a;
            }
            ln(x, g1, nullptr);
            for (int n = 0; n < d; ++n) x[n] += b1[n] + stepv[n] + per[(size_t)(m / Tq) * d + n];
            ln(x, g2, &b2);
            for (int n = 0; n < d; ++n) hp[n] = bf2f(f2bf((float)x[n]));
            for (int j = 0; j < ff; ++j) {
                double a = bf1[j];
                for (int k = 0; k < d; ++k) a += hp[k] * W1[(size_t)j * d + k];
                hid[j] = bf2f(f2bf((float)(0.5 * a * (1.0 + std::erf(a / std::sqrt(2.0))))));
            }
            for (int n = 0; n < d; ++n) {
                double a = bf2[n] + hp[n];
                for (int j = 0; j < ff; ++j) a += hid[j] * W2[(size_t)n * ff + j];
                y[n] = a;
            }
            ln(y, g3, &b3);
            for (int n = 0; n < d; ++n) {
                const double got = bf2f(out[((size_t)(n / 32) * M + m) * 32 + n % 32]);
                const double e = std::fabs(got - y[n]);
                worst = e > worst ? e : worst; sum += e; ++cnt;
            }
        }
        printf("  check (%zu rows vs fp64 host): max abs err %.3e, mean abs err %.3e  %s\n", rows.size(), worst, sum / cnt,
               (worst < 6e-2 && sum / cnt < 6e-3) ? "OK" : "MISMATCH");
        return (worst < 6e-2 && sum / cnt < 6e-3) ? 0 : 2;
    }
    return 0;
}
