// Needs rgn_mlp.hip with g_ml_prof[4096] and
//   #define RGN_MT(i) if ((i) == 0 || (i) == 1 || (i) == 5) { if (threadIdx.x == 0 && blockIdx.x < 1024) g_ml_prof[((i) == 5 ? 2 : (i)) * 1024 + blockIdx.x] = wall_clock64(); }
#include "rgn_internal.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
using namespace rgn;
namespace rgn { void ml_prof_read(long long* out); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 15360, iters = 20, d = 512, ff = 1024, Tq = 60;
    std::mt19937 rng(1);
    std::uniform_int_distribution<int> U(0x3c00, 0x3eff);
    auto bf = [&](size_t n) { std::vector<uint16_t> v(n); for (auto& x : v) x = (uint16_t)(U(rng) | ((rng() & 1) << 15)); void* p; CK(hipMalloc(&p, n * 2)); CK(hipMemcpy(p, v.data(), n * 2, hipMemcpyHostToDevice)); return (__bf16*)p; };
    auto f32 = [&](size_t n, float s) { std::vector<float> v(n); std::uniform_real_distribution<float> R(-s, s); for (auto& x : v) x = R(rng); void* p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, v.data(), n * 4, hipMemcpyHostToDevice)); return (float*)p; };
    MlpArgs g{};
    g.att = bf((size_t)M * d); g.h = bf((size_t)M * d); g.out = bf((size_t)M * d); g.rows = M; g.M = M;
    g.Wo = bf((size_t)d * d); g.W1 = bf((size_t)ff * d); g.W2 = bf((size_t)d * ff);
    g.bo = f32(d, 0.1f); g.bf1 = f32(ff, 0.1f); g.bf2 = f32(d, 0.1f);
    g.g1 = f32(d, 1.f); g.b1 = f32(d, .1f); g.g2 = f32(d, 1.f); g.b2 = f32(d, .1f); g.g3 = f32(d, 1.f); g.b3 = f32(d, .1f);
    g.pervec = f32((size_t)(M / Tq + 1) * d, 1.f); g.ldper = d; g.stepvec = f32(d, 1.f); g.ldstep = d; g.Tq = Tq;
    int* ds; CK(hipMalloc(&ds, 4)); CK(hipMemset(ds, 0, 4)); g.d_step = ds;
    CK(configure_mlp());
    for (int i = 0; i < iters; ++i) CK(launch_mlp(g, nullptr));
    CK(hipDeviceSynchronize());
    static long long t[4096]; ml_prof_read(t);
    const int nb = (M + 63) / 64;
    long long t0 = t[0]; for (int b = 0; b < nb; ++b) t0 = std::min(t0, t[b]);
    std::vector<double> st, en, life, wait;
    for (int b = 0; b < nb; ++b) { st.push_back((t[b] - t0) * 0.01); en.push_back((t[2048 + b] - t0) * 0.01); life.push_back((t[2048 + b] - t[b]) * 0.01); wait.push_back((t[1024 + b] - t[b]) * 0.01); }
    auto q = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
    printf("%d WGs (us): start min %.2f med %.2f max %.2f | tile wait min %.2f med %.2f max %.2f | lifetime min %.2f med %.2f p90 %.2f max %.2f | end min %.2f med %.2f max %.2f\n", nb,
           q(st, 0), q(st, .5), q(st, 1), q(wait, 0), q(wait, .5), q(wait, 1), q(life, 0), q(life, .5), q(life, .9), q(life, 1), q(en, 0), q(en, .5), q(en, 1));
    for (int x = 0; x < 8; ++x) { double m = 0, w = 0; int n = 0; for (int b = x; b < nb; b += 8) { m = std::max(m, en[b]); w += wait[b]; ++n; } printf("  xcd %d: %d WGs, mean tile wait %.2f, last end %.2f\n", x, n, w / n, m); }
    return 0;
}
