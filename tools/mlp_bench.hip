// Stand-alone micro-benchmark of the row-persistent layer-tail kernel (tools only; refcheck lives in the parity tests).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRGN_ML_PROF=5] -I regennet_amd/csrc tools/mlp_bench.hip regennet_amd/csrc/rgn_mlp.hip -o tools/bin/mlp_bench
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace rgn;
#ifdef RGN_ML_PROF
namespace rgn { void ml_prof_read(long long* out); }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 15360, iters = argc > 2 ? atoi(argv[2]) : 20, d = 512, ff = 1024, Tq = 60;
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
    for (int i = 0; i < 3; ++i) CK(launch_mlp(g, nullptr));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) CK(launch_mlp(g, nullptr));
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters, fl = 2.0 * M * (d * d + 2.0 * d * ff);
    printf("k_mlp M=%d: %.1f us  %.1f TF\n", M, us, fl / us * 1e-6);
#ifdef RGN_ML_PROF
    long long t[16]; ml_prof_read(t);
    printf("  cycles: tile DMA + wait %lld | out_proj k-loop %lld | LN1+vec+LN2+image %lld | ffn (2 x (linear1, gelu, linear2)) %lld | LN3 + store %lld | total %lld\n",
           t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
#endif
    return 0;
}
