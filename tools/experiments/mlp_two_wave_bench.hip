#include "rgn_internal.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
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
    long long t[64]; ml_prof_read(t);
    for (int w = 0; w < 2; ++w) {
        const long long* u = t + 32 * w; const long long z = t[0];
        printf("wave %d: tiles in %lld | out_proj %lld | LN1 %lld | LN2 %lld | img+bar %lld | L1(0) %lld | gelu+img %lld | bar %lld | L2(0) %lld | L1(1) %lld | gelu+img+2bar %lld | L2(1) %lld | LN3+store %lld | total %lld\n",
               4 * w, u[1] - z, u[2] - u[1], u[15] - u[2], u[17] - u[15], u[3] - u[17], u[7] - u[3], u[9] - u[7], u[10] - u[9], u[6] - u[10], u[11] - u[6], u[12] - u[11], u[4] - u[12], u[5] - u[4], u[5] - z);
    }
    return 0;
}
