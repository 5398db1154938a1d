// Stand-alone micro-benchmark of the fused in_proj + attention kernel (tools only; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRGN_QA_PROF=100] -I regennet_amd/csrc tools/qkv_attn_bench.hip \
//         regennet_amd/csrc/rgn_qkv_attn.hip -o tools/bin/qkv_attn_bench
// Times k_qkv_attn at Bm samples x Tq tokens (default 256 x 60, d = 512, H = 4); with -DRGN_QA_PROF=<block> it also
// prints the cycle stamps of that workgroup's phases (GEMM loop / operand split + partial scores / reduction + softmax + PV).
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace rgn;
#ifdef RGN_QA_PROF
namespace rgn { void qa_prof_read(long long* out); }
#endif
#ifdef RGN_QL_PROF
namespace rgn { void ql_prof_read(long long* out); }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    const int Bm = argc > 1 ? atoi(argv[1]) : 256, Tq = argc > 2 ? atoi(argv[2]) : 60, iters = argc > 3 ? atoi(argv[3]) : 50;
    const int d = 512, H = 4, M = Bm * Tq, Kp = d;
    const bool x3 = getenv("BF16") == nullptr;   // BF16=1: the plain-bf16 phase build (hi planes only)
    std::mt19937 rng(1);
    std::uniform_int_distribution<int> U(0x3c00, 0x3fff);   // bf16 bit patterns in [0.0078, 2)
    auto fill = [&](size_t n) { std::vector<uint16_t> v(n); for (auto& x : v) x = (uint16_t)U(rng); return v; };
    auto up = [&](const std::vector<uint16_t>& v) { void* p; CK(hipMalloc(&p, v.size() * 2)); CK(hipMemcpy(p, v.data(), v.size() * 2, hipMemcpyHostToDevice)); return (__bf16*)p; };
    __bf16 *Ahi = up(fill((size_t)M * Kp)), *Alo = up(fill((size_t)M * Kp)), *Whi = up(fill((size_t)3 * d * Kp)), *Wlo = up(fill((size_t)3 * d * Kp));
    __bf16 *Ohi, *Olo; float* bias;
    CK(hipMalloc(&Ohi, (size_t)M * d * 2)); CK(hipMalloc(&Olo, (size_t)M * d * 2)); CK(hipMalloc(&bias, 3 * d * 4)); CK(hipMemset(bias, 0, 3 * d * 4));
    QkvAttnArgs g{};
    g.Ahi = Ahi; g.Alo = Alo; g.a_rows = M; g.Whi = Whi; g.Wlo = Wlo; g.bias = bias;
    if (getenv("RS")) g.Wfr = Whi;   // RS=1 (with BF16=1): the register-streamed weight loop (any bytes do for timing)
    g.out.hi = Ohi; g.out.lo = x3 ? Olo : nullptr; g.out.rows = M; g.Bm = Bm; g.Kp = Kp; g.d = d; g.H = H; g.Tq = Tq; g.qscale = 0.0884f;
    CK(configure_qkv_attn());
    const bool longk = getenv("LONG") != nullptr;   // LONG=1 BF16=1: k_qkv_attn_long (give Tq = 150)
    if (longk) { CK(configure_qkv_attn_long()); g.Wfr = Whi; g.out.lo = nullptr; }
    auto launch = [&] { return longk ? launch_qkv_attn_long(g, nullptr) : launch_qkv_attn(g, x3, nullptr); };
    for (int i = 0; i < 3; ++i) CK(launch());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) CK(launch());
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters, fl = 2.0 * M * 3 * d * d + 4.0 * Bm * H * Tq * Tq * 128;
    printf("k_qkv_attn Bm=%d Tq=%d: %.1f us  %.1f TF(alg)\n", Bm, Tq, us, fl / us * 1e-6);
#ifdef RGN_QA_PROF
    long long pr[64]; qa_prof_read(pr);
    for (int h = 0; h < 2; ++h) {
        const long long* t = pr + h * 8;
        printf("  head %d cycles: gemm %lld | split + S partials + barrier %lld | reduce %lld | softmax %lld | PV + store %lld | barrier %lld\n", h,
               t[1] - t[0], t[2] - t[1], t[4] - t[2], t[5] - t[4], t[6] - t[5], t[3] - t[6]);
    }
    printf("  total cycles %lld\n", pr[8 + 3] - pr[0]);
#endif
#ifdef RGN_QL_PROF
    if (longk) {
        long long t[16 * 8]; ql_prof_read(t);
        for (int w : {0, 4, 5, 11}) printf("  wave %2d cycles: gemm %lld | slabs + barrier %lld | attention %lld | barrier + output %lld | total %lld\n", w, t[w * 8 + 1] - t[w * 8], t[w * 8 + 2] - t[w * 8 + 1], t[w * 8 + 3] - t[w * 8 + 2], t[w * 8 + 4] - t[w * 8 + 3], t[w * 8 + 4] - t[w * 8]);
    }
#endif
    return 0;
}
