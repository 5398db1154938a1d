// Stand-alone micro-benchmark of the fused in_proj + attention kernel (tools only; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DRGN_QA_PROF=100] -I regennet_amd/csrc tools/qkv_attn_bench.hip \
//         regennet_amd/csrc/rgn_qkv_attn.hip regennet_amd/csrc/rgn_qkv_attn_long.hip -o tools/bin/qkv_attn_bench
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
#ifdef RGN_QA_LIFE
namespace rgn { void qa_life_read(long long* out, int n); }
#include <algorithm>
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
    {   // FNV-1a of the output plane: variants that must be bit-identical are compared by this
        std::vector<uint16_t> o((size_t)M * d);
        CK(hipMemcpy(o.data(), Ohi, o.size() * 2, hipMemcpyDeviceToHost));
        unsigned long long hsh = 1469598103934665603ull;
        for (uint16_t v : o) { hsh ^= v; hsh *= 1099511628211ull; }
        printf("  output hash %016llx\n", hsh);
    }
#ifdef RGN_QA_PROF
    long long pr[64]; qa_prof_read(pr);
    for (int h = 0; h < 2; ++h) {
        const long long* t = pr + h * 8;
        printf("  head %d cycles: gemm %lld | split + S partials + barrier %lld | reduce %lld | softmax %lld | PV + store %lld | barrier %lld\n", h,
               t[1] - t[0], t[2] - t[1], t[4] - t[2], t[5] - t[4], t[6] - t[5], t[3] - t[6]);
    }
    printf("  total cycles %lld\n", pr[8 + 3] - pr[0]);
    if (pr[32]) {
        printf("  prologue (RS build) cycles from entry: setup done %lld | head 0: ring requested %lld, k-block 0 in LDS %lld, barriers of k-steps 0-3 passed %lld %lld %lld %lld\n",
               pr[33] - pr[32], pr[34] - pr[32], pr[35] - pr[32], pr[36] - pr[32], pr[37] - pr[32], pr[38] - pr[32], pr[39] - pr[32]);
        printf("  head 1 from its start: ring requested %lld, k-block 0 in LDS %lld, barriers of k-steps 0-3 passed %lld %lld %lld %lld\n",
               pr[42] - pr[8], pr[43] - pr[8], pr[44] - pr[8], pr[45] - pr[8], pr[46] - pr[8], pr[47] - pr[8]);
    }
#endif
#ifdef RGN_QA_LIFE
    {   // -DRGN_QA_LIFE (RS=1 BF16=1): start / first MFMA operands landed / end of every workgroup of the last launch, 10 ns units
        const int nwg = (getenv("REGENNET_QKV_NS") && atoi(getenv("REGENNET_QKV_NS")) == 2) ? (Bm + 1) / 2 * 2 : Bm * (getenv("REGENNET_QKV_HSPLIT") ? atoi(getenv("REGENNET_QKV_HSPLIT")) : 2);   // grid = sample groups x 2 head halves
        std::vector<long long> t(nwg * 3); qa_life_read(t.data(), nwg * 3);
        long long t0 = t[0], t1 = t[2];
        for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, t[3 * i]); t1 = std::max(t1, t[3 * i + 2]); }
        std::vector<double> st, ld, life;
        for (int i = 0; i < nwg; ++i) { st.push_back(0.01 * (t[3 * i] - t0)); ld.push_back(0.01 * (t[3 * i + 1] - t[3 * i])); life.push_back(0.01 * (t[3 * i + 2] - t[3 * i])); }
        auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
        printf("  %d workgroups: first start -> last end %.2f us | start offset min/med/max %.2f %.2f %.2f | first operands landed after %.2f %.2f %.2f | lifetime %.2f %.2f %.2f\n",
               nwg, 0.01 * (t1 - t0), q(st, 0), q(st, .5), q(st, 1), q(ld, 0), q(ld, .5), q(ld, 1), q(life, 0), q(life, .5), q(life, 1));
    }
#endif
#ifdef RGN_QL_PROF
    if (longk) {
        long long t[16 * 8]; ql_prof_read(t);
        for (int w : {0, 4, 5, 11}) printf("  wave %2d cycles: gemm %lld | slabs + barrier %lld | attention %lld | barrier + output %lld | total %lld\n", w, t[w * 8 + 1] - t[w * 8], t[w * 8 + 2] - t[w * 8 + 1], t[w * 8 + 3] - t[w * 8 + 2], t[w * 8 + 4] - t[w * 8 + 3], t[w * 8 + 4] - t[w * 8]);
    }
#endif
    return 0;
}
