// Stand-alone check + timing of the small-batch column-split GEMM (rgn_sb.hip) against an fp64 host reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I regennet_amd/csrc tools/sb_bench.hip regennet_amd/csrc/rgn_sb.hip -o tools/bin/sb_bench
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace rgn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
template <typename T> static T* up(const std::vector<T>& v) { void* p; CK(hipMalloc(&p, v.size() * sizeof(T) + 256)); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return (T*)p; }
static std::mt19937 rng(7);
static std::vector<float> rnd(size_t n, float s) { std::vector<float> v(n); std::uniform_real_distribution<float> R(-s, s); for (auto& x : v) x = R(rng); return v; }
// K32-blocked hi/lo planes of a row-major [rows, K] matrix (K padded to Kp)
static void planes(const std::vector<float>& a, int rows, int K, int Kp, std::vector<uint16_t>& hi, std::vector<uint16_t>& lo) {
    hi.assign((size_t)rows * Kp, 0); lo.assign((size_t)rows * Kp, 0);
    for (int r = 0; r < rows; ++r) for (int k = 0; k < K; ++k) {
        const size_t o = ((size_t)(k / 32) * rows + r) * 32 + k % 32;
        hi[o] = f2bf(a[(size_t)r * K + k]); lo[o] = f2bf(a[(size_t)r * K + k] - bf2f(hi[o]));
    }
}
static double gelu(double v) { return 0.5 * v * (1.0 + erf(v / sqrt(2.0))); }

static double time_us(const SbArgs& g, int pre, int post, bool x3, int iters = 200) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK(launch_sb_gemm(g, pre, post, x3, nullptr));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) CK(launch_sb_gemm(g, pre, post, x3, nullptr));
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / iters;
}

// PRE 0 / POST 0: C = A.W^T + bias + resid
static void test_planes(int M, int N, int K, bool x3) {
    const int Kp = (K + 31) / 32 * 32;
    auto A = rnd((size_t)M * K, 1.f), W = rnd((size_t)N * K, 0.05f), bias = rnd(N, 0.1f), resid = rnd((size_t)M * N, 1.f);
    std::vector<uint16_t> ah, al, wh, wl;
    planes(A, M, K, Kp, ah, al); planes(W, N, K, Kp, wh, wl);
    SbArgs g{};
    g.Ahi = (__bf16*)up(ah); g.Alo = (__bf16*)up(al); g.a_rows = M;
    g.Whi = (__bf16*)up(wh); g.Wlo = (__bf16*)up(wl); g.w_rows = N;
    g.bias = up(bias); g.resid = up(resid); g.ldr = N; g.M = M; g.N = N; g.Kp = Kp;
    std::vector<float> C((size_t)M * N, -7.f);
    g.C = up(C); g.ldc = N;
    CK(launch_sb_gemm(g, 0, 0, x3, nullptr)); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), g.C, C.size() * 4, hipMemcpyDeviceToHost));
    double err = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
        double s = bias[n] + resid[(size_t)m * N + n];
        for (int k = 0; k < K; ++k) {
            const double a = x3 ? (double)A[(size_t)m * K + k] : bf2f(f2bf(A[(size_t)m * K + k])), w = x3 ? (double)W[(size_t)n * K + k] : bf2f(f2bf(W[(size_t)n * K + k]));
            s += a * w;
        }
        err = fmax(err, fabs(s - C[(size_t)m * N + n]));
    }
    printf("planes->f32  M=%4d N=%4d K=%4d %s: max err %.2e   %.2f us\n", M, N, K, x3 ? "x3  " : "bf16", err, time_us(g, 0, 0, x3));
}

// PRE 1: LN_b(LN_a(src) + stepvec + pervec) -> POST 0 / 1 / 2
static void test_ln(int M, int N, int post, bool two, bool x3, int Tq) {
    const int d = 512, H = 4, dh = 128, Tqp = (Tq + 31) / 32 * 32, nb = (M + Tq - 1) / Tq;
    auto src = rnd((size_t)M * d, 2.f), W = rnd((size_t)N * d, 0.05f), bias = rnd(N, 0.1f), ga = rnd(d, 1.f), ba = rnd(d, .2f), gb = rnd(d, 1.f), bb = rnd(d, .2f),
         pv = rnd((size_t)nb * d, 1.f), sv = rnd((size_t)3 * d, 1.f);
    std::vector<uint16_t> wh, wl;
    planes(W, N, d, d, wh, wl);
    SbArgs g{};
    g.src = up(src); g.ga = up(ga); g.ba = up(ba);
    if (two) { g.gb = up(gb); g.bb = up(bb); g.pervec = up(pv); g.ldper = d; g.stepvec = up(sv); g.ldstep = d; int one = 1; int* ds; CK(hipMalloc(&ds, 4)); CK(hipMemcpy(ds, &one, 4, hipMemcpyHostToDevice)); g.d_step = ds; }
    g.Tq = Tq;
    std::vector<float> xo((size_t)M * d, -7.f);
    g.xout = up(xo);
    g.Whi = (__bf16*)up(wh); g.Wlo = (__bf16*)up(wl); g.w_rows = N; g.bias = up(bias); g.M = M; g.N = N; g.Kp = d;
    std::vector<float> C((size_t)M * N, -7.f);
    std::vector<uint16_t> ph((size_t)M * N, 0), pl((size_t)M * N, 0);
    const size_t qn = (size_t)nb * H * Tqp * dh;
    std::vector<uint16_t> q[6];
    if (post == 0) { g.C = up(C); g.ldc = N; }
    if (post == 1) { g.Chi = (__bf16*)up(ph); g.Clo = (__bf16*)up(pl); g.c_rows = M; }
    if (post == 2) {
        for (auto& v : q) v.assign(qn, 0);
        g.Qhi = (__bf16*)up(q[0]); g.Qlo = (__bf16*)up(q[1]); g.Khi = (__bf16*)up(q[2]); g.Klo = (__bf16*)up(q[3]); g.Vhi = (__bf16*)up(q[4]); g.Vlo = (__bf16*)up(q[5]);
        g.d = d; g.H = H; g.dh = dh; g.Tqp = Tqp; g.qscale = 1.0f / sqrtf((float)dh);
    }
    CK(launch_sb_gemm(g, 1, post, x3, nullptr)); CK(hipDeviceSynchronize());
    // reference
    std::vector<double> X((size_t)M * d);
    auto ln = [&](double* r, const std::vector<float>& gm, const std::vector<float>& bt) {
        double mu = 0, var = 0;
        for (int k = 0; k < d; ++k) mu += r[k];
        mu /= d;
        for (int k = 0; k < d; ++k) var += (r[k] - mu) * (r[k] - mu);
        var /= d;
        for (int k = 0; k < d; ++k) r[k] = (r[k] - mu) / sqrt(var + 1e-5) * gm[k] + bt[k];
    };
    for (int m = 0; m < M; ++m) {
        double* r = &X[(size_t)m * d];
        for (int k = 0; k < d; ++k) r[k] = src[(size_t)m * d + k];
        ln(r, ga, ba);
        if (two) { for (int k = 0; k < d; ++k) r[k] += sv[d + k] + pv[(size_t)(m / Tq) * d + k]; ln(r, gb, bb); }
    }
    CK(hipMemcpy(xo.data(), g.xout, xo.size() * 4, hipMemcpyDeviceToHost));
    double ex = 0, err = 0;
    const int xcols = N >= 512 ? 512 : N / 32 * 32;     // only the first N/32 slices exist
    for (int m = 0; m < M; ++m) for (int k = 0; k < xcols; ++k) ex = fmax(ex, fabs(X[(size_t)m * d + k] - xo[(size_t)m * d + k]));
    if (post == 0) CK(hipMemcpy(C.data(), g.C, C.size() * 4, hipMemcpyDeviceToHost));
    if (post == 1) { CK(hipMemcpy(ph.data(), g.Chi, ph.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(pl.data(), g.Clo, pl.size() * 2, hipMemcpyDeviceToHost)); }
    if (post == 2) { __bf16* dp[6] = {g.Qhi, g.Qlo, g.Khi, g.Klo, g.Vhi, g.Vlo}; for (int i = 0; i < 6; ++i) CK(hipMemcpy(q[i].data(), dp[i], qn * 2, hipMemcpyDeviceToHost)); }
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
        double s = bias[n];
        for (int k = 0; k < d; ++k) {
            const double a = x3 ? X[(size_t)m * d + k] : bf2f(f2bf((float)X[(size_t)m * d + k])), w = x3 ? (double)W[(size_t)n * d + k] : bf2f(f2bf(W[(size_t)n * d + k]));
            s += a * w;
        }
        double got;
        if (post == 0) got = C[(size_t)m * N + n];
        else if (post == 1) { s = gelu(s); const size_t o = ((size_t)(n / 32) * M + m) * 32 + n % 32; got = (double)bf2f(ph[o]) + bf2f(pl[o]); }
        else {
            const int which = n / d, cin = n % d, hd = cin / dh, c = cin % dh, b = m / Tq, t = m % Tq;
            if (which == 0) s *= 1.0 / sqrt((double)dh);
            const size_t o = (((size_t)b * H + hd) * Tqp + t) * dh + c;
            got = (double)bf2f(q[which * 2][o]) + bf2f(q[which * 2 + 1][o]);
        }
        err = fmax(err, fabs(s - got));
    }
    printf("LN%d->post%d   M=%4d N=%4d %s: max err %.2e (normalised rows %.2e)   %.2f us\n", two ? 2 : 1, post, M, N, x3 ? "x3  " : "bf16", err, ex, time_us(g, 1, post, x3));
}

int main() {
    CK(configure_sb());
    for (bool x3 : {true, false}) {
        test_planes(60, 512, 512, x3);
        test_planes(60, 512, 1024, x3);
        test_planes(60, 512, 336, x3);
        test_planes(150, 512, 512, x3);
        test_planes(480, 512, 1024, x3);
        test_ln(60, 1536, 2, false, x3, 60);
        test_ln(60, 1024, 1, true, x3, 60);
        test_ln(60, 150, 0, false, x3, 60);
        test_ln(60, 336, 0, false, x3, 60);
        test_ln(302, 1536, 2, false, x3, 151);
        test_ln(480, 1024, 1, true, x3, 60);
    }
    return 0;
}
