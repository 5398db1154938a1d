#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long n, long long* out) {
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    while (__builtin_readcyclecounter() - t0 < n) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = __builtin_readcyclecounter() - t0; out[1] = wall_clock64() - w0; }
}
__global__ void mfma_spin(int iters, long long* out) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(i * 0.5f); }
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 12345.f) out[7] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[2] = t1 - t0; out[3] = w1 - w0; }
}
int main() {
    long long* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); spin<<<1, 64>>>(100000000LL, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("idle spin: %lld ticks, wall_clock64 %lld, %.3f ms -> tick rate %.3f GHz, wall_clock rate %.3f MHz\n", h[0], h[1], ms, h[0] / ms * 1e-6, h[1] / ms * 1e-3);
    for (int wpb = 256; wpb <= 512; wpb += 256) {
      const int iters = 200000;
      hipEventRecord(e0); mfma_spin<<<256, wpb>>>(iters, d); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
      const double mf = 4.0 * iters * (wpb / 64) / 4;   // MFMAs per SIMD
      printf("mfma full chip, %d waves/SIMD: %lld ticks, %.3f ms -> tick rate %.3f GHz; %.1f ticks per MFMA per SIMD; %.1f ns per MFMA; TF %.0f\n", wpb / 256, h[2], ms, h[2] / ms * 1e-6, h[2] / mf, ms * 1e6 / mf, 256.0 * 4 * mf * 32768 / (ms * 1e-3) * 1e-12);
    }
    return 0;
}
