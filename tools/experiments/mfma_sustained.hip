#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void mfma_spin(long long iters, float* out) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(((threadIdx.x * 37 + i * 11) % 97) * 0.01f - 0.5f); y[i] = (__bf16)(((threadIdx.x * 13 + i * 7) % 89) * 0.01f - 0.4f); }
    for (long long i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 12345.f) out[0] = s;
}
int main(int argc, char** argv) {
    float* d; (void)hipMalloc(&d, 64);
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long long iters = 60000000LL / wps;
    (void)hipEventRecord(e0); mfma_spin<<<256, 256 * wps>>>(iters, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%d waves/SIMD: %.1f ms, %.0f TF\n", wps, ms, 256.0 * 4 * wps * 4.0 * iters * 32768 / (ms * 1e-3) * 1e-12);
    return 0;
}
