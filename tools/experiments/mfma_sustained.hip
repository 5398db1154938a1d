// Sustained rate of v_mfma_f32_32x32x16_{bf16,f16} on all 1024 SIMDs with full-entropy operands (the shader clock under a matrix load is
// power-managed: what the matrix pipe sustains depends on what its multipliers toggle). tools/experiments/mfma_sustained [waves per SIMD]
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_sustained tools/experiments/mfma_sustained.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool F16>
__global__ void mfma_spin(long long iters, const u32x4* ops, float* out) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const u32x4 xr = ops[threadIdx.x & 255], yr = ops[256 + (threadIdx.x & 255)];
    for (long long i = 0; i < iters; ++i) {
        if constexpr (F16) {
            const f16x8 x = __builtin_bit_cast(f16x8, xr), y = __builtin_bit_cast(f16x8, yr);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a3, 0, 0, 0);
        } else {
            const bf16x8 x = __builtin_bit_cast(bf16x8, xr), y = __builtin_bit_cast(bf16x8, yr);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a3, 0, 0, 0);
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 12345.f) out[0] = s;
}
// 16-bit patterns of N(0, 1)-like values with random mantissas: sign random, exponent in [bias - 4, bias], mantissa uniform
static unsigned short rnd16(bool f16) {
    const unsigned sign = rand() & 1, e = rand() % 5;
    if (f16) return (unsigned short)((sign << 15) | ((15 - e) << 10) | (rand() & 0x3ff));
    return (unsigned short)((sign << 15) | ((127 - e) << 7) | (rand() & 0x7f));
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const double secs = argc > 2 ? atof(argv[2]) : 2.0;
    float* d; (void)hipMalloc(&d, 64);
    u32x4* ops; (void)hipMalloc(&ops, 512 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long long iters = (long long)(secs * 13.5e6) / wps;
    for (int rep = 0; rep < 3; ++rep)
        for (int f16 = 0; f16 < 2; ++f16) {
            std::vector<unsigned short> h(512 * 8);
            srand(1234 + rep);
            for (auto& v : h) v = rnd16(f16 != 0);
            (void)hipMemcpy(ops, h.data(), 512 * 16, hipMemcpyHostToDevice);
            (void)hipEventRecord(e0);
            if (f16) mfma_spin<true><<<256, 256 * wps>>>(iters, ops, d); else mfma_spin<false><<<256, 256 * wps>>>(iters, ops, d);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s %d waves/SIMD: %.1f ms, %.0f TF\n", f16 ? "f16 " : "bf16", wps, ms, 256.0 * 4 * wps * 4.0 * iters * 32768 / (ms * 1e-3) * 1e-12);
        }
    return 0;
}
