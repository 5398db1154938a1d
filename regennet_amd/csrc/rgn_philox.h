// Device-side Philox4x32-10 + Box-Muller shared by the kernels that draw the sampler's noise (rgn_kernels.hip, rgn_step.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef RGN_PHILOX_ROUNDS
#define RGN_PHILOX_ROUNDS 10
#endif

namespace rgn {

// ---- Philox4x32-10 + Box-Muller: counter = (element/4, loop index, sample lo, sample hi), key = seed
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < RGN_PHILOX_ROUNDS; ++r) {
        // one 32 x 32 -> 64 multiply (v_mad_u64_u32) per product instead of a v_mul_hi + v_mul_lo pair: the integer multiplies are
        // quarter-rate instructions and were most of the draw's cost
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// Box-Muller on the hardware transcendentals: -2 ln u1 = -2 ln 2 * v_log_f32(u1) (log2), v_sqrt_f32, and v_sin_f32 / v_cos_f32, which
// take their argument in REVOLUTIONS (sin(2 pi u2) = v_sin_f32(u2)): 6 instructions per pair of normals instead of the ~80 of
// logf / sqrtf / sincospif (the draw is ~10 us of the 47 us step boundary at B = 256: 5.2 M normals per step). ~1 ulp transforms of
// uniform 24-bit inputs; one definition for every kernel that draws (k_randn, k_update, k_step), so all of them see one stream.
__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& n0, float& n1) {
    const float u1 = ((ra >> 8) + 1u) * 5.9604644775390625e-08f;   // (0,1]
    const float u2 = (rb >> 8) * 5.9604644775390625e-08f;          // [0,1)
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    n0 = rad * __builtin_amdgcn_cosf(u2);
    n1 = rad * __builtin_amdgcn_sinf(u2);
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned long long sample, uint32_t stream,
                                               uint32_t elem) {
    uint32_t r[4];
    philox4x32_10(elem >> 2, stream, (uint32_t)sample, (uint32_t)(sample >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int pair = (elem >> 1) & 1;
    float n0, n1;
    box_muller(r[2 * pair], r[2 * pair + 1], n0, n1);
    return (elem & 1) ? n1 : n0;
}

}  // namespace rgn
