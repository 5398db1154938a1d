// Device-side Philox4x32-10 + Box-Muller shared by the kernels that draw the sampler's noise (rgn_kernels.hip, rgn_step.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rgn {

// ---- Philox4x32-10 + Box-Muller: counter = (element/4, loop index, sample lo, sample hi), key = seed
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned long long sample, uint32_t stream,
                                               uint32_t elem) {
    uint32_t r[4];
    philox4x32_10(elem >> 2, stream, (uint32_t)sample, (uint32_t)(sample >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int pair = (elem >> 1) & 1;
    const float u1 = ((r[2 * pair] >> 8) + 1u) * 5.9604644775390625e-08f;   // (0,1]
    const float u2 = (r[2 * pair + 1] >> 8) * 5.9604644775390625e-08f;      // [0,1)
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    return rad * ((elem & 1) ? sn : cs);
}


}  // namespace rgn
