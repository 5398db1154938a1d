// Device helpers shared by the small-batch kernels (rgn_sb.hip, rgn_sb_attn.hip): DPP row sums, the 16-lane-per-row
// LayerNorm of a 512-wide fp32 row, the erf GELU of the split-bf16 epilogues.
#pragma once
#include <hip/hip_runtime.h>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SB_D = 512;        // model width of the PRE_LN variants (row = 64 lanes x 8 floats)

template <int CTRL>
__device__ __forceinline__ float sb_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// wave-wide sum on the VALU (DPP inside rows of 16 lanes, the four row totals through SGPRs), as in rgn_rowgemm.hip
__device__ __forceinline__ float sb_wave_sum(float v) {
    v += sb_dpp<0xB1>(v);
    v += sb_dpp<0x4E>(v);
    v += sb_dpp<0x141>(v);
    v += sb_dpp<0x140>(v);
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}

// erf: Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7), the split-bf16 GEMM epilogue's form
__device__ __forceinline__ float sb_gelu(float v) {
    const float x = v * 0.70710678118654752440f, ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = copysignf(1.0f - p * t * __expf(-ax * ax), x);
    return v * 0.5f * (1.0f + e);
}

__device__ __forceinline__ void sb_ld8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// LayerNorm phase register layout: a 512-wide row lives in ONE 16-lane DPP row (lane c of it holds columns
// 128 j + 8 c .. + 8 for j = 0..3), a wave normalises 4 rows at once, and a row's sum is 4 DPP adds with the result
// in every lane of the row: no cross-row traffic, no SGPR round trips.
__device__ __forceinline__ float sb_row16_sum(float v) {
    v += sb_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
    v += sb_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
    v += sb_dpp<0x141>(v);    // row_half_mirror
    v += sb_dpp<0x140>(v);    // row_mirror
    return v;
}
// two-pass LayerNorm, eps = 1e-5, like k_layernorm; gamma / beta from the workgroup's LDS copy (pointers at this lane's columns)
__device__ __forceinline__ void sb_ln_row(float (&v)[4][8], const float* gv, const float* bv) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[j][i];
    const float mean = sb_row16_sum(s) * (1.0f / SB_D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float c = v[j][i] - mean;
            q += c * c;
        }
    const float rstd = 1.0f / sqrtf(sb_row16_sum(q) * (1.0f / SB_D) + 1e-5f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float ga[8], ba[8];
        sb_ld8(gv + j * 128, ga);
        sb_ld8(bv + j * 128, ba);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[j][i] = (v[j][i] - mean) * rstd * ga[i] + ba[i];
    }
}
__device__ __forceinline__ void sb_ldvec(const float* p, int lc, float (&v)[4][8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sb_ld8(p + j * 128 + lc * 8, v[j]);
}

}  // namespace

}  // namespace rgn
