// Row-complete split-bf16 GEMM with the residual LayerNorm(s) fused into the epilogue.
//
//   t   = (Ahi+Alo) . (Whi+Wlo)^T + bias + resid                      [M, 512]   (out_proj / linear2 of a decoder layer)
//   y   = LN_a(t)                                                      norm1 / norm3
//   out = LN_b(y + pervec[row / Tq] + stepvec[*d_step])   (optional)   norm2 after the folded 1-token cross-attention
//
// Replaces k_gemm_x3 + k_layernorm for the two N = d = 512 GEMMs of every layer: the pre-norm tensor never goes to
// HBM (saves a 4 B/element write + read and one launch per LayerNorm). A workgroup owns 64 COMPLETE rows:
// tile 64 x 512, 8 waves as 2 (M) x 4 (N), wave tile 32 x 128 = 1 x 4 MFMA tiles (v_mfma_f32_32x32x16_bf16, three
// products per tile pair as in k_gemm_x3), two LDS stages of 72 KiB fed by direct-to-LDS DMA from K32-blocked planes.
// Epilogue: the accumulators are parked in an LDS row buffer (aliasing the dead stages); every wave then normalises
// 8 complete rows with the one-wave-per-row scheme of k_layernorm and writes coalesced rows.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

namespace rgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

constexpr int LN_BM = 64, LN_BN = 512, LN_NT = 512;

template <bool X3>
__global__ __launch_bounds__(LN_NT, 2) void k_gemm_x3_ln(GemmLnArgs g) {
    constexpr int NPL = X3 ? 2 : 1;
    constexpr int A_BYTES = LN_BM * 64, W_BYTES = LN_BN * 64;
    constexpr int STAGE = NPL * (A_BYTES + W_BYTES);                    // 72 KiB (x3)
    constexpr int A_IT = 1, W_IT = LN_BN * 4 / LN_NT;                   // A: only the first 256 threads carry a chunk
    constexpr int LPT_A = NPL, LPT_W = NPL * W_IT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = blockIdx.x * LN_BM;
    const bool a_loader = wave < 4;                                     // 256 chunk-lanes cover the 64 x 4 chunks of an A plane tile

    size_t a_src = 0, w_src[W_IT];
    {
        const int q = tid & 255, r = q >> 2, c = (q & 3) ^ ((r >> 2) & 3);
        int m = m0 + r;
        m = m < g.M ? m : g.M - 1;
        a_src = (size_t)m * 32 + c * 8;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int q = it * LN_NT + tid, r = q >> 2, c = (q & 3) ^ ((r >> 2) & 3);
        w_src[it] = (size_t)r * 32 + c * 8;                             // N == 512 exactly: every row exists
    }
    auto issue = [&](int kt, int stage) {
        char* sb = smem + stage * STAGE;
        const size_t ka = (size_t)kt * g.a_rows * 32, kw = (size_t)kt * LN_BN * 32;
        if (a_loader) {
            const int lo = (tid & ~63) * 16;
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.Ahi + a_src + ka), (RGN_AS3 void*)(sb + lo), 16, 0, 0);
            if (X3)
                __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.Alo + a_src + ka), (RGN_AS3 void*)(sb + A_BYTES + lo), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int lo = (it * LN_NT + (tid & ~63)) * 16;
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.Whi + w_src[it] + kw), (RGN_AS3 void*)(sb + NPL * A_BYTES + lo), 16, 0, 0);
            if (X3)
                __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.Wlo + w_src[it] + kw), (RGN_AS3 void*)(sb + NPL * A_BYTES + W_BYTES + lo), 16, 0, 0);
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

    const int l31 = lane & 31, kh = lane >> 5;
    int a_off[2], w_off[4][2];
    {
        const int rr = wm * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rr = wn * 128 + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    const int nk = g.Kp / 32;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            issue(kt + 1, (kt + 1) & 1);
            // tile kt landed; tile kt+1's DMA (a loader wave has LPT_A more in flight than the others) stays in flight
            if (a_loader) {
                if (X3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                if (X3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const char* sb = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah, al, bh[4], bl[4];
            ah = *reinterpret_cast<const bf16x8*>(sb + a_off[ks]);
            if (X3) al = *reinterpret_cast<const bf16x8*>(sb + A_BYTES + a_off[ks]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bh[t] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + w_off[t][ks]);
                if (X3) bl[t] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + W_BYTES + w_off[t][ks]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (X3) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    static_assert(LPT_A + LPT_W == (X3 ? 10 : 5), "vmcnt literals above assume W_IT == 4");

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // (1) accumulators (+ bias) go to an LDS row buffer [64][516] fp32 that aliases the dead pipeline stages: in the
    //     C/D layout a half-wave writes 32 consecutive columns of one row (conflict-free);
    // (2) every wave then owns 8 complete rows, one at a time, 8 columns per lane (lane + 64 j) exactly like k_layernorm:
    //     + residual (coalesced fp32 row reads), two-pass LayerNorm a, (+ per-sample / per-step vectors, LayerNorm b),
    //     coalesced stores of the fp32 residual stream and of the split planes.
    constexpr int RLD = LN_BN + 4;
    float* rowbuf = reinterpret_cast<float*>(smem);
    __builtin_amdgcn_s_barrier();                                       // every wave is done reading the pipeline stages
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = wn * 128 + t * 32 + l31;
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) rowbuf[(wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh) * RLD + n] = acc[t][i] + bias;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int VPL = LN_BN / 64;                                     // 8
    float ga[VPL], ba[VPL], gb[VPL], bb[VPL], sv[VPL];
    const int step = g.stepvec ? *g.d_step : 0;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int n = lane + 64 * j;
        ga[j] = g.ga[n];
        ba[j] = g.ba[n];
        gb[j] = g.gb ? g.gb[n] : 0.f;
        bb[j] = g.gb ? g.bb[n] : 0.f;
        sv[j] = g.stepvec ? g.stepvec[(size_t)step * g.ldstep + n] : 0.f;
    }
    const float invn = 1.0f / (float)LN_BN;
    for (int rr = 0; rr < LN_BM / 8; ++rr) {
        const int r = wave * (LN_BM / 8) + rr, m = m0 + r;
        if (m >= g.M) break;                                            // wave-uniform
        float v[VPL], s = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            v[j] = rowbuf[r * RLD + lane + 64 * j] + g.resid[(size_t)m * LN_BN + lane + 64 * j];
            s += v[j];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        float mean = s * invn, q = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const float c = v[j] - mean;
            q += c * c;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        float rstd = 1.0f / sqrtf(q * invn + 1e-5f);
#pragma unroll
        for (int j = 0; j < VPL; ++j) v[j] = (v[j] - mean) * rstd * ga[j] + ba[j];
        if (g.gb) {
            const float* pv = g.pervec ? g.pervec + (size_t)(m / g.Tq) * g.ldper : nullptr;
            s = 0.f;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                v[j] += sv[j] + (pv ? pv[lane + 64 * j] : 0.f);
                s += v[j];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            mean = s * invn;
            q = 0.f;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const float c = v[j] - mean;
                q += c * c;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            rstd = 1.0f / sqrtf(q * invn + 1e-5f);
#pragma unroll
            for (int j = 0; j < VPL; ++j) v[j] = (v[j] - mean) * rstd * gb[j] + bb[j];
        }
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int n = lane + 64 * j;
            g.out[(size_t)m * LN_BN + n] = v[j];
            if (g.ohi) {
                const size_t o = ((size_t)(n >> 5) * g.o_rows + m) * 32 + (n & 31);
                const __bf16 h = (__bf16)v[j];
                g.ohi[o] = h;
                if (g.olo) g.olo[o] = (__bf16)(v[j] - (float)h);
            }
        }
    }
}

bool gemm_ln_supported(int N) { return N == LN_BN; }
hipError_t configure_gemm_ln() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_x3_ln<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * (LN_BM * 64 + LN_BN * 64));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_x3_ln<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (LN_BM * 64 + LN_BN * 64));
}
hipError_t launch_gemm_ln(const GemmLnArgs& g, bool x3, hipStream_t s) {
    const int lds = 2 * (x3 ? 2 : 1) * (LN_BM * 64 + LN_BN * 64);
    const dim3 grid((g.M + LN_BM - 1) / LN_BM), block(LN_NT);
    if (x3)
        hipLaunchKernelGGL((k_gemm_x3_ln<true>), grid, block, lds, s, g);
    else
        hipLaunchKernelGGL((k_gemm_x3_ln<false>), grid, block, lds, s, g);
    return hipGetLastError();
}

}  // namespace rgn
