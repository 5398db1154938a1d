// Small-batch ("latency") GEMMs of the decoder layers: the denoiser evaluation for a handful of motions.
//
// At B = 1 (BASELINE configs[0], the reference CLI's own default range) one evaluation has 60 token rows: a single 64-row
// tile. The throughput kernels (k_qkv_attn, k_mlp, k_rowgemm) give every tile ALL columns of a layer, so one workgroup
// streams whole weight matrices through one CU (~4 MB per layer at the ~30-60 B/clk a CU pulls from L2): 0.5 ms per step
// with 255 CUs idle. Here the split is the other way round: a workgroup owns 64 rows x 32 output columns, a launch has
// N/32 (16-48) workgroups per row tile, each reads a 32-64 KiB weight slice, and the step becomes a chain of short
// kernels whose cost is the dependent-launch boundary (~1.5 us, MI355X_MICROARCH.md "boundary") plus one L2 round trip.
// In-kernel grid barriers are priced at 4-7 us on this chip, so kernel boundaries ARE the cheap barrier.
//
// LayerNorm cannot sit in the epilogue of a column-split GEMM (a row's statistics span all workgroups), so it moves to the
// CONSUMER: the producing GEMM writes the pre-norm sum (GEMM + bias + residual) as fp32 rows, and every workgroup of the
// next GEMM re-normalises the full 64 x 512 tile on its way into LDS (128 KiB of L2 reads, 8 rows per wave, DPP sums)
// before it forms its A fragments. The first 16 column slices also write their 32 columns of the normalised rows back as
// the fp32 residual stream of the next pre-norm sum.
//
//   per layer:  qkv  = PRE_LN(norm3 of the previous layer | identity) -> in_proj -> attention-ready q/k/v planes
//               attn = k_attn_x3 (rgn_attn_x3.hip)
//               proj = PRE_PLANES(attention output) -> out_proj + bias + residual -> fp32 pre-norm rows
//               ff1  = PRE_LN(norm1, + folded cross-attention vectors, norm2) -> linear1 + GELU -> planes
//               ff2  = PRE_PLANES -> linear2 + bias + residual -> fp32 pre-norm rows
//
// MFMA: v_mfma_f32_32x32x16_bf16, computed transposed (weights are the A operand, activations the B operand) so a lane
// holds runs of 4 consecutive output columns. 8 waves = 2 row patches of 32 x 4 quarters of K: a wave's whole operand set
// (its 32 x K/4 weight and activation fragments, straight from the K32-blocked planes [K/32][rows][32]) is in flight at
// once, 64-128 VGPRs. The four k-quarters are summed through LDS in a fixed order, so every output element is accumulated
// in an order that does not depend on which rows share its tile: results are bit-identical under any batch composition
// (sharding, chains).
// Precision follows the phase of the schedule: X3 = three MFMAs per product on hi/lo planes, else hi planes only.
#include "rgn_internal.h"
#include "rgn_sb_common.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace rgn {

namespace {
static int g_sb_small_rows = 128;             // REGENNET_SB_SMALL_ROWS / REGENNET_SB_WIDE_ROWS override (tools; read once in configure_sb)
static int g_sb_wide_rows = 512;
}  // namespace

// PRE: 0 = A fragments from K32-blocked planes, 1 = A = LayerNorm(s) of fp32 rows (through an LDS image)
// POST: 0 = fp32 rows (+ bias + residual), 1 = GELU -> K32-blocked planes, 2 = attention-ready q / k / v planes
// NP: row patches of 32 per workgroup (1: 32-row tiles / 4 waves for the smallest evaluations, 2: 64-row tiles / 8 waves)
// NC: column blocks of 32 per workgroup (2 above 128 rows: half the workgroups re-normalise each row tile, half the LayerNorm
//     phase traffic, at twice the weight slice per workgroup)
template <int PRE, int POST, bool X3, int NP, int NC>
__global__ __launch_bounds__(256 * NP) void k_sb_gemm(SbArgs g) {
    constexpr int CH = (X3 && NC == 2) ? 2 : ((X3 || PRE == 1 || NC == 2) ? 4 : 8);   // k32-blocks per register chunk of a wave (PRE 1: K = 512, 4 per wave)
    constexpr bool W_LATE = PRE == 1 && NC == 2;                     // two column blocks: the first weight chunk is requested after the LayerNorm phase (registers)
    constexpr int ROWS = 32 * NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int patch = w % NP, kq = w / NP, l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * ROWS, n0 = blockIdx.x * 32 * NC;
    const int KB = g.Kp >> 5;
    const int kb0 = (kq * KB) >> 2, kb1 = ((kq + 1) * KB) >> 2;      // this wave's quarter of the k32-blocks
    const int mrow = min(m0 + patch * 32 + l31, g.M - 1);            // the rows / weight rows whose fragments this lane loads
    int nrow[NC];                                                    // (clamped: the surplus is masked at the store)
#pragma unroll
    for (int c = 0; c < NC; ++c) nrow[c] = min(n0 + c * 32 + l31, g.N - 1);
    const int m = m0 + (tid >> 3), nb = n0 + (tid & 7) * 4;          // this THREAD's outputs after the reduction: row m, columns nb + 32 c .. + 3
    // the loop index of the sampling step (address of the per-step vector of the second norm): a scalar load, waited for here - as a
    // vector load inside the staging below it put two dependent memory round trips in front of the LayerNorm phase
    int step = 0;
    if constexpr (PRE == 1) {
        if (g.gb && g.stepvec) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(step) : "s"(g.d_step) : "memory");
    }

    // ---- weight fragments of the first chunk: nothing depends on them until the MFMAs, so their L2 round trip overlaps
    //      the LayerNorm phase / the activation fragment loads
    bf16x8 wh[NC][CH][2], wl[NC][X3 ? CH : 1][2];
    auto load_w = [&](int kc) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (kc + j < kb1) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const unsigned o = ((unsigned)(kc + j) * g.w_rows + nrow[c]) * 32 + ks * 16 + kh * 8;   // (32-bit offsets: SGPR base + one VGPR per address)
                        wh[c][j][ks] = *reinterpret_cast<const bf16x8*>(g.Whi + o);
                        if constexpr (X3) wl[c][j][ks] = *reinterpret_cast<const bf16x8*>(g.Wlo + o);
                    }
                }
    };
    if constexpr (!W_LATE) load_w(kb0);

    // ---- epilogue operands, requested up front as well
    float bias[NC][4], res[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nb + 32 * c + i;
            bias[c][i] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
            res[c][i] = 0.f;
            if constexpr (POST == 0) {
                if (g.resid && m < g.M && n < g.N) res[c][i] = g.resid[(size_t)m * g.ldr + n];
            }
        }

    __bf16* img = reinterpret_cast<__bf16*>(smem);                   // PRE 1: [planes][16][ROWS][32], 16-byte chunk c of row r at c ^ ((r >> 2) & 3)
    if constexpr (PRE == 1) {
        // ---- LayerNorm phase: wave w owns rows 8w .. 8w+7 of the tile, 4 at a time (one per 16-lane row)
        const int rr = lane >> 4, lc = lane & 15;
        float x[2][4][8];
        int mr[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            mr[p] = m0 + w * 8 + p * 4 + rr;
            sb_ldvec(g.src + (unsigned)min(mr[p], g.M - 1) * SB_D, lc, x[p]);
        }
        // the per-sample vectors of the second norm are requested now, with the rows (behind the first norm they were a third
        // exposed round trip of this phase)
        // (only in the smallest build - 32-row tiles, one column block, the B <= 2 regime where this phase is pure latency: the
        //  64 live registers spill in the larger ones)
        constexpr bool PV_EARLY = NP == 1 && NC == 1 && !X3;
        float pv[2][4][8];
        if constexpr (PV_EARLY) {
            if (g.gb && g.pervec) {
#pragma unroll
                for (int p = 0; p < 2; ++p) sb_ldvec(g.pervec + (unsigned)(min(mr[p], g.M - 1) / g.Tq) * g.ldper, lc, pv[p]);
            }
        }
        // gamma / beta of both norms and the per-step vector: one LDS copy per workgroup (each lane reads 32 values of each;
        // as per-lane global loads they cost up to 160 VGPRs of live range and four redundant fetches per wave)
        float* vec = reinterpret_cast<float*>(smem + (size_t)(X3 ? 2 : 1) * 16 * ROWS * 32 * 2);   // [5][512]: ga, ba, gb, bb, stepvec
        if (tid < 128) {
            const float* srcs[5] = {g.ga, g.ba, g.gb, g.bb, (g.gb && g.stepvec) ? g.stepvec + (size_t)step * g.ldstep : nullptr};
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (srcs[k]) *reinterpret_cast<f32x4*>(vec + k * SB_D + tid * 4) = *reinterpret_cast<const f32x4*>(srcs[k] + tid * 4);
        }
        __syncthreads();
        const float* lv = vec + lc * 8;
        if (g.ga) {
#pragma unroll
            for (int p = 0; p < 2; ++p) sb_ln_row(x[p], lv, lv + SB_D);
        }
        if (g.gb) {
            if (g.stepvec) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float sv[8];
                    sb_ld8(lv + 4 * SB_D + j * 128, sv);
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int i = 0; i < 8; ++i) x[p][j][i] += sv[i];
                }
            }
            if (g.pervec) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if constexpr (!PV_EARLY) sb_ldvec(g.pervec + (unsigned)(min(mr[p], g.M - 1) / g.Tq) * g.ldper, lc, pv[p]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < 8; ++i) x[p][j][i] += pv[p][j][i];
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) sb_ln_row(x[p], lv + 2 * SB_D, lv + 3 * SB_D);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = w * 8 + p * 4 + rr;
            const bool valid = mr[p] < g.M;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // this workgroup's columns of the residual stream (those of its slice that the lane holds)
                if (g.xout && valid && (j * 128 + lc * 8) / (32 * NC) == (int)blockIdx.x) {
                    float* xo = g.xout + (size_t)mr[p] * SB_D + j * 128 + lc * 8;
                    *reinterpret_cast<f32x4*>(xo) = f32x4{x[p][j][0], x[p][j][1], x[p][j][2], x[p][j][3]};
                    *reinterpret_cast<f32x4*>(xo + 4) = f32x4{x[p][j][4], x[p][j][5], x[p][j][6], x[p][j][7]};
                }
                bf16x8 h, l;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = valid ? x[p][j][i] : 0.f;
                    h[i] = (__bf16)v;
                    l[i] = (__bf16)(v - (float)h[i]);
                }
                const int o = ((j * 4 + (lc >> 2)) * ROWS + r) * 32 + (((lc & 3) ^ ((r >> 2) & 3)) * 8);
                *reinterpret_cast<bf16x8*>(img + o) = h;
                if constexpr (X3) *reinterpret_cast<bf16x8*>(img + 16 * ROWS * 32 + o) = l;
            }
        }
        if constexpr (W_LATE) load_w(kb0);
        __syncthreads();
    }

    f32x16 a0[NC], a1[NC], a2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) a0[c][i] = a1[c][i] = a2[c][i] = 0.f;
    for (int kc = kb0; kc < kb1; kc += CH) {
        if (kc != kb0) load_w(kc);
        bf16x8 ah[CH][2], al[X3 ? CH : 1][2];
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (kc + j < kb1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if constexpr (PRE == 0) {
                        const unsigned o = ((unsigned)(kc + j) * g.a_rows + mrow) * 32 + ks * 16 + kh * 8;
                        ah[j][ks] = *reinterpret_cast<const bf16x8*>(g.Ahi + o);
                        if constexpr (X3) al[j][ks] = *reinterpret_cast<const bf16x8*>(g.Alo + o);
                    } else {
                        const int r = patch * 32 + l31;
                        const int o = ((kc + j) * ROWS + r) * 32 + (((ks * 2 + kh) ^ ((r >> 2) & 3)) * 8);
                        ah[j][ks] = *reinterpret_cast<const bf16x8*>(img + o);
                        if constexpr (X3) al[j][ks] = *reinterpret_cast<const bf16x8*>(img + 16 * ROWS * 32 + o);
                    }
                }
            }
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (kc + j < kb1) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        a0[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[c][j][ks], ah[j][ks], a0[c], 0, 0, 0);
                        if constexpr (X3) {
                            a1[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[c][j][ks], al[j][ks], a1[c], 0, 0, 0);
                            a2[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[c][j][ks], ah[j][ks], a2[c], 0, 0, 0);
                        }
                    }
            }
    }
    if constexpr (X3) {
#pragma unroll
        for (int c = 0; c < NC; ++c) a0[c] += a1[c] + a2[c];
    }

    // ---- sum the four k-quarters through LDS in a fixed order. Computed transposed: a lane holds row (= lane & 31) of its
    //      patch and the output columns (i & 3) + 8 (i >> 2) + 4 kh of a column block, i.e. four runs of 4 consecutive columns.
    constexpr int RLD = 36;                                          // padded row stride (floats) of a partial patch
    constexpr int RCB = 4 * NP * 32 * RLD;                           // floats per column block
    float* red = reinterpret_cast<float*>(smem);                     // [NC][kq 4][patch NP][32][RLD] (PRE 1: over the dead image)
    if constexpr (PRE == 1) __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(red + c * RCB + ((kq * NP + patch) * 32 + l31) * RLD + 8 * q + 4 * kh) =
                f32x4{a0[c][4 * q], a0[c][4 * q + 1], a0[c][4 * q + 2], a0[c][4 * q + 3]};
    __syncthreads();
    if (m >= g.M) return;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int n = nb + 32 * c;
        if (n >= g.N) continue;
        float v[4];
        {
            const float* rp = red + c * RCB + (tid >> 3) * RLD + (tid & 7) * 4;    // (tid >> 3 = patch * 32 + row in the patch)
            f32x4 sum = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
            for (int k = 1; k < 4; ++k) sum += *reinterpret_cast<const f32x4*>(rp + k * NP * 32 * RLD);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = sum[i] + bias[c][i];
        }
        if constexpr (POST == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += res[c][i];
            float* cp = g.C + (size_t)m * g.ldc + n;
            if (n + 3 < g.N && (g.ldc & 3) == 0) {
                *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (n + i < g.N) cp[i] = v[i];
            }
        } else {
            float sc = 1.0f;
            __bf16 *ph, *pl;
            size_t o;
            if constexpr (POST == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = sb_gelu(v[i]);
                ph = g.Chi; pl = g.Clo;
                o = ((size_t)(n >> 5) * g.c_rows + m) * 32 + (n & 31);
            } else {
                const int which = n / g.d, cin = n - which * g.d, hd = cin / g.dh, cc = cin - hd * g.dh;
                const int b = m / g.Tq, t = m - b * g.Tq;
                ph = which == 0 ? g.Qhi : (which == 1 ? g.Khi : g.Vhi);
                pl = which == 0 ? g.Qlo : (which == 1 ? g.Klo : g.Vlo);
                if (which == 0) sc = g.qscale;
                o = (((size_t)b * g.H + hd) * g.Tqp + t) * g.dh + cc;
            }
            bf16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = v[i] * sc;
                h[i] = (__bf16)x;
                l[i] = (__bf16)(x - (float)h[i]);
            }
            *reinterpret_cast<bf16x4*>(ph + o) = h;
            if (pl) *reinterpret_cast<bf16x4*>(pl + o) = l;
        }
    }
}

bool sb_supported(int d, int ff, int dh) { return d == SB_D && ff % 32 == 0 && dh % 4 == 0; }

template <int PRE, int POST, bool X3, int NP, int NC>
static hipError_t sb_launch(const SbArgs& g, hipStream_t s, bool cfg) {
    const size_t red = (size_t)NC * 4 * NP * 32 * 36 * 4, img = (size_t)(X3 ? 2 : 1) * 16 * 32 * NP * 32 * 2;
    const size_t lds = PRE == 1 ? (img + 5 * 512 * 4 > red ? img + 5 * 512 * 4 : red) : red;
    if (cfg) return hipFuncSetAttribute(reinterpret_cast<const void*>(k_sb_gemm<PRE, POST, X3, NP, NC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_sb_gemm<PRE, POST, X3, NP, NC>), dim3((g.N + 32 * NC - 1) / (32 * NC), (g.M + 32 * NP - 1) / (32 * NP)), dim3(256 * NP), lds, s, g);
    return hipGetLastError();
}
// 32-row x 32-column tiles while the evaluation has at most 128 rows (B <= 2 at 60 frames: the most workgroups, the shortest
// LayerNorm phase each), 64 x 32 up to 512 rows, 64 x 64 beyond (half the workgroups re-normalising each row tile). Measured
// at 60 frames, ms per 1000-step call, 64 x 32 vs 64 x 64: B = 4: 357 / 412, B = 8: 421 / 426, B = 12: 549 / 497.
template <int PRE, int POST>
static hipError_t sb_go(const SbArgs& g, bool x3, hipStream_t s, bool cfg) {
    if (cfg) {
        hipError_t e = sb_launch<PRE, POST, true, 1, 1>(g, s, true);
        if (e == hipSuccess) e = sb_launch<PRE, POST, true, 2, 2>(g, s, true);
        if (e == hipSuccess) e = sb_launch<PRE, POST, false, 1, 1>(g, s, true);
        if (e == hipSuccess) e = sb_launch<PRE, POST, false, 2, 2>(g, s, true);
        if (e == hipSuccess) e = sb_launch<PRE, POST, true, 2, 1>(g, s, true);
        if (e == hipSuccess) e = sb_launch<PRE, POST, false, 2, 1>(g, s, true);
        return e;
    }
    const bool small = g.M <= g_sb_small_rows, wide = g.M > g_sb_wide_rows;
    if (x3) return small ? sb_launch<PRE, POST, true, 1, 1>(g, s, false) : (wide ? sb_launch<PRE, POST, true, 2, 2>(g, s, false) : sb_launch<PRE, POST, true, 2, 1>(g, s, false));
    return small ? sb_launch<PRE, POST, false, 1, 1>(g, s, false) : (wide ? sb_launch<PRE, POST, false, 2, 2>(g, s, false) : sb_launch<PRE, POST, false, 2, 1>(g, s, false));
}
static hipError_t sb_dispatch(const SbArgs& g, int pre, int post, bool x3, hipStream_t s, bool cfg) {
    if (pre == 0 && post == 0) return sb_go<0, 0>(g, x3, s, cfg);
    if (pre == 1 && post == 0) return sb_go<1, 0>(g, x3, s, cfg);
    if (pre == 1 && post == 1) return sb_go<1, 1>(g, x3, s, cfg);
    if (pre == 1 && post == 2) return sb_go<1, 2>(g, x3, s, cfg);
    return hipErrorInvalidValue;
}
hipError_t configure_sb() {
    if (const char* e = getenv("REGENNET_SB_SMALL_ROWS")) g_sb_small_rows = atoi(e);
    if (const char* e = getenv("REGENNET_SB_WIDE_ROWS")) g_sb_wide_rows = atoi(e);
    SbArgs g{};
    const int combos[4][2] = {{0, 0}, {1, 0}, {1, 1}, {1, 2}};
    for (auto& c : combos) {
        hipError_t e = sb_dispatch(g, c[0], c[1], false, nullptr, true);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_sb_gemm(const SbArgs& g, int pre, int post, bool x3, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || (g.Kp & 31) || (pre == 1 && g.Kp != SB_D) || (post != 0 && (g.N & 31))) return hipErrorInvalidValue;
    return sb_dispatch(g, pre, post, x3, s, false);
}

}  // namespace rgn
