// Small-batch engine: LayerNorm prologue + in_proj + causal self-attention of ONE (sample, head) per workgroup.
//
// rgn_sb.hip runs a layer's first half as two launches - the column-split in_proj (k_sb_gemm<1, 2>, 48 workgroups per row
// tile, q / k / v planes through L2) and k_attn_x3 (one workgroup per (sample, head)) - and at B = 1 each of the two is a
// 7 - 8 us dependent link of the step's chain (profiles/r03_kernel_stats_cfg1_B1.csv). Here a workgroup owns a head of a
// sample outright: it re-normalises the sample's 64 x 512 pre-norm rows into LDS like every k_sb_gemm<1, .> workgroup does,
// multiplies them with the head's 384 in_proj rows (q, k, v: 12 column blocks of 32, 393 KB of weights per plane), keeps
// q / k / v in LDS and runs the attention of rgn_attn_x3.hip on them: one launch, no q / k / v round trip.
// (reference: the decoder layer's self-attention block, nn.TransformerDecoderLayer as built by cmdm.py:75-81; norm3 of the
// previous layer is the prologue's LayerNorm.)
//
// 8 waves = 2 K-halves x 4 column-block quarters; wave (khalf, cbq) owns the column blocks q_cbq, k_cbq, v_cbq for BOTH
// 32-row patches, so every weight fragment is loaded by exactly one wave (96 VGPRs per chunk of 4 k32-blocks, the first chunk
// requested before the LayerNorm phase). q and k are computed transposed (weights = A operand: a lane holds a row's runs of 4
// consecutive columns -> 8-byte writes into the row-major Q / K slabs), v the other way round (activations = A operand: a
// lane holds a column's runs of 4 consecutive keys -> 8-byte writes into the transposed V^T slab). The two K-halves are summed
// through LDS in a fixed order and each (sample, head) is its own workgroup: bit-identical under any batch composition.
// X3 = split bf16 (three MFMAs per product, one accumulator, small terms first), else hi planes only.
#include "rgn_internal.h"
#include "rgn_sb_common.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace rgn {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int SA_DH = 128, SA_ROWS = 64;
constexpr int SA_KLD = SA_DH + 8;             // Q / K slab row stride (bf16): conflict-free ds_read_b128
constexpr int SA_VLD = SA_ROWS + 4;           // V^T slab row stride (bf16)
constexpr int SA_OLD = SA_DH + 4;             // fp32 output patch row stride
constexpr int SA_QK_PLANE = SA_ROWS * SA_KLD, SA_VT_PLANE = SA_DH * SA_VLD;
constexpr int SA_EXCH = 4 * 24 * 64 * 16;     // bytes: the khalf = 1 waves' partial sums, [cbq][quad 24][lane] x 16 B
constexpr int SA_BIAS = 3 * SA_DH * 4;        // the head's q / k / v bias, at the top of the allocation
template <bool X3>
constexpr int sa_lds() {
    constexpr int npl = X3 ? 2 : 1;
    constexpr int img = npl * 16 * SA_ROWS * 32 * 2 + 2 * SB_D * 4;                                  // image + gamma / beta
    constexpr int slabs = npl * (2 * SA_QK_PLANE + SA_VT_PLANE) * 2 + 2 * 32 * SA_OLD * 4;           // Q, K, V^T + two output patches
    constexpr int m = img > SA_EXCH ? (img > slabs ? img : slabs) : (SA_EXCH > slabs ? SA_EXCH : slabs);
    return m + SA_BIAS;
}
}  // namespace

template <bool X3>
__global__ __launch_bounds__(512) void k_sb_qkv_attn(SbArgs g) {
    constexpr int NPL = X3 ? 2 : 1;
    constexpr int CH = X3 ? 2 : 4;                                   // k32-blocks per weight chunk (96 VGPRs either way)
    constexpr int IMG_PLANE = 16 * SA_ROWS * 32;                     // elements per image plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int khalf = w & 1, cbq = w >> 1;
    const int hd = blockIdx.x, b = blockIdx.y, Tq = g.Tq;
    const unsigned row0 = (unsigned)b * Tq;
    const int kb0 = khalf * 8;

    // ---- the wave's weight fragments: rows (= output columns) which * d + hd * 128 + 32 cbq + l31 of q, k, v
    //      (buffer loads: one 32-bit lane offset per column block, the k32-block in the scalar offset)
    const __amdgpu_buffer_rsrc_t whr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Whi), 0, (g.Kp >> 5) * g.w_rows * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t wlr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(X3 ? g.Wlo : g.Whi), 0, (g.Kp >> 5) * g.w_rows * 64, 0x00020000);
    int voff[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) voff[c] = (c * g.d + hd * SA_DH + 32 * cbq + l31) * 64 + kh * 16;
    bf16x8 wh[3][CH][2], wl[3][X3 ? CH : 1][2];
    auto load_w = [&](int kc) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int so = (kb0 + kc + j) * g.w_rows * 64;           // bytes to the k32-block
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    wh[c][j][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(whr, voff[c] + ks * 32, so, 0));
                    if constexpr (X3) wl[c][j][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wlr, voff[c] + ks * 32, so, 0));
                }
        }
    };
    load_w(0);

    // ---- LayerNorm phase (k_sb_gemm<1, .>'s, 64 rows): wave w owns rows 8w .. 8w+7 of the sample, 4 at a time
    __bf16* img = reinterpret_cast<__bf16*>(smem);                   // [planes][16][64][32], 16-byte chunk c of row r at c ^ ((r >> 2) & 3)
    float* vec = reinterpret_cast<float*>(smem + (size_t)NPL * IMG_PLANE * 2);   // [2][512]: gamma, beta
    float* hbias = reinterpret_cast<float*>(smem + sa_lds<X3>() - SA_BIAS);      // [3][128]
    {
        const int rr = lane >> 4, lc = lane & 15;
        float x[2][4][8];
        int rl[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            rl[p] = w * 8 + p * 4 + rr;
            sb_ldvec(g.src + (row0 + (unsigned)min(rl[p], Tq - 1)) * SB_D, lc, x[p]);
        }
        if (tid < 128) {
            const float* srcs[2] = {g.ga, g.ba};
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (srcs[k]) *reinterpret_cast<f32x4*>(vec + k * SB_D + tid * 4) = *reinterpret_cast<const f32x4*>(srcs[k] + tid * 4);
        } else if (tid < 128 + 96) {
            const int i = tid - 128, which = i >> 5, c4 = (i & 31) * 4;
            *reinterpret_cast<f32x4*>(hbias + which * SA_DH + c4) = *reinterpret_cast<const f32x4*>(g.bias + which * g.d + hd * SA_DH + c4);
        }
        __syncthreads();
        const float* lv = vec + lc * 8;
        if (g.ga) {
#pragma unroll
            for (int p = 0; p < 2; ++p) sb_ln_row(x[p], lv, lv + SB_D);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = rl[p];
            const bool valid = r < Tq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (g.xout && valid && j == hd) {                    // this head's 128 columns of the residual stream
                    float* xo = g.xout + (size_t)(row0 + r) * SB_D + j * 128 + lc * 8;
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
                const int o = ((j * 4 + (lc >> 2)) * SA_ROWS + r) * 32 + (((lc & 3) ^ ((r >> 2) & 3)) * 8);
                *reinterpret_cast<bf16x8*>(img + o) = h;
                if constexpr (X3) *reinterpret_cast<bf16x8*>(img + IMG_PLANE + o) = l;
            }
        }
        __syncthreads();
    }

    // ---- in_proj: this wave's K half of (q_cbq, k_cbq, v_cbq) x both row patches
    f32x16 acc[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][p][i] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 8; kc += CH) {
        if (kc) load_w(kc);
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int r = p * 32 + l31;
                    const int o = ((kb0 + kc + j) * SA_ROWS + r) * 32 + (((ks * 2 + kh) ^ ((r >> 2) & 3)) * 8);
                    ah[p] = *reinterpret_cast<const bf16x8*>(img + o);
                    if constexpr (X3) al[p] = *reinterpret_cast<const bf16x8*>(img + IMG_PLANE + o);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        if (c < 2) {                                 // q, k: D[column][row]
                            if constexpr (X3) {
                                acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[c][j][ks], ah[p], acc[c][p], 0, 0, 0);
                                acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[c][j][ks], al[p], acc[c][p], 0, 0, 0);
                            }
                            acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[c][j][ks], ah[p], acc[c][p], 0, 0, 0);
                        } else {                                     // v: D[row][column]
                            if constexpr (X3) {
                                acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[p], wl[c][j][ks], acc[c][p], 0, 0, 0);
                                acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[p], wh[c][j][ks], acc[c][p], 0, 0, 0);
                            }
                            acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[p], wh[c][j][ks], acc[c][p], 0, 0, 0);
                        }
                    }
            }
    }
    __syncthreads();                                                 // the image is dead

    // ---- the upper K half -> LDS, the lower half's wave adds it (own + other: one fixed order)
    char* exch = smem + (size_t)cbq * (24 * 64 * 16) + lane * 16;
    if (khalf) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
                    *reinterpret_cast<f32x4*>(exch + ((c * 2 + p) * 4 + i4) * 1024) =
                        f32x4{acc[c][p][4 * i4], acc[c][p][4 * i4 + 1], acc[c][p][4 * i4 + 2], acc[c][p][4 * i4 + 3]};
    }
    __syncthreads();
    if (!khalf) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(exch + ((c * 2 + p) * 4 + i4) * 1024);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][p][4 * i4 + e] += o[e];
                }
    }
    __syncthreads();                                                 // the exchange is dead

    // ---- q (pre-scaled), k, v^T slabs (+ bias), split into hi / lo planes
    __bf16* sQ = reinterpret_cast<__bf16*>(smem);                    // [planes][64][SA_KLD]
    __bf16* sK = sQ + NPL * SA_QK_PLANE;
    __bf16* sV = sK + NPL * SA_QK_PLANE;                             // [planes][128][SA_VLD]
    if (!khalf) {
        const float bv = hbias[2 * SA_DH + 32 * cbq + l31];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int n = 32 * cbq + 8 * i4 + 4 * kh;            // q, k: this lane's run of 4 columns of the head
                const f32x4 bq = *reinterpret_cast<const f32x4*>(hbias + n), bk = *reinterpret_cast<const f32x4*>(hbias + SA_DH + n);
                bf16x4 qh, ql, kkh, kl, vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q = (acc[0][p][4 * i4 + e] + bq[e]) * g.qscale, k = acc[1][p][4 * i4 + e] + bk[e], v = acc[2][p][4 * i4 + e] + bv;
                    qh[e] = (__bf16)q; ql[e] = (__bf16)(q - (float)qh[e]);
                    kkh[e] = (__bf16)k; kl[e] = (__bf16)(k - (float)kkh[e]);
                    vh[e] = (__bf16)v; vl[e] = (__bf16)(v - (float)vh[e]);
                }
                const int oqk = (32 * p + l31) * SA_KLD + n;         // row 32 p + l31
                const int ov = (32 * cbq + l31) * SA_VLD + 32 * p + 8 * i4 + 4 * kh;   // v: column 32 cbq + l31, keys 32 p + 8 i4 + 4 kh ..
                *reinterpret_cast<bf16x4*>(sQ + oqk) = qh;
                *reinterpret_cast<bf16x4*>(sK + oqk) = kkh;
                *reinterpret_cast<bf16x4*>(sV + ov) = vh;
                if constexpr (X3) {
                    *reinterpret_cast<bf16x4*>(sQ + SA_QK_PLANE + oqk) = ql;
                    *reinterpret_cast<bf16x4*>(sK + SA_QK_PLANE + oqk) = kl;
                    *reinterpret_cast<bf16x4*>(sV + SA_VT_PLANE + ov) = vl;
                }
            }
    }
    __syncthreads();
    if (w >= 2) return;

    // ---- attention of query patch w (k_attn_x3<2, 128, X3> on the slabs): S^T = K . Q^T, softmax over keys, O^T = V^T . P^T
    constexpr int NS = SA_DH / 16, ND = SA_DH / 32;
    const int qrow = 32 * w + l31;
    f32x16 st[2];
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kj][i] = 0.f;
        if (kj <= w) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int oq = qrow * SA_KLD + 16 * s + 8 * kh, ok = (32 * kj + l31) * SA_KLD + 16 * s + 8 * kh;
                const bf16x8 qfh = *reinterpret_cast<const bf16x8*>(sQ + oq), kfh = *reinterpret_cast<const bf16x8*>(sK + ok);
                if constexpr (X3) {
                    const bf16x8 qfl = *reinterpret_cast<const bf16x8*>(sQ + SA_QK_PLANE + oq), kfl = *reinterpret_cast<const bf16x8*>(sK + SA_QK_PLANE + ok);
                    st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qfh, st[kj], 0, 0, 0);
                    st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qfl, st[kj], 0, 0, 0);
                }
                st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qfh, st[kj], 0, 0, 0);
            }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
        if (kj <= w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 32 * kj + (i & 3) + 8 * (i >> 2) + 4 * kh;
                const bool ok = (key <= qrow) && (key < Tq);
                st[kj][i] = ok ? st[kj][i] : -INFINITY;
                mx = fmaxf(mx, st[kj][i]);
            }
        }
    mx = half_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
        if (kj <= w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __expf(st[kj][i] - mx);
                st[kj][i] = e;
                sum += e;
            }
        }
    sum = half_sum(sum);
    const float inv = 1.0f / sum;
    f32x16 oa[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oa[dt][i] = 0.f;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
        if (kj <= w) {
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                bf16x8 ph, pl;                                       // B operand: this lane's 8 keys = registers 8 step .. 8 step + 7
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = st[kj][8 * step + j];
                    ph[j] = (__bf16)x;
                    pl[j] = (__bf16)(x - (float)ph[j]);
                }
                const int kb = 32 * kj + 16 * step + 4 * kh;         // keys kb .. kb+3 and kb+8 .. kb+11
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const int o = (32 * dt + l31) * SA_VLD + kb;
                    u32x4 vh;
                    vh.lo = *reinterpret_cast<const u32x2*>(sV + o);
                    vh.hi = *reinterpret_cast<const u32x2*>(sV + o + 8);
                    const bf16x8 vfh = __builtin_bit_cast(bf16x8, vh);
                    if constexpr (X3) {
                        u32x4 vl;
                        vl.lo = *reinterpret_cast<const u32x2*>(sV + SA_VT_PLANE + o);
                        vl.hi = *reinterpret_cast<const u32x2*>(sV + SA_VT_PLANE + o + 8);
                        const bf16x8 vfl = __builtin_bit_cast(bf16x8, vl);
                        oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, oa[dt], 0, 0, 0);
                        oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl, oa[dt], 0, 0, 0);
                    }
                    oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph, oa[dt], 0, 0, 0);
                }
            }
        }
    }
    // ---- O^T -> rows through this wave's private patch (behind the slabs) -> the out_proj GEMM's K32-blocked planes
    float* patch = reinterpret_cast<float*>(smem + (size_t)NPL * (2 * SA_QK_PLANE + SA_VT_PLANE) * 2) + w * (32 * SA_OLD);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) patch[l31 * SA_OLD + 32 * dt + (i & 3) + 8 * (i >> 2) + 4 * kh] = oa[dt][i] * inv;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int C4 = SA_DH / 4;
    for (int idx = lane; idx < 32 * C4; idx += 64) {
        const int r = idx / C4, c = (idx - r * C4) * 4;
        const int q = 32 * w + r;
        if (q < Tq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&patch[r * SA_OLD + c]);
            const int col = hd * SA_DH + c;
            const size_t o = ((size_t)(col >> 5) * g.att.rows + row0 + q) * 32 + (col & 31);
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (__bf16)v[e];
                l[e] = (__bf16)(v[e] - (float)h[e]);
            }
            *reinterpret_cast<bf16x4*>(g.att.hi + o) = h;
            if (g.att.lo) *reinterpret_cast<bf16x4*>(g.att.lo + o) = l;
        }
    }
}

bool sb_qkv_attn_supported(int d, int dh, int Tq) {
    // Supported shapes. The engine takes it only on request (REGENNET_SB_FUSED_ATTN=1). Measured (profiles/r04_sb_fused_attn.txt): 18.0 us per launch at B = 1 against 5.8 + 8.3 us for
    // k_sb_gemm<1, 2> + k_attn_x3 - four workgroups each pull a head's 393 KB of weights through ONE CU's memory path, where the
    // column-split in_proj spreads 32 KB slices over 96 CUs - so 1000 steps at B = 1 take 0.294 s instead of 0.261 s; B = 4: 0.351 / 0.339 s;
    // it wins from B ~ 6 up (B = 8: 0.368 / 0.408 s). The two forms round differently and a motion must not depend on the batch it was
    // drawn in (test_small_batch_engine_is_bit_exact_under_batch_composition), so the choice cannot follow the batch size: off.
    // (the switch is read when an engine is created: rgn_pack.cpp)
    return d == SB_D && dh == SA_DH && Tq <= SA_ROWS;
}
hipError_t configure_sb_qkv_attn() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sb_qkv_attn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, sa_lds<true>());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_sb_qkv_attn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, sa_lds<false>());
}
// g: the in_proj's SbArgs (src / ga / ba / xout, weight planes, bias, d, Tq, qscale) + att; Bm samples, H heads
hipError_t launch_sb_qkv_attn(const SbArgs& g, int Bm, bool x3, hipStream_t s) {
    if (Bm <= 0 || g.d != SB_D || g.dh != SA_DH || g.H * SA_DH != g.d || g.Tq <= 0 || g.Tq > SA_ROWS || g.Kp != SB_D || !g.bias || !g.att.hi) return hipErrorInvalidValue;
    if (x3)
        hipLaunchKernelGGL((k_sb_qkv_attn<true>), dim3(g.H, Bm), dim3(512), sa_lds<true>(), s, g);
    else
        hipLaunchKernelGGL((k_sb_qkv_attn<false>), dim3(g.H, Bm), dim3(512), sa_lds<false>(), s, g);
    return hipGetLastError();
}

}  // namespace rgn
