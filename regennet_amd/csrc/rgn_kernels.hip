// Device code for the ReGenNet sampling hot path on MI355X (gfx950, wave64, MFMA).
//
// Layouts:
//   boundary tensors   x, cmotion, out : fp32 [B, F=njoints*nfeats, T]   (frames contiguous)
//   token-major        xin  [B*Tq, F]   h [Bm*Tq, d]   qkv [Bm*Tq, 3d]   att [Bm*Tq, d]   ffn [Bm*Tq, ff]
//                      row = b*Tq + etd + t  (a sample's tokens are contiguous -> attention reads one slab)
//   weights            nn.Linear layout [N, Kp] (K contiguous, zero padded to a multiple of 16)
//
// Reference arithmetic replaced (file:line in the upstream repo) is cited per kernel.
#include "rgn_internal.h"
#include "rgn_philox.h"

#include <hip/hip_runtime.h>
#include <math.h>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// element (row, col) of a K32-blocked plane [cols/32][rows][32]
__device__ __forceinline__ size_t plane_off(int rows, int row, int col) {
    return ((size_t)(col >> 5) * rows + row) * 32 + (col & 31);
}
__device__ __forceinline__ void plane_put(const Planes& p, int row, int col, float v) {
    const size_t o = plane_off(p.rows, row, col);
    const __bf16 h = (__bf16)v;
    p.hi[o] = h;
    if (p.lo) p.lo[o] = (__bf16)(v - (float)h);
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));  // F.gelu, exact erf
    if (act == 2) return v / (1.0f + __expf(-v));                                  // nn.SiLU (cmdm.py:293)
    if (act == 3) return fmaxf(v, 0.f);                                            // nn.ReLU (ST-GCN evaluator)
    return v;
}

// =================================================================================================
// GEMM  C[M,N] = act( A[M,K] * W[N,K]^T + bias[N] + add[r % add_mod, N] )
// Replaces every nn.Linear on the path (cmdm.py:61,291-295,307,337 and the in_proj/out_proj/linear1/
// linear2 of nn.TransformerDecoderLayer constructed at cmdm.py:75-81).
//
// F32 precision mode: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
// Block = 128x128 output tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 acc VGPRs).
// K is walked in 16-wide LDS tiles; rows are padded to 20 floats so that the 16 lanes of a
// ds_read_b128 group (consecutive rows, 80-byte stride) fall on disjoint bank quads.
// An 8-wide k chunk is contracted by 4 MFMAs: lanes 0-31 feed k = kc+j, lanes 32-63 feed k = kc+4+j
// for A and B alike (any consistent permutation of the contraction index is legal).
// =================================================================================================
constexpr int G_BM = 128, G_BN = 128, G_BK = 16, G_LD = G_BK + 4;

template <bool VEC_A>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[G_BM * G_LD];
    __shared__ __attribute__((aligned(16))) float Ws[G_BN * G_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * G_BM, n0 = blockIdx.x * G_BN;
    const int lr = tid >> 2, lk = (tid & 3) * 4;  // staging: row (0..63, +64), k quad

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    f32x4 ra[2], rw[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + lr + 64 * i, k = k0 + lk;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < g.M) {
                const float* p = g.A + (size_t)m * g.lda + k;
                if (VEC_A) {
                    if (k < g.K) v = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < g.K) v[j] = p[j];
                }
            }
            ra[i] = v;
            const int n = n0 + lr + 64 * i;
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (n < g.N) w = *reinterpret_cast<const f32x4*>(g.W + (size_t)n * g.Kp + k);
            rw[i] = w;
        }
    };
    const int nk = g.Kp / G_BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&As[(lr + 64 * i) * G_LD + lk]) = ra[i];
            *reinterpret_cast<f32x4*>(&Ws[(lr + 64 * i) * G_LD + lk]) = rw[i];
        }
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * G_BK);
#pragma unroll
        for (int kc = 0; kc < G_BK; kc += 8) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + t * 32 + (lane & 31)) * G_LD + kc + 4 * (lane >> 5)]);
                b[t] = *reinterpret_cast<const f32x4*>(&Ws[(wn * 64 + t * 32 + (lane & 31)) * G_LD + kc + 4 * (lane >> 5)]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta][j], b[tb][j], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int n = n0 + wn * 64 + tb * 32 + (lane & 31);
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * 64 + ta * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = acc[ta][tb][i] + bias;
                if (g.add) {
                    const int ar = g.add_mod ? (m % g.add_mod) : m;
                    v += g.add[(size_t)ar * g.ldadd + n];
                }
                g.C[(size_t)m * g.ldc + n] = act_apply(v, g.act);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// BF16X3 / BF16 precision modes: v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//   BF16X3: every fp32 operand is split x = hi + lo (both bf16, round-to-nearest-even) and the product
//           is formed as ah*bh + ah*bl + al*bh (the al*bl term, ~2^-18 relative, is dropped): three MFMAs
//           per tile pair, ~2^-16 relative accuracy per product — inside the 1e-3 end-to-end tolerance
//           where plain bf16 (2^-9) is not (SURVEY.md §7 "Precision vs. the 1e-3 abs tolerance").
//   Weights are pre-split on the host (Whi/Wlo [N,Kp] bf16); activations stay fp32 in HBM and are
//   split while being staged into LDS (v_cvt_pk_bf16_f32), so the residual stream never loses bits.
// Tile 128x128x32, 4 waves (2x2), wave tile 64x64 = 2x2 MFMA tiles; LDS rows padded to 40 bf16 (80 B)
// -> the 16 lanes of each ds_read_b128 group hit 16 disjoint bank quads.
// Blocks are remapped so that the column tiles of one A row-block run on the same XCD (shared L2).
// -------------------------------------------------------------------------------------------------
constexpr int H_BM = 128, H_BN = 128, H_BK = 32, H_LD = H_BK + 8;   // bf16 elements per LDS row

__device__ __forceinline__ void split_bf16(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)v[j];
        hi[j] = h;
        lo[j] = (__bf16)(v[j] - (float)h);
    }
}

template <bool X3, bool VEC_A>
__global__ __launch_bounds__(256) void k_gemm_bf16(GemmArgs g, int nbx, int nby) {
    __shared__ __attribute__((aligned(16))) __bf16 Ah[H_BM * H_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Al[X3 ? H_BM * H_LD : 8];
    __shared__ __attribute__((aligned(16))) __bf16 Wh[H_BN * H_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Wl[X3 ? H_BN * H_LD : 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware bijective remap: hardware places block b on XCD b%8; give each XCD a contiguous range of
    // virtual ids (n-tile fastest) so one A row-block is fetched into a single XCD's L2.
    const int nwg = nbx * nby, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (vid / nbx) * H_BM, n0 = (vid % nbx) * H_BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int ar = tid >> 3, ak = (tid & 7) * 4;     // A staging: 8 threads x float4 cover one 32-float row
    const int wr = tid >> 2, wk = (tid & 3) * 8;     // W staging: 4 threads x 8 bf16 cover one 32-elem row
    f32x4 ra[4];
    u32x4 rwh[2], rwl[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ar + 32 * i, k = k0 + ak;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < g.M) {
                const float* p = g.A + (size_t)m * g.lda + k;
                if (VEC_A) {
                    if (k < g.K) v = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < g.K) v[j] = p[j];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + wr + 64 * i;
            u32x4 h = {0u, 0u, 0u, 0u}, l = {0u, 0u, 0u, 0u};
            if (n < g.N) {
                const size_t o = (size_t)n * g.Kp + k0 + wk;
                h = *reinterpret_cast<const u32x4*>(g.Whi + o);
                if (X3) l = *reinterpret_cast<const u32x4*>(g.Wlo + o);
            }
            rwh[i] = h;
            rwl[i] = l;
        }
    };
    const int nk = g.Kp / H_BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4 hi, lo;
            split_bf16(ra[i], hi, lo);
            *reinterpret_cast<bf16x4*>(&Ah[(ar + 32 * i) * H_LD + ak]) = hi;
            if (X3) *reinterpret_cast<bf16x4*>(&Al[(ar + 32 * i) * H_LD + ak]) = lo;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4*>(&Wh[(wr + 64 * i) * H_LD + wk]) = rwh[i];
            if (X3) *reinterpret_cast<u32x4*>(&Wl[(wr + 64 * i) * H_LD + wk]) = rwl[i];
        }
        __syncthreads();
        if (kt + 1 < nk) gload((kt + 1) * H_BK);
#pragma unroll
        for (int ks = 0; ks < H_BK; ks += 16) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ao = (wm * 64 + t * 32 + (lane & 31)) * H_LD + ks + 8 * (lane >> 5);
                const int bo = (wn * 64 + t * 32 + (lane & 31)) * H_LD + ks + 8 * (lane >> 5);
                ah[t] = *reinterpret_cast<const bf16x8*>(&Ah[ao]);
                bh[t] = *reinterpret_cast<const bf16x8*>(&Wh[bo]);
                if (X3) {
                    al[t] = *reinterpret_cast<const bf16x8*>(&Al[ao]);
                    bl[t] = *reinterpret_cast<const bf16x8*>(&Wl[bo]);
                }
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) {
                    if (X3) {   // small terms first, then the leading product
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ta], bh[tb], acc[ta][tb], 0, 0, 0);
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ta], bl[tb], acc[ta][tb], 0, 0, 0);
                    }
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ta], bh[tb], acc[ta][tb], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int n = n0 + wn * 64 + tb * 32 + (lane & 31);
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * 64 + ta * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float v = acc[ta][tb][i] + bias;
                if (g.add) {
                    const int arow = g.add_mod ? (m % g.add_mod) : m;
                    v += g.add[(size_t)arow * g.ldadd + n];
                }
                g.C[(size_t)m * g.ldc + n] = act_apply(v, g.act);
            }
        }
    }
}

hipError_t launch_gemm(const GemmArgs& g, int precision, hipStream_t s) {
    const bool vec = (g.lda % 4 == 0) && (g.K % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
    if (precision == 0) {
        dim3 grid((g.N + G_BN - 1) / G_BN, (g.M + G_BM - 1) / G_BM);
        if (vec)
            hipLaunchKernelGGL(k_gemm_f32<true>, grid, dim3(256), 0, s, g);
        else
            hipLaunchKernelGGL(k_gemm_f32<false>, grid, dim3(256), 0, s, g);
        return hipGetLastError();
    }
    const int nbx = (g.N + H_BN - 1) / H_BN, nby = (g.M + H_BM - 1) / H_BM;
    dim3 grid(nbx * nby);
    if (precision == 1) {
        if (vec)
            hipLaunchKernelGGL((k_gemm_bf16<true, true>), grid, dim3(256), 0, s, g, nbx, nby);
        else
            hipLaunchKernelGGL((k_gemm_bf16<true, false>), grid, dim3(256), 0, s, g, nbx, nby);
    } else {
        if (vec)
            hipLaunchKernelGGL((k_gemm_bf16<false, true>), grid, dim3(256), 0, s, g, nbx, nby);
        else
            hipLaunchKernelGGL((k_gemm_bf16<false, false>), grid, dim3(256), 0, s, g, nbx, nby);
    }
    return hipGetLastError();
}

// =================================================================================================
// Causal self-attention, one workgroup per (sample, head). v1: fp32 VALU, K/V slab in LDS.
// Replaces nn.MultiheadAttention inside TransformerDecoderLayer._sa_block with the mask of
// generate_square_subsequent_mask (cmdm.py:168-171,220-227): softmax(q k^T / sqrt(dh) + causal) v.
// =================================================================================================
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, float* __restrict__ out, Planes op, Dims dm) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, hd = blockIdx.y;
    const int Tq = dm.Tq, dh = dm.dh, d = dm.d, ldk = dh + 1;
    float* Ks = smem;                       // [Tq][dh+1]
    float* Vs = Ks + Tq * ldk;              // [Tq][dh+1]
    float* qs = Vs + Tq * ldk;              // [4][dh]
    float* ps = qs + 4 * dh;                // [4][Tq]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)b * Tq;
    for (int idx = tid; idx < Tq * dh; idx += 256) {
        const int j = idx / dh, c = idx - j * dh;
        const float* base = qkv + (row0 + j) * (size_t)(3 * d) + hd * dh + c;
        Ks[j * ldk + c] = base[d];
        Vs[j * ldk + c] = base[2 * d];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)dh);
    float* q = qs + wave * dh;
    float* p = ps + wave * Tq;
    for (int i0 = 0; i0 < Tq; i0 += 4) {   // uniform trip count: block barriers order the LDS hand-offs
        const int i = i0 + wave;
        const bool on = i < Tq;
        if (on)
            for (int c = lane; c < dh; c += 64) q[c] = qkv[(row0 + i) * (size_t)(3 * d) + hd * dh + c] * scale;
        __syncthreads();
        float inv = 0.f;
        if (on) {
            float mx = -INFINITY;
            for (int j = lane; j <= i; j += 64) {
                float s = 0.f;
                const float* kr = Ks + j * ldk;
                for (int c = 0; c < dh; ++c) s = fmaf(q[c], kr[c], s);
                p[j] = s;
                mx = fmaxf(mx, s);
            }
            mx = wave_max(mx);
            float sum = 0.f;
            for (int j = lane; j <= i; j += 64) {
                const float e = __expf(p[j] - mx);
                p[j] = e;
                sum += e;
            }
            inv = 1.0f / wave_sum(sum);
        }
        __syncthreads();
        if (on)
            for (int c = lane; c < dh; c += 64) {
                float o = 0.f;
                for (int j = 0; j <= i; ++j) o = fmaf(p[j], Vs[j * ldk + c], o);
                if (out) out[(row0 + i) * (size_t)d + hd * dh + c] = o * inv;
                if (op.hi) plane_put(op, (int)(row0 + i), hd * dh + c, o * inv);
            }
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------
// MFMA attention (fp32-input v_mfma_f32_32x32x2_f32, exact fp32 products), one workgroup per (sample, head),
// one wave per 32-query tile. Everything is computed TRANSPOSED so that no register shuffle is needed
// between the two GEMMs:
//   S^T[key, query] = K . Q^T     A = K tile from LDS, B = Q held in registers (pre-scaled by 1/sqrt(dh))
//   softmax over keys             a query's scores live in ONE lane pair (lane, lane^32): in-register max/sum
//                                 + one cross-half exchange
//   O^T[dh, query]  = V^T . P^T   B = the S^T accumulator registers as they are (C/D layout == B layout up
//                                 to a permutation of the contraction index, which A follows), A = V from LDS
// K and V time-share one LDS slab [32*NT][DH+4] (pad 4 floats: the 16 lanes of a ds_read_b128 group land on
// 16 different bank quads). Causality: wave w only visits key tiles 0..w; the diagonal tile is masked.
// The result is transposed through the (by then dead) slab so global stores are row-contiguous.
// -------------------------------------------------------------------------------------------------
template <int NT, int DH>
__global__ __launch_bounds__(64 * NT) void k_attn_mfma(const float* __restrict__ qkv, float* __restrict__ out, Planes op, Dims dm) {
    constexpr int DP = (DH < 32 ? 32 : DH);       // padded head dim (PV works on 32-wide dh tiles)
    constexpr int LD = DP + 4;
    constexpr int NC = DH / 8;                    // 8-wide contraction chunks of QK^T
    constexpr int ND = DP / 32;                   // dh tiles of PV
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* slab = smem;                           // [32*NT][LD]
    const int b = blockIdx.x, hd = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int Tq = dm.Tq, d = dm.d;
    const size_t row0 = (size_t)b * Tq;
    const float* qbase = qkv + row0 * (size_t)(3 * d) + hd * DH;

    auto load_slab = [&](int which) {             // which: 1 = K, 2 = V
        constexpr int C4 = DP / 4;
        for (int idx = tid; idx < 32 * NT * C4; idx += 64 * NT) {
            const int r = idx / C4, c = (idx - r * C4) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < Tq && c < DH) v = *reinterpret_cast<const f32x4*>(qbase + (size_t)r * (3 * d) + which * d + c);
            *reinterpret_cast<f32x4*>(&slab[r * LD + c]) = v;
        }
    };
    load_slab(1);
    // Q fragments: lane (query, half) holds Q[query][8c + 4*half + 0..3], scaled
    const float scale = 1.0f / sqrtf((float)DH);
    const int qrow = 32 * w + l31;
    f32x4 qf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qrow < Tq) v = *reinterpret_cast<const f32x4*>(qbase + (size_t)qrow * (3 * d) + 8 * c + 4 * half);
        qf[c] = v * scale;
    }
    __syncthreads();
    f32x16 st[NT];
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kj][i] = 0.f;
        if (kj <= w) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&slab[(32 * kj + l31) * LD + 8 * c + 4 * half]);
#pragma unroll
                for (int j = 0; j < 4; ++j) st[kj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[c][j], st[kj], 0, 0, 0);
            }
        }
    }
    // softmax over keys (rows of S^T): key = 32*kj + (i&3) + 8*(i>>2) + 4*half, query = 32*w + l31
    float mx = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
        if (kj <= w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 32 * kj + (i & 3) + 8 * (i >> 2) + 4 * half;
                const bool ok = (key <= qrow) && (key < Tq);
                st[kj][i] = ok ? st[kj][i] : -INFINITY;
                mx = fmaxf(mx, st[kj][i]);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
        if (kj <= w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __expf(st[kj][i] - mx);
                st[kj][i] = e;
                sum += e;
            }
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    __syncthreads();          // every wave is done with K
    load_slab(2);
    __syncthreads();
    f32x16 oa[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oa[dt][i] = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
        if (kj <= w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 32 * kj + (i & 3) + 8 * (i >> 2) + 4 * half;   // the key this lane's P register belongs to
                const float* vr = &slab[key * LD + l31];
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
                    oa[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32 * dt], st[kj][i], oa[dt], 0, 0, 0);
            }
        }
    }
    __syncthreads();          // every wave is done with V: reuse the slab rows of this wave's queries
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = 32 * dt + (i & 3) + 8 * (i >> 2) + 4 * half;       // dh index (row of O^T)
            slab[(32 * w + l31) * LD + c] = oa[dt][i] * inv;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    constexpr int C4 = DH / 4;            // float4 per output row
    for (int idx = lane; idx < 32 * C4; idx += 64) {
        const int r = idx / C4, c = (idx - r * C4) * 4;
        const int q = 32 * w + r;
        if (q < Tq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&slab[(32 * w + r) * LD + c]);
            if (out) *reinterpret_cast<f32x4*>(out + (row0 + q) * (size_t)d + hd * DH + c) = v;
            if (op.hi) {   // 4 consecutive columns stay inside one 32-column block
                const size_t o = plane_off(op.rows, (int)(row0 + q), hd * DH + c);
                bf16x4 h, l;
                split_bf16(v, h, l);
                *reinterpret_cast<bf16x4*>(op.hi + o) = h;
                if (op.lo) *reinterpret_cast<bf16x4*>(op.lo + o) = l;
            }
        }
    }
}

template <int NT, int DH>
static hipError_t attn_mfma_go(const float* qkv, float* out, Planes op, const Dims& dm, hipStream_t s, bool configure_only) {
    constexpr int DP = (DH < 32 ? 32 : DH);
    const size_t lds = (size_t)32 * NT * (DP + 4) * sizeof(float);
    if (configure_only)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_mfma<NT, DH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_attn_mfma<NT, DH>), dim3(dm.Bm, dm.H), dim3(64 * NT), lds, s, qkv, out, op, dm);
    return hipGetLastError();
}
template <int DH>
static hipError_t attn_mfma_nt(int nt, const float* qkv, float* out, Planes op, const Dims& dm, hipStream_t s, bool cfg) {
    switch (nt) {
        case 1: return attn_mfma_go<1, DH>(qkv, out, op, dm, s, cfg);
        case 2: return attn_mfma_go<2, DH>(qkv, out, op, dm, s, cfg);
        case 3: return attn_mfma_go<3, DH>(qkv, out, op, dm, s, cfg);
        case 4: return attn_mfma_go<4, DH>(qkv, out, op, dm, s, cfg);
        case 5: return attn_mfma_go<5, DH>(qkv, out, op, dm, s, cfg);
    }
    return hipErrorInvalidValue;
}
static bool attn_mfma_ok(int Tq, int dh) { return Tq <= 160 && (dh == 16 || dh == 32 || dh == 64 || dh == 128); }
static hipError_t attn_mfma(const float* qkv, float* out, Planes op, const Dims& dm, hipStream_t s, bool cfg) {
    const int nt = (dm.Tq + 31) / 32;
    switch (dm.dh) {
        case 16: return attn_mfma_nt<16>(nt, qkv, out, op, dm, s, cfg);
        case 32: return attn_mfma_nt<32>(nt, qkv, out, op, dm, s, cfg);
        case 64: return attn_mfma_nt<64>(nt, qkv, out, op, dm, s, cfg);
        case 128: return attn_mfma_nt<128>(nt, qkv, out, op, dm, s, cfg);
    }
    return hipErrorInvalidValue;
}

static size_t attn_lds_bytes(int Tq, int dh) { return ((size_t)2 * Tq * (dh + 1) + 4 * dh + 4 * Tq) * sizeof(float); }
// Called once at finalize (never during graph capture): allow > 64 KiB of dynamic LDS.
hipError_t configure_attention(int Tq, int dh) {
    if (attn_mfma_ok(Tq, dh)) {
        Dims dm{};
        dm.Tq = Tq;
        dm.dh = dh;
        return attn_mfma(nullptr, nullptr, Planes{nullptr, nullptr, 0}, dm, nullptr, true);
    }
    const size_t lds = attn_lds_bytes(Tq, dh);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}
hipError_t launch_attention(const float* qkv, float* out, Planes op, const Dims& dm, hipStream_t s) {
    if (attn_mfma_ok(dm.Tq, dm.dh)) return attn_mfma(qkv, out, op, dm, s, false);
    hipLaunchKernelGGL(k_attention, dim3(dm.Bm, dm.H), dim3(256), attn_lds_bytes(dm.Tq, dm.dh), s, qkv, out, op, dm);
    return hipGetLastError();
}

// =================================================================================================
// Residual LayerNorm(s): one wave per token row, row held in registers (lane owns VPL = d/64 CONSECUTIVE columns, so
// every access is one or two 16-byte vectors per lane: fp32 rows as float4, plane rows as 8 bf16).
//   out = LN_b( LN_a(in + resid) + addvec[row / Tq] + stepvec[*d_step] )   or   out = LN_a(in + resid)
// F32 mode: `in` already contains the residual (added in the producing GEMM's epilogue), resid is empty and the
// result goes to the fp32 residual stream `out`. bf16 modes: the residual stream exists only as split planes; it is
// added here (hi + lo) and the result is written as planes only (out == nullptr).
// Replaces norm1/norm2/norm3 of TransformerDecoderLayer (post-norm, eps=1e-5) and the add of the
// 1-token cross-attention result, which is constant over the sequence.
// =================================================================================================
template <int N>
__device__ __forceinline__ void ld_f32(const float* __restrict__ p, float* v) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int j = 0; j < N / 4; ++j) {
            const f32x4 t = reinterpret_cast<const f32x4*>(p)[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * j + e] = t[e];
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = p[j];
    }
}
template <int N>
__device__ __forceinline__ void st_f32(float* __restrict__ p, const float* v) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int j = 0; j < N / 4; ++j) {
            const f32x4 t = {v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
            reinterpret_cast<f32x4*>(p)[j] = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) p[j] = v[j];
    }
}
template <int N>
__device__ __forceinline__ void ld_bf16_add(const __bf16* __restrict__ p, float* v) {   // v += p[0..N)
    typedef __bf16 bf16xN __attribute__((ext_vector_type(N)));
    if constexpr (N == 1) {
        v[0] += (float)p[0];
    } else if constexpr (N <= 8) {
        const bf16xN t = *reinterpret_cast<const bf16xN*>(p);
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] += (float)t[e];
    } else {
        ld_bf16_add<8>(p, v);
        ld_bf16_add<N - 8>(p + 8, v + 8);
    }
}
template <int N>
__device__ __forceinline__ void st_split(__bf16* __restrict__ hi, __bf16* __restrict__ lo, const float* v) {
    typedef __bf16 bf16xN __attribute__((ext_vector_type(N)));
    if constexpr (N == 1) {
        const __bf16 h = (__bf16)v[0];
        hi[0] = h;
        if (lo) lo[0] = (__bf16)(v[0] - (float)h);
    } else if constexpr (N <= 8) {
        bf16xN h, l;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            h[e] = (__bf16)v[e];
            l[e] = (__bf16)(v[e] - (float)h[e]);
        }
        *reinterpret_cast<bf16xN*>(hi) = h;
        if (lo) *reinterpret_cast<bf16xN*>(lo) = l;
    } else {
        st_split<8>(hi, lo, v);
        st_split<N - 8>(hi + 8, lo ? lo + 8 : nullptr, v + 8);
    }
}

template <int VPL>
__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ in, Planes resid, float* __restrict__ out, Planes op, int M, int d,
                                                    const float* __restrict__ ga, const float* __restrict__ ba,
                                                    const float* __restrict__ addvec, int ldadd,
                                                    const float* __restrict__ stepvec, int ldstep, const int* __restrict__ d_step,
                                                    int Tq, const float* __restrict__ gb, const float* __restrict__ bb) {
    const int lane = threadIdx.x & 63;
    // XCD-affine mapping: hardware places block b on XCD b%8; give every XCD one contiguous range of rows, the same
    // range whose tiles the neighbouring GEMMs compute on that XCD (their L2 still holds what this kernel reads/writes)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int row = vid * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int c0 = lane * VPL;                                   // first of this lane's VPL consecutive columns
    float v[VPL], w[VPL], bsh[VPL];
    ld_f32<VPL>(in + (size_t)row * d + c0, v);
    if (resid.hi) {
        const size_t o = plane_off(resid.rows, row, c0);          // VPL divides 32: the run stays inside one K32 block
        ld_bf16_add<VPL>(resid.hi + o, v);
        if (resid.lo) ld_bf16_add<VPL>(resid.lo + o, v);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += v[i];
    const float invd = 1.0f / (float)d;
    float mean = wave_sum(s) * invd;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float c = v[i] - mean;
        q += c * c;
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) * invd + 1e-5f);
    ld_f32<VPL>(ga + c0, w);
    ld_f32<VPL>(ba + c0, bsh);
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = (v[i] - mean) * rstd * w[i] + bsh[i];
    if (gb) {
        if (stepvec) {
            ld_f32<VPL>(stepvec + (size_t)(*d_step) * ldstep + c0, w);
#pragma unroll
            for (int i = 0; i < VPL; ++i) v[i] += w[i];
        }
        if (addvec) {
            ld_f32<VPL>(addvec + (size_t)(row / Tq) * ldadd + c0, w);
#pragma unroll
            for (int i = 0; i < VPL; ++i) v[i] += w[i];
        }
        s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) s += v[i];
        mean = wave_sum(s) * invd;
        q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const float c = v[i] - mean;
            q += c * c;
        }
        rstd = 1.0f / sqrtf(wave_sum(q) * invd + 1e-5f);
        ld_f32<VPL>(gb + c0, w);
        ld_f32<VPL>(bb + c0, bsh);
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = (v[i] - mean) * rstd * w[i] + bsh[i];
    }
    if (out) st_f32<VPL>(out + (size_t)row * d + c0, v);
    if (op.hi) {
        const size_t o = plane_off(op.rows, row, c0);
        st_split<VPL>(op.hi + o, op.lo ? op.lo + o : nullptr, v);
    }
}

hipError_t launch_layernorm(const float* in, Planes resid, float* out, Planes op, int M, int d, const float* ga, const float* ba,
                            const float* addvec, int ldadd, const float* stepvec, int ldstep, const int* d_step, int Tq,
                            const float* gb, const float* bb, hipStream_t s) {
    dim3 grid((M + 3) / 4), block(256);
#define RGN_LN(V) hipLaunchKernelGGL(k_layernorm<V>, grid, block, 0, s, in, resid, out, op, M, d, ga, ba, addvec, ldadd, stepvec, ldstep, d_step, Tq, gb, bb)
    switch (d / 64) {
        case 1: RGN_LN(1); break;
        case 2: RGN_LN(2); break;
        case 4: RGN_LN(4); break;
        case 8: RGN_LN(8); break;
        case 16: RGN_LN(16); break;
        default: return hipErrorInvalidValue;
    }
#undef RGN_LN
    return hipGetLastError();
}

// =================================================================================================
// Timestep embedding input: out[r,:] = pe[t_r,:]  (TimestepEmbedder.forward gather, cmdm.py:298).
// t_r comes from the device step table (sampling loop; bit-exact timestep_map lookup of
// _WrappedModel.__call__, respace.py:124-129) or from external int64 timesteps (rgn_denoise).
// =================================================================================================
__global__ void k_gather_pe(const float* __restrict__ pe, const StepCoef* __restrict__ tab, const int* __restrict__ d_step,
                            const SampleParams* __restrict__ sp, float* __restrict__ out, int Bm, int B, int d, int pe_len) {
    const int r = blockIdx.x;
    long long t;
    if (sp->t_ext)
        t = sp->t_ext[r % B];
    else
        t = tab[*d_step].t_model;
    t = t < 0 ? 0 : (t >= pe_len ? pe_len - 1 : t);   // never read outside the table (the Python boundary raises IndexError first)
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[(size_t)r * d + c] = pe[(size_t)t * d + c];
}
hipError_t launch_gather_pe(const float* pe, const StepCoef* tab, const int* d_step, const SampleParams* sp,
                            float* out, int Bm, int B, int d, int pe_len, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_pe, dim3(Bm), dim3(d >= 256 ? 256 : 64), 0, s, pe, tab, d_step, sp, out, Bm, B, d, pe_len);
    return hipGetLastError();
}

// schedule-time variant: out[i,:] = pe[timestep_map[i],:] for every loop index i (the timestep embedding of the
// sampling loop depends on the step only, so its MLP and the folded cross-attention GEMM run once per schedule)
__global__ void k_gather_pe_all(const float* __restrict__ pe, const StepCoef* __restrict__ tab, float* __restrict__ out, int d) {
    const int i = blockIdx.x;
    const long long t = tab[i].t_model;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[(size_t)i * d + c] = pe[(size_t)t * d + c];
}
hipError_t launch_gather_pe_all(const float* pe, const StepCoef* tab, float* out, int S, int d, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_pe_all, dim3(S), dim3(d >= 256 ? 256 : 64), 0, s, pe, tab, out, d);
    return hipGetLastError();
}

// emb_trans_dec: token 0 of every sample is emb[b] (+ pe[0])            (cmdm.py:212-218)
__global__ void k_emb_rows(const float* __restrict__ emb, const float* __restrict__ stepemb, const int* __restrict__ d_step,
                           const float* __restrict__ pe, float* __restrict__ h, Planes hp, Dims dm, int wo_pos) {
    const int b = blockIdx.x;
    const float* se = stepemb ? stepemb + (size_t)(*d_step) * dm.d : nullptr;   // sampling loop: time part per step
    for (int c = threadIdx.x; c < dm.d; c += blockDim.x) {
        const float v = (emb ? emb[(size_t)b * dm.d + c] : 0.f) + (se ? se[c] : 0.f) + (wo_pos ? 0.f : pe[c]);
        h[(size_t)b * dm.Tq * dm.d + c] = v;
        if (hp.hi) plane_put(hp, b * dm.Tq, c, v);
    }
}
hipError_t launch_emb_rows(const float* emb, const float* stepemb, const int* d_step, const float* pe, float* h, Planes hp,
                           const Dims& dm, int wo_pos, hipStream_t s) {
    hipLaunchKernelGGL(k_emb_rows, dim3(dm.Bm), dim3(256), 0, s, emb, stepemb, d_step, pe, h, hp, dm, wo_pos);
    return hipGetLastError();
}

// c0[b*Tq + j, :] += pe[j, :]   (sequence_pos_encoder, cmdm.py:278-281) — hoisted, once per condition
__global__ void k_add_pe(float* __restrict__ c0, const float* __restrict__ pe, Dims dm) {
    const int r = blockIdx.x, j = r % dm.Tq;
    for (int c = threadIdx.x; c < dm.d; c += blockDim.x) c0[(size_t)r * dm.d + c] += pe[(size_t)j * dm.d + c];
}
hipError_t launch_add_pe(float* c0, const float* pe, const Dims& dm, hipStream_t s) {
    hipLaunchKernelGGL(k_add_pe, dim3(dm.B * dm.Tq), dim3(256), 0, s, c0, pe, dm);
    return hipGetLastError();
}

// =================================================================================================
// Boundary layout <-> token-major transposes, fused with the sampler arithmetic.
// =================================================================================================
// InputProcess permute (cmdm.py:312): x [B,F,T] -> xin[(b*Tq+etd+t), f]; 32x32 LDS tiles, both sides coalesced.
__global__ __launch_bounds__(256) void k_pack_x(const float* __restrict__ x, float* __restrict__ xin, Planes xp, int copies,
                                                 Dims dm) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int f = f0 + i, t = t0 + tx;
        tile[i][tx] = (f < dm.F && t < dm.T) ? x[((size_t)b * dm.F + f) * dm.T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        if (t < dm.T && f < dm.F) {
            const int row = b * dm.Tq + dm.etd + t;
            if (xin) xin[(size_t)row * dm.F + f] = tile[tx][i];
            if (xp.hi)
                for (int cpy = 0; cpy < copies; ++cpy) plane_put(xp, row + cpy * dm.B * dm.Tq, f, tile[tx][i]);
        }
    }
}
hipError_t launch_pack_x(const float* x, float* xin, Planes xp, int copies, const Dims& dm, hipStream_t s) {
    dim3 grid((dm.T + 31) / 32, (dm.F + 31) / 32, dm.B);
    hipLaunchKernelGGL(k_pack_x, grid, dim3(256), 0, s, x, xin, xp, copies, dm);
    return hipGetLastError();
}

// The element counter is (feature * 4096 + frame), not the flat index: the draw for (sample, step, feature, frame) does
// not depend on the sequence length, so a run truncated to the first frames (auto_regressive evaluation: frame f only
// needs tokens 0..f of a causal decoder) sees the same noise as the full-length run.
__global__ void k_randn(float* __restrict__ x, int B, int FT, int T, unsigned long long seed, unsigned long long off, uint32_t stream) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * FT) return;
    const int b = (int)(idx / FT), e = (int)(idx - (size_t)b * FT);
    x[idx] = philox_normal(seed, off + b, stream, (uint32_t)((e / T) * 4096 + e % T));   // stream 0xFFFFFFFF = x_T draw, k = loop index k's noise
}
hipError_t launch_randn(float* x, int B, int FT, int T, unsigned long long seed, unsigned long long off, uint32_t stream, hipStream_t s) {
    const size_t n = (size_t)B * FT;
    hipLaunchKernelGGL(k_randn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, B, FT, T, seed, off, stream);
    return hipGetLastError();
}

// Sampler update, one 32(f) x 32(t) tile per block.
//   x0 = x0_c                                   (plain)          OutputProcess permute cmdm.py:353-354
//   x0 = x0_u + scale_b * (x0_c - x0_u)         (guided)         cfg_sampler.py:31
//   DDPM  x' = (c1*x0 + c2*x) + sig*eps                          gaussian_diffusion.py:265-276,559
//   DDIM  e = (sr*x - x0)/srm1 ; x' = (x0*ca + cb*e) + sig*eps   gaussian_diffusion.py:419-423,785-793
// Products and sums are rounded separately (no FMA contraction) to follow the reference's op order.
// Writes the new state in boundary layout [B,F,T] and token-major xin for the next step's GEMM.
// In sampling mode the LAST block to finish (of all k_update launches of the step: the chains' launches together cover the
// B samples once) also moves the device-side loop index on: *d_step -= 1 (ticket counters behind d_step[4]). Every block read *d_step before it took its
// ticket, and every other reader of *d_step (the layer kernels of a chain) precedes that chain's k_update in stream order.
__global__ __launch_bounds__(256) void k_update(const float* __restrict__ x0tok, const float* __restrict__ scale,
                                                 const StepCoef* __restrict__ tab, int* d_step,
                                                 const SampleParams* __restrict__ spp, float* __restrict__ xin, Planes xp,
                                                 Dims dm, int b0) {
    __shared__ float tc[32][33];
    __shared__ float tu[32][33];
    const SampleParams sp = *spp;
    const int b = b0 + blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;   // samples [b0, b0 + gridDim.z)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t half = (size_t)dm.B * dm.Tq * dm.F;
    for (int i = ty; i < 32; i += 8) {   // token-major read, coalesced along f
        const int t = t0 + i, f = f0 + tx;
        float c = 0.f, u = 0.f;
        if (t < dm.T && f < dm.F) {
            const size_t o = ((size_t)b * dm.Tq + dm.etd + t) * dm.F + f;
            c = x0tok[o];
            if (sp.guided) u = x0tok[half + o];
        }
        tc[i][tx] = c;
        tu[i][tx] = u;
    }
    __syncthreads();
    const int step = (sp.mode == 0) ? *d_step : 0;
    StepCoef k;
    if (sp.mode == 0) k = tab[step];
    const size_t FT = (size_t)dm.F * dm.T;
    const float sc = sp.guided ? scale[b] : 0.f;
    for (int i = ty; i < 32; i += 8) {   // boundary layout, coalesced along t
        const int f = f0 + i, t = t0 + tx;
        float nv = 0.f;
        if (f < dm.F && t < dm.T) {
            float x0 = tc[tx][i];
            if (sp.guided) {
                const float u = tu[tx][i];
                x0 = __fadd_rn(u, __fmul_rn(sc, __fsub_rn(x0, u)));
            }
            if (sp.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            const size_t o = (size_t)b * FT + (size_t)f * dm.T + t;
            if (sp.x0_out) sp.x0_out[o] = x0;
            if (sp.mode == 0) {
                const float xv = sp.x[o];
                float eps;
                const int bn = sp.const_noise ? 0 : b;              // const_noise: motion 0's draw for every motion (gaussian_diffusion.py:546-547)
                if (sp.noise)
                    eps = sp.noise[(size_t)(sp.first_index - step) * dm.B * FT + (size_t)bn * FT + (size_t)f * dm.T + t];
                else
                    eps = philox_normal(sp.seed, sp.sample_offset + bn, (uint32_t)step, (uint32_t)(f * 4096 + t));   // (feature, frame): independent of T
                if (sp.sampler == 0) {
                    const float mean = __fadd_rn(__fmul_rn(k.c1, x0), __fmul_rn(k.c2, xv));
                    nv = __fadd_rn(mean, __fmul_rn(k.sig_ddpm, eps));
                } else {
                    const float e = __fdiv_rn(__fsub_rn(__fmul_rn(k.sr, xv), x0), k.srm1);
                    const float mean = __fadd_rn(__fmul_rn(x0, k.ca), __fmul_rn(k.cb, e));
                    nv = __fadd_rn(mean, __fmul_rn(k.sig_ddim, eps));
                }
                sp.x[o] = nv;
            }
        }
        tc[tx][i] = nv;
    }
    if (sp.mode != 0) return;
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, f = f0 + tx;
        if (t < dm.T && f < dm.F) {
            const int row = b * dm.Tq + dm.etd + t;
            if (xin) xin[(size_t)row * dm.F + f] = tc[i][tx];
            if (xp.hi) {
                plane_put(xp, row, f, tc[i][tx]);
                if (sp.guided) plane_put(xp, row + dm.B * dm.Tq, f, tc[i][tx]);   // uncond half sees the same x_t
            }
        }
    }
    if (threadIdx.x == 0) {   // two-level tickets (one address takes ~88 atomics per us: 5632 blocks on one counter cost 60 us)
        int* tick = d_step + 4;                                       // [0]: samples done this step, [1 + b]: blocks of sample b done
        if (atomicAdd(tick + 1 + b, 1) == (int)(gridDim.x * gridDim.y) - 1) {
            tick[1 + b] = 0;
            if (atomicAdd(tick, 1) == dm.B - 1) {
                tick[0] = 0;
                d_step[0] = step - 1;
            }
        }
    }
}
hipError_t launch_update(const float* x0tok, const float* scale, const StepCoef* tab, int* d_step,
                         const SampleParams* sp, float* xin, Planes xp, const Dims& dm, int b0, int nb, hipStream_t s) {
    dim3 grid((dm.T + 31) / 32, (dm.F + 31) / 32, nb);
    hipLaunchKernelGGL(k_update, grid, dim3(256), 0, s, x0tok, scale, tab, d_step, sp, xin, xp, dm, b0);
    return hipGetLastError();
}

__global__ void k_advance(int* d_step) { *d_step -= 1; }
hipError_t launch_advance(int* d_step, hipStream_t s) {
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, s, d_step);
    return hipGetLastError();
}

// EmbedAction.forward (cmdm.py:363-365): out[b,:] = table[action[b],:]
__global__ void k_cond_rows(const float* __restrict__ table, const int64_t* __restrict__ action, float* __restrict__ out,
                            int B, int d, int num_actions) {
    const int b = blockIdx.x;
    long long a = action[b];
    a = a < 0 ? 0 : (a >= num_actions ? num_actions - 1 : a);   // never read outside the table (the Python boundary raises IndexError first)
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[(size_t)b * d + c] = table[(size_t)a * d + c];
}
hipError_t launch_cond_rows(const float* table, const int64_t* action, float* out, int B, int d, int num_actions, hipStream_t s) {
    hipLaunchKernelGGL(k_cond_rows, dim3(B), dim3(256), 0, s, table, action, out, B, d, num_actions);
    return hipGetLastError();
}
__global__ void k_fill_rows(float* __restrict__ out, const float* __restrict__ row, int rows, int d) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[(size_t)r * d + c] = row ? row[c] : 0.f;
}
// fp32 -> bf16 (round to nearest even), 8 values per thread: the bf16 copy of the hoisted condition rows k_step adds (rgn_step.hip)
__global__ void k_cvt_bf16(const float* __restrict__ in, __bf16* __restrict__ out, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const f32x4 a = reinterpret_cast<const f32x4*>(in)[2 * i], b = reinterpret_cast<const f32x4*>(in)[2 * i + 1];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)a[e]; o[4 + e] = (__bf16)b[e]; }
    reinterpret_cast<bf16x8*>(out)[i] = o;
}
hipError_t launch_cvt_bf16(const float* in, __bf16* out, size_t n, hipStream_t s) {   // n % 8 == 0
    if (n == 0) return hipSuccess;
    const long long n8 = (long long)(n / 8);
    hipLaunchKernelGGL(k_cvt_bf16, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, in, out, n8);
    return hipGetLastError();
}
// fp32 -> fp16 (rne) and bf16 -> fp16 in place: the fp16-operand form of k_layers (rgn_layers.hip, LayersArgs::f16) reads its condition rows and
// the residual-stream planes the up-front embedding wrote in that format (a bf16 value inside fp16's normal range converts exactly)
__global__ void k_cvt_f16(const float* __restrict__ in, _Float16* __restrict__ out, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const f32x4 a = reinterpret_cast<const f32x4*>(in)[2 * i], b = reinterpret_cast<const f32x4*>(in)[2 * i + 1];
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (_Float16)a[e]; o[4 + e] = (_Float16)b[e]; }
    reinterpret_cast<f16x8*>(out)[i] = o;
}
__global__ void k_bf16_to_f16(__bf16* __restrict__ io, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const bf16x8 a = reinterpret_cast<const bf16x8*>(io)[i];
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)(float)a[e];
    reinterpret_cast<f16x8*>(io)[i] = o;
}
hipError_t launch_cvt_f16(const float* in, _Float16* out, size_t n, hipStream_t s) {   // n % 8 == 0
    if (n == 0) return hipSuccess;
    const long long n8 = (long long)(n / 8);
    hipLaunchKernelGGL(k_cvt_f16, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, in, out, n8);
    return hipGetLastError();
}
hipError_t launch_bf16_to_f16(__bf16* inout, size_t n, hipStream_t s) {   // n % 8 == 0
    if (n == 0) return hipSuccess;
    const long long n8 = (long long)(n / 8);
    hipLaunchKernelGGL(k_bf16_to_f16, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, inout, n8);
    return hipGetLastError();
}
hipError_t launch_fill_rows(float* out, const float* row, int rows, int d, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_rows, dim3(rows), dim3(256), 0, s, out, row, rows, d);
    return hipGetLastError();
}

// =================================================================================================
// next-1 row: rotation_6d_to_matrix (utils/rotation_conversions.py:513-534) — Gram-Schmidt.
// F.normalize semantics: v / max(||v||, 1e-12).
// =================================================================================================
__global__ void k_rot6d(const float* __restrict__ d6, float* __restrict__ mat, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = d6 + i * 6;
    float a1x = p[0], a1y = p[1], a1z = p[2], a2x = p[3], a2y = p[4], a2z = p[5];
    float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float dot = b1x * a2x + b1y * a2y + b1z * a2z;
    float b2x = a2x - dot * b1x, b2y = a2y - dot * b1y, b2z = a2z - dot * b1z;
    const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
    b2x /= n2; b2y /= n2; b2z /= n2;
    float* m = mat + i * 9;
    m[0] = b1x; m[1] = b1y; m[2] = b1z;
    m[3] = b2x; m[4] = b2y; m[5] = b2z;
    m[6] = b1y * b2z - b1z * b2y;
    m[7] = b1z * b2x - b1x * b2z;
    m[8] = b1x * b2y - b1y * b2x;
}
hipError_t launch_rot6d(const float* d6, float* mat, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rot6d, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d6, mat, n);
    return hipGetLastError();
}

// next-2 row: scipy.ndimage.gaussian_filter1d(x, sigma, axis=-1, mode='reflect', truncate=4)
// (sample/cgenerate.py:142). One thread per output sample; taps recomputed (<= 2*4*sigma+1 of them).
__global__ void k_gauss1d(const float* __restrict__ x, float* __restrict__ out, long long rows, int T, float sigma,
                          int radius) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * T) return;
    const long long r = idx / T;
    const int t = (int)(idx - r * T);
    const float* xr = x + r * T;
    double wsum = 0.0, acc = 0.0;
    const double s2 = -0.5 / ((double)sigma * sigma);
    const int period = 2 * T;
    for (int j = -radius; j <= radius; ++j) {
        const double w = exp(s2 * j * j);
        int q = (t + j) % period;
        if (q < 0) q += period;
        if (q >= T) q = period - 1 - q;
        wsum += w;
        acc += w * (double)xr[q];
    }
    out[idx] = (float)(acc / wsum);
}
hipError_t launch_gauss1d(const float* x, float* out, long long rows, int T, float sigma, hipStream_t s) {
    const long long n = rows * T;
    if (n <= 0) return hipSuccess;
    const int radius = (int)(4.0f * sigma + 0.5f);
    hipLaunchKernelGGL(k_gauss1d, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, rows, T, sigma, radius);
    return hipGetLastError();
}

}  // namespace rgn
