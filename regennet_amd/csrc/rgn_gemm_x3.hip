// Split-bf16 ("bf16x3") MFMA GEMM on PRE-SPLIT operands — the dominant kernel of the hot path.
//
//   C[M,N] = act( (Ahi+Alo)[M,K] . (Whi+Wlo)[N,K]^T + bias + add )     with  a.b ~ ah.bh + ah.bl + al.bh
//
// Every activation that feeds a GEMM is stored by its producer as two bf16 planes (hi = rne(x),
// lo = rne(x - hi)), so this kernel is a pure bf16 GEMM with four operand planes: no VALU conversion in
// the main loop and all four planes go HBM/L2 -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4,
// 1 KiB per wave-instruction, no VGPR round trip).
//
// Operand memory layout ("K32-blocked planes"): element (row, k) of a plane lives at
//   ((k/32) * rows + row) * 32 + k%32          i.e. [Kp/32][rows][32] bf16
// so the [rows x 32] slice a block needs for one k-step is ONE contiguous run of 64-byte rows: every DMA
// wave-instruction (16 rows x 64 B) reads a contiguous, 1 KiB-aligned span = 8 full 128-byte lines.
// (With a plain row-major [rows][K] plane each instruction would touch 16 half-used lines.)
// The producers (LayerNorm, GEMM epilogues, attention, sampler update) write this layout directly.
//
// LDS image of one plane tile: [rows][32] bf16, 64-byte rows, written lane-linear by the DMA. To make the
// ds_read_b128 fragment reads conflict-free the 16-byte chunk c of row r is stored at chunk position
// c ^ ((r>>2)&3): the permutation is applied to the per-lane global SOURCE address and again on the read
// (both-sides-or-neither rule for global_load_lds).
//
// Loop (shipped flavour): 128x128 tile, 4 waves, 2 LDS stages of 32 KiB -> two workgroups per CU, one hiding the
// other's barriers and epilogue. Per k-step: issue tile kt+1's DMA, counted wait (vmcnt) for tile kt, barrier, fragments
// + 24 MFMAs per wave, barrier. Measured with per-step cycle stamps (tools/gemm_bench -DRGN_GEMM_PROF): the step is
// bound by the ~20 B/clk a CU can pull from L2 through the vector-memory path (the 64 KiB the two resident workgroups
// request per step take ~3000 clk to land, the MFMAs ~1500), not by the matrix pipe. A 256x256 / 8-wave tile with one
// barrier per step and the DMA issued between MFMA groups (ILV = true, tools only) halves the bytes per MFMA and its
// loop runs at ~80 % MFMA occupancy, but at M = 15360 it leaves only 120-240 tiles for 256 CUs and nothing to hide its
// 256 KiB-per-tile epilogue behind: end to end it only ties (70 vs 72 us on linear1), so it is not shipped.
#include <algorithm>
#include <cstdlib>
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7) on fast exp/rcp: ~12 VALU instead of ~30 for erff();
// far inside the fp32 noise of the surrounding GEMM and of the 1e-3 tolerance.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(e, x);
}
__device__ __forceinline__ float x3_act(float v, int act) {
    if (act == 1) return v * 0.5f * (1.0f + fast_erf(v * 0.70710678118654752440f));
    if (act == 2) return v / (1.0f + __expf(-v));
    if (act == 3) return fmaxf(v, 0.f);
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}

// ---- epilogue ------------------------------------------------------------------------------------------
// Straight from the MFMA C/D layout (col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)): for a fixed register
// the two half-waves write two full 128-byte row segments (fp32) or two 64-byte segments of the K32-blocked bf16
// planes. Interior blocks take a branch-free path (CHECK = false): one pointer per tile, constant row strides,
// residual loads batched per tile; only edge blocks pay per-element bounds checks.
// mw / nw: first row / column of this wave's TM x TN tiles of 32x32.
// EPI: 0 plain, 1 the attention-ready scatter of the packed in_proj output, 2 the ST-GCN evaluator's GEMMs ((row % add_mod) addend)
template <int TM, int TN, int EPI, bool CHECK>
__device__ __forceinline__ void x3_epilogue(const GemmX3Args& g, f32x16 (&acc)[TM][TN], int mw, int nw, int lane) {
    constexpr bool QKV = EPI == 1;
    const int l31 = lane & 31, kh = lane >> 5;
        #pragma unroll
    for (int tb = 0; tb < TN; ++tb) {
        const int n = nw + tb * 32 + l31;
        const bool n_ok = !CHECK || n < g.N;
        const float bias = (g.bias && n_ok) ? g.bias[n] : 0.f;
#pragma unroll
        for (int ta = 0; ta < TM; ++ta) {
            const int mb = mw + ta * 32 + 4 * kh;      // row of register 0
            float r[16];
            if (EPI == 2 && g.add && g.add_mod > 0) {      // addend row = row % add_mod (a per-vertex bias: rows run (frame, vertex); add_mod >= 28)
                const int base = (int)((unsigned)mb % (unsigned)g.add_mod);
                const float* ap = g.add + n;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ro = (i & 3) + 8 * (i >> 2);
                    int rr = base + ro;
                    rr = rr >= g.add_mod ? rr - g.add_mod : rr;
                    r[i] = n_ok ? ap[rr * g.ldadd] : 0.f;
                }
            } else if (g.add) {
                const float* ap = g.add + (size_t)mb * g.ldadd + n;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ro = (i & 3) + 8 * (i >> 2);
                    r[i] = (!CHECK || (n_ok && mb + ro < g.M)) ? ap[(size_t)ro * g.ldadd] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = acc[ta][tb][i] + bias;
                if (g.add) v += r[i];
                r[i] = x3_act(v, g.act);
            }
            if constexpr (QKV) {   // attention-ready scatter of the packed in_proj output (see GemmX3Args)
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const int which = n / g.d, cin = n - which * g.d, hd = cin / g.dh, c = cin - hd * g.dh;
                __bf16* ph = which == 0 ? g.Qhi : (which == 1 ? g.Khi : g.Vthi);
                __bf16* pl = which == 0 ? g.Qlo : (which == 1 ? g.Klo : g.Vtlo);
                if (!CHECK) {                                          // pair adjacent columns across lane^1 -> 4-byte stores
                    const float sc = which == 0 ? g.qscale : 1.0f;
                    const bool odd = lane & 1;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float mine = (odd ? r[i + 8] : r[i]) * sc;
                        const float give = (odd ? r[i] : r[i + 8]) * sc;
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));   // lane ^ 1: quad_perm [1, 0, 3, 2]
                        const float c0 = odd ? got : mine, c1 = odd ? mine : got;
                        const int ii = odd ? i + 8 : i;
                        const unsigned m = (unsigned)(mb + (ii & 3) + 8 * (ii >> 2));
                        const unsigned bb = __umulhi(m, g.tq_magic), tt = m - bb * (unsigned)g.Tq;   // m / Tq, m % Tq (exact: m*Tq < 2^32)
                        const size_t o = (((size_t)bb * g.H + hd) * g.Tqp + tt) * g.dh + (c & ~1);
                        const __bf16 h0 = (__bf16)c0, h1 = (__bf16)c1;
                        bf16x2 hv = {h0, h1};
                        *reinterpret_cast<bf16x2*>(ph + o) = hv;
                        if (pl) {
                            bf16x2 lv = {(__bf16)(c0 - (float)h0), (__bf16)(c1 - (float)h1)};
                            *reinterpret_cast<bf16x2*>(pl + o) = lv;
                        }
                    }
                } else {                                              // edge blocks
                    const float sc = which == 0 ? g.qscale : 1.0f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int m = mb + (i & 3) + 8 * (i >> 2);
                        if (!n_ok || m >= g.M) continue;
                        const int bb = m / g.Tq, tt = m - bb * g.Tq;
                        const size_t sl = (size_t)bb * g.H + hd;
                        const float x = r[i] * sc;
                        const __bf16 h = (__bf16)x;
                        const size_t o = (sl * g.Tqp + tt) * g.dh + c;
                        ph[o] = h;
                        if (pl) pl[o] = (__bf16)(x - (float)h);
                    }
                }
            }
            if (!QKV && g.C) {
                float* cp = g.C + (size_t)mb * g.ldc + n;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ro = (i & 3) + 8 * (i >> 2);
                    if (!CHECK || (n_ok && mb + ro < g.M)) cp[(size_t)ro * g.ldc] = r[i];
                }
            }
            if (!QKV && g.Chi) {   // K32-blocked planes [N/32][c_rows][32]: a 32-column tile is one contiguous run of rows
                if (!CHECK) {
                    // pair adjacent columns across lanes (lane^1) so every store is a packed bf16x2 (4 B):
                    // even lanes store rows of registers 0..7, odd lanes those of registers 8..15
                    const size_t o = ((size_t)(n >> 5) * g.c_rows + mb) * 32 + (n & 30);
                    const bool odd = lane & 1;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float mine = odd ? r[i + 8] : r[i];          // value this lane contributes to its own store
                        const float give = odd ? r[i] : r[i + 8];          // value the partner needs
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));   // lane ^ 1: quad_perm [1, 0, 3, 2]
                        const float lo_col = odd ? got : mine, hi_col = odd ? mine : got;   // columns n&~1, (n&~1)+1
                        const int ii = odd ? i + 8 : i;
                        const int ro = (ii & 3) + 8 * (ii >> 2);
                        const __bf16 h0 = (__bf16)lo_col, h1 = (__bf16)hi_col;
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        bf16x2 hv = {h0, h1};
                        *reinterpret_cast<bf16x2*>(g.Chi + o + ro * 32) = hv;
                        if (g.Clo) {
                            bf16x2 lv = {(__bf16)(lo_col - (float)h0), (__bf16)(hi_col - (float)h1)};
                            *reinterpret_cast<bf16x2*>(g.Clo + o + ro * 32) = lv;
                        }
                    }
                } else {
                    const size_t o = ((size_t)(n >> 5) * g.c_rows + mb) * 32 + (n & 31);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ro = (i & 3) + 8 * (i >> 2);
                        if (n_ok && mb + ro < g.M) {
                            const __bf16 h = (__bf16)r[i];
                            g.Chi[o + ro * 32] = h;
                            if (g.Clo) g.Clo[o + ro * 32] = (__bf16)(r[i] - (float)h);
                        }
                    }
                }
            }
        }
    }
}

#ifdef RGN_GEMM_PROF
__device__ long long g_prof[1024];   // tools only: per-k-step cycle stamps of one workgroup
#endif

// BM x BN block tile, WM x WN waves, each wave (BM/WM) x (BN/WN) = TM x TN tiles of 32x32.
// ILV = false: two barriers per k-step, the whole next tile's DMA issued at the top of the step (128x128, 2 WG / CU).
// ILV = true : one barrier per k-step, DMA pieces interleaved with the MFMAs of the first K half, fragments fetched one
//              MFMA group ahead (256x256, 1 WG / CU).
template <int BM, int BN, int WM, int WN, bool X3, bool ILV, int EPI>
__global__ __launch_bounds__(64 * WM * WN, (2 * 2 * (BM + BN) * 64 <= 80 * 1024) ? 2 : 1) void k_gemm_x3(GemmX3Args g, int nbx, int nby) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NPL = X3 ? 2 : 1;                         // planes per operand
    constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64;     // one plane tile (32 bf16 per row)
    constexpr int STAGE = NPL * (A_BYTES + W_BYTES);
    constexpr int A_IT = BM * 4 / NT, W_IT = BN * 4 / NT;   // DMA instructions per thread per plane
    constexpr int LPT = NPL * (A_IT + W_IT);                // ... per thread per tile
    static_assert(BM * 4 % NT == 0 && BN * 4 % NT == 0, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][Ahi|Alo|Whi|Wlo]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = nbx * nby, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int m0 = (vid / nbx) * BM, n0 = (vid % nbx) * BN;

    // per-thread DMA source byte offsets inside one k-block of a plane (k-invariant, 32-bit: a plane k-block is
    // rows x 64 B); the k-dependent part is a wave-uniform base pointer, so the DMA uses the SGPR-base + VGPR-offset form
    unsigned a_src[A_IT], w_src[W_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int q = it * NT + tid, r = q >> 2, c = (q & 3) ^ ((r >> 2) & 3);
        int m = m0 + r;
        m = m < g.M ? m : g.M - 1;
        a_src[it] = (unsigned)m * 64u + c * 16u;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int q = it * NT + tid, r = q >> 2, c = (q & 3) ^ ((r >> 2) & 3);
        int n = n0 + r;
        n = n < g.N ? n : g.N - 1;
        w_src[it] = (unsigned)n * 64u + c * 16u;
    }
    // DMA piece idx of tile kt into stage buffer sb (idx is a compile-time constant after unrolling)
    auto piece = [&](int idx, int kt, char* sb) {
        ptrdiff_t ka = (ptrdiff_t)kt * g.a_rows * 64;
        if constexpr (EPI == 2) {   // temporal convolution as one GEMM: k-block -> (tap, channel block), the tap shifts the rows
            if (g.a_taps > 0) {
                const int cb = kt / g.a_taps;
                ka = (ptrdiff_t)cb * g.a_rows * 64 + (ptrdiff_t)g.a_tap[kt - cb * g.a_taps];
            }
        }
        const size_t kw = (size_t)kt * g.N * 64;
        if (idx < NPL * A_IT) {
            const int it = idx / NPL, pl = idx % NPL;
            const int lo = (it * NT + (tid & ~63)) * 16;   // wave-uniform LDS byte offset of this 1 KiB piece
            const char* base = reinterpret_cast<const char*>(pl ? g.Alo : g.Ahi) + ka;
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(base + a_src[it]), (RGN_AS3 void*)(sb + pl * A_BYTES + lo), 16, 0, 0);
        } else {
            const int j = idx - NPL * A_IT, it = j / NPL, pl = j % NPL;
            const int lo = (it * NT + (tid & ~63)) * 16;
            const char* base = reinterpret_cast<const char*>(pl ? g.Wlo : g.Whi) + kw;
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(base + w_src[it]), (RGN_AS3 void*)(sb + NPL * A_BYTES + pl * W_BYTES + lo), 16, 0, 0);
        }
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int idx = 0; idx < LPT; ++idx) piece(idx, kt, smem + stage * STAGE);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // fragment read offsets: row rr (within the block tile), chunk c = 2*ks + khalf at position c ^ ((rr>>2)&3)
    const int l31 = lane & 31, kh = lane >> 5;
    int a_off[TM][2], w_off[TN][2];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int rr = wm * (BM / WM) + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int rr = wn * (BN / WN) + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    // the three products of one C tile (a.w ~ al.wh + ah.wl + ah.wh, small terms first)
    auto mma3 = [&](f32x16& c, const bf16x8& a_h, const bf16x8& a_l, const bf16x8& w_h, const bf16x8& w_l) {
        if (X3) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, w_h, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, w_l, c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, w_h, c, 0, 0, 0);
    };

#ifdef RGN_GEMM_PROF
    const bool prof = (bid == RGN_GEMM_PROF) && (tid == 0 || tid == NT - 64);
    long long* pp = g_prof + (tid ? 512 : 0);
#define RGN_T(i) if (prof) pp[kt * 6 + i] = clock64();
#else
#define RGN_T(i)
#endif
    const int nk = g.Kp / 32;
    if constexpr (ILV) {
        constexpr int NG = 2 * TN;                               // MFMA groups per k-step: (K half, N tile), TM tiles each
        constexpr int PPG = (LPT + TN - 1) / TN;                 // DMA pieces after each group of the first K half
        auto step = [&](int kt, bool more) {
            const char* sb = smem + (kt & 1) * STAGE;
            char* nb = smem + ((kt + 1) & 1) * STAGE;
            bf16x8 ah[2][TM], al[2][TM], wh[2], wl[2];
            auto fetch = [&](int grp) {                          // fragments of group grp = ks*TN + tb
                const int ks = grp / TN, tb = grp % TN;
                if (tb == 0) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        ah[ks][t] = *reinterpret_cast<const bf16x8*>(sb + a_off[t][ks]);
                        if (X3) al[ks][t] = *reinterpret_cast<const bf16x8*>(sb + A_BYTES + a_off[t][ks]);
                    }
                }
                wh[grp & 1] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + w_off[tb][ks]);
                if (X3) wl[grp & 1] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + W_BYTES + w_off[tb][ks]);
            };
            fetch(0);
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                if (grp + 1 < NG) fetch(grp + 1);
#pragma unroll
                for (int ta = 0; ta < TM; ++ta) mma3(acc[ta][grp % TN], ah[grp / TN][ta], al[grp / TN][ta], wh[grp & 1], wl[grp & 1]);
                if (grp < TN && more) {
#pragma unroll
                    for (int q = 0; q < PPG; ++q)
                        if (grp * PPG + q < LPT) piece(grp * PPG + q, kt + 1, nb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        issue(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            RGN_T(0)
            wait_vmcnt<0>();                      // my pieces of tile kt landed
            RGN_T(1)
            __builtin_amdgcn_s_barrier();         // ... everyone's did, and everyone is done reading tile kt-1's stage
            RGN_T(2)
            step(kt, kt + 1 < nk);
            RGN_T(3)
            RGN_T(4)
            RGN_T(5)
        }
    } else {
        auto compute = [&](const char* sb) {
            bf16x8 ah[2][TM], al[2][TM], wh[2][TN], wl[2][TN];
            auto frags = [&](int ks, int buf) {
#pragma unroll
                for (int t = 0; t < TM; ++t) {
                    ah[buf][t] = *reinterpret_cast<const bf16x8*>(sb + a_off[t][ks]);
                    if (X3) al[buf][t] = *reinterpret_cast<const bf16x8*>(sb + A_BYTES + a_off[t][ks]);
                }
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    wh[buf][t] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + w_off[t][ks]);
                    if (X3) wl[buf][t] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + W_BYTES + w_off[t][ks]);
                }
            };
            frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 0) frags(1, 1);
#pragma unroll
                for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                    for (int tb = 0; tb < TN; ++tb) mma3(acc[ta][tb], ah[ks][ta], al[ks][ta], wh[ks][tb], wl[ks][tb]);
            }
        };
        issue(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            RGN_T(0)
            if (kt + 1 < nk) {
                issue(kt + 1, (kt + 1) & 1);
                RGN_T(1)
                wait_vmcnt<LPT>();                // tile kt landed, tile kt+1 stays in flight
            } else {
                RGN_T(1)
                wait_vmcnt<0>();
            }
            RGN_T(2)
            __builtin_amdgcn_s_barrier();
            RGN_T(3)
            compute(smem + (kt & 1) * STAGE);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RGN_T(4)
            __builtin_amdgcn_s_barrier();         // stage (kt&1) may now be overwritten by tile kt+2
            RGN_T(5)
        }
    }
#undef RGN_T

    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
    if (interior) x3_epilogue<TM, TN, EPI, false>(g, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
    else x3_epilogue<TM, TN, EPI, true>(g, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
}

template <int BM, int BN, int WM, int WN, bool ILV>
static hipError_t x3_launch(const GemmX3Args& g, bool x3, hipStream_t s, bool configure_only) {
    const int lds = 2 * (x3 ? 2 : 1) * (BM * 64 + BN * 64);
    if (configure_only) {
        const int big = 2 * 2 * (BM * 64 + BN * 64), small = 2 * (BM * 64 + BN * 64);
        hipError_t e;
#define RGN_CFG(X3V, QV, BYTES)                                                                                        \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_x3<BM, BN, WM, WN, X3V, ILV, QV>),                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, BYTES);                                        \
        if (e != hipSuccess) return e;
        RGN_CFG(true, 0, big) RGN_CFG(false, 0, small)
        if constexpr (BM == 128 && !ILV) { RGN_CFG(true, 1, big) RGN_CFG(false, 1, small) }
#undef RGN_CFG
        return hipSuccess;
    }
    const int nbx = (g.N + BN - 1) / BN, nby = (g.M + BM - 1) / BM;
    const dim3 grid(nbx * nby), block(64 * WM * WN);
    const bool qkv = g.Qhi != nullptr;
    if constexpr (BM == 128 && !ILV) {   // the attention-ready scatter exists for the default tile only
        if (qkv) {
            if (x3) hipLaunchKernelGGL((k_gemm_x3<BM, BN, WM, WN, true, ILV, 1>), grid, block, lds, s, g, nbx, nby);
            else hipLaunchKernelGGL((k_gemm_x3<BM, BN, WM, WN, false, ILV, 1>), grid, block, lds, s, g, nbx, nby);
            return hipGetLastError();
        }
    } else if (qkv) {
        return hipErrorInvalidValue;
    }
    if (x3) hipLaunchKernelGGL((k_gemm_x3<BM, BN, WM, WN, true, ILV, 0>), grid, block, lds, s, g, nbx, nby);
    else hipLaunchKernelGGL((k_gemm_x3<BM, BN, WM, WN, false, ILV, 0>), grid, block, lds, s, g, nbx, nby);
    return hipGetLastError();
}

// ---- the ST-GCN evaluator's GEMMs (rgn_stgcn.hip): millions of rows x 64 / 128 / 256 channels, split-bf16 throughout. 256-row tiles with
//      the interleaved one-barrier loop; the tile is as wide as the layer (a 64-channel layer on a 128-wide tile would idle half its MFMAs)
template <int BN, int WN>
static hipError_t sg_launch(const GemmX3Args& g, hipStream_t s, bool configure_only) {
    constexpr int BM = 256, WM = 4;
    const int lds = 2 * 2 * (BM * 64 + BN * 64);
    if (configure_only)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_x3<BM, BN, WM, WN, true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int nbx = (g.N + BN - 1) / BN, nby = (g.M + BM - 1) / BM;
    hipLaunchKernelGGL((k_gemm_x3<BM, BN, WM, WN, true, true, 2>), dim3(nbx * nby), dim3(64 * WM * WN), lds, s, g, nbx, nby);
    return hipGetLastError();
}
hipError_t launch_gemm_x3_sg(const GemmX3Args& g, hipStream_t s) {
    if (g.N <= 64) return sg_launch<64, 1>(g, s, false);
    // 256 channels: two 128-wide column blocks. The 256-wide tile (123 spilled SGPRs + 576 B of scratch in this form) was measured anyway: 23.2 / 53.2 ms
    // per forward at 60 / 150 frames against 23.0 / 53.2 - the wide blocks are not bound by operand bytes per MFMA (profiles/r05/sg_tile256.txt)
    return sg_launch<128, 2>(g, s, false);
}
hipError_t configure_gemm_x3_sg() {
    GemmX3Args g{};
    hipError_t e = sg_launch<64, 1>(g, nullptr, true);
    return e != hipSuccess ? e : sg_launch<128, 2>(g, nullptr, true);
}


// variant 0 = 128x128 / 4 waves (two workgroups per CU): the default. variant 1 = 256x256 / 8 waves / interleaved DMA (one
// workgroup per CU): its loop is ~1.45x faster per output but it needs several tiles per CU to hide its 256 KiB-per-tile
// epilogue, so the engine picks it only for launches of >= 7000 rows per chain (rgn_plan.cpp). tools/gemm_bench builds
// with RGN_GEMM_TOOLS and can also time 2 = 256x256 with the two-barrier loop, 3 = 128x128 with the interleaved loop,
// 4 / 5 = 128x256 / 256x128 interleaved.
hipError_t launch_gemm_x3(const GemmX3Args& g, bool x3, int variant, hipStream_t s) {
    if (variant == 1) return x3_launch<256, 256, 4, 2, true>(g, x3, s, false);
#ifdef RGN_GEMM_TOOLS
    if (variant == 2) return x3_launch<256, 256, 4, 2, false>(g, x3, s, false);
    if (variant == 3) return x3_launch<128, 128, 2, 2, true>(g, x3, s, false);
    if (variant == 4) return x3_launch<128, 256, 2, 4, true>(g, x3, s, false);
    if (variant == 5) return x3_launch<256, 128, 4, 2, true>(g, x3, s, false);
#endif
    return x3_launch<128, 128, 2, 2, false>(g, x3, s, false);
}
hipError_t configure_gemm_x3() {
    GemmX3Args g{};
    hipError_t e = x3_launch<128, 128, 2, 2, false>(g, true, nullptr, true);
    if (e != hipSuccess) return e;
#ifdef RGN_GEMM_TOOLS
    e = x3_launch<256, 256, 4, 2, false>(g, true, nullptr, true);
    if (e != hipSuccess) return e;
    e = x3_launch<128, 128, 2, 2, true>(g, true, nullptr, true);
    if (e != hipSuccess) return e;
    e = x3_launch<128, 256, 2, 4, true>(g, true, nullptr, true);
    if (e != hipSuccess) return e;
    e = x3_launch<256, 128, 4, 2, true>(g, true, nullptr, true);
#endif
    if (e != hipSuccess) return e;
    return x3_launch<256, 256, 4, 2, true>(g, true, nullptr, true);
}

#ifdef RGN_GEMM_PROF
void gemm_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 1024); }
#endif

}  // namespace rgn
