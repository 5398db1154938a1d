// Row-complete bf16 GEMM for the plain-bf16 phase of the precision schedule, with the layer's elementwise tail fused:
//
//   EPI_LN  : t = A . W^T + bias + resid ;  y = LN_a(t) ;  out = LN_b(y + pervec[row / Tq] + stepvec[*d_step]) (optional)
//             -> out_proj + residual + norm1 (+ folded 1-token cross-attention + norm2), linear2 + residual + norm3
//                (nn.TransformerDecoderLayer post-norm blocks constructed at model/cmdm.py:75-81, called at :227)
//   EPI_ACT : out = act(A . W^T + bias + add)  as bf16 planes and / or fp32
//             -> linear1 + GELU, InputProcess / fuse (cmdm.py:201-218), OutputProcess.poseFinal (cmdm.py:337)
//
// Why a second GEMM family next to k_gemm_x3: with single-plane bf16 operands the 128 x 128 kernel spends more than half
// of its time outside the MFMA loop (C tile write, LayerNorm round trip through an fp32 tensor, launch), and its loop
// is bound by how many operand bytes a CU keeps in flight towards L2 (tools/l2_paths_bench: ~4 B/clk per wave with 4-8 KiB
// outstanding, i.e. latency-bound; ~60-70 B/clk/CU only with > 100 KiB outstanding). This kernel is built around that:
//
//   * A workgroup owns 64 COMPLETE rows x 512 columns (8 waves, wave w = columns [64 w, 64 w + 64)), so LayerNorm
//     statistics never leave the workgroup and the pre-norm tensor never exists in memory.
//   * The activation tile (64 rows x K, 64 KiB at K = 512, 128 KiB at K = 1024) is DMA'd into LDS ONCE (direct-to-LDS,
//     same swizzled image as k_gemm_x3) and stays resident: ONE barrier before the k-loop, none inside it.
//   * Every weight row is needed by exactly one wave of the workgroup, so weights bypass LDS: each lane loads its MFMA
//     fragment straight from the K32-blocked plane (global_load_dwordx4) into a 4-deep register ring, 3 k-steps
//     (12 KiB per wave, 96 KiB per CU) ahead of the MFMAs that consume it; counted s_waitcnt vmcnt keeps the ring full.
//   * Nothing leaves the accumulator layout through global memory (scattered 8-byte accesses from it measured ~45 k cycles
//     per tile, 3x the k-loop): EPI_LN parks acc + bias in an fp32 LDS row buffer that aliases the dead activation image
//     and then runs one wave per row exactly like k_layernorm (16-byte plane runs; the residual rows are requested while
//     the last k-steps run); EPI_ACT writes its bf16 result into a swizzled LDS image of the output plane tile and copies
//     it out as contiguous 1 KiB wave-stores.
//
// Used only when the evaluation runs the plain-bf16 phase (RGN_PREC_BF16_X3TAIL, loop indices >= tail) and d == 512;
// the split-bf16 tail and other widths keep k_gemm_x3 + k_layernorm.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

#ifndef RGN_RG_D_LN
#define RGN_RG_D_LN 3
#endif
constexpr int RG_BM = 64, RG_BN = 512, RG_NT = 512;
// weight prefetch distance in k-steps (register ring of D + 1 slots of 16 VGPRs): per epilogue kind, the ACT build has to fit 128 VGPRs
template <int EPI> struct RgDepth { static constexpr int D = EPI == 0 ? RGN_RG_D_LN : 3; };

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) for the plain-bf16 phase: erf as an odd degree-15 polynomial in u = clamp(x / sqrt 2,
// +-3.2) (weighted least-squares fit, max abs error 1.6e-4 -> relative GELU error <= 8e-5, 25x below the bf16 rounding of the
// result), evaluated two elements at a time with packed fp32 FMAs: no transcendental instruction (v_exp / v_rcp issue at
// quarter rate; the A&S form of k_gemm_x3 spends ~40 % of its VALU time in them).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 rg_gelu2(f32x2 x) {
    f32x2 u = x * 0.70710678118654752440f;
    u = __builtin_elementwise_min(__builtin_elementwise_max(u, f32x2{-3.2f, -3.2f}), f32x2{3.2f, 3.2f});
    const f32x2 z = u * u;
    f32x2 p = f32x2{-2.6911866e-07f, -2.6911866e-07f};
    p = __builtin_elementwise_fma(p, z, f32x2{1.2661994e-05f, 1.2661994e-05f});
    p = __builtin_elementwise_fma(p, z, f32x2{-2.5566161e-04f, -2.5566161e-04f});
    p = __builtin_elementwise_fma(p, z, f32x2{2.9286479e-03f, 2.9286479e-03f});
    p = __builtin_elementwise_fma(p, z, f32x2{-2.1317327e-02f, -2.1317327e-02f});
    p = __builtin_elementwise_fma(p, z, f32x2{1.0528564e-01f, 1.0528564e-01f});
    p = __builtin_elementwise_fma(p, z, f32x2{-3.7135834e-01f, -3.7135834e-01f});
    p = __builtin_elementwise_fma(p, z, f32x2{1.1274883e+00f, 1.1274883e+00f});
    const f32x2 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, p * u, hx);                  // 0.5 x (1 + erf)
}

// wave-wide sum on the VALU (DPP within rows of 16 lanes, then the four row totals through SGPRs): ~15 instructions
// with no LDS round trip; ds_bpermute butterflies (what __shfl_xor compiles to) cost ~1.4 k cycles per row here.
template <int CTRL>
__device__ __forceinline__ float rg_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float rg_wave_sum(float v) {
    v += rg_dpp<0xB1>(v);                                             // quad_perm [1,0,3,2]: lane ^ 1
    v += rg_dpp<0x4E>(v);                                             // quad_perm [2,3,0,1]: lane ^ 2
    v += rg_dpp<0x141>(v);                                            // row_half_mirror: sums of 8
    v += rg_dpp<0x140>(v);                                            // row_mirror: sums of 16
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
           __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}

template <int N>
__device__ __forceinline__ void rg_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}

#ifdef RGN_RG_PROF
__device__ long long g_rg_prof[8 * 16];   // tools only: phase cycle stamps (s_memtime) of wave w of workgroup RGN_RG_PROF
#define RGN_RT(i) if (blockIdx.x == RGN_RG_PROF && blockIdx.y == 0 && lane == 0) g_rg_prof[wave * 16 + (i)] = __builtin_readcyclecounter();
#else
#define RGN_RT(i)
#endif

// EPI: 0 = EPI_LN, 1 = EPI_ACT. NK = Kp / 32 is a template parameter: the k-loop is fully unrolled, so the compiler's own
// s_waitcnt bookkeeping stays exact (across a loop back-edge it falls back to vmcnt(0) and drains the prefetch ring).
// linear1 + GELU (K = 512, hi-only output: 64 KiB of LDS) is built for 128 VGPRs so that two workgroups share a CU: one's
// prologue / GELU epilogue overlaps the other's k-loop (measured 31.6 -> 28.4 us at M = 15360)
#ifndef RGN_RG_ACT_WAVES
#define RGN_RG_ACT_WAVES 4
#endif
template <int EPI, int NK>
__global__ __launch_bounds__(RG_NT, (EPI == 1 && NK <= 16) ? RGN_RG_ACT_WAVES : 2) void k_rowgemm(RowGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [nk][64 rows][64 B] activation image | 8 KiB reduction scratch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * RG_BM, nblk0 = blockIdx.y * RG_BN;
    const int nw = nblk0 + wave * 64;                                // first column of this wave
    constexpr int nk = NK;
    constexpr int RG_D = RgDepth<EPI>::D, RING = RG_D + 1;
    static_assert(NK > RG_D && NK <= 32, "prefetch distance; activation image <= 128 KiB");

    RGN_RT(0)
    // ---- activation tile -> LDS, once: k-block kb = 4 wave-instructions of 1 KiB (16 rows each); wave w issues the
    //      pieces p = w, w + 8, ... of the nk * 4 pieces (p = 4 kb + q: rows [16 q, 16 q + 16) of k-block kb)
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < nk / 2; ++j) {                            // compile-time trip count: the waitcnt pass can count
            const int p = wave + 8 * j;
            const int kb = p >> 2, q = p & 3, r = q * 16 + r16;
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            const __bf16* src = g.A + ((size_t)kb * g.a_rows + m) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)src, (RGN_AS3 void*)(smem + p * 1024), 16, 0, 0);
        }
    }
    // ---- weight fragments, from the FRAGMENT-ORDERED plane [Kp/32][N/32][2 ks][64 lanes][8]: lane (l31, kh) of column
    //      block nb holds W[32 nb + l31][32 kt + 16 ks + 8 kh .. + 8], i.e. exactly its MFMA A-operand fragment, and one
    //      wave-load is one contiguous 1 KiB run (loading the same fragments from the K32-blocked plane touches sixteen
    //      half-used lines per instruction and ran the k-loop at 27 B/clk/CU)
    const int nb_all = g.N >> 5;                                      // N % 32 == 0 (host-checked)
    int nbw[2];                                                       // this wave's two column blocks (clamped: N < 512)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int nb = (nw >> 5) + nt;
        nbw[nt] = nb < nb_all ? nb : nb_all - 1;
    }
    bf16x8 wf[RG_D + 1][2][2];                                        // [ring slot][ks][nt]
    const unsigned lane8 = (unsigned)lane * 8u;
    auto load_w = [&](int kt, int slot) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const __bf16* base = g.W + ((size_t)kt * nb_all + nbw[nt]) * 1024;   // wave-uniform: scalar base + the lane's 32-bit offset
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[slot][ks][nt] = *reinterpret_cast<const bf16x8*>(base + ks * 512 + lane8);
        }
    };
    int a_off[2][2];                                                  // [mt][ks]: B-operand fragment of token 32 mt + l31
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int rr = 32 * mt + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[mt][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    f32x16 acc[2][2];                                                 // [nt][mt]: rows (registers) = columns n, lane = token
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

#pragma unroll
    for (int s = 0; s < RG_D; ++s) load_w(s, s);                      // nk >= 4 (host-checked)
    RGN_RT(1)
    rg_wait_vmcnt<4 * RG_D>();                                        // in order: the activation pieces landed, weights may still fly
    RGN_RT(2)
    __builtin_amdgcn_s_barrier();
    RGN_RT(3)

    // ---- LN epilogue inputs that can travel early: wave w normalises the rows 8 w .. 8 w + 7 of the tile, lane = 8
    //      consecutive columns (one 16-byte run of a plane row). Their residual values are requested while the last
    //      k-steps run, so the row phase starts with them in registers.
    constexpr int RLD = RG_BN + 4;                                    // fp32 row buffer stride (floats)
    const int c0 = 8 * lane;                                          // LN row phase: this lane's first column
    bf16x8 rh[8], rl[8];
    auto load_resid = [&] {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int m = m0 + 8 * wave + j;
            m = m < g.M ? m : g.M - 1;
            const size_t o = ((size_t)(c0 >> 5) * g.r_rows + m) * 32 + (c0 & 31);
            rh[j] = *reinterpret_cast<const bf16x8*>(g.Rhi + o);
            if (g.Rlo) rl[j] = *reinterpret_cast<const bf16x8*>(g.Rlo + o);
        }
    };

    // one k-step: activation fragments from LDS, then (optionally) the weight prefetch D steps ahead, then the MFMAs.
    // The prefetch is issued AFTER the fragment reads: the first ds_read behind the tile DMA makes the compiler drain
    // vmcnt completely (it cannot prove the read does not alias the DMA'd bytes) - with this order that one drain
    // covers only the prologue loads, never a prefetch.
    auto step = [&](int kt, int slot, int pf_kt, int pf_slot, auto wait) {
        const char* sb = smem + kt * 4096;
        bf16x8 af[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[ks][mt] = *reinterpret_cast<const bf16x8*>(sb + a_off[mt][ks]);
        asm volatile("" ::: "memory");
        if (pf_kt >= 0) load_w(pf_kt, pf_slot);
        wait();                                                       // this step's weight fragments are in; later ones stay in flight
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[slot][ks][nt], af[ks][mt], acc[nt][mt], 0, 0, 0);
    };
    // fully unrolled (ring slots are compile-time: slot = k-step % 4)
#pragma unroll
    for (int kt = 0; kt < nk - RG_D; ++kt) step(kt, kt % RING, kt + RG_D, (kt + RG_D) % RING, [] { rg_wait_vmcnt<4 * RG_D>(); });
    if constexpr (EPI == 0) load_resid();                             // behind the last weight prefetch; the drain steps cover its latency
#pragma unroll
    for (int kt = nk - RG_D; kt < nk; ++kt) step(kt, kt % RING, -1, 0, [] {});   // drain: straight-line code, the compiler's own counts are exact
    RGN_RT(4)

    // ---- epilogue. Accumulator layout: lane (l31, kh) holds, for token m = m0 + 32 mt + l31, the columns
    //      n = nw + 32 nt + 8 i4 + 4 kh + e (register i = 4 i4 + e). Scattered 8-byte global accesses from this layout cost
    //      ~45 k cycles per tile (measured), so everything leaves through LDS and fully coalesced 16-byte accesses.
    const float4* bias4 = reinterpret_cast<const float4*>(g.bias);
    auto col4 = [&](int nt, int i4) { return nw + 32 * nt + 8 * i4 + 4 * kh; };   // first of 4 consecutive columns
    __builtin_amdgcn_s_barrier();                                     // every wave is done reading the activation image
    if constexpr (EPI == 1) {
        // (1) act(acc + bias) as bf16 into an LDS image of the output plane tile: [16 column blocks][64 rows][64 B], the
        //     16-byte chunk c of row r at chunk position c ^ ((r >> 2) & 3) (conflict-free 8-byte writes and 16-byte reads)
        char* img_hi = smem;
        char* img_lo = smem + 16 * 4096;
        const int nloc = wave * 64;                                   // first tile-local column of this wave
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int n = col4(nt, i4);
                const float4 b = (g.bias && n < g.N) ? bias4[n >> 2] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int r = 32 * mt + l31;
                    float v[4] = {acc[nt][mt][4 * i4] + b.x, acc[nt][mt][4 * i4 + 1] + b.y, acc[nt][mt][4 * i4 + 2] + b.z,
                                  acc[nt][mt][4 * i4 + 3] + b.w};
                    if (g.act == 1) {
                        const f32x2 g0 = rg_gelu2(f32x2{v[0], v[1]}), g1 = rg_gelu2(f32x2{v[2], v[3]});
                        v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
                    } else if (g.act == 2 && blockIdx.y == 0) {       // packed in_proj: column chunk 0 = q, pre-scaled by 1/sqrt(dh)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= g.qscale;
                    }
                    bf16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h[e] = (__bf16)v[e];
                        l[e] = (__bf16)(v[e] - (float)h[e]);
                    }
                    const int blk = (nloc >> 5) + nt;
                    const int off = blk * 4096 + r * 64 + ((i4 ^ ((r >> 2) & 3)) << 4) + 8 * kh;
                    *reinterpret_cast<bf16x4*>(img_hi + off) = h;
                    if (g.Clo) *reinterpret_cast<bf16x4*>(img_lo + off) = l;
                }
            }
        RGN_RT(5)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RGN_RT(6)
        // (2) copy out: a column block of the tile is ONE contiguous 4 KiB run of the K32-blocked plane (64 rows x 64 B);
        //     every wave-instruction moves 1 KiB (16 rows)
        const int nblk_tile = (((g.N - nblk0) < RG_BN ? (g.N - nblk0) : RG_BN) + 31) >> 5;   // column blocks that exist
        const int r16 = lane >> 2, c = lane & 3;
        if (g.act == 2) {
            // packed in_proj of a long sequence (N = 3 d, d = 512 = one column chunk each for q, k, v): the "attention-ready"
            // layout [sample * H + head][Tqp][dh] k_attn_x3 reads; a row's 32-column block is a 64-byte run of its head's row
            __bf16* dst = blockIdx.y == 0 ? g.Qhi : (blockIdx.y == 1 ? g.Khi : g.Vhi);
            const int bpk = g.dh >> 5;                                 // 32-column blocks per head
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = wave * 8 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
                const int m = m0 + r;
                if (m < g.M) {
                    const int b = m / g.Tq, t = m - b * g.Tq, head = blk / bpk, cc = (blk - head * bpk) * 32 + c * 8;
                    const int off = blk * 4096 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                    const size_t o = (((size_t)b * g.H + head) * g.Tqp + t) * g.dh + cc;
                    *reinterpret_cast<bf16x8*>(dst + o) = *reinterpret_cast<const bf16x8*>(img_hi + off);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = wave * 8 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
                const int m = m0 + r;
                if (blk < nblk_tile && m < g.M) {
                    const int off = blk * 4096 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                    const size_t o = ((size_t)((nblk0 >> 5) + blk) * g.c_rows + m) * 32 + c * 8;
                    *reinterpret_cast<bf16x8*>(g.Chi + o) = *reinterpret_cast<const bf16x8*>(img_hi + off);
                    if (g.Clo) *reinterpret_cast<bf16x8*>(g.Clo + o) = *reinterpret_cast<const bf16x8*>(img_lo + off);
                }
            }
        }
        RGN_RT(7)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RGN_RT(8)
    } else {
        // (1) acc + bias -> fp32 row buffer [64][RLD] (aliases the dead activation image): lane = row, 4 consecutive
        //     floats per write, row stride 516 floats -> conflict-free ds_write_b128
        float* rowbuf = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int n = col4(nt, i4);
                const float4 b = bias4[n >> 2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f32x4 v = {acc[nt][mt][4 * i4] + b.x, acc[nt][mt][4 * i4 + 1] + b.y, acc[nt][mt][4 * i4 + 2] + b.z,
                                     acc[nt][mt][4 * i4 + 3] + b.w};
                    *reinterpret_cast<f32x4*>(rowbuf + (32 * mt + l31) * RLD + n) = v;
                }
            }
        RGN_RT(5)
        // per-column vectors of this lane's 8 columns (coalesced 32-byte runs)
        float ga[8], ba[8], gb[8], bb[8], sv[8];
        auto ld8 = [&](const float* p, float* v) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p + c0), b = *reinterpret_cast<const f32x4*>(p + c0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = a[e];
                v[4 + e] = b[e];
            }
        };
        ld8(g.ga, ga);
        ld8(g.ba, ba);
        if (g.gb) {
            ld8(g.gb, gb);
            ld8(g.bb, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) sv[e] = 0.f;
            if (g.stepvec) ld8(g.stepvec + (size_t)(*g.d_step) * g.ldstep, sv);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RGN_RT(6)
        // (2) one wave per row, 8 rows per wave: + residual, LayerNorm a (two-pass, like k_layernorm), optional
        //     + per-step / per-sample vectors and LayerNorm b, result -> residual-stream planes (in place)
        const float invn = 1.0f / (float)RG_BN;
        auto wsum = [](float v) { return rg_wave_sum(v); };
        // the wave's 8 consecutive rows belong to at most two samples: per-step + per-sample vector of each, combined once
        float spa[8], spb[8];
        int samp_a = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) spa[e] = spb[e] = sv[e];
        if (g.gb && g.pervec) {
            const int ma = m0 + 8 * wave, mb = ma + 7;
            float pa[8], pb[8];
            samp_a = (ma < g.M ? ma : g.M - 1) / g.Tq;
            ld8(g.pervec + (size_t)samp_a * g.ldper, pa);
            ld8(g.pervec + (size_t)((mb < g.M ? mb : g.M - 1) / g.Tq) * g.ldper, pb);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                spa[e] += pa[e];
                spb[e] += pb[e];
            }
        }
        // y = LN(v): two-pass statistics (like k_layernorm), hardware rsqrt (1 ulp; this phase rounds its result to bf16)
        auto norm = [&](float (&v)[8], const float (&gam)[8], const float (&bet)[8]) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e];
            const float mean = wsum(s) * invn;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] -= mean;
                q = fmaf(v[e], v[e], q);
            }
            const float rstd = __builtin_amdgcn_rsqf(wsum(q) * invn + 1e-5f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], rstd * gam[e], bet[e]);
        };
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * wave + j, m = m0 + r;
            float v[8];
            {
                const f32x4 a = *reinterpret_cast<const f32x4*>(rowbuf + r * RLD + c0), b = *reinterpret_cast<const f32x4*>(rowbuf + r * RLD + c0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = a[e];
                    v[4 + e] = b[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rh[j][e];
            if (g.Rlo) {                                              // wave-uniform
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rl[j][e];
            }
            norm(v, ga, ba);
            if (g.gb) {
                if ((m < g.M ? m : g.M - 1) / g.Tq == samp_a) {       // wave-uniform
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += spa[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += spb[e];
                }
                norm(v, gb, bb);
            }
            if (m < g.M) {
                const size_t o = ((size_t)(c0 >> 5) * g.o_rows + m) * 32 + (c0 & 31);
                bf16x8 h;
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x8*>(g.Ohi + o) = h;
                if (g.Olo) {
                    bf16x8 l;
#pragma unroll
                    for (int e = 0; e < 8; ++e) l[e] = (__bf16)(v[e] - (float)h[e]);
                    *reinterpret_cast<bf16x8*>(g.Olo + o) = l;
                }
            }
        }
        RGN_RT(7)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RGN_RT(8)
    }
}

#ifdef RGN_RG_PROF
void rg_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rg_prof), sizeof(long long) * 8 * 16); }
#endif

bool rowgemm_supported(int N, int Kp, bool ln) {
    const int nk = Kp / 32;
    if (Kp % 32 || (nk != 12 && nk != 16 && nk != 32)) return false;   // instantiated depths: K = 384, 512, 1024
    if (N % 32) return false;                                          // whole column blocks (fragment-ordered weights)
    return ln ? N == RG_BN : N >= 32;
}
static int rg_lds(int nk, bool ln, bool lo) {   // activation image, re-used by the epilogue: LN fp32 row buffer [64][516] / ACT plane images hi (+ lo)
    const int img = nk * 4096, epi = ln ? 64 * (RG_BN + 4) * 4 : (lo ? 2 : 1) * 16 * 4096;
    return img > epi ? img : epi;
}
template <int EPI, int NK>
static hipError_t rg_go(const RowGemmArgs& g, hipStream_t s, bool configure_only) {
    if (configure_only)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowgemm<EPI, NK>), hipFuncAttributeMaxDynamicSharedMemorySize, rg_lds(NK, EPI == 0, true));
    const dim3 grid((g.M + RG_BM - 1) / RG_BM, (g.N + RG_BN - 1) / RG_BN), block(RG_NT);
    hipLaunchKernelGGL((k_rowgemm<EPI, NK>), grid, block, rg_lds(NK, EPI == 0, g.Clo != nullptr), s, g);
    return hipGetLastError();
}
template <int EPI>
static hipError_t rg_nk(const RowGemmArgs& g, int nk, hipStream_t s, bool cfg) {
    switch (nk) {
        case 12: return rg_go<EPI, 12>(g, s, cfg);
        case 16: return rg_go<EPI, 16>(g, s, cfg);
        case 32: return rg_go<EPI, 32>(g, s, cfg);
    }
    return hipErrorInvalidValue;
}
hipError_t configure_rowgemm() {
    RowGemmArgs g{};
    for (int nk : {12, 16, 32}) {
        hipError_t e = rg_nk<0>(g, nk, nullptr, true);
        if (e != hipSuccess) return e;
        e = rg_nk<1>(g, nk, nullptr, true);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_rowgemm(const RowGemmArgs& g, bool ln, hipStream_t s) {
    if (!rowgemm_supported(g.N, g.Kp, ln)) return hipErrorInvalidValue;
    if (!ln && g.act == 2) {                                         // attention-ready q / k / v scatter of the packed in_proj
        if (g.N != 3 * RG_BN || !g.Qhi || !g.Khi || !g.Vhi || g.Clo || g.dh % 32 || g.H * g.dh != RG_BN || g.Tq <= 0) return hipErrorInvalidValue;
    } else if (!ln && (g.add || g.C || !g.Chi || g.N % 32)) {
        return hipErrorInvalidValue;                                 // EPI_ACT writes planes only
    }
    return ln ? rg_nk<0>(g, g.Kp / 32, s, false) : rg_nk<1>(g, g.Kp / 32, s, false);
}

}  // namespace rgn
