// The whole decoder stack of one denoiser evaluation as ONE kernel: a workgroup owns ONE sample (Tq <= 64 tokens, padded to a
// 64-row tile) and carries its residual stream through all L layers without leaving the CU
//
//   for every layer:  a   = SelfAttention(h)            in_proj (q | k | v), causal softmax, p . v        (model/cmdm.py:227 ->
//                     h'  = LN2( LN1( a . Wo^T + bo + h ) + call_time[step] + call_cond[sample] )          TransformerDecoderLayer,
//                     h   = LN3( gelu( h' . W1^T + b1 ) . W2^T + b2 + h' )                                  built at :75-81)
//
// Causal attention never looks across samples and everything else is row-local, so a sample's tokens close on themselves: no
// workgroup ever needs another one's data. What that buys over the kernel-per-stage form (k_qkv_attn_rs + k_mlp per layer: 16
// launches per evaluation): no kernel boundaries, no chip-wide tile burst at the head of every kernel (the XCD L2s are
// invalidated at boundaries, so every hand-over went through the memory-side cache: ~13 % of the step), no attention-output /
// residual planes in HBM at all - only the weights stream (L2-resident per layer, every CU of an XCD reads the same bytes at about
// the same time) and B workgroups of 8 waves for B samples (256 samples = 256 CUs).
//
// 8 waves. LDS (160 KiB):
//   X (64 KiB)   the residual stream as a bf16 MFMA-operand image [16 k-blocks][64 rows][64 B] (16-byte chunks swizzled by the
//                row): h -> h' -> next h, updated IN PLACE by the wave that owns the columns; A operand of in_proj and linear1
//   Y (64 KiB)   attention output image (A operand of out_proj) -> GELU(hidden half) images (A operand of linear2)
//   Y + 32 KiB   while the attention runs: the 96 KiB fp32 exchange of the S^T partials (two heads at a time x 4 dh tiles)
//   last 32 KiB  layer tail: LayerNorm statistics exchange (8 KiB) + wave-private per-column vectors (20 KiB)
// Attention: heads two at a time, waves 0-3 / 4-7 one head each, wave = 32 dh columns of q, k, v for all 64 tokens (96 x 64
// accumulators); q, k, v, the scores and p never leave the register file (rgn_qkv_attn.hip: the C/D layouts of q^T, k^T, v and
// of the softmaxed S^T agree as MFMA operands by construction); the activation operand comes from the RESIDENT image X, so the
// k-loop has no barrier and no staging at all. Layer tail: the structure of rgn_mlp2.hip (MT = 2) on the resident images.
// Weights: fragment-ordered planes streamed into register rings through buffer loads (scalar resource, compile-time offsets).
//
// Three instantiations: k_layers<false> - the stack of ONE evaluation (planes in, planes out); k_layers<true> - whole runs of sampler steps: after the
// stack the step boundary of rgn_step.hip in per-sample form (output projection, sampler update of x in place - diffusion/gaussian_diffusion.py:508-560,
// 744-794 - and the next evaluation's input embedding straight into X), looped `steps` times, the device-side loop index moved on by the last
// workgroup; k_layers<true, true> - the same under classifier-free guidance (model/cfg_sampler.py:22-31): a workgroup owns a MOTION and runs its
// conditional and its unconditional evaluation back to back, the conditional x0 parked in global scratch meanwhile.
#include "rgn_internal.h"
#include "rgn_philox.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>
#include <utility>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))
#ifndef RGN_LY_ST_AUX
#define RGN_LY_ST_AUX 16   // output stores write-through (sc1)
#endif

constexpr int LY_NTH = 512, LY_KB = 4096;
constexpr int LY_X = 0, LY_Y = 64 * 1024, LY_EXCH = LY_Y, LY_RED = 128 * 1024, LY_REDF = 2 * 8 * 64, LY_VEC = LY_RED + 2 * LY_REDF * 4, LY_VECW = 640,
              LY_LDS = 160 * 1024;
constexpr int LY_PF = LY_VEC + 8 * LY_VECW * 4;   // 4 KiB nobody reads: the landing zone of the step-boundary weight prefetch (8 waves x 256 B)
static_assert(LY_PF + 8 * 256 <= LY_LDS, "LDS map");
// wave-private vector region (floats, 64 columns each)
enum { V_BO = 0, V_G1 = 64, V_G2 = 128, V_B2 = 192, V_SPV = 256, V_BF1 = 320 /* 2 x 64: hidden halves */, V_BF2 = 448, V_G3 = 512, V_B3 = 576 };
constexpr int LY_RDA = 6;   // in_proj weight ring: granules (half k-steps) of 3 fragments (q | k | v): 72 registers
constexpr int LY_RDM = 8;   // layer-tail weight ring: granules of 2 fragments: 64 registers

#ifdef RGN_LY_STAMPS
__device__ long long g_ly_st[1024][16];
#define RGN_LYT(i)                                                                                   \
    {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 1024 && l == RGN_LY_STAMPS) g_ly_st[blockIdx.x][i] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    }
#define RGN_LYS(i)                                                                                   \
    {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 1024 && it == 1) g_ly_st[blockIdx.x][i] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    }
#else
#define RGN_LYT(i)
#define RGN_LYS(i)
#endif

// GELU (erf form), see rgn_mlp2.hip
__device__ __forceinline__ f32x2 ly_gelu2(f32x2 x) {
    const f32x2 t = {__builtin_amdgcn_fmed3f(x[0], -3.9f, 3.9f), __builtin_amdgcn_fmed3f(x[1], -3.9f, 3.9f)};
    const f32x2 z = t * t;
    f32x2 p = f32x2{3.214928057e-08f, 3.214928057e-08f};
    p = __builtin_elementwise_fma(p, z, f32x2{-2.075321845e-06f, -2.075321845e-06f});
    p = __builtin_elementwise_fma(p, z, f32x2{5.740237248e-05f, 5.740237248e-05f});
    p = __builtin_elementwise_fma(p, z, f32x2{-9.056383278e-04f, -9.056383278e-04f});
    p = __builtin_elementwise_fma(p, z, f32x2{9.218782187e-03f, 9.218782187e-03f});
    p = __builtin_elementwise_fma(p, z, f32x2{-6.556465477e-02f, -6.556465477e-02f});
    p = __builtin_elementwise_fma(p, z, f32x2{3.986084461e-01f, 3.986084461e-01f});
    return x * __builtin_elementwise_fma(t, p, f32x2{0.5f, 0.5f});
}

// step boundary (STEPS build): fp32 x0 tile [64][356] over the dead images, x' image (A operand of the input embedding) behind it
constexpr int LY_XLD = 356, LY_TILE = 0, LY_XIMG = 92160, LY_NKX = 11;
static_assert(64 * LY_XLD * 4 <= LY_XIMG && LY_XIMG + LY_NKX * 4096 <= LY_LDS, "step boundary LDS map");
template <int... Is, class F>
__device__ __forceinline__ void ly_static_for_seq(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void ly_static_for(F&& f) { ly_static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

template <bool STEPS, bool GUIDED = false, bool F16 = false>
__global__ __launch_bounds__(LY_NTH, 2) void k_layers(LayersArgs g) {
    static_assert(STEPS || !GUIDED, "guidance inside the launch needs the step boundary");
    using OP = OpFmt<F16>;
    using op_t = typename OP::t;
    using op8 = typename OP::v8;
    using op4 = typename OP::v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, hg = wave >> 2;                     // attention roles: dh tile, head of the pair
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = xcd_affine(blockIdx.x, gridDim.x), Tq = g.Tq;
    const size_t row0 = (size_t)b * Tq;
#ifdef RGN_LY_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(RGN_LY_PRIO);      // the second-dispatched half loses every age arbitration on its SIMD otherwise (MI355X_MICROARCH.md, two waves per SIMD, item 4)
#endif
    float* vec = reinterpret_cast<float*>(smem + LY_VEC) + wave * LY_VECW;   // this wave's private region
    const int lane16 = lane * 16;
    const int swz = (l31 >> 2) & 3;
    // ---- the sample's residual rows -> X by DMA: 16 k-blocks x 4 pieces of 1 KiB (16 rows x 64 B); padding rows replicate the
    //      last token (row-local everywhere, masked as keys)
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave + 8 * j, kb = p >> 2, r = (p & 3) * 16 + r16;
            const int rr = r < Tq ? r : Tq - 1;
            const size_t src = ((size_t)kb * g.rows + row0 + rr) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.h + src), (RGN_AS3 void*)(smem + LY_X + p * 1024), 16, 0, 0);
        }
    }
    const int first_step = (g.stepvec || STEPS) ? *g.d_step : 0;
    // B-operand fragment of token l31 (+ 32 ta: an immediate offset of 2 KiB) inside a k-block image, per 16-wide k-half; reads of
    // the second image want their own base registers (16-bit ds_read offsets)
    int a_off[2], a_offy[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_off[ks] = l31 * 64 + (((2 * ks + kh) ^ swz) << 4);
        a_offy[ks] = a_off[ks] + LY_Y;
    }
    // element (token 32 mt + l31, column 64 wave + 32 nt + 8 i4 + 4 kh + e) <-> register acc[nt][mt][4 i4 + e] of the layer tail;
    // its 8-byte run inside an image
    auto col4 = [&](int nt, int i4) { return 32 * nt + 8 * i4 + 4 * kh; };
    int img_base = (2 * wave) * LY_KB + l31 * 64 + 8 * kh;
    asm volatile("" : "+v"(img_base));
    auto img_off = [&](int nt, int i4, int mt) { return img_base + nt * LY_KB + mt * 2048 + ((i4 ^ swz) << 4); };
    int red_base = LY_RED + 4 * l31;
    asm volatile("" : "+v"(red_base));
    const float invn = 1.0f / 512.f;
    const float qs2 = g.qscale * 1.44269504088896340736f;        // 1 / sqrt(dh) in log2 units, applied INSIDE the softmax's exponent: exp2(qs2 s - qs2 max), one FMA where the subtraction was

    struct Pass { __amdgpu_buffer_rsrc_t rs; int kstride, hs0; };
    op8 wf[LY_RDM][2];
    auto load_g = [&](const Pass& ps, int hs_rel, int slot) {
        const int hs = ps.hs0 + hs_rel;
        int soff;
        asm volatile("s_mov_b32 %0, %1" : "=s"(soff) : "i"((hs >> 1) * ps.kstride + (hs & 1) * 1024));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            wf[slot][nt] = __builtin_bit_cast(op8, __builtin_amdgcn_raw_buffer_load_b128(ps.rs, lane16, soff + nt * 2048, 0));
    };
    // one GEMM pass over K = 512 from the image at byte offsets aoff (see rgn_mlp2.hip): the ring never drains between passes
    auto gemm_n = [&](f32x16 (&acc)[2][2], const int (&aoff)[2], const Pass& cur, const Pass& nxt, auto chain, auto extra, auto ngran) {
        constexpr int EX = decltype(extra)::value, AH = LY_RDM - 1, NG = decltype(ngran)::value;   // NG granules = NG / 2 k-blocks
        constexpr bool CH = decltype(chain)::value;
        __builtin_amdgcn_sched_barrier(0);
        op8 af[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const op8*>(smem + aoff[0] + mt * 2048);
#pragma unroll
        for (int hs = 0; hs < NG; ++hs) {
            op8 afn[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                afn[mt] = af[mt];
                if (hs + 1 < NG) afn[mt] = *reinterpret_cast<const op8*>(smem + ((hs + 1) >> 1) * LY_KB + aoff[(hs + 1) & 1] + mt * 2048);
            }
            if (hs + AH < NG) load_g(cur, hs + AH, (hs + AH) % LY_RDM);
            else if (CH) load_g(nxt, hs + AH - NG, (hs + AH) % LY_RDM);
            if (hs + AH < NG || CH) {
                if (hs < AH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AH + EX) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AH) : "memory");
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = OP::mfma(wf[hs % LY_RDM][nt], af[mt], acc[nt][mt]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = afn[mt];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto init_bias = [&](f32x16 (&acc)[2][2], const float* bias) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + col4(nt, i4));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] = bb[e];
            }
    };
    // LayerNorm over the 512 columns of every token, in place: one exchange of (sum, sum of squares) - rgn_mlp2.hip
    int red_base2 = red_base + 128 * kh;                         // post-barrier reads: lane (l31, kh) reduces token 32 kh + l31
    asm volatile("" : "+v"(red_base2));
    auto layernorm = [&](f32x16 (&acc)[2][2], const float* gam, auto slot, auto shift /* (nt, i4) -> f32x4 */) {
        const char* buf = smem + red_base + decltype(slot)::value * LY_REDF * 4;
        const char* buf2 = smem + red_base2 + decltype(slot)::value * LY_REDF * 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x2 s2 = f32x2{0.f, 0.f}, q2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 v = f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]};
                    s2 += v;
                    q2 = __builtin_elementwise_fma(v, v, q2);
                }
            float s = s2[0] + s2[1], q = q2[0] + q2[1];
            half_swap(s, q);                                        // s = [s.lo | q.lo], q = [s.hi | q.hi]
            *reinterpret_cast<float*>(const_cast<char*>(buf) + (kh * 512 + wave * 64 + 32 * mt) * 4) = s + q;   // kh = 0: the sum, kh = 1: the sum of squares
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // the halves share the work: lane (l31, kh) reduces the eight partials of token 32 kh + l31, then the two results change hands
        f32x2 rs[2], nm[2];
        {
            float p[2][8];
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) p[st][ww] = *reinterpret_cast<const float*>(buf2 + (st * 512 + ww * 64) * 4);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int dd = 1; dd < 8; dd *= 2)
#pragma unroll
                    for (int ww = 0; ww < 8; ww += 2 * dd) p[st][ww] += p[st][ww + dd];
            const float mean = p[0][0] * invn;
            const float var = __builtin_fmaxf(p[1][0] * invn - mean * mean, 0.f);
            float r0 = __builtin_amdgcn_rsqf(var + 1e-5f), n0 = -mean * r0;
            float r1 = r0, n1 = n0;
            asm volatile("" : "+v"(r1), "+v"(n1));               // (copies in registers of their own)
            half_swap(r0, r1);                                      // r0 = token l31's (tile 0), r1 = token 32 + l31's (tile 1), in every lane
            half_swap(n0, n1);
            rs[0] = f32x2{r0, r0}; rs[1] = f32x2{r1, r1};
            nm[0] = f32x2{n0, n0}; nm[1] = f32x2{n1, n1};
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x4 ga[4], sh[4];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                ga[i4] = *reinterpret_cast<const f32x4*>(gam + col4(nt, i4));
                sh[i4] = shift(nt, i4);
            }
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 t = __builtin_elementwise_fma(f32x2{acc[nt][mt][4 * i4 + e], acc[nt][mt][4 * i4 + e + 1]}, rs[mt], nm[mt]);   // (v - mean) rstd
                        const f32x2 o = __builtin_elementwise_fma(t, f32x2{ga[i4][e], ga[i4][e + 1]}, f32x2{sh[i4][e], sh[i4][e + 1]});
                        acc[nt][mt][4 * i4 + e] = o[0];
                        acc[nt][mt][4 * i4 + e + 1] = o[1];
                    }
        }
    };
    auto store_img = [&](const f32x16 (&acc)[2][2], int img) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    op4 hh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hh[e] = (op_t)acc[nt][mt][4 * i4 + e];
                    *reinterpret_cast<op4*>(smem + img + img_off(nt, i4, mt)) = hh;
                }
    };
    // acc += bf16 residual from the image X (this wave's own columns)
    auto add_resid = [&](f32x16 (&acc)[2][2]) {
        op4 rr[2][2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) rr[nt][mt][i4] = *reinterpret_cast<const op4*>(smem + LY_X + img_off(nt, i4, mt));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] += (float)rr[nt][mt][i4][e];
    };

    auto gemm32 = [&](f32x16 (&acc)[2][2], const int (&aoff)[2], const Pass& cur, const Pass& nxt, auto chain, auto extra) {
        gemm_n(acc, aoff, cur, nxt, chain, extra, std::integral_constant<int, 32>{});
    };

    // ---- in_proj bias of the NEXT round, requested a phase ahead (before the previous layer's norm3 / during the previous round's softmax), so the
    //      round's first MFMA does not wait an L2 round trip for its accumulators' start value. q: register i <-> dh 8 (i >> 2) + 4 kh + (i & 3);
    //      v: lane = dh. The k bias is not applied at all: q . (k + b_k) = q . k + q . b_k adds the same number to every score of a query's row,
    //      which the softmax removes (exact in real arithmetic; the reference's output does not depend on it either)
    f32x4 qb[4];
    float vb1;
    auto load_qbias = [&](const float* bqkv, int r) {
        const float* bq = bqkv + (2 * r + hg) * 128 + wn * 32;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) qb[i4] = *reinterpret_cast<const f32x4*>(bq + 8 * i4 + 4 * kh);
        vb1 = bq[1024 + l31];
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_qbias(g.lw[0].bqkv, 0);

    const int nit = STEPS ? g.steps : 1;
    // the output projection of the step boundary: acc = h . Wout^T from the image X (N = F <= 352: waves 0-5; a column block past the plane
    // reads in-bounds garbage or zeros and is never stored)
    auto out_proj = [&](f32x16 (&acc)[2][2], int wv) {
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a2][b2][i] = 0.f;
        if (wv < 6) {
            const int cb0 = 2 * wv;
            const Pass p_out{__builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Wout) + (size_t)cb0 * 1024, 0, (16 * g.nb_out - cb0) * 2048, 0x00020000), LY_NKX * 2048, 0};
#pragma unroll
            for (int s2 = 0; s2 < LY_RDM - 1; ++s2) load_g(p_out, s2, s2);
            gemm_n(acc, a_off, p_out, p_out, std::false_type{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 32>{});
        }
    };
    for (int it = 0; it < nit; ++it) {
    const int step = first_step - it;
    for (int pass = 0; pass < (GUIDED ? 2 : 1); ++pass) {
    if (GUIDED && pass == 1) {
        // ---- guidance: the conditional evaluation is done - its x0 goes to the parking buffer (accumulator layout: every lane reads back
        //      what it wrote), the unconditional evaluation's input (written to the planes by the previous step boundary / the up-front
        //      embedding; sc1: past this CU's L1, which may still hold last step's lines) takes its place in X
        f32x16 accp[2][2];
        // (ids made opaque per pass: the park / reload addresses below are invariant across the step loop - hoisted out of it they stay live across the
        //  whole layer loop and are spilled)
        int wave_p = wave, lane_p = lane, b_p = b;
        asm volatile("" : "+s"(wave_p), "+s"(b_p));
        asm volatile("" : "+v"(lane_p));
        out_proj(accp, wave_p);
        if (wave_p < 6) {
            float* pk = g.park + ((size_t)b_p * 6 + wave_p) * 4096 + lane_p * 4;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4)
                        *reinterpret_cast<f32x4*>(pk + ((nt * 2 + mt) * 4 + i4) * 256) = f32x4{accp[nt][mt][4 * i4], accp[nt][mt][4 * i4 + 1], accp[nt][mt][4 * i4 + 2], accp[nt][mt][4 * i4 + 3]};
        }
        __builtin_amdgcn_s_barrier();                                     // every wave is done reading the image
        {
            const int r16 = lane_p >> 2, c = lane_p & 3;
            const size_t row0_p = (size_t)b_p * Tq;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = wave_p + 8 * j, kb = p >> 2, r = (p & 3) * 16 + r16;
                const int rr = r < Tq ? r : Tq - 1;
                const size_t src = ((size_t)kb * g.rows + g.half + row0_p + rr) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
                __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.h + src), (RGN_AS3 void*)(smem + LY_X + p * 1024), 16, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    for (int l = 0; l < g.L; ++l) {
        const LayerWts& w = g.lw[l];
        RGN_LYT(0)
        // ========================= self-attention: in_proj + causal softmax + p . v, two heads at a time ===================
        op4 attk[2][2][4];                                    // [round][query tile][run of 4 dh]: this wave's O^T tiles as bf16
        float bo_r = 0.f;
        {
            struct PassA { __amdgpu_buffer_rsrc_t rs; };
            auto qrs = [&](int r) {
                const int blk = (2 * r + hg) * 4 + wn;           // first column block (32 columns) of this wave's q slice
                return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(w.Wqkv) + (size_t)blk * 1024, 0, 3 * 512 * 512 * 2 - blk * 2048, 0x00020000);
            };
            const PassA pa[2] = {{qrs(0)}, {qrs(1)}};
            op8 wq[LY_RDA][3];
            // granule hs = 2 kt + ks of the plane [16 k-blocks][48 column blocks][2][64][8]: the q, k, v fragments sit 16 column blocks apart
            auto load_ga = [&](const PassA& ps, int hs, int slot) {
                // (the granule offset is materialised by a volatile s_mov right here: as plain literals the ~300 offsets of a layer are hoisted
                // out of the layer loop as loop invariants and live in spilled SGPRs)
                int soff;
                asm volatile("s_mov_b32 %0, %1" : "=s"(soff) : "i"((hs >> 1) * (48 * 2048) + (hs & 1) * 1024));
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    wq[slot][t] = __builtin_bit_cast(op8, __builtin_amdgcn_raw_buffer_load_b128(ps.rs, lane16, soff + t * (16 * 2048), 0));
            };
            constexpr int AH = LY_RDA - 1;
#pragma unroll
            for (int s = 0; s < AH; ++s) load_ga(pa[0], s, s);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int head = 2 * r + hg;
                // accumulators start from the in_proj bias (requested a phase ago: load_qbias): q^T, k^T (lane = token, registers = dh), v (lane = dh)
                f32x16 acc[2][3];
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[ta][0][4 * i4 + e] = qb[i4][e];
                            acc[ta][1][4 * i4 + e] = 0.f;
                            acc[ta][2][4 * i4 + e] = vb1;
                        }
                // ---- in_proj: [64 tokens] x [q | k | v of this wave's 32 dh columns] over K = 512, from the resident image X
                __builtin_amdgcn_sched_barrier(0);
                {
                    op8 af[2];
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta) af[ta] = *reinterpret_cast<const op8*>(smem + a_off[0] + ta * 2048);
#pragma unroll
                    for (int hs = 0; hs < 32; ++hs) {
                        op8 afn[2];
#pragma unroll
                        for (int ta = 0; ta < 2; ++ta) {
                            afn[ta] = af[ta];
                            if (hs + 1 < 32) afn[ta] = *reinterpret_cast<const op8*>(smem + ((hs + 1) >> 1) * LY_KB + a_off[(hs + 1) & 1] + ta * 2048);
                        }
                        if (hs + AH < 32) load_ga(pa[r], hs + AH, (32 * r + hs + AH) % LY_RDA);
                        else if (r == 0) load_ga(pa[1], hs + AH - 32, (32 * r + hs + AH) % LY_RDA);
                        if (hs + AH < 32 || r == 0) {
                            if (hs < AH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * AH + 5) : "memory");   // (+ the 5 bias loads: behind round 1's chained granules)
                            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * AH) : "memory");
                        }
                        const int slot = (32 * r + hs) % LY_RDA;
#pragma unroll
                        for (int t = 0; t < 3; ++t)
#pragma unroll
                            for (int ta = 0; ta < 2; ++ta) {
                                if (t < 2) acc[ta][t] = OP::mfma(wq[slot][t], af[ta], acc[ta][t]);
                                else acc[ta][t] = OP::mfma(af[ta], wq[slot][t], acc[ta][t]);
                            }
#pragma unroll
                        for (int ta = 0; ta < 2; ++ta) af[ta] = afn[ta];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                RGN_LYT(1 + 3 * r)
                // ---- attention straight from the accumulators (rgn_qkv_attn.hip qa_attention, plain-bf16 form)
                op8 qh[2][2], kf[2][2], vh[2][2];             // [token tile][16-slice of the register index]
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int i = 8 * sl + j;
                            qh[ta][sl][j] = (op_t)acc[ta][0][i];
                            kf[ta][sl][j] = (op_t)acc[ta][1][i];
                            vh[ta][sl][j] = (op_t)acc[ta][2][i];
                        }
                if (r == 0) load_qbias(w.bqkv, 1);                // (the accumulators are dead: round 1's start value lands under this round's softmax)
                else bo_r = w.bo[64 * wave + lane];              // out_proj's bias = its accumulators' start value: lands under round 1's softmax
                // causal tiles of S^T: 0 = (keys 0-31, queries 0-31), 1 = (keys 0-31, queries 32-63), 2 = (keys 32-63, queries 32-63)
                f32x16 st[3];
#pragma unroll
                for (int tl = 0; tl < 3; ++tl) {
                    const int kj = tl >> 1, qtile = tl ? 1 : 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[tl][i] = 0.f;
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl) st[tl] = OP::mfma(kf[kj][sl], qh[qtile][sl], st[tl]);
                }
                // sum the partials of the four dh tiles: [head of the pair][tile][wave wn][i / 4][lane] float4
                f32x4* sred = reinterpret_cast<f32x4*>(smem + LY_EXCH);
#pragma unroll
                for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const f32x4 v = {st[tl][4 * i4], st[tl][4 * i4 + 1], st[tl][4 * i4 + 2], st[tl][4 * i4 + 3]};
                        sred[(((hg * 3 + tl) * 4 + wn) * 4 + i4) * 64 + lane] = v;
                    }
                if (r == 0) { RGN_LYT(12) }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (r == 0) { RGN_LYT(13) }
                // ---- softmax, each score ONCE: wave wn takes the queries {8 wn .. 8 wn + 7} of both query tiles, lane = (query qi, key group
                //      kg of 16 keys), so a row lives in 4 adjacent lanes x 16 registers. Every exchange entry [..][lane'] is read by exactly
                //      one wave - the one that owns query lane' & 31 - which is what lets the normalised probabilities go back INTO the
                //      exchange (slots [tile][wave 0][run = 16-key slice][lane'], 8 bf16 = one PV operand each) without another buffer.
                {
                    const int kg = lane & 3, qi = lane >> 2, qt = qi >> 3, ql = 8 * wn + (qi & 7), Q = 32 * qt + ql;
                    const int kj = kg >> 1, sl = kg & 1;
                    const bool live = !(qt == 0 && kj == 1);               // keys 32-63 never reach queries 0-31
                    const int tl = live ? qt + kj : 0;
                    const char* ex = smem + LY_EXCH + ((hg * 3 + tl) * 4 * 4 + 2 * sl) * 1024 + ql * 16;   // [hg][tl][w][i4][lane'] x 16 B
                    f32x4 sv[4];                                            // keys 16 sl + {0-3 | 4-7 | 8-11 | 12-15} of key tile kj
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            f32x4 v = *reinterpret_cast<const f32x4*>(ex + a2 * 1024 + h2 * 512);
#pragma unroll
                            for (int ww = 1; ww < 4; ++ww) {
                                const f32x4 u = *reinterpret_cast<const f32x4*>(ex + ww * 4096 + a2 * 1024 + h2 * 512);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += u[e];
                            }
                            sv[2 * a2 + h2] = v;
                        }
                    // visible iff key <= min(Q, Tq - 1); key = 32 kj + 16 sl + x. One per-lane limit, opaque: see the note at the mask of
                    // rgn_qkv_attn.hip (hoisted compares -> spilled lane masks)
                    int xlim = live ? (Q < Tq - 1 ? Q : Tq - 1) - 32 * kj - 16 * sl : -1;
                    asm volatile("" : "+v"(xlim));
                    float mx = -INFINITY;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            sv[c4][e] = (4 * c4 + e <= xlim) ? sv[c4][e] : -INFINITY;
                            mx = fmaxf(mx, sv[c4][e]);
                        }
                    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, mx), 0xB1, 0xf, 0xf, true)));   // quad_perm [1,0,3,2]
                    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, mx), 0x4E, 0xf, 0xf, true)));   // quad_perm [2,3,0,1]
                    float sum = 0.f;
                    const float nmx = -mx * qs2;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            sv[c4][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[c4][e], qs2, nmx));
                            sum += sv[c4][e];
                        }
                    sum += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sum), 0xB1, 0xf, 0xf, true));
                    sum += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sum), 0x4E, 0xf, 0xf, true));
                    const float inv = 1.0f / sum;
                    // PV operand of (tile, slice sl), lane half h2: keys 16 sl + 4 h2 + {0-3} and 16 sl + 8 + 4 h2 + {0-3}
                    if (live) {
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            op8 pp;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                pp[e] = (op_t)(sv[h2][e] * inv);
                                pp[4 + e] = (op_t)(sv[2 + h2][e] * inv);
                            }
                            *reinterpret_cast<op8*>(smem + LY_EXCH + (((hg * 3 + tl) * 4 + 0) * 4 + sl) * 1024 + (h2 * 32 + ql) * 16) = pp;
                        }
                    }
                }
                if (r == 0) { RGN_LYT(14) }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (r == 0) { RGN_LYT(15) }
                // O^T[dh tile wn, queries] = V (A operand, registers = keys) x P^T (B operand: this lane's slot of the probabilities)
#pragma unroll
                for (int qtile = 0; qtile < 2; ++qtile) {
                    f32x16 oa;
#pragma unroll
                    for (int i = 0; i < 16; ++i) oa[i] = 0.f;
#pragma unroll
                    for (int tl = qtile; tl <= 2 * qtile; ++tl) {
                        const int kj = tl >> 1;
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl) {
                            const op8 ph = *reinterpret_cast<const op8*>(smem + LY_EXCH + (((hg * 3 + tl) * 4 + 0) * 4 + sl) * 1024 + lane * 16);
                            oa = OP::mfma(vh[kj][sl], ph, oa);
                        }
                    }
                    // O^T tile: lane = query, registers = 16 dh indices -> 4 runs of 4 consecutive dh
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) attk[r][qtile][i4][e] = (op_t)oa[4 * i4 + e];
                }
                RGN_LYT(2 + 3 * r)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                    // every wave has read this round's sums: the exchange may be overwritten
                RGN_LYT(3 + 3 * r)
            }
        }
        // ========================= layer tail on the resident images (rgn_mlp2.hip, MT = 2) ==================================
        auto wrs = [&](const __bf16* W, int cb0, int bytes) {
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W) + (size_t)cb0 * 1024, 0, bytes - cb0 * 2048, 0x00020000);
        };
        const Pass p_wo{wrs(w.Wo, 2 * wave, 512 * 512 * 2), 16 * 2048, 0}, p_w1a{wrs(w.W1, 2 * wave, 1024 * 512 * 2), 32 * 2048, 0},
            p_w1b{wrs(w.W1, 16 + 2 * wave, 1024 * 512 * 2), 32 * 2048, 0}, p_w2a{wrs(w.W2, 2 * wave, 512 * 1024 * 2), 16 * 2048, 0},
            p_w2b{p_w2a.rs, 16 * 2048, 32};
        // ---- out_proj's first fragments and this layer's per-column vectors (this wave's 64 columns: lane = column)
        vec[V_BO + lane] = bo_r;                                 // (wave-private; the exchange it lies in is dead: every wave passed the barrier behind the last round)
#pragma unroll
        for (int s = 0; s < LY_RDM - 1; ++s) load_g(p_wo, s, s);
        int cw = 64 * wave + lane;
        asm volatile("" : "+v"(cw));                             // (per layer: the per-column vector addresses are step-loop invariants - hoisted, they are spilled in the guided form)
        float vv[12];
        {
            const float* src[9] = {w.bo, w.g1, w.g2, w.b2, w.bf1, w.bf1 + 512, w.bf2, w.g3, w.b3};
            vv[1] = src[1][cw]; vv[2] = src[2][cw]; vv[3] = src[3][cw];
            vv[4] = w.b1[cw] + (g.stepvec ? g.stepvec[(size_t)step * g.ldstep + (size_t)l * 512 + cw] : 0.f) +
                    (g.pervec ? g.pervec[((size_t)b + (size_t)pass * g.B) * g.ldper + (size_t)l * 512 + cw] : 0.f);   // norm1's beta + call_time[step] + call_cond[sample]
            vv[5] = src[4][cw]; vv[6] = src[5][cw]; vv[7] = src[6][cw]; vv[8] = src[7][cw]; vv[9] = src[8][cw];
        }
        asm volatile("" ::: "memory");
        // ---- the attention output -> image Y (the exchange is dead: every wave passed the barrier behind the last round)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int qtile = 0; qtile < 2; ++qtile)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
                    *reinterpret_cast<op4*>(smem + LY_Y + ((2 * r + hg) * 4 + wn) * LY_KB + (32 * qtile + l31) * 64 + ((i4 ^ swz) << 4) + 8 * kh) = attk[r][qtile][i4];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x16 acc[2][2];
        init_bias(acc, vec + V_BO);
        RGN_LYT(7)
        gemm32(acc, a_offy, p_wo, p_w1a, std::true_type{}, std::integral_constant<int, 11>{});
        RGN_LYT(8)
        // vectors -> the wave's LDS region (wave-private: program order suffices)
        vec[V_G1 + lane] = vv[1]; vec[V_G2 + lane] = vv[2]; vec[V_B2 + lane] = vv[3]; vec[V_SPV + lane] = vv[4];
        vec[V_BF1 + lane] = vv[5]; vec[V_BF1 + 64 + lane] = vv[6]; vec[V_BF2 + lane] = vv[7]; vec[V_G3 + lane] = vv[8]; vec[V_B3 + lane] = vv[9];
        add_resid(acc);
        layernorm(acc, vec + V_G1, std::integral_constant<int, 0>{}, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(vec + V_SPV + col4(nt, i4)); });
        layernorm(acc, vec + V_G2, std::integral_constant<int, 1>{}, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(vec + V_B2 + col4(nt, i4)); });
        store_img(acc, LY_X);                                    // h' replaces h in place (this wave's columns: it read them above)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RGN_LYT(9)
        f32x16 acc2[2][2];
        init_bias(acc2, vec + V_BF2);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            init_bias(acc, vec + V_BF1 + 64 * c);
            gemm32(acc, a_off, c ? p_w1b : p_w1a, c ? p_w2b : p_w2a, std::true_type{}, std::integral_constant<int, 0>{});   // hidden columns [512 c, 512 c + 512)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        const f32x2 gl = ly_gelu2(f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]});
                        acc[nt][mt][i] = gl[0];
                        acc[nt][mt][i + 1] = gl[1];
                    }
            if (c == 1) __builtin_amdgcn_s_barrier();             // every wave is done reading the first half's image
            store_img(acc, LY_Y);                                 // (c == 0: Y holds the attention output, dead since out_proj)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c == 0) gemm32(acc2, a_offy, p_w2a, p_w1b, std::true_type{}, std::integral_constant<int, 0>{});
            else gemm32(acc2, a_offy, p_w2b, p_w2b, std::false_type{}, std::integral_constant<int, 0>{});
        }
        load_qbias(g.lw[l + 1 < g.L ? l + 1 : 0].bqkv, 0);      // the next layer's (next step's first layer's) round 0: lands under norm3
#ifndef RGN_LY_NO_PREFETCH
        if constexpr (STEPS) {
            // ---- the step boundary's weights (Wout: 16 x nb_out KiB-pairs, Wx: 352 KiB) were evicted from this XCD's L2 by the 33 MB of layer
            //      weights that streamed through it since the last step: the CUs of the XCD touch them once more, each 1 / nsl of the lines,
            //      while the last layer's norm3 runs - the output projection then finds them in L2 instead of waiting a fabric round trip per granule
            if (l == g.L - 1) {
                const int nsl = (int)(gridDim.x >> 3) >= 32 ? 32 : ((int)(gridDim.x >> 3) > 0 ? (int)(gridDim.x >> 3) : 1);
                const int sl = (int)(blockIdx.x >> 3) % nsl;
                const int n_out = 16 * g.nb_out * 16, n_all = n_out + LY_NKX * 16 * 16;           // 128-byte lines
                // (direct-to-LDS loads into 2 KiB of LDS nothing ever reads: a load into a VGPR would land asynchronously, long after the compiler
                //  has given that register to something else. As inline asm: behind the BUILTIN the compiler drains vmcnt(0) before the next
                //  ds_read - norm3's residual reads - and the prefetch would be waited for instead of flying under norm3)
                const unsigned pf_lds = (unsigned)(size_t)((RGN_AS3 char*)(smem + LY_PF)) + (unsigned)wave * 256u;
                for (int ln = sl + nsl * tid; ln < n_all; ln += nsl * LY_NTH) {
                    const char* pp = ln < n_out ? reinterpret_cast<const char*>(g.Wout) + (size_t)ln * 128 : reinterpret_cast<const char*>(g.Wx) + (size_t)(ln - n_out) * 128;
                    // (m0 - the LDS base of a direct-to-LDS load - is the compiler's too: it keeps the base of its own global_load_lds builtins there and does not
                    //  honour a clobber of a reserved register, so the asm saves and restores it; tools/check_m0.py reads the ISA)
                    int m0_keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(pp), "s"(pf_lds) : "memory");
                }
            }
        }
#endif
        RGN_LYT(10)
        add_resid(acc2);
        layernorm(acc2, vec + V_G3, std::integral_constant<int, 0>{}, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(vec + V_B3 + col4(nt, i4)); });
        store_img(acc2, LY_X);                                   // the next layer's input, in place
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RGN_LYT(11)
    }
    }
    if constexpr (STEPS) {
        // ========================= step boundary, per sample (rgn_step.hip k_step<11, false>) =====================================
        //   x0 = h . Wout^T + bout (OutputProcess, cmdm.py:353); x' = sampler(x, x0, eps) in place (gaussian_diffusion.py:508-560,
        //   744-794; same arithmetic and Philox stream as k_update); h' = x' . Wx'^T + c0 (InputProcess / fuse / positional part hoisted
        //   into c0, cmdm.py:201-218) -> the resident image X: the next step's layer 0 reads it without leaving the CU
        // (lane / wave / sample ids made opaque PER ITERATION: everything below is invariant across the step loop, and hoisted out of it
        //  the ~90 addresses of the update phase alone spill hundreds of registers)
        int lane_s = lane, wave_s = wave, b_s = b;
        asm volatile("" : "+v"(lane_s));
        asm volatile("" : "+s"(wave_s), "+s"(b_s));
        const int l31_s = lane_s & 31, kh_s = lane_s >> 5;
        const size_t row0_s = (size_t)b_s * Tq;
        auto col4s = [&](int nt, int i4) { return 32 * nt + 8 * i4 + 4 * kh_s; };
        RGN_LYS(12)
        const SampleParams sp = *g.sp;
        const StepCoef k = g.tab[step];
        const int T = Tq, gb = g.s0 + b_s;                                  // frames = tokens (no emb_trans_dec token on this path); motion index
        const float gscale = GUIDED ? g.scale[gb] : 0.f;
        const __amdgpu_buffer_rsrc_t park_rs = __builtin_amdgcn_make_buffer_rsrc(GUIDED ? g.park + ((size_t)b_s * 6 + (wave_s < 6 ? wave_s : 0)) * 4096 : nullptr, 0, 4096 * 4, 0x00020000);
        f32x16 acc[2][2];
        // ---- A: x0 = h . Wout^T from the image X (guided: of the unconditional evaluation)
        out_proj(acc, wave_s);
        __builtin_amdgcn_s_barrier();                                     // every wave is done reading the image
        // ---- B: x0 + bias -> fp32 tile [64][356]
        float* tile = reinterpret_cast<float*>(smem + LY_TILE);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int n = 64 * wave_s + col4s(nt, i4);
                if (n < 352) {
                    f32x4 bb = {0.f, 0.f, 0.f, 0.f};
                    if (n < g.F) bb = *reinterpret_cast<const f32x4*>(g.bout + n);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        f32x4 v = {acc[nt][mt][4 * i4] + bb[0], acc[nt][mt][4 * i4 + 1] + bb[1], acc[nt][mt][4 * i4 + 2] + bb[2], acc[nt][mt][4 * i4 + 3] + bb[3]};
                        if constexpr (GUIDED) {   // x0 = x0_u + scale_b (x0_c - x0_u), cfg_sampler.py:31, rounded like k_update / k_step
                            // (sc1: served by L2, past this CU's L1 - the same addresses were read a step ago and rewritten since)
                            const f32x4 cc = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(park_rs, (lane_s * 4 + ((nt * 2 + mt) * 4 + i4) * 256) * 4, 0, 16));
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], __fmul_rn(gscale, __fsub_rn(cc[e] + bb[e], v[e])));
                        }
                        *reinterpret_cast<f32x4*>(tile + (32 * mt + l31_s) * LY_XLD + n) = v;
                    }
                }
            }
        const Pass p_wx{__builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Wx) + (size_t)(2 * wave_s) * 1024, 0, (LY_NKX * 16 - 2 * wave_s) * 2048, 0x00020000), 16 * 2048, 0};
#pragma unroll
        for (int s2 = 0; s2 < LY_RDM - 1; ++s2) load_g(p_wx, s2, s2);   // the embedding's first fragments fly under the update phase
        // the condition rows the embedding adds, in the accumulator layout (requested now, used behind the GEMM)
        op4 c0v[2][2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int r = 32 * mt + l31_s, rr = r < T ? r : T - 1;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
                    c0v[nt][mt][i4] = *reinterpret_cast<const op4*>(g.c0 + (row0_s + rr) * 512 + 64 * wave_s + col4s(nt, i4));
        }
        op4 c0u[GUIDED ? 2 : 1][2][4];                                 // guided: the unconditional evaluation's condition rows
        if constexpr (GUIDED) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int r = 32 * mt + l31_s, rr = r < T ? r : T - 1;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4)
                        c0u[nt][mt][i4] = *reinterpret_cast<const op4*>(g.c0 + ((size_t)g.half + row0_s + rr) * 512 + 64 * wave_s + col4s(nt, i4));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RGN_LYS(13)
        // ---- C: sampler update. lane = frame, the waves stride the features; Philox per quad of lanes where the quad is a run of four
        //      frames (see rgn_step.hip), per element otherwise (bit-identical values)
        {
            const int t = lane_s;
            const bool valid = t < T;
            const size_t FT = (size_t)g.F * T;
            const int bn = sp.const_noise ? 0 : gb;
            char* ximg = smem + LY_XIMG;
            const int q = lane_s & 3, tq = t - q;
            const bool run4 = valid && (tq & 3) == 0 && tq + 3 < T;
            const bool quads = !sp.noise && !g.no_quads && __all(run4 || tq >= T);   // (a quad past the last frame does nothing)
            float xpre[LY_NKX][4];
#pragma unroll
            for (int i2 = 0; i2 < LY_NKX; ++i2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = 4 * (wave_s + 8 * i2) + j;
                    xpre[i2][j] = (valid && f < g.F) ? sp.x[(size_t)gb * FT + (size_t)f * T + t] : 0.f;
                }
            // the four features 4 fg + {0 .. 3} of this lane's frame: one 16-byte read of the x0 tile, four x updates, ONE 8-byte store of
            // the bf16 x' run (feature 32 i2 + 4 wave + j sits in k-block i2, 16-byte chunk wave >> 1, bytes 8 (wave & 1) + 2 j of its row)
            const int ximg_lane = lane_s * 64 + ((((wave_s >> 1) ^ ((lane_s >> 2) & 3)) << 4) + 8 * (wave_s & 1));
            auto update4 = [&](int i2, int fg, const float (&eps_in)[4], const float (&xv)[4]) {
                op4 nvb;
#pragma unroll
                for (int j = 0; j < 4; ++j) nvb[j] = (op_t)0.f;
                if (valid && 4 * fg < g.F) {                              // (F % 4 == 0: whole groups)
                    const f32x4 x04 = *reinterpret_cast<const f32x4*>(tile + lane_s * LY_XLD + 4 * fg);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = 4 * fg + j;
                        float x0 = x04[j];
                        if (sp.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                        const size_t o = (size_t)gb * FT + (size_t)f * T + t;
                        if (sp.x0_out) sp.x0_out[o] = x0;
                        float eps = eps_in[j];
                        if (sp.noise)
                            eps = sp.noise[(size_t)(sp.first_index - step) * g.B * FT + (size_t)bn * FT + (size_t)f * T + t];
                        else if (!quads)
                            eps = philox_normal(sp.seed, sp.sample_offset + bn, (uint32_t)step, (uint32_t)(f * 4096 + t));
                        float nv;
                        if (sp.sampler == 0) {
                            const float mean = __fadd_rn(__fmul_rn(k.c1, x0), __fmul_rn(k.c2, xv[j]));
                            nv = __fadd_rn(mean, __fmul_rn(k.sig_ddpm, eps));
                        } else {
                            const float e = __fdiv_rn(__fsub_rn(__fmul_rn(k.sr, xv[j]), x0), k.srm1);
                            const float mean = __fadd_rn(__fmul_rn(x0, k.ca), __fmul_rn(k.cb, e));
                            nv = __fadd_rn(mean, __fmul_rn(k.sig_ddim, eps));
                        }
                        sp.x[o] = nv;
                        nvb[j] = (op_t)nv;
                    }
                }
                // x' (0 in the K padding columns and the surplus rows) -> the K32-blocked image of the embedding's A operand
                *reinterpret_cast<op4*>(ximg + i2 * 4096 + ximg_lane) = nvb;
            };
            ly_static_for<LY_NKX>([&](auto IT) __attribute__((always_inline)) {   // groups of 4 features
                constexpr int i2 = decltype(IT)::value;
                const int fg = wave_s + 8 * i2;
                float eps4[4] = {0.f, 0.f, 0.f, 0.f};
                if (quads) {                                               // wave-uniform
                    const uint32_t elem = (uint32_t)((4 * fg + q) * 4096 + tq);
                    const unsigned long long sample = sp.sample_offset + bn;
                    uint32_t rr4[4];
                    philox4x32_10(elem >> 2, (uint32_t)step, (uint32_t)sample, (uint32_t)(sample >> 32), (uint32_t)sp.seed, (uint32_t)(sp.seed >> 32), rr4);
                    float n4[4];
#pragma unroll
                    for (int pair = 0; pair < 2; ++pair) box_muller(rr4[2 * pair], rr4[2 * pair + 1], n4[2 * pair], n4[2 * pair + 1]);
                    // 4 x 4 transpose inside the quad (this lane - frame tq + q - needs, for feature 4 fg + j, element q of lane j's n4): two
                    // butterfly stages of a conditional swap with the lane q ^ 1, then q ^ 2 (DPP quad_perm): 16 operations instead of 32
                    auto stage = [&](float& lo_r, float& hi_r, bool bit, auto ctrl) {
                        const float send = bit ? lo_r : hi_r;
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), decltype(ctrl)::value, 0xf, 0xf, true));
                        lo_r = bit ? recv : lo_r;
                        hi_r = bit ? hi_r : recv;
                    };
                    const bool b0 = (q & 1) != 0, b1 = (q & 2) != 0;
                    stage(n4[0], n4[1], b0, std::integral_constant<int, 0xB1>{});   // quad_perm [1, 0, 3, 2]
                    stage(n4[2], n4[3], b0, std::integral_constant<int, 0xB1>{});
                    stage(n4[0], n4[2], b1, std::integral_constant<int, 0x4E>{});   // quad_perm [2, 3, 0, 1]
                    stage(n4[1], n4[3], b1, std::integral_constant<int, 0x4E>{});
#pragma unroll
                    for (int j = 0; j < 4; ++j) eps4[j] = n4[j];
                }
                update4(i2, fg, eps4, xpre[i2]);
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RGN_LYS(14)
        // ---- D: h' = x' . Wx'^T + c0 -> image X (the fp32 tile is dead since the barrier above; the x' image lies behind X)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a2][b2][i] = 0.f;
        {
            int a_offx[2] = {a_off[0] + LY_XIMG, a_off[1] + LY_XIMG};
            gemm_n(acc, a_offx, p_wx, p_wx, std::false_type{}, std::integral_constant<int, 8 + 16>{}, std::integral_constant<int, 2 * LY_NKX>{});
        }
        if constexpr (GUIDED) {   // the same embedding with the unconditional condition rows -> image Y -> the planes (read back at the next step's pass 1)
            __builtin_amdgcn_s_barrier();                                 // Y overlaps the x' image: every wave must be out of the embedding's k-loop first
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        op4 hh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hh[e] = (op_t)(acc[nt][mt][4 * i4 + e] + (float)c0u[nt][mt][i4][e]);
                        *reinterpret_cast<op4*>(smem + LY_Y + img_off(nt, i4, mt)) = hh;
                    }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] += (float)c0v[nt][mt][i4][e];
        store_img(acc, LY_X);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (GUIDED) {
            const __amdgpu_buffer_rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.rows * 512 * 2), 0x00020000);
            const int r16 = lane_s >> 2, c = lane_s & 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = wave_s * 8 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
                if (r < T) {
                    const int off = blk * LY_KB + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                    __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(smem + LY_Y + off), u_rs,
                                                           (int)((((size_t)blk * g.rows + g.half + row0_s + r) * 32 + c * 8) * 2), 0, RGN_LY_ST_AUX);
                }
            }
        }
        RGN_LYS(15)
    }
    }
    // ---- the sample's rows -> output planes (write-through)
    {
        const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.rows * 512 * 2), 0x00020000);
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave * 8 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
            if (r < Tq) {
                const int off = blk * LY_KB + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(smem + LY_X + off), o_rs,
                                                       (int)((((size_t)blk * g.rows + row0 + r) * 32 + c * 8) * 2), 0, RGN_LY_ST_AUX);
            }
        }
    }
    if constexpr (STEPS) {
        if (tid == 0) {   // ticket: the last workgroup of the launch moves the device-side loop index on by the steps it ran
            int* tick = g.d_stepw + 4;
            if (atomicAdd(tick, 1) == (int)gridDim.x - 1) {
                tick[0] = 0;
                g.d_stepw[0] = first_step - g.steps;
            }
        }
    }
}

#ifdef RGN_LY_STAMPS
void ly_stamps_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ly_st), sizeof(long long) * 1024 * 16); }
#endif

bool layers_supported(int d, int ff, int H, int Tq, int L) { return d == 512 && ff == 1024 && H == 4 && Tq >= 1 && Tq <= 64 && L >= 1 && L <= LY_MAXL; }
bool layers_steps_supported(int d, int F, int Kpx) { return d == 512 && F % 4 == 0 && F <= 352 && Kpx == 32 * LY_NKX; }
template <class K>
static hipError_t ly_lds(K kern) { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LY_LDS); }
hipError_t configure_layers() {
    hipError_t e = ly_lds(k_layers<false>);
    if (e == hipSuccess) e = ly_lds(k_layers<true>);
    if (e == hipSuccess) e = ly_lds(k_layers<true, true>);
    if (e == hipSuccess) e = ly_lds(k_layers<true, false, true>);
    if (e == hipSuccess) e = ly_lds(k_layers<true, true, true>);
    if (e == hipSuccess) e = ly_lds(k_layers<false, false, true>);
    return e;
}
hipError_t launch_layers(const LayersArgs& g, hipStream_t s) {
    if (g.steps > 0 && g.scale) {
        if (g.f16) hipLaunchKernelGGL((k_layers<true, true, true>), dim3(g.Bm), dim3(LY_NTH), LY_LDS, s, g);
        else hipLaunchKernelGGL((k_layers<true, true>), dim3(g.Bm), dim3(LY_NTH), LY_LDS, s, g);
    } else if (g.steps > 0) {
        if (g.f16) hipLaunchKernelGGL((k_layers<true, false, true>), dim3(g.Bm), dim3(LY_NTH), LY_LDS, s, g);
        else hipLaunchKernelGGL(k_layers<true>, dim3(g.Bm), dim3(LY_NTH), LY_LDS, s, g);
    } else if (g.f16) hipLaunchKernelGGL((k_layers<false, false, true>), dim3(g.Bm), dim3(LY_NTH), LY_LDS, s, g);
    else hipLaunchKernelGGL(k_layers<false>, dim3(g.Bm), dim3(LY_NTH), LY_LDS, s, g);
    return hipGetLastError();
}

}  // namespace rgn
