// Row-persistent decoder-layer tail for the SPLIT-bf16 phase (the last steps of the precision schedule, uniform split-bf16 mode): for a tile of
// 32 complete token rows ONE workgroup runs
//
//   h' = LN2( LN1( att . Wo^T + bo + h ) + call_time[step] + call_cond[sample] )        out_proj, norm1, folded cross-attn, norm2
//   y  = LN3( gelu( h' . W1^T + b1 ) . W2^T + b2 + h' )                                 linear1, GELU, linear2, norm3
//
// (nn.TransformerDecoderLayer post-norm blocks constructed at model/cmdm.py:75-81, called at :227) with every intermediate on chip - the
// structure of rgn_mlp2.hip in the arithmetic of rgn_gemm_x3.hip: every operand is a (hi, lo) pair of bf16 planes, a product is three MFMAs
// (a_hi w_lo + a_lo w_hi + a_hi w_hi, fp32 accumulate; the lo lo term, 2^-18, is dropped), LayerNorm is two-pass (mean, then the centred sum of
// squares), GELU is the erf form with |erf error| <= 1.5e-7 - the same accuracy class as the kernels it replaces (k_gemm_x3 x 3 + k_layernorm x 2
// per layer, five launches and four round trips of the activations through HBM).
//
// Why 32 rows: the images are pairs. LDS: XH | XL (att tile -> GELU(hidden half) -> output), YH | YL (h': A operand of linear1, residual of
// norm3), 32 KiB each = 128 KiB, + statistics exchange + wave-private vectors = 144 KiB: one workgroup of 8 waves per CU, wave w = output columns
// [64 w, 64 w + 64) (two accumulator tiles of 32 x 32). A weight fragment PAIR (2 KiB) feeds three MFMAs, so the L2 -> register weight stream
// (4 B per weight per 32 rows) bounds the loops, not the matrix pipe; what the kernel removes is everything else the five launches carried.
// Weights: fragment-ordered hi and lo planes [K/32][N/32][2][64][8] (rgn_pack.cpp pack_linear), streamed through a register ring of RD granules
// (half k-steps: 2 column blocks x (hi, lo) = 4 buffer loads) that never drains between the five GEMM passes.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

#ifndef RGN_MX_RD
#define RGN_MX_RD 6        // weight ring depth in granules of 4 fragments (16 registers each)
#endif

namespace {

struct MX {
    static constexpr int R = 32, NW = 8, NTH = 64 * NW, NT = 2, CW = 64, RD = RGN_MX_RD;
    static constexpr int KB = R * 64, IMG = 16 * KB;      // bytes of one k-block [32 rows][64 B] and of an image
    static constexpr int NSAMP = 2;                        // samples a tile can touch (Tq >= 32)
    static constexpr int XH = 0, XL = IMG, YH = 2 * IMG, YL = 3 * IMG, RED = 4 * IMG, REDF = NW * R /* floats per exchange buffer */,
                         VEC = RED + 4 * REDF * 4, VECW = (4 + NSAMP) * CW, LDS = VEC + NW * VECW * 4;
    static constexpr int A_BO = 0, A_G1 = CW, A_G2 = 2 * CW, A_B2 = 3 * CW, A_SPV = 4 * CW /* NSAMP x CW */;
    static constexpr int B_BF1 = 0 /* 2 x CW */, B_BF2 = 2 * CW, B_G3 = 3 * CW, B_B3 = 4 * CW;
};
static_assert(MX::LDS == 144 * 1024 && MX::LDS <= 160 * 1024, "LDS map");

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7) on fast exp / rcp: rgn_gemm_x3.hip's
__device__ __forceinline__ float mx_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // (1 ulp: below the fit's own 1.5e-7)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(e, x);
}
__device__ __forceinline__ float mx_gelu(float v) { return v * 0.5f * (1.0f + mx_erf(v * 0.70710678118654752440f)); }

}  // namespace

__global__ __launch_bounds__(MX::NTH, 2) void k_mlp_x3(MlpX3Args gx) {
    using C = MX;
    constexpr int NT = C::NT, NW = C::NW, R = C::R, CW = C::CW, RD = C::RD, KB = C::KB, NSAMP = C::NSAMP;
    const MlpArgs& g = gx.p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = xcd_affine(blockIdx.x, gridDim.x) * R;
    float* vec = reinterpret_cast<float*>(smem + C::VEC) + wave * C::VECW;   // this wave's private region
    float* red = reinterpret_cast<float*>(smem + C::RED);
    // ---- att tile (hi, lo) -> XH | XL by DMA: per image 16 k-blocks x 2 pieces of 1 KiB (16 rows x 64 B); wave w issues the pieces w, w + 8, ...
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = wave + NW * j, lo = q >> 5, p = q & 31, kb = p >> 1, r = (p & 1) * 16 + r16;
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            const size_t src = ((size_t)kb * g.rows + m) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)((lo ? gx.att_lo : g.att) + src), (RGN_AS3 void*)(smem + (lo ? C::XL : C::XH) + p * 1024), 16, 0, 0);
        }
    }
    asm volatile("" ::: "memory");                                    // (nothing below is issued ahead of the DMA pieces: the count further down relies on it)
    // B-operand fragment of token l31 inside a k-block image [32 rows][64 B] (16-byte chunks swizzled by the row), per 16-wide k-half
    int a_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[ks] = l31 * 64 + (((2 * ks + kh) ^ ((l31 >> 2) & 3)) << 4);

    // ---- weight ring: granule = half a k-step (16 k) of this wave's 2 column blocks, hi and lo planes
    struct Pass { __amdgpu_buffer_rsrc_t hi, lo; int kstride, hs0; };   // kstride = nb_all * 2048 bytes per k-block
    bf16x8 wh[RD][NT], wl[RD][NT];
    const int lane16 = lane * 16;
    auto load_g = [&](const Pass& ps, int hs_rel, int slot) {
        const int hs = ps.hs0 + hs_rel;
        const int soff = (hs >> 1) * ps.kstride + (hs & 1) * 1024;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            wh[slot][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ps.hi, lane16, soff + nt * 2048, 0));
            wl[slot][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ps.lo, lane16, soff + nt * 2048, 0));
        }
    };
    // one GEMM pass over K = 512: acc[nt] += A_image(16 k-blocks, hi at imgh, lo at imgl) . W[the wave's column blocks, granules hs0 .. hs0 + 31]^T.
    // The ring never drains between passes: the tail of a pass requests the first RD - 1 granules of the NEXT pass (chain). BASE: the ring slot of
    // this pass's granule 0 (32 is not a multiple of RD: the slots rotate from pass to pass).
    auto gemm32 = [&](f32x16 (&acc)[NT], const char* imgh, const char* imgl, const Pass& cur, const Pass& nxt, auto chain, auto base) {
        constexpr int AH = RD - 1, BASE = decltype(base)::value;
        constexpr bool CH = decltype(chain)::value;
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 ah = *reinterpret_cast<const bf16x8*>(imgh + a_off[0]), al = *reinterpret_cast<const bf16x8*>(imgl + a_off[0]);
#pragma unroll
        for (int hs = 0; hs < 32; ++hs) {
            bf16x8 ahn = ah, aln = al;
            if (hs + 1 < 32) {                                        // one granule ahead
                ahn = *reinterpret_cast<const bf16x8*>(imgh + ((hs + 1) >> 1) * KB + a_off[(hs + 1) & 1]);
                aln = *reinterpret_cast<const bf16x8*>(imgl + ((hs + 1) >> 1) * KB + a_off[(hs + 1) & 1]);
            }
            if (hs + AH < 32) load_g(cur, hs + AH, (BASE + hs + AH) % RD);
            else if (CH) load_g(nxt, hs + AH - 32, (BASE + hs + AH) % RD);
            // this granule is in, the next RD - 1 stay in flight (the statement also keeps the compiler from hoisting later granules' loads up here)
            if (hs + AH < 32 || CH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NT * AH) : "memory");
            const int slot = (BASE + hs) % RD;
            // small terms first, then hi . hi (rgn_gemm_x3.hip); the two accumulators alternate so that no MFMA waits for its predecessor
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[slot][nt], ah, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[slot][nt], al, acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[slot][nt], ah, acc[nt], 0, 0, 0);
            ah = ahn;
            al = aln;
            // granule by granule: left alone the compiler runs ONE accumulator's chain through the whole pass and sinks the other behind it
            // (every fragment of the pass then lives in scratch until that second sweep): both accumulators are pinned here
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // element (token l31, column 64 wave + 32 nt + 8 i4 + 4 kh + e) <-> register acc[nt][4 i4 + e]
    auto col4 = [&](int nt, int i4) { return 32 * nt + 8 * i4 + 4 * kh; };          // inside the wave's column slice
    int img_base = (NT * wave) * KB + l31 * 64 + 8 * kh;
    asm volatile("" : "+v"(img_base));
    const int swz = (l31 >> 2) & 3;
    auto img_off = [&](int nt, int i4) { return img_base + nt * KB + ((i4 ^ swz) << 4); };   // its 8-byte run inside an image [16][32 rows][64 B]
    auto init_bias = [&](f32x16 (&acc)[NT], const float* bias) {                      // bias: the wave's column slice in LDS
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + col4(nt, i4));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nt][4 * i4 + e] = b[e];
            }
    };
    // LayerNorm over the 512 columns of every token, in place, two-pass like k_layernorm: mean, then the centred sum of squares. Two exchanges
    // ([wave][token] floats; the halves of a wave hold the same token's two column groups: lane ^ 32 sum first), four alternating buffers.
    const float invn = 1.0f / 512.f;
    auto layernorm = [&](f32x16 (&acc)[NT], const float* gam, auto slot, auto shift /* (nt, i4) -> f32x4 */) {
        float* b0 = red + (2 * decltype(slot)::value) * C::REDF;
        float* b1 = b0 + C::REDF;
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[nt][i];
        s = half_sum(s);
        b0[wave * R + l31] = s;                                       // (both halves write the same value)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += b0[w * R + l31];
        const float mean = tot * invn;
        float q = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float c = acc[nt][i] - mean;
                acc[nt][i] = c;
                q = fmaf(c, c, q);
            }
        q = half_sum(q);
        b1[wave * R + l31] = q;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float qt = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) qt += b1[w * R + l31];
        const float rstd = 1.0f / sqrtf(qt * invn + 1e-5f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + col4(nt, i4));
                const f32x4 sh = shift(nt, i4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nt][4 * i4 + e] = acc[nt][4 * i4 + e] * rstd * ga[e] + sh[e];
            }
    };
    // fp32 accumulators -> a (hi, lo) pair of images
    auto store_img = [&](const f32x16 (&acc)[NT], int imgh, int imgl) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                bf16x4 hh, ll;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[nt][4 * i4 + e];
                    hh[e] = (__bf16)v;
                    ll[e] = (__bf16)(v - (float)hh[e]);
                }
                *reinterpret_cast<bf16x4*>(smem + imgh + img_off(nt, i4)) = hh;
                *reinterpret_cast<bf16x4*>(smem + imgl + img_off(nt, i4)) = ll;
            }
    };

    // =============== stage 1: out_proj + residual + norm1 + folded cross-attention + norm2 -> h' (Y) ====================
    auto wrs = [&](const __bf16* W, int cb0, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W) + (size_t)cb0 * 1024, 0, bytes - cb0 * 2048, 0x00020000);
    };
    const int cb = NT * wave;
    const Pass p_wo{wrs(g.Wo, cb, 512 * 512 * 2), wrs(gx.Wo_lo, cb, 512 * 512 * 2), 16 * 2048, 0},
        p_w1a{wrs(g.W1, cb, 1024 * 512 * 2), wrs(gx.W1_lo, cb, 1024 * 512 * 2), 32 * 2048, 0},
        p_w1b{wrs(g.W1, 16 + cb, 1024 * 512 * 2), wrs(gx.W1_lo, 16 + cb, 1024 * 512 * 2), 32 * 2048, 0},
        p_w2a{wrs(g.W2, cb, 512 * 1024 * 2), wrs(gx.W2_lo, cb, 512 * 1024 * 2), 16 * 2048, 0}, p_w2b{p_w2a.hi, p_w2a.lo, 16 * 2048, 32};
#pragma unroll
    for (int s = 0; s < RD - 1; ++s) load_g(p_wo, s, s);             // right behind the att DMA
    const int cw = CW * wave + lane;                                   // this lane's column of every vector slice
    // phase A vectors of the wave's columns (staged to the wave's LDS region after the out_proj loop), the residual tile straight into registers,
    // phase B vectors (held in registers until the wave is past norm2)
    float va[4], sv, pv[NSAMP], vb[5];
    int step = 0;
    {
        va[0] = g.bo[cw]; va[1] = g.g1[cw]; va[2] = g.g2[cw]; va[3] = g.b2[cw];
        if (g.stepvec) step = *g.d_step;
        const int s0 = m0 / g.Tq, slast = (g.M - 1) / g.Tq;
        sv = g.b1[cw];                                                 // norm1's beta, folded into the per-sample vector
#pragma unroll
        for (int j = 0; j < NSAMP; ++j) {
            const int sidx = s0 + j < slast ? s0 + j : slast;
            pv[j] = g.pervec ? g.pervec[(size_t)sidx * g.ldper + cw] : 0.f;
        }
    }
    bf16x4 rh[NT][4], rl[NT][4];
    {
        int m = m0 + l31;
        m = m < g.M ? m : g.M - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const size_t o = ((size_t)(cb + nt) * g.rows + m) * 32 + 8 * i4 + 4 * kh;
                rh[nt][i4] = *reinterpret_cast<const bf16x4*>(g.h + o);
                rl[nt][i4] = *reinterpret_cast<const bf16x4*>(gx.h_lo + o);
            }
    }
    vb[0] = g.bf1[cw]; vb[1] = g.bf1[512 + cw]; vb[2] = g.bf2[cw]; vb[3] = g.g3[cw]; vb[4] = g.b3[cw];
    const float tv = g.stepvec ? g.stepvec[(size_t)step * g.ldstep + cw] : 0.f;
    // the att images are complete once EVERY wave's DMA pieces have landed: they are this wave's oldest vector-memory operations; at least
    // 46 younger ones follow (ring 20, vectors 5 + 5, residual 16; the per-sample / step vectors may be absent), which may stay in flight
    asm volatile("s_waitcnt vmcnt(46)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    gemm32(acc, smem + C::XH, smem + C::XL, p_wo, p_w1a, std::true_type{}, std::integral_constant<int, 0>{});
    // phase A vectors -> the wave's LDS region (wave-private: program order suffices)
#pragma unroll
    for (int v = 0; v < 4; ++v) vec[CW * v + lane] = va[v];
#pragma unroll
    for (int j = 0; j < NSAMP; ++j) vec[C::A_SPV + CW * j + lane] = (tv + sv) + pv[j];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(vec + C::A_BO + col4(nt, i4));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][4 * i4 + e] = (acc[nt][4 * i4 + e] + b[e]) + ((float)rh[nt][i4][e] + (float)rl[nt][i4][e]);
        }
    {   // norm1 (gamma) + norm1.beta + call_time[step] + call_cond[sample of the token] (pre-summed per sample in LDS)
        const int m = m0 + l31;
        const float* spv = vec + C::A_SPV + ((m < g.M ? m : g.M - 1) / g.Tq - m0 / g.Tq) * CW;
        layernorm(acc, vec + C::A_G1, std::integral_constant<int, 0>{}, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(spv + col4(nt, i4)); });
    }
    layernorm(acc, vec + C::A_G2, std::integral_constant<int, 1>{}, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(vec + C::A_B2 + col4(nt, i4)); });
    store_img(acc, C::YH, C::YL);
    // phase B vectors over phase A (wave-private)
#pragma unroll
    for (int v = 0; v < 5; ++v) vec[CW * v + lane] = vb[v];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // h' images complete

    // =============== stage 2: linear1 + GELU + linear2, the hidden 1024 columns in two halves ==============================
    f32x16 acc2[NT];
    init_bias(acc2, vec + C::B_BF2);
    // ring slot of a pass's first granule: 32 granules per pass, RD slots
    constexpr int B1 = 32 % RD, B2 = 64 % RD, B3 = 96 % RD, B4 = 128 % RD;
    {
        init_bias(acc, vec + C::B_BF1);
        gemm32(acc, smem + C::YH, smem + C::YL, p_w1a, p_w2a, std::true_type{}, std::integral_constant<int, B1>{});   // hidden columns [0, 512)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = mx_gelu(acc[nt][i]);
        store_img(acc, C::XH, C::XL);                                 // (X still holds the att tile, dead since stage 1's loop: every wave passed the h' barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        gemm32(acc2, smem + C::XH, smem + C::XL, p_w2a, p_w1b, std::true_type{}, std::integral_constant<int, B2>{});   // linear2 over hidden k-blocks [0, 16)
    }
    {
        init_bias(acc, vec + C::B_BF1 + CW);
        gemm32(acc, smem + C::YH, smem + C::YL, p_w1b, p_w2b, std::true_type{}, std::integral_constant<int, B3>{});   // hidden columns [512, 1024)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = mx_gelu(acc[nt][i]);
        __builtin_amdgcn_s_barrier();                                 // every wave is done reading the first half's images
        store_img(acc, C::XH, C::XL);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        gemm32(acc2, smem + C::XH, smem + C::XL, p_w2b, p_w2b, std::false_type{}, std::integral_constant<int, B4>{});  // hidden k-blocks [16, 32)
    }

    // =============== stage 3: + residual h' + norm3 -> output planes =====================================================
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const bf16x4 a = *reinterpret_cast<const bf16x4*>(smem + C::YH + img_off(nt, i4)), b = *reinterpret_cast<const bf16x4*>(smem + C::YL + img_off(nt, i4));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2[nt][4 * i4 + e] += (float)a[e] + (float)b[e];
        }
    layernorm(acc2, vec + C::B_G3, std::integral_constant<int, 0>{}, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(vec + C::B_B3 + col4(nt, i4)); });   // (its barriers also fence the last reads of X)
    store_img(acc2, C::XH, C::XL);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const int bytes = (int)((size_t)g.rows * 512 * 2);
        const __amdgpu_buffer_rsrc_t oh = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, bytes, 0x00020000), ol = __builtin_amdgcn_make_buffer_rsrc(gx.out_lo, 0, bytes, 0x00020000);
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = wave * 8 + j, lo = q >> 5, p = q & 31, blk = p >> 1, r = (p & 1) * 16 + r16;   // waves 0-3: the hi image, 4-7: the lo image
            const int m = m0 + r;
            if (m < g.M) {
                const int off = blk * KB + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(smem + (lo ? C::XL : C::XH) + off), lo ? ol : oh,
                                                       (int)((((size_t)blk * g.rows + m) * 32 + c * 8) * 2), 0, 16);   // write-through (sc1)
            }
        }
    }
}

bool mlp_x3_supported(int d, int ff, int Tq) { return d == 512 && ff == 1024 && 31 / Tq + 2 <= MX::NSAMP; }
hipError_t configure_mlp_x3() { return hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_x3), hipFuncAttributeMaxDynamicSharedMemorySize, MX::LDS); }
hipError_t launch_mlp_x3(const MlpX3Args& g, hipStream_t s) {
    hipLaunchKernelGGL(k_mlp_x3, dim3((g.p.M + 31) / 32), dim3(MX::NTH), MX::LDS, s, g);
    return hipGetLastError();
}

}  // namespace rgn
