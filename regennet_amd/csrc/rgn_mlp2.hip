// Row-persistent decoder-layer tail for the plain-bf16 phase (second build: round 2's k_mlp is gone, its measurements are DESIGN.md 4.0b): for a tile of R
// complete token rows ONE workgroup runs
//
//   h' = LN2( LN1( att . Wo^T + bo + h ) + call_time[step] + call_cond[sample] )        out_proj, norm1, folded cross-attn, norm2
//   y  = LN3( gelu( h' . W1^T + b1 ) . W2^T + b2 + h' )                                 linear1, GELU, linear2, norm3
//
// (nn.TransformerDecoderLayer post-norm blocks constructed at model/cmdm.py:75-81, called at :227) with every intermediate on chip.
// Two instantiations of one template:
//   MT = 2: R = 64 rows, 8 waves, wave w = output columns [64 w, 64 w + 64) (2 x 2 accumulator tiles of 32 x 32), one workgroup
//           per CU - a weight fragment feeds two MFMAs; the shape for launches that fill the chip
//   MT = 1: R = 32 rows, 4 waves, wave w = columns [128 w, 128 w + 128) (4 x 1 tiles), TWO independently scheduled workgroups
//           per CU - a weight fragment feeds one MFMA, i.e. twice the L2 -> register weight stream per row: measured slower
//           whenever the 64-row tiles fill the chip (the stream is what bounds the loops), faster for launches of few tiles
// What changed against round 2's k_mlp (all measured in the sampling loop):
//   * the layer input tile h (residual of norm1) never touches LDS: every lane loads the 64 values it will add straight into
//     registers BEHIND the att DMA and the first weight fragments, and the first MFMA waits for the att tile only
//   * ONE continuous weight stream: buffer loads (scalar resource + compile-time offsets, the lane contributes lane * 16), a ring of
//     half-k-step granules that never drains between the five GEMM passes - the tail of a pass requests the head of the next one, so
//     the epilogues run with the next pass's first fragments in flight
//   * per-column vectors: every wave stages ITS OWN column slices (wave-private LDS, no barrier); phase A (out_proj bias, norm1 /
//     norm2, per-sample vectors) is overwritten by phase B (linear1 / linear2 biases, norm3) once the wave is past norm2
//   * LayerNorm statistics as sum and sum of squares in ONE exchange (fp32; plain-bf16 phase only - the split-bf16 tail keeps the
//     two-pass kernels), mean folded into the final FMA; exchange layout [stat][wave][token]: conflict-free both ways
//   LDS X: att tile image (A operand of out_proj) -> GELU(hidden half) image (A operand of linear2) -> output image
//   LDS Y: h' image (A operand of linear1, residual of norm3)
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#ifndef RGN_M2_HRES
#define RGN_M2_HRES 0      // where the residual tile is requested: 0 = in the prologue behind the att DMA, 1 = behind the att barrier
#endif
#ifndef RGN_M2_ST_AUX
#define RGN_M2_ST_AUX 16   // output stores write-through (sc1): nothing left dirty in the XCD L2s for the end-of-kernel write-back
#endif

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

template <int MT>
struct M2 {
    static constexpr int R = 32 * MT, NW = 4 * MT, NTH = 64 * NW, NT = 4 / MT, CW = 512 / NW;   // rows, waves, threads, column blocks and columns per wave
    static constexpr int RD = 4 * MT;                 // weight ring depth in granules (half k-steps of NT fragments): 64 registers either way
    static constexpr int KB = R * 64, IMG = 16 * KB;  // bytes of one k-block [R rows][64 B] and of an image
    static constexpr int NSAMP = MT == 1 ? 2 : 4;     // samples a tile can touch (mlp2_supported)
    // LDS map: X | Y | statistics exchange (2 buffers x [2 stats][NW waves][R tokens]) | NW wave-private vector regions
    static constexpr int X = 0, Y = IMG, RED = 2 * IMG, REDF = 2 * NW * R, VEC = RED + 2 * REDF * 4, VECW = (4 + NSAMP) * CW, LDS = VEC + NW * VECW * 4;
    // wave-private vector region (floats): phase A (stage 1) and phase B (stages 2, 3)
    static constexpr int A_BO = 0, A_G1 = CW, A_G2 = 2 * CW, A_B2 = 3 * CW, A_SPV = 4 * CW /* NSAMP x CW */;
    static constexpr int B_BF1 = 0 /* 2 x CW: hidden halves */, B_BF2 = 2 * CW, B_G3 = 3 * CW, B_B3 = 4 * CW;
    static_assert(32 % RD == 0, "ring slots continue across passes");
    static_assert((2 / MT) * LDS <= 160 * 1024, "LDS map");
};
static_assert(M2<1>::LDS == 78 * 1024 && M2<2>::LDS == 152 * 1024, "LDS map");

#ifdef RGN_M2_STAMPS
// tools/mlp_bench -DRGN_M2_STAMPS: wave 0 of EVERY workgroup stamps s_memtime at the phase boundaries, plus where it runs
// (HW_ID: SIMD / CU / SE, XCC_ID), so that the host can put the workgroups of one CU next to each other
__device__ long long g_m2_st[1024][12];
#define RGN_M2T(i)                                                                                                  \
    {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                                \
            g_m2_st[blockIdx.x][i] = __builtin_readcyclecounter();                                                  \
            if (i == 0) {                                                                                           \
                unsigned hw, xcc;                                                                                   \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                    \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                  \
                g_m2_st[blockIdx.x][10] = hw;                                                                       \
                g_m2_st[blockIdx.x][11] = xcc;                                                                       \
            }                                                                                                       \
        }                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
#else
#define RGN_M2T(i)
#endif

// GELU (erf form): x (0.5 + t Q(t^2)) with t = clamp(x, +-3.9) and t Q(t^2) ~ Phi(t) - 0.5: an odd degree-13 minimax polynomial on
// [0, 3.9] (max abs error 8.3e-5 in Phi with the clamp's 4.8e-5 beyond it; max abs error of the GELU 3.2e-4 over [-8, 8], evaluated in
// fp32 like here - the degree-15 fit on [0, 4.53] this replaces had 8.1e-5 and 6.4e-4: the wider interval bought nothing the bf16
// rounding of the result does not hide 10x over), the 1/sqrt 2 and the 0.5 folded into the coefficients: 11 instructions per pair
__device__ __forceinline__ f32x2 m2_gelu2(f32x2 x) {
    const f32x2 t = {__builtin_amdgcn_fmed3f(x[0], -3.9f, 3.9f), __builtin_amdgcn_fmed3f(x[1], -3.9f, 3.9f)};   // (no canonicalising v_max in front, unlike min(max()))
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

template <int MT, bool F16 = false>
__global__ __launch_bounds__(M2<MT>::NTH, 2) void k_mlp2(MlpArgs g) {
    using C = M2<MT>;
    using OP = OpFmt<F16>;                // bf16 or fp16 operands (rgn_internal.h): planes in, planes out, weight planes
    using op_t = typename OP::t;
    using op8 = typename OP::v8;
    using op4 = typename OP::v4;
    constexpr int NT = C::NT, NW = C::NW, R = C::R, CW = C::CW, RD = C::RD, KB = C::KB, NSAMP = C::NSAMP, VK = CW / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = xcd_affine(blockIdx.x, gridDim.x) * R;
    float* vec = reinterpret_cast<float*>(smem + C::VEC) + wave * C::VECW;   // this wave's private region
    RGN_M2T(0)
    // ---- att tile -> X by DMA: 16 k-blocks x R/16 pieces of 1 KiB (16 rows x 64 B), wave w issues the pieces w, w + NW, ...
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave + NW * j, kb = p / (R / 16), r = (p % (R / 16)) * 16 + r16;
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            const size_t src = ((size_t)kb * g.rows + m) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.att + src), (RGN_AS3 void*)(smem + C::X + p * 1024), 16, 0, 0);
        }
    }
    // B-operand fragment of token l31 (+ 32 mt: an immediate offset of 2 KiB) inside a k-block image [R rows][64 B] (16-byte chunks
    // swizzled by the row), per 16-wide k-half
    int a_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[ks] = l31 * 64 + (((2 * ks + kh) ^ ((l31 >> 2) & 3)) << 4);

    // ---- weight ring: granule = half a k-step (16 k) of this wave's NT column blocks; W: fragment-ordered plane
    //      [K/32][nb_all][2][64][8] (rgn_rowgemm.hip). Granule index hs = 2 kt + ks. Buffer loads: the resource (based at the wave's
    //      first column block) is scalar and every granule offset a compile-time constant, the lane contributes lane * 16
    struct Pass { __amdgpu_buffer_rsrc_t rs; int kstride, hs0; };   // kstride = nb_all * 2048 bytes per k-block
    op8 wf[RD][NT];
    const int lane16 = lane * 16;
    auto load_g = [&](const Pass& ps, int hs_rel, int slot) {
        const int hs = ps.hs0 + hs_rel;
        const int soff = (hs >> 1) * ps.kstride + (hs & 1) * 1024;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wf[slot][nt] = __builtin_bit_cast(op8, __builtin_amdgcn_raw_buffer_load_b128(ps.rs, lane16, soff + nt * 2048, 0));
    };
    // one GEMM pass over K = 512: acc[nt][mt] += A_image(16 k-blocks at img) . W[the wave's column blocks, granules hs0 .. hs0 + 31]^T.
    // The ring never drains between passes: the tail of a pass requests the first RD - 1 granules of the NEXT pass (chain). `extra`:
    // vector-memory operations issued between the granules RD - 2 and RD - 1 of this pass that may stay in flight (stage 1: the
    // residual tile and the phase-B vectors).
    auto gemm32 = [&](f32x16 (&acc)[NT][MT], const char* img, const Pass& cur, const Pass& nxt, auto chain, auto extra) {
        constexpr int EX = decltype(extra)::value, AH = RD - 1;
        constexpr bool CH = decltype(chain)::value;
        __builtin_amdgcn_sched_barrier(0);
        op8 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const op8*>(img + a_off[0] + mt * 2048);
#pragma unroll
        for (int hs = 0; hs < 32; ++hs) {
            op8 afn[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                afn[mt] = af[mt];
                if (hs + 1 < 32) afn[mt] = *reinterpret_cast<const op8*>(img + ((hs + 1) >> 1) * KB + a_off[(hs + 1) & 1] + mt * 2048);   // one granule ahead
            }
            if (hs + AH < 32) load_g(cur, hs + AH, (hs + AH) % RD);
            else if (CH) load_g(nxt, hs + AH - 32, (hs + AH) % RD);
            if (hs + AH < 32 || CH) {
                if (hs < AH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT * AH + EX) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT * AH) : "memory");   // this granule is in; the next RD - 1 stay in flight
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = OP::mfma(wf[hs % RD][nt], af[mt], acc[nt][mt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = afn[mt];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // element (token 32 mt + l31, column CW wave + 32 nt + 8 i4 + 4 kh + e) <-> register acc[nt][mt][4 i4 + e]
    auto col4 = [&](int nt, int i4) { return 32 * nt + 8 * i4 + 4 * kh; };          // inside the wave's column slice
    auto img_off = [&](int nt, int i4, int mt) {                                      // its 8-byte run inside an image [16][R rows][64 B]
        return (NT * wave + nt) * KB + mt * 2048 + l31 * 64 + ((i4 ^ ((l31 >> 2) & 3)) << 4) + 8 * kh;
    };
    auto init_bias = [&](f32x16 (&acc)[NT][MT], const float* bias) {                  // bias: the wave's column slice in LDS
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + col4(nt, i4));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] = b[e];
            }
    };
    // LayerNorm over the 512 columns of every token, in place: o = v (rstd gamma) + (shift - mean rstd gamma). One exchange of
    // (sum, sum of squares): halves by lane ^ 32, the NW column slices through LDS ([stat][wave][token], conflict-free both ways)
    // (exchange addresses = ONE opaque base register + immediates: left to itself the compiler forms every address with v_or
    // into a register of its own and keeps them all alive across the kernel)
    int red_base = C::RED + 4 * l31;
    asm volatile("" : "+v"(red_base));
    const float invn = 1.0f / 512.f;
    int red_base2 = red_base + (MT == 2 ? 128 * kh : 0);        // MT = 2, post-barrier reads: lane (l31, kh) reduces token 32 kh + l31
    asm volatile("" : "+v"(red_base2));
    auto layernorm = [&](f32x16 (&acc)[NT][MT], const float* gam, auto slot, auto shift /* (nt, i4, mt) -> f32x4 */) {
        const char* buf = smem + red_base + decltype(slot)::value * C::REDF * 4;   // two alternating buffers: a barrier separates each write from its reads
        const char* buf2 = smem + red_base2 + decltype(slot)::value * C::REDF * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x2 s2 = f32x2{0.f, 0.f}, q2 = f32x2{0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 v = f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]};
                    s2 += v;
                    q2 = __builtin_elementwise_fma(v, v, q2);
                }
            float s = s2[0] + s2[1], q = q2[0] + q2[1];
            half_swap(s, q);                                        // s = [s.lo | q.lo], q = [s.hi | q.hi]
            *reinterpret_cast<float*>(const_cast<char*>(buf) + (kh * (NW * R) + wave * R + 32 * mt) * 4) = s + q;   // kh = 0: the sum, kh = 1: the sum of squares
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x2 rs[MT], nm[MT];
        {   // MT = 2: the halves share the work - lane (l31, kh) reduces the partials of token 32 kh + l31, two swaps hand the results over
            float p[2][NW];
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int w = 0; w < NW; ++w) p[st][w] = *reinterpret_cast<const float*>(buf2 + (st * (NW * R) + w * R) * 4);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int d = 1; d < NW; d *= 2)
#pragma unroll
                    for (int w = 0; w < NW; w += 2 * d) p[st][w] += p[st][w + d];
            const float mean = p[0][0] * invn;
            const float var = __builtin_fmaxf(p[1][0] * invn - mean * mean, 0.f);
            float r0 = __builtin_amdgcn_rsqf(var + 1e-5f), n0 = -mean * r0;
            if constexpr (MT == 2) {
                float r1 = r0, n1 = n0;
                asm volatile("" : "+v"(r1), "+v"(n1));           // (copies in registers of their own)
                half_swap(r0, r1);
                half_swap(n0, n1);
                rs[MT - 1] = f32x2{r1, r1};
                nm[MT - 1] = f32x2{n1, n1};
            }
            rs[0] = f32x2{r0, r0};
            nm[0] = f32x2{n0, n0};
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 ga[4], sh[4][MT];                                   // the LDS reads of a column block first, then the arithmetic
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                ga[i4] = *reinterpret_cast<const f32x4*>(gam + col4(nt, i4));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) sh[i4][mt] = shift(nt, i4, mt);
            }
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 t = __builtin_elementwise_fma(f32x2{acc[nt][mt][4 * i4 + e], acc[nt][mt][4 * i4 + e + 1]}, rs[mt], nm[mt]);   // (v - mean) rstd
                        const f32x2 o = __builtin_elementwise_fma(t, f32x2{ga[i4][e], ga[i4][e + 1]}, f32x2{sh[i4][mt][e], sh[i4][mt][e + 1]});
                        acc[nt][mt][4 * i4 + e] = o[0];
                        acc[nt][mt][4 * i4 + e + 1] = o[1];
                    }
        }
    };
    auto store_img = [&](const f32x16 (&acc)[NT][MT], char* img) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    op4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (op_t)acc[nt][mt][4 * i4 + e];
                    *reinterpret_cast<op4*>(img + img_off(nt, i4, mt)) = h;
                }
    };

    // =============== stage 1: out_proj + residual + norm1 + folded cross-attention + norm2 -> h' (Y) ====================
    auto wrs = [&](const __bf16* W, int cb0, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W) + (size_t)cb0 * 1024, 0, bytes - cb0 * 2048, 0x00020000);
    };
    const Pass p_wo{wrs(g.Wo, NT * wave, 512 * 512 * 2), 16 * 2048, 0}, p_w1a{wrs(g.W1, NT * wave, 1024 * 512 * 2), 32 * 2048, 0},
        p_w1b{wrs(g.W1, 16 + NT * wave, 1024 * 512 * 2), 32 * 2048, 0}, p_w2a{wrs(g.W2, NT * wave, 512 * 1024 * 2), 16 * 2048, 0},
        p_w2b{p_w2a.rs, 16 * 2048, 32};
    f32x16 acc[NT][MT];
#pragma unroll
    for (int s = 0; s < RD - 1; ++s) load_g(p_wo, s, s);           // right behind the att DMA: the first MFMA needs both, and nothing else
    RGN_M2T(6)
    const int cw = CW * wave + lane;                                 // this lane's column(s) of every vector slice: cw (+ 64)
    // phase A vectors of the wave's columns: requested now, staged to the wave's LDS region after the out_proj loop - nothing
    // before the first MFMA depends on them (the accumulators start from zero; out_proj's bias is added with the residual)
    float va[4][VK], sv[VK], pv[NSAMP][VK];
    int step = 0;
    {
        const float* srcA[4] = {g.bo, g.g1, g.g2, g.b2};
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int k = 0; k < VK; ++k) va[v][k] = srcA[v][cw + 64 * k];
        if (g.stepvec) step = *g.d_step;
        const int s0 = m0 / g.Tq, slast = (g.M - 1) / g.Tq;
#pragma unroll
        for (int k = 0; k < VK; ++k) {
            sv[k] = g.b1[cw + 64 * k];                               // norm1's beta, folded into the per-sample vector
#pragma unroll
            for (int j = 0; j < NSAMP; ++j) {
                const int sidx = s0 + j < slast ? s0 + j : slast;
                pv[j][k] = g.pervec ? g.pervec[(size_t)sidx * g.ldper + cw + 64 * k] : 0.f;
            }
        }
    }
    asm volatile("" ::: "memory");
    // the residual tile straight into registers (needed after the loop: NOT waited for before the first MFMA), then phase B
    op4 hres[NT][MT][4];
    auto load_hres = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int m = m0 + 32 * mt + l31;
            m = m < g.M ? m : g.M - 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
                    hres[nt][mt][i4] = *reinterpret_cast<const op4*>(g.h + ((size_t)(NT * wave + nt) * g.rows + m) * 32 + 8 * i4 + 4 * kh);
        }
    };
#if RGN_M2_HRES == 0
    load_hres();
#endif
    float vb[5][VK];                                                 // phase B, held in registers until the wave is past norm2
#pragma unroll
    for (int k = 0; k < VK; ++k) {
        vb[0][k] = g.bf1[cw + 64 * k];
        vb[1][k] = g.bf1[512 + cw + 64 * k];
        vb[2][k] = g.bf2[cw + 64 * k];
        vb[3][k] = g.g3[cw + 64 * k];
        vb[4][k] = g.b3[cw + 64 * k];
    }
    asm volatile("" ::: "memory");
    constexpr int EXTRA = 16 + 6 * VK;                               // residual + phase B + step vector loads
    // the att image is complete once EVERY wave's DMA pieces have landed: they are the oldest vector-memory operations of this
    // wave, so the count below leaves everything younger than the first weight fragments in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RGN_M2_HRES == 0 ? 16 : 0) + 5 * VK) : "memory");
    RGN_M2T(7)
    __builtin_amdgcn_s_barrier();
#if RGN_M2_HRES == 1
    load_hres();
#endif
    float tv[VK];
#pragma unroll
    for (int k = 0; k < VK; ++k) tv[k] = g.stepvec ? g.stepvec[(size_t)step * g.ldstep + cw + 64 * k] : 0.f;   // (the loop index landed long ago)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;
    RGN_M2T(1)
    gemm32(acc, smem + C::X, p_wo, p_w1a, std::true_type{}, std::integral_constant<int, EXTRA>{});
    RGN_M2T(2)
    // phase A vectors -> the wave's LDS region (wave-private: no barrier, the exchange barrier of norm1 is far behind the writes)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int k = 0; k < VK; ++k) vec[CW * v + 64 * k + lane] = va[v][k];
#pragma unroll
    for (int j = 0; j < NSAMP; ++j)
#pragma unroll
        for (int k = 0; k < VK; ++k) vec[C::A_SPV + CW * j + 64 * k + lane] = (tv[k] + sv[k]) + pv[j][k];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(vec + C::A_BO + col4(nt, i4));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] += (float)hres[nt][mt][i4][e] + b[e];
        }
    {   // norm1 (gamma only) + norm1.beta + call_time[step] + call_cond[sample of the token] (pre-summed per sample in LDS)
        const float* spv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + 32 * mt + l31;
            spv[mt] = vec + C::A_SPV + ((m < g.M ? m : g.M - 1) / g.Tq - m0 / g.Tq) * CW;
        }
        layernorm(acc, vec + C::A_G1, std::integral_constant<int, 0>{}, [&](int nt, int i4, int mt) { return *reinterpret_cast<const f32x4*>(spv[mt] + col4(nt, i4)); });
    }
    layernorm(acc, vec + C::A_G2, std::integral_constant<int, 1>{}, [&](int nt, int i4, int) { return *reinterpret_cast<const f32x4*>(vec + C::A_B2 + col4(nt, i4)); });
    store_img(acc, smem + C::Y);
    // phase B vectors over phase A (wave-private: program order suffices)
#pragma unroll
    for (int v = 0; v < 5; ++v)
#pragma unroll
        for (int k = 0; k < VK; ++k) vec[CW * v + 64 * k + lane] = vb[v][k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // h' image complete
    RGN_M2T(3)

    // =============== stage 2: linear1 + GELU + linear2, the hidden 1024 columns in two halves ==============================
    f32x16 acc2[NT][MT];
    init_bias(acc2, vec + C::B_BF2);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        init_bias(acc, vec + C::B_BF1 + CW * c);
        gemm32(acc, smem + C::Y, c ? p_w1b : p_w1a, c ? p_w2b : p_w2a, std::true_type{}, std::integral_constant<int, 0>{});   // hidden columns [512 c, 512 c + 512)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 gl = m2_gelu2(f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]});
                    acc[nt][mt][i] = gl[0];
                    acc[nt][mt][i + 1] = gl[1];
                }
        if (c == 1) __builtin_amdgcn_s_barrier();                     // every wave is done reading the first half's image
        store_img(acc, smem + C::X);                                  // (c == 0: X still holds the att tile, dead since stage 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c == 0) gemm32(acc2, smem + C::X, p_w2a, p_w1b, std::true_type{}, std::integral_constant<int, 0>{});   // linear2 over hidden k-blocks [16 c, 16 c + 16)
        else gemm32(acc2, smem + C::X, p_w2b, p_w2b, std::false_type{}, std::integral_constant<int, 0>{});
    }
    RGN_M2T(4)

    // =============== stage 3: + residual h' + norm3 -> output planes =====================================================
    {
        op4 r[NT][MT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) r[nt][mt][i4] = *reinterpret_cast<const op4*>(smem + C::Y + img_off(nt, i4, mt));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[nt][mt][4 * i4 + e] += (float)r[nt][mt][i4][e];
    }
    layernorm(acc2, vec + C::B_G3, std::integral_constant<int, 0>{}, [&](int nt, int i4, int) { return *reinterpret_cast<const f32x4*>(vec + C::B_B3 + col4(nt, i4)); });   // (its barrier also fences the last reads of X)
    store_img(acc2, smem + C::X);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.rows * 512 * 2), 0x00020000);
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave * 8 + j, blk = p / (R / 16), r = (p % (R / 16)) * 16 + r16;
            const int m = m0 + r;
            if (m < g.M) {
                const int off = blk * KB + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(smem + C::X + off), o_rs,
                                                       (int)((((size_t)blk * g.rows + m) * 32 + c * 8) * 2), 0, RGN_M2_ST_AUX);
            }
        }
    }
    RGN_M2T(5)
}

#ifdef RGN_M2_STAMPS
void m2_stamps_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m2_st), sizeof(long long) * 1024 * 12); }
#endif

// samples a tile of `rows` rows can touch
bool mlp2_supported(int rows, int d, int ff, int Tq) {
    if (d != 512 || ff != 1024) return false;
    return rows == 32 ? 31 / Tq + 2 <= M2<1>::NSAMP : rows == 64 ? 63 / Tq + 2 <= M2<2>::NSAMP : false;
}
hipError_t configure_mlp2() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp2<1>), hipFuncAttributeMaxDynamicSharedMemorySize, M2<1>::LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp2<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, M2<2>::LDS);
    return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp2<2>), hipFuncAttributeMaxDynamicSharedMemorySize, M2<2>::LDS);
}
hipError_t launch_mlp2(int rows, const MlpArgs& g, hipStream_t s) {
    if (g.f16 && rows != 64) return hipErrorInvalidValue;         // (fp16 operands: the 64-row form only)
    if (rows == 32) hipLaunchKernelGGL(k_mlp2<1>, dim3((g.M + 31) / 32), dim3(M2<1>::NTH), M2<1>::LDS, s, g);
    else if (g.f16) hipLaunchKernelGGL((k_mlp2<2, true>), dim3((g.M + 63) / 64), dim3(M2<2>::NTH), M2<2>::LDS, s, g);
    else hipLaunchKernelGGL(k_mlp2<2>, dim3((g.M + 63) / 64), dim3(M2<2>::NTH), M2<2>::LDS, s, g);
    return hipGetLastError();
}

// ---- the layer tail as the engine sees it: 64-row tiles by default; REGENNET_MLP_ROWS=32 selects the two-workgroups-per-CU form where the
//      sequence allows it (tools, A/B: it loses wherever the 64-row tiles fill the chip)
bool mlp_supported(int d, int ff, int Tq) { return mlp2_supported(64, d, ff, Tq); }
hipError_t configure_mlp() { return configure_mlp2(); }
hipError_t launch_mlp(const MlpArgs& g, hipStream_t s) {
    static const int rows = [] {
        const char* e = getenv("REGENNET_MLP_ROWS");
        return e && atoi(e) == 32 ? 32 : 64;
    }();
    return launch_mlp2(rows == 32 && !g.f16 && mlp2_supported(32, 512, 1024, g.Tq) ? 32 : 64, g, s);
}

}  // namespace rgn
