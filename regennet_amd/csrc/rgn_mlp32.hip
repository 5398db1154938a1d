// The decoder-layer tail of rgn_mlp.hip for tiles of 32 complete token rows, FOUR waves per workgroup and TWO independently
// scheduled workgroups per CU (the structure that paid on k_qkv_attn_rs: one workgroup's VALU epilogues, statistics exchanges and
// tile wait run under the other one's MFMA loops):
//
//   h' = LN2( LN1( att . Wo^T + bo + h ) + call_time[step] + call_cond[sample] )        out_proj, norm1, folded cross-attn, norm2
//   y  = LN3( gelu( h' . W1^T + b1 ) . W2^T + b2 + h' )                                 linear1, GELU, linear2, norm3
//
// (nn.TransformerDecoderLayer post-norm blocks constructed at model/cmdm.py:75-81, called at :227).
//
// Wave w owns output columns [128 w, 128 w + 128) of whichever GEMM is running (four 32x32 accumulator tiles, lane = token,
// registers = columns); a weight fragment feeds ONE MFMA here (two in the 64-row kernel), so the L2 -> register weight stream per
// row doubles - that is the price of the structure, DESIGN.md 4.0b has the measurement.
//   LDS X (32 KiB): att tile image (A operand of out_proj) -> GELU(hidden half) image (A operand of linear2) -> output image
//   LDS Y (32 KiB): h' image (A operand of linear1, residual of norm3)
//   the layer input tile h (residual of norm1) never touches LDS: every lane loads the 64 values it will add straight into
//   registers, behind the att DMA and the first weight fragments, and needs them only after the out_proj loop
//   per-column vectors: every wave stages ITS OWN 128-column slices (wave-private LDS, no barrier): phase A (out_proj bias, norm1 /
//   norm2, the per-sample vectors) is overwritten by phase B (linear1 / linear2 biases, norm3) once the wave is past norm2
//   LayerNorm statistics: sum and sum of squares in ONE exchange (fp32; the plain-bf16 phase only - the split-bf16 tail keeps the
//   two-pass kernels), mean folded into the final FMA
// 78 KiB of LDS per workgroup.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <type_traits>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#ifndef RGN_M3_ST_AUX
#define RGN_M3_ST_AUX 16   // output stores write-through (sc1), as in rgn_mlp.hip
#endif
#ifndef RGN_M3_RD
#define RGN_M3_RD 4        // weight ring depth in half k-steps (4 fragments = 16 registers each)
#endif

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

constexpr int M3_BM = 32, M3_D = 512, M3_NT = 256, M3_RD = RGN_M3_RD;
constexpr int M3_NSAMP = 2;                                    // samples a 32-row tile can touch (Tq >= 32)
// LDS map: X | Y | statistics exchange (2 buffers x [2 stats][4 waves][32 tokens]) | 4 wave-private vector regions of 768 floats
constexpr int M3_X = 0, M3_Y = 32 * 1024, M3_RED = 64 * 1024, M3_VEC = M3_RED + 2 * 1024, M3_VECW = 768, M3_LDS = M3_VEC + 4 * M3_VECW * 4;
static_assert(2 * M3_LDS <= 160 * 1024, "two workgroups per CU");
// wave-private vector region, phase A (stage 1) and phase B (stages 2, 3): offsets in floats
enum { A_BO = 0, A_G1 = 128, A_G2 = 256, A_B2 = 384, A_SPV = 512 /* M3_NSAMP x 128 */ };
enum { B_BF1 = 0 /* 2 x 128: hidden halves */, B_BF2 = 256, B_G3 = 384, B_B3 = 512 };

#ifdef RGN_M3_STAMPS
// tools/mlp_bench -DRGN_M3_STAMPS: wave 0 of EVERY workgroup stamps s_memtime at the phase boundaries, plus where it runs
// (HW_ID: SIMD / CU / SE, XCC_ID), so that the host can put the two workgroups of one CU next to each other
__device__ long long g_m3_st[1024][8];
#define RGN_M3T(i)                                                                                                  \
    {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                                \
            g_m3_st[blockIdx.x][i] = __builtin_readcyclecounter();                                                  \
            if (i == 0) {                                                                                           \
                unsigned hw, xcc;                                                                                   \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                    \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                  \
                g_m3_st[blockIdx.x][6] = hw;                                                                        \
                g_m3_st[blockIdx.x][7] = xcc;                                                                       \
            }                                                                                                       \
        }                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
#else
#define RGN_M3T(i)
#endif

// GELU (erf form), see rgn_mlp.hip
__device__ __forceinline__ f32x2 m3_gelu2(f32x2 x) {
    const f32x2 t = {__builtin_amdgcn_fmed3f(x[0], -4.5254834f, 4.5254834f), __builtin_amdgcn_fmed3f(x[1], -4.5254834f, 4.5254834f)};
    const f32x2 z = t * t;
    f32x2 p = f32x2{-7.433422766e-10f, -7.433422766e-10f};
    p = __builtin_elementwise_fma(p, z, f32x2{6.994829249e-08f, 6.994829249e-08f});
    p = __builtin_elementwise_fma(p, z, f32x2{-2.824688409e-06f, -2.824688409e-06f});
    p = __builtin_elementwise_fma(p, z, f32x2{6.471458619e-05f, 6.471458619e-05f});
    p = __builtin_elementwise_fma(p, z, f32x2{-9.421016439e-04f, -9.421016439e-04f});
    p = __builtin_elementwise_fma(p, z, f32x2{9.306023829e-03f, 9.306023829e-03f});
    p = __builtin_elementwise_fma(p, z, f32x2{-6.564749777e-02f, -6.564749777e-02f});
    p = __builtin_elementwise_fma(p, z, f32x2{3.986273110e-01f, 3.986273110e-01f});
    return x * __builtin_elementwise_fma(t, p, f32x2{0.5f, 0.5f});
}

__global__ __launch_bounds__(M3_NT, 2) void k_mlp32(MlpArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = xcd_affine(blockIdx.x, gridDim.x) * M3_BM;
    float* red = reinterpret_cast<float*>(smem + M3_RED);
    float* vec = reinterpret_cast<float*>(smem + M3_VEC) + wave * M3_VECW;   // this wave's private region
    RGN_M3T(0)
    // ---- att tile -> X by DMA: 16 k-blocks x 2 pieces of 1 KiB (16 rows x 64 B), wave w issues the pieces w, w + 4, ...
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave + 4 * j, kb = p >> 1, r = (p & 1) * 16 + r16;
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            const size_t src = ((size_t)kb * g.rows + m) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.att + src), (RGN_AS3 void*)(smem + M3_X + p * 1024), 16, 0, 0);
        }
    }
    // B-operand fragment of token l31 inside a k-block image [32 rows][64 B] (16-byte chunks swizzled by the row), per 16-wide k-half
    int a_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[ks] = l31 * 64 + (((2 * ks + kh) ^ ((l31 >> 2) & 3)) << 4);

    // ---- weight ring: granule = half a k-step (16 k) of this wave's four column blocks; W: fragment-ordered plane
    //      [K/32][nb_all][2][64][8] (rgn_rowgemm.hip). Granule index hs = 2 kt + ks. Buffer loads: the resource and the granule's
    //      byte offset are scalar, the lane contributes lane * 16 - no 64-bit vector addresses anywhere in the stream
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    struct Pass { __amdgpu_buffer_rsrc_t rs; int kstride, hs0; };   // rs: based at the wave's first column block in k-block 0 (every offset below is a compile-time constant); kstride = nb_all * 2048
    bf16x8 wf[M3_RD][4];
    const int lane16 = lane * 16;
    auto load_g = [&](const Pass& ps, int hs_rel, int slot) {
        const int hs = ps.hs0 + hs_rel;
        const int soff = (hs >> 1) * ps.kstride + (hs & 1) * 1024;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            wf[slot][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ps.rs, lane16, soff + nt * 2048, 0));
    };
    auto gemm_prefetch = [&](const Pass& ps) {
#pragma unroll
        for (int s = 0; s < M3_RD - 1; ++s) load_g(ps, s, s);
    };
    // one GEMM pass over K = 512: acc[nt] += A_image(16 k-blocks at img) . W[column blocks cb0 + nt, granules hs0 .. hs0 + 31]^T.
    // The ring never drains between passes: the tail of a pass requests the first RD - 1 granules of the NEXT pass (nxt.W != null;
    // 32 % RD == 0, so ring slots continue seamlessly). `extra`: vector-memory operations issued between the granules RD - 2 and
    // RD - 1 of this pass that may stay in flight (stage 1: the residual tile and the phase-B vectors).
    auto gemm32 = [&](f32x16 (&acc)[4], const char* img, const Pass& cur, const Pass& nxt, auto chain, auto extra) {
        constexpr int EX = decltype(extra)::value;
        constexpr bool CH = decltype(chain)::value;
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 af = *reinterpret_cast<const bf16x8*>(img + a_off[0]);
#pragma unroll
        for (int hs = 0; hs < 32; ++hs) {
            bf16x8 afn = af;
            if (hs + 1 < 32) afn = *reinterpret_cast<const bf16x8*>(img + ((hs + 1) >> 1) * 2048 + a_off[(hs + 1) & 1]);   // one granule ahead
            constexpr int AH = M3_RD - 1;
            if (hs + AH < 32) load_g(cur, hs + AH, (hs + AH) % M3_RD);
            else if (CH) load_g(nxt, hs + AH - 32, (hs + AH) % M3_RD);
            if (hs + AH < 32 || CH) {
                if (hs < AH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * AH + EX) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * AH) : "memory");   // this granule is in; the next RD - 1 stay in flight
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[hs % M3_RD][nt], af, acc[nt], 0, 0, 0);
            af = afn;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // element (token l31, column 128 wave + 32 nt + 8 i4 + 4 kh + e) <-> register acc[nt][4 i4 + e]
    auto col4 = [&](int nt, int i4) { return 32 * nt + 8 * i4 + 4 * kh; };          // inside the wave's 128-column slice
    auto img_off = [&](int nt, int i4) {                                              // its 8-byte run inside an image [16][32 rows][64 B]
        return (4 * wave + nt) * 2048 + l31 * 64 + ((i4 ^ ((l31 >> 2) & 3)) << 4) + 8 * kh;
    };
    auto init_bias = [&](f32x16 (&acc)[4], const float* bias) {                       // bias: the wave's 128-column slice in LDS
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + col4(nt, i4));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nt][4 * i4 + e] = b[e];
            }
    };
    // LayerNorm over the 512 columns of every token, in place: o = v (rstd gamma) + (shift - mean rstd gamma). One exchange of
    // (sum, sum of squares): halves by lane ^ 32, the four column slices through LDS ([stat][wave][token], conflict-free both ways)
    int red_slot = 0;
    const float invn = 1.0f / (float)M3_D;
    auto layernorm = [&](f32x16 (&acc)[4], const float* gam, auto shift /* (nt, i4) -> f32x4 */) {
        f32x2 s2 = f32x2{0.f, 0.f}, q2 = f32x2{0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 v = f32x2{acc[nt][i], acc[nt][i + 1]};
                s2 += v;
                q2 = __builtin_elementwise_fma(v, v, q2);
            }
        float s = s2[0] + s2[1], q = q2[0] + q2[1];
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        float* buf = red + (red_slot & 1) * 256;   // two alternating buffers: a barrier separates each write from its reads
        ++red_slot;
        buf[kh * 128 + wave * 32 + l31] = kh ? q : s;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float st = (buf[l31] + buf[32 + l31]) + (buf[64 + l31] + buf[96 + l31]);
        const float qt = (buf[128 + l31] + buf[160 + l31]) + (buf[192 + l31] + buf[224 + l31]);
        const float mean = st * invn;
        const float var = __builtin_fmaxf(qt * invn - mean * mean, 0.f);
        const float rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
        const f32x2 rs = f32x2{rstd, rstd}, nm = f32x2{-mean, -mean};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 ga[4], sh[4];                                       // the LDS reads of a column block first, then the arithmetic
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                ga[i4] = *reinterpret_cast<const f32x4*>(gam + col4(nt, i4));
                sh[i4] = shift(nt, i4);
            }
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const f32x2 rg = f32x2{ga[i4][e], ga[i4][e + 1]} * rs;
                    const f32x2 b = __builtin_elementwise_fma(nm, rg, f32x2{sh[i4][e], sh[i4][e + 1]});
                    const f32x2 o = __builtin_elementwise_fma(f32x2{acc[nt][4 * i4 + e], acc[nt][4 * i4 + e + 1]}, rg, b);
                    acc[nt][4 * i4 + e] = o[0];
                    acc[nt][4 * i4 + e + 1] = o[1];
                }
        }
    };
    auto store_img = [&](const f32x16 (&acc)[4], char* img) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (__bf16)acc[nt][4 * i4 + e];
                *reinterpret_cast<bf16x4*>(img + img_off(nt, i4)) = h;
            }
    };

    // =============== stage 1: out_proj + residual + norm1 + folded cross-attention + norm2 -> h' (Y) ====================
    static_assert(32 % M3_RD == 0, "ring slots continue across passes");
    auto wrs = [&](const __bf16* W, int cb0, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W) + (size_t)cb0 * 1024, 0, bytes - cb0 * 2048, 0x00020000);
    };
    const Pass p_wo{wrs(g.Wo, 4 * wave, 512 * 512 * 2), 16 * 2048, 0}, p_w1a{wrs(g.W1, 4 * wave, 1024 * 512 * 2), 32 * 2048, 0},
        p_w1b{wrs(g.W1, 16 + 4 * wave, 1024 * 512 * 2), 32 * 2048, 0}, p_w2a{wrs(g.W2, 4 * wave, 512 * 1024 * 2), 16 * 2048, 0},
        p_w2b{p_w2a.rs, 16 * 2048, 32};
    f32x16 acc[4];
    gemm_prefetch(p_wo);                          // right behind the att DMA: the first MFMA needs both, and nothing else
    const int cw = 128 * wave + lane;                                // this lane's two columns of every vector slice: cw, cw + 64
    // phase A vectors of the wave's 128 columns (staged to LDS below)
    float va[8], sv[2], pv[M3_NSAMP][2];
    {
        const float* srcA[4] = {g.bo, g.g1, g.g2, g.b2};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            va[2 * v] = srcA[v][cw];
            va[2 * v + 1] = srcA[v][cw + 64];
        }
        const int step = g.stepvec ? *g.d_step : 0;
        const int s0 = m0 / g.Tq, slast = (g.M - 1) / g.Tq;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            sv[k] = (g.stepvec ? g.stepvec[(size_t)step * g.ldstep + cw + 64 * k] : 0.f) + g.b1[cw + 64 * k];   // norm1's beta folded in
#pragma unroll
            for (int j = 0; j < M3_NSAMP; ++j) {
                const int sidx = s0 + j < slast ? s0 + j : slast;
                pv[j][k] = g.pervec ? g.pervec[(size_t)sidx * g.ldper + cw + 64 * k] : 0.f;
            }
        }
    }
    asm volatile("" ::: "memory");
    // the residual tile straight into registers (needed after the loop: NOT waited for before the first MFMA), then phase B
    bf16x4 hres[4][4];
    {
        int m = m0 + l31;
        m = m < g.M ? m : g.M - 1;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                hres[nt][i4] = *reinterpret_cast<const bf16x4*>(g.h + ((size_t)(4 * wave + nt) * g.rows + m) * 32 + 8 * i4 + 4 * kh);
    }
    float vb[10];                                                    // phase B, held in registers until the wave is past norm2
    vb[0] = g.bf1[cw]; vb[1] = g.bf1[cw + 64]; vb[2] = g.bf1[512 + cw]; vb[3] = g.bf1[512 + cw + 64];
    vb[4] = g.bf2[cw]; vb[5] = g.bf2[cw + 64]; vb[6] = g.g3[cw]; vb[7] = g.g3[cw + 64]; vb[8] = g.b3[cw]; vb[9] = g.b3[cw + 64];
    asm volatile("" ::: "memory");
    constexpr int M3_EXTRA = 16 + 10;                                // residual + phase B loads
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        vec[128 * v + lane] = va[2 * v];
        vec[128 * v + 64 + lane] = va[2 * v + 1];
    }
#pragma unroll
    for (int j = 0; j < M3_NSAMP; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) vec[A_SPV + 128 * j + 64 * k + lane] = sv[k] + pv[j][k];
    // the att image is complete once EVERY wave's DMA pieces have landed: they are the oldest vector-memory operations of this
    // wave, so the count below leaves the residual and the phase-B vectors in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(M3_EXTRA) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    init_bias(acc, vec + A_BO);
    RGN_M3T(1)
    gemm32(acc, smem + M3_X, p_wo, p_w1a, std::true_type{}, std::integral_constant<int, M3_EXTRA>{});
    RGN_M3T(2)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][4 * i4 + e] += (float)hres[nt][i4][e];
    {   // norm1 (gamma only) + norm1.beta + call_time[step] + call_cond[sample of the token] (pre-summed per sample in LDS)
        const int m = m0 + l31;
        const float* spv = vec + A_SPV + ((m < g.M ? m : g.M - 1) / g.Tq - m0 / g.Tq) * 128;
        layernorm(acc, vec + A_G1, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(spv + col4(nt, i4)); });
    }
    layernorm(acc, vec + A_G2, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(vec + A_B2 + col4(nt, i4)); });
    store_img(acc, smem + M3_Y);
    // phase B vectors over phase A (wave-private: program order suffices)
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        vec[128 * v + lane] = vb[2 * v];
        vec[128 * v + 64 + lane] = vb[2 * v + 1];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // h' image complete
    RGN_M3T(3)

    // =============== stage 2: linear1 + GELU + linear2, the hidden 1024 columns in two halves ==============================
    f32x16 acc2[4];
    init_bias(acc2, vec + B_BF2);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        init_bias(acc, vec + B_BF1 + 128 * c);
        gemm32(acc, smem + M3_Y, c ? p_w1b : p_w1a, c ? p_w2b : p_w2a, std::true_type{}, std::integral_constant<int, 0>{});   // hidden columns [512 c, 512 c + 512)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 gl = m3_gelu2(f32x2{acc[nt][i], acc[nt][i + 1]});
                acc[nt][i] = gl[0];
                acc[nt][i + 1] = gl[1];
            }
        if (c == 1) __builtin_amdgcn_s_barrier();                     // every wave is done reading the first half's image
        store_img(acc, smem + M3_X);                                  // (c == 0: X still holds the att tile, dead since stage 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c == 0) gemm32(acc2, smem + M3_X, p_w2a, p_w1b, std::true_type{}, std::integral_constant<int, 0>{});   // linear2 over hidden k-blocks [16 c, 16 c + 16)
        else gemm32(acc2, smem + M3_X, p_w2b, p_w2b, std::false_type{}, std::integral_constant<int, 0>{});
    }
    RGN_M3T(4)

    // =============== stage 3: + residual h' + norm3 -> output planes =====================================================
    {
        bf16x4 r[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) r[nt][i4] = *reinterpret_cast<const bf16x4*>(smem + M3_Y + img_off(nt, i4));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc2[nt][4 * i4 + e] += (float)r[nt][i4][e];
    }
    layernorm(acc2, vec + B_G3, [&](int nt, int i4) { return *reinterpret_cast<const f32x4*>(vec + B_B3 + col4(nt, i4)); });   // (its barrier also fences the last reads of X)
    store_img(acc2, smem + M3_X);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.rows * 512 * 2), 0x00020000);
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave * 8 + j, blk = p >> 1, r = (p & 1) * 16 + r16;
            const int m = m0 + r;
            if (m < g.M) {
                const int off = blk * 2048 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(smem + M3_X + off), o_rs,
                                                       (int)((((size_t)blk * g.rows + m) * 32 + c * 8) * 2), 0, RGN_M3_ST_AUX);
            }
        }
    }
    RGN_M3T(5)
}

#ifdef RGN_M3_STAMPS
void m3_stamps_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_m3_st), sizeof(long long) * 1024 * 8); }
#endif

bool mlp32_supported(int d, int ff, int Tq) { return d == M3_D && ff == 2 * M3_D && 31 / Tq + 2 <= M3_NSAMP; }
hipError_t configure_mlp32() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp32), hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS);
}
hipError_t launch_mlp32(const MlpArgs& g, hipStream_t s) {
    hipLaunchKernelGGL(k_mlp32, dim3((g.M + M3_BM - 1) / M3_BM), dim3(M3_NT), M3_LDS, s, g);
    return hipGetLastError();
}

}  // namespace rgn
