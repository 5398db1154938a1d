// Row-persistent decoder-layer tail for the plain-bf16 phase: for a tile of 64 complete token rows ONE workgroup runs
//
//   h' = LN2( LN1( att . Wo^T + bo + h ) + call_time[step] + call_cond[sample] )        out_proj, norm1, folded cross-attn, norm2
//   y  = LN3( gelu( h' . W1^T + b1 ) . W2^T + b2 + h' )                                 linear1, GELU, linear2, norm3
//
// (nn.TransformerDecoderLayer post-norm blocks constructed at model/cmdm.py:75-81, called at :227) with every intermediate
// resident on chip: the [M, 1024] hidden tensor and the h' round trip never reach memory, the LayerNorms run on the accumulators.
//
// Structure (8 waves, accumulators transposed: lane = token, registers = columns):
//   LDS Y (64 KiB): h tile image (residual of norm1: consumed when the accumulators are initialised) -> h' image (A operand of
//                   linear1, residual of norm3: consumed into linear2's accumulator while linear1 starts)
//   LDS X (64 KiB): att tile image (A operand of out_proj) -> two 32 KiB images of GELU(hidden QUARTER) (A operands of linear2)
//                   -> output image
//   stage 1  out_proj (K = 512; wave w = columns [64 w, 64 w + 64)) from X on accumulators that start at bo + h; LN1; LN2 -> Y
//   stage 2  the hidden 1024 columns in four quarters q: linear1(q) (wave w = hidden columns 256 q + [32 w, 32 w + 32), both
//            token tiles: a 32-register accumulator, two of them alternating), GELU(q) -> image q & 1, linear2 accumulates the
//            quarter's 8 k-blocks into ONE 64 x 512 accumulator. The passes are software-pipelined INSIDE every wave: GELU(q)
//            is issued between the MFMAs of linear1(q + 1) (GELU(3): of linear2(2)), the residual + bias initialisation of
//            linear2's accumulator between those of linear1(0) - the VALU work of the FFN runs in the shadow of the matrix
//            pipe instead of in phases of its own (round 2: 13 k of 46 k cycles of this stage had the matrix cores idle).
//   stage 3  LN3 on linear2's accumulator -> bf16 image -> contiguous 1 KiB wave-stores into the residual planes
// ONE weight stream for the whole kernel: 80 items of four 1 KiB fragments per wave (16 out_proj k-steps, then per FFN pass 8
// items = 8 MFMAs each), fragment-ordered planes (rgn_rowgemm.hip) -> a 4-slot register ring, always three items ahead, across
// pass boundaries, barriers and the LayerNorm phase (weights depend on nothing).
// LayerNorm statistics: every lane reduces its 32 columns to (mean, M2) in registers, the 16 partials of a token are merged
// Chan-style (equal counts) through one lane^32 exchange and ONE 4 KiB LDS exchange per LayerNorm - two-pass accuracy with one
// barrier instead of two.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) (the 64-item FFN pipeline is too
// large for `#pragma unroll` to be honoured; its register arrays must be indexed by constants)
template <int... Is, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

constexpr int ML_BM = 64, ML_D = 512, ML_NT = 512;
// LDS map: X | Y | statistics exchange (2 x 4 KiB) | per-column vectors g1 g2 b2 bf2 g3 b3 (6 x 512) + bf1 (1024) floats |
// sv + pv of the (at most ML_NSAMP) samples a 64-row tile touches (Tq >= 22)  = 160 KiB
constexpr int ML_NSAMP = 4;
constexpr int ML_X = 0, ML_Y = 64 * 1024, ML_RED = 128 * 1024, ML_VEC = ML_RED + 2 * 4096, ML_SPV = ML_VEC + (6 * 512 + 1024) * 4,
              ML_LDS = ML_SPV + ML_NSAMP * 512 * 4;
static_assert(ML_LDS <= 160 * 1024, "LDS map");
enum { V_G1 = 0, V_G2 = 512, V_B2 = 1024, V_BF2 = 1536, V_G3 = 2048, V_B3 = 2560, V_BF1 = 3072 };
// the FFN passes: linear1 (1) / linear2 (2) of which quarter, which quarter's GELU rides along, barrier behind the pass
constexpr int ML_NITEM = 16 + 64;
constexpr int ML_PT[8] = {1, 1, 2, 1, 2, 1, 2, 2};
constexpr int ML_PQ[8] = {0, 1, 0, 2, 1, 3, 2, 3};
constexpr int ML_PG[8] = {-1, 0, -1, 1, -1, 2, 3, -1};

#ifdef RGN_ML_PROF
__device__ long long g_ml_prof[32];
#define RGN_MT(i) if (blockIdx.x == RGN_ML_PROF && threadIdx.x == 0) g_ml_prof[i] = __builtin_readcyclecounter();
#else
#define RGN_MT(i)
#endif

// GELU (erf form): x (0.5 + 0.5 erf(x / sqrt 2)) with 0.5 erf(x / sqrt 2) = t Q(t^2), t = clamp(x, +-3.2 sqrt 2): rgn_rowgemm.hip's
// odd degree-15 polynomial of erf (max abs error 1.6e-4) with the 1/sqrt 2, the 1/2^k of u^2 = x^2 / 2 and the 0.5 folded into
// the coefficients: 12 instructions per pair of values
__device__ __forceinline__ f32x2 ml_gelu2(f32x2 x) {
    const f32x2 t = {__builtin_amdgcn_fmed3f(x[0], -4.5254834f, 4.5254834f), __builtin_amdgcn_fmed3f(x[1], -4.5254834f, 4.5254834f)};   // (no canonicalising v_max in front, unlike min(max()))
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

__global__ __launch_bounds__(ML_NT, 2) void k_mlp(MlpArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = xcd_affine(blockIdx.x, gridDim.x) * ML_BM;
    float* red = reinterpret_cast<float*>(smem + ML_RED);
    float* vec = reinterpret_cast<float*>(smem + ML_VEC);
    float* spv = reinterpret_cast<float*>(smem + ML_SPV);
    RGN_MT(0)
    auto col4 = [&](int nt, int i4) { return 64 * wave + 32 * nt + 8 * i4 + 4 * kh; };
    // ---- prologue, in the order the counted waits rely on (vmcnt retires in order):
    //   (1) h tile -> Y by DMA   (2) per-column vectors + out_proj's bias -> registers   (3) att tile -> X by DMA   (4) weight items 0-2
    // then the vectors go to LDS (their loads are in => so is Y), the accumulators are initialised from Y + bo while X lands.
    // 16 k-blocks x 4 pieces of 1 KiB per image, wave w issues the pieces p = w, w + 8, ... (coalesced 1 KiB runs of the planes)
    auto tile_dma = [&](const __bf16* src_plane, int dst) {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave + 8 * j, kb = p >> 2, r = (p & 3) * 16 + r16;
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            const size_t src = ((size_t)kb * g.rows + m) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(src_plane + src), (RGN_AS3 void*)(smem + dst + p * 1024), 16, 0, 0);
        }
    };
    // the loop index of the sampling step: a scalar load up front (as a vector load its consumer would wait for vmcnt(0), i.e. for
    // every tile piece issued before it)
    int step = 0;
    if (g.stepvec) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(step) : "s"(g.d_step) : "memory");
    tile_dma(g.h, ML_Y);
    asm volatile("" ::: "memory");
    const int s0 = m0 / g.Tq, slast = (g.M - 1) / g.Tq;
    f32x4 bo4[2][4];                                                  // out_proj's bias stays in registers (used once)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) bo4[nt][i4] = *reinterpret_cast<const f32x4*>(g.bo + col4(nt, i4));
    float vv[8], pv[ML_NSAMP], sv, b1v;                               // (no arithmetic on these before the att tile is requested)
    {
        const float* src[6] = {g.g1, g.g2, g.b2, g.bf2, g.g3, g.b3};
#pragma unroll
        for (int v = 0; v < 6; ++v) vv[v] = src[v][tid];
        vv[6] = g.bf1[tid];
        vv[7] = g.bf1[512 + tid];
        sv = g.stepvec ? g.stepvec[(size_t)step * g.ldstep + tid] : 0.f;
        b1v = g.b1[tid];
#pragma unroll
        for (int j = 0; j < ML_NSAMP; ++j) {
            const int sidx = s0 + j < slast ? s0 + j : slast;
            pv[j] = g.pervec ? g.pervec[(size_t)sidx * g.ldper + tid] : 0.f;
        }
    }
    asm volatile("" ::: "memory");
    tile_dma(g.att, ML_X);
    // B-operand fragment of token 32 mt + l31 inside a k-block: row * 64 B + the 16-byte chunk (2 ks + kh) ^ ((row >> 2) & 3). The
    // swizzle term does not depend on mt, so ONE register per ks serves both token tiles (mt * 2048 rides in the instruction's
    // immediate offset) - and one more pair for Y: ds_read offsets are 16 bits, the second 64 KiB needs its own base.
    int a_off[2], a_off_y[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_off[ks] = l31 * 64 + (((2 * ks + kh) ^ ((l31 >> 2) & 3)) << 4);
        a_off_y[ks] = a_off[ks] + ML_Y;
    }
    // ---- the weight stream: item j -> ring slot j & 3 (four fragments = 16 registers), requested three items ahead -------------
    //   j < 16              out_proj k-step j:             Wo  [16 k][16 cb], column blocks 2 w + nt          -> wf[slot][nt][ks]
    //   linear1 (q, i)      k-steps 2 i, 2 i + 1 of W1 [16 k][32 cb], column block 8 q + w                    -> wf[slot][kl][ks]
    //   linear2 (q, i)      hidden k-block 8 q + i of W2 [32 k][16 cb], column blocks 2 w + nt                -> wf[slot][nt][ks]
    bf16x8 wf[4][2][2];
    // buffer loads: descriptor (base advanced to the wave's column blocks) + compile-time byte offset in SGPRs, ONE address VGPR per
    // lane for the whole stream (with flat addresses the compiler keeps a 64-bit address pair per unrolled item and spills)
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Wo) + (size_t)(2 * wave) * 1024, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.W1) + (size_t)wave * 1024, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.W2) + (size_t)(2 * wave) * 1024, 0, -1, 0x00020000);
    const int lane16 = lane * 16;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load_item = [&](int j) {
        const int slot = j & 3;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (j < 16) {                                                                                            // a = nt
                    wf[slot][a][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_o, lane16, ((j * 16 + a) * 1024 + ks * 512) * 2, 0));
                } else {
                    const int p = (j - 16) >> 3, i = (j - 16) & 7, q = ML_PQ[p];
                    if (ML_PT[p] == 1)                                                                                   // a = kl
                        wf[slot][a][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_1, lane16, (((2 * i + a) * 32 + 8 * q) * 1024 + ks * 512) * 2, 0));
                    else                                                                                                 // a = nt
                        wf[slot][a][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_2, lane16, (((8 * q + i) * 16 + a) * 1024 + ks * 512) * 2, 0));
                }
            }
    };
    // item j is about to run: request item j + 3, then wait until item j's fragments are in (the later ones stay in flight)
    auto advance = [&](int j) {
        if (j + 3 < ML_NITEM) {
            load_item(j + 3);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else if (j + 3 == ML_NITEM) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (j + 2 == ML_NITEM) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // element (token 32 mt + l31, column 64 wave + 32 nt + 8 i4 + 4 kh + e) <-> register acc[nt][mt][4 i4 + e]; its 8-byte
    // run inside a [16 column blocks][64 rows][64 B] swizzled image:
    int i_off[4];                                                     // per i4; nt * 4096 + mt * 2048 ride in the immediate offset
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) i_off[i4] = 2 * wave * 4096 + l31 * 64 + ((i4 ^ ((l31 >> 2) & 3)) << 4) + 8 * kh;
    auto img_off = [&](int nt, int i4, int mt) { return i_off[i4] + nt * 4096 + mt * 2048; };

    // ---- LayerNorm over the 512 columns of the tile's tokens, in place on a 64 x 64 accumulator set -----------------------------
    // per lane and token tile mt: (mean, M2) of its 32 columns (two-pass, in registers; the values stay centred on the LOCAL mean),
    // merged with the lane^32 partner, then across the 8 waves through LDS: ONE barrier per LayerNorm.
    //   out = ((xc + (m_local - mean)) rstd) gamma + beta,   beta: none | per-column vector | per-(sample, column) vector
    int red_slot = 0;
    auto layernorm = [&](f32x16 (&acc)[2][2], const float* gam, const float* bet, const int* bsel, auto beta_kind) {
        constexpr int BK = decltype(beta_kind)::value;                // 0: no beta, 1: bet[col], 2: bet[bsel[mt] + col]
        f32x2 s2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; i += 2) s2[mt] += f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]};
        float lm[2], M2[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) lm[mt] = (s2[mt][0] + s2[mt][1]) * (1.0f / 32.0f);
        f32x2 q2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x2 nm = f32x2{-lm[mt], -lm[mt]};
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 dlt = f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]} + nm;
                    acc[nt][mt][i] = dlt[0];
                    acc[nt][mt][i + 1] = dlt[1];
                    q2[mt] = __builtin_elementwise_fma(dlt, dlt, q2[mt]);
                }
            }
        float* buf = red + (red_slot & 1) * 1024;                      // two alternating buffers: a barrier separates each write from its reads
        ++red_slot;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            M2[mt] = q2[mt][0] + q2[mt][1];
            const float pm = __shfl_xor(lm[mt], 32, 64), pq = __shfl_xor(M2[mt], 32, 64);
            const float dm = pm - lm[mt];
            if (kh == 0)                                               // (mean, M2) of the token's 64 columns in this wave
                *reinterpret_cast<f32x2*>(buf + ((32 * mt + l31) * 8 + wave) * 2) = f32x2{lm[mt] + 0.5f * dm, M2[mt] + pq + 16.0f * dm * dm};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x2 ab[2];                                                   // per token tile: out = xc * ab[0] + ab[1] before gamma / beta
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(buf + (32 * mt + l31) * 16 + 4 * k);
            const float mean = (((v[0][0] + v[0][2]) + (v[1][0] + v[1][2])) + ((v[2][0] + v[2][2]) + (v[3][0] + v[3][2]))) * 0.125f;
            float m2 = ((v[0][1] + v[0][3]) + (v[1][1] + v[1][3])) + ((v[2][1] + v[2][3]) + (v[3][1] + v[3][3]));
            float dev = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d0 = v[k][0] - mean, d1 = v[k][2] - mean;
                dev = __builtin_fmaf(d0, d0, dev);
                dev = __builtin_fmaf(d1, d1, dev);
            }
            m2 = __builtin_fmaf(64.0f, dev, m2);
            const float rstd = __builtin_amdgcn_rsqf(m2 * (1.0f / (float)ML_D) + 1e-5f);
            ab[mt] = f32x2{rstd, (lm[mt] - mean) * rstd};
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + col4(nt, i4));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 be;
                    if constexpr (BK == 1) be = *reinterpret_cast<const f32x4*>(bet + col4(nt, i4));
                    if constexpr (BK == 2) be = *reinterpret_cast<const f32x4*>(bet + bsel[mt] + col4(nt, i4));
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 v = f32x2{acc[nt][mt][4 * i4 + e], acc[nt][mt][4 * i4 + e + 1]};
                        const f32x2 n = __builtin_elementwise_fma(v, f32x2{ab[mt][0], ab[mt][0]}, f32x2{ab[mt][1], ab[mt][1]});
                        f32x2 o;
                        if constexpr (BK == 0) o = n * f32x2{ga[e], ga[e + 1]};
                        else o = __builtin_elementwise_fma(n, f32x2{ga[e], ga[e + 1]}, f32x2{be[e], be[e + 1]});
                        acc[nt][mt][4 * i4 + e] = o[0];
                        acc[nt][mt][4 * i4 + e + 1] = o[1];
                    }
                }
            }
    };
    auto store_img = [&](const f32x16 (&acc)[2][2], char* img) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    bf16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (__bf16)acc[nt][mt][4 * i4 + e];
                    *reinterpret_cast<bf16x4*>(img + img_off(nt, i4, mt)) = h;
                }
    };
    // acc[nt][mt][4 i4 ..] = bias + bf16 image value (the residual), for one (nt, i4, mt) group
    auto init_group = [&](f32x16 (&acc)[2][2], const f32x4 b, const char* img, int nt, int i4, int mt) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(img + img_off(nt, i4, mt));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] = b[e] + (float)r[e];
    };

    // =============== stage 1: out_proj + residual + norm1 + folded cross-attention + norm2 -> h' (Y) ====================
    f32x16 acc[2][2];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 3; ++j) load_item(j);
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");                // in order: the h tile and the vectors are in; att and the weights may still fly
#pragma unroll
    for (int v = 0; v < 6; ++v) vec[v * 512 + tid] = vv[v];
    vec[V_BF1 + tid] = vv[6];
    vec[V_BF1 + 512 + tid] = vv[7];
#pragma unroll
    for (int j = 0; j < ML_NSAMP; ++j) spv[j * 512 + tid] = sv + b1v + pv[j];   // norm1's beta folded in
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) init_group(acc, bo4[nt][i4], smem + ML_Y, nt, i4, mt);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                // the att tile landed
    __builtin_amdgcn_s_barrier();
    RGN_MT(1)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const char* sb = smem + ML_X + j * 4096;
        bf16x8 af[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[ks][mt] = *reinterpret_cast<const bf16x8*>(sb + a_off[ks] + mt * 2048);
        asm volatile("" ::: "memory");
        advance(j);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j & 3][nt][ks], af[ks][mt], acc[nt][mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    RGN_MT(2)
    int sj[2];                                                        // the token's sample inside the tile -> its row of spv
    {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + 32 * mt + l31;
            int rel = 0;
#pragma unroll
            for (int k = 1; k < ML_NSAMP; ++k) rel += (m >= (s0 + k) * g.Tq) ? 1 : 0;
            sj[mt] = (rel < slast - s0 ? rel : slast - s0) * 512;     // (rows past M repeat the last sample's vector)
        }
    }
    RGN_MT(14)
    layernorm(acc, vec + V_G1, spv, sj, std::integral_constant<int, 2>{});       // + norm1.beta + call_time[step] + call_cond[sample] (pre-summed per sample)
    RGN_MT(15)
    layernorm(acc, vec + V_G2, vec + V_B2, nullptr, std::integral_constant<int, 1>{});
    RGN_MT(16)
    store_img(acc, smem + ML_Y);                                      // h' replaces h (consumed when acc was initialised)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RGN_MT(3)

    // =============== stage 2: linear1 + GELU + linear2, the hidden 1024 columns in four software-pipelined quarters ==============
    f32x16 acc1[2][2];                                                // [quarter & 1][mt]: hidden columns 256 q + 32 w + (8 i4 + 4 kh + e)
    f32x16 acc2[2][2];                                                // [nt][mt]: linear2, all four quarters
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // GELU image addresses: element (token 32 mt + l31, hidden column 32 w + 8 i4 + 4 kh + e of the quarter) -> k-block w of the
    // quarter's image, 8-byte run = i_off[i4] - wave * 4096 (one scalar subtract per store instead of four more registers)
    const int wave4k = wave * 4096;
    // GELU of one (mt, i4) group (four values of this lane) of quarter qg: + bias, GELU, bf16, 8-byte store into image qg & 1 -
    // cut into 8 chunks of <= 4 VALU instructions (the two pairs' dependent chains alternate), one chunk behind each of an item's
    // 8 MFMAs: in an in-order wave the matrix pipe is only fed while the NEXT MFMA can issue, so the VALU work has to sit between
    // the MFMAs (the compiler's own schedule clustered it behind them; sched_group_barrier pipelines pushed the bias read last)
    struct GeluState { f32x2 x, t, z, p; unsigned h0; };
    auto gelu_chunk = [&](GeluState& gs, const f32x4& gb, auto QG, auto GI, auto C) __attribute__((always_inline)) {
        constexpr int qg = decltype(QG)::value, gi = decltype(GI)::value, c = decltype(C)::value;
        constexpr int mt = gi >> 2, i4 = gi & 3, bb = qg & 1, pr = c >> 2, cc = c & 3;   // chunks 0-3: values 0, 1; chunks 4-7: values 2, 3
        auto lvl = [&](float k) __attribute__((always_inline)) { gs.p = __builtin_elementwise_fma(gs.p, gs.z, f32x2{k, k}); };
        if constexpr (cc == 0) {
            gs.x = f32x2{acc1[bb][mt][4 * i4 + 2 * pr], acc1[bb][mt][4 * i4 + 2 * pr + 1]} + f32x2{gb[2 * pr], gb[2 * pr + 1]};
            gs.t = f32x2{__builtin_amdgcn_fmed3f(gs.x[0], -4.5254834f, 4.5254834f), __builtin_amdgcn_fmed3f(gs.x[1], -4.5254834f, 4.5254834f)};
            gs.z = gs.t * gs.t;
        } else if constexpr (cc == 1) {
            gs.p = __builtin_elementwise_fma(f32x2{-7.433422766e-10f, -7.433422766e-10f}, gs.z, f32x2{6.994829249e-08f, 6.994829249e-08f});
            lvl(-2.824688409e-06f);
            lvl(6.471458619e-05f);
            lvl(-9.421016439e-04f);
        } else if constexpr (cc == 2) {
            lvl(9.306023829e-03f);
            lvl(-6.564749777e-02f);
            lvl(3.986273110e-01f);
            gs.p = __builtin_elementwise_fma(gs.t, gs.p, f32x2{0.5f, 0.5f});
        } else {
            gs.x = gs.x * gs.p;
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 hh;
            hh[0] = (__bf16)gs.x[0]; hh[1] = (__bf16)gs.x[1];
            const unsigned hv = __builtin_bit_cast(unsigned, hh);
            if constexpr (pr == 0) gs.h0 = hv;
            else {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(smem + (i_off[i4] - wave4k) + ML_X + bb * 32768 + mt * 2048) = u32x2{gs.h0, hv};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    static_for<64>([&](auto JJ) __attribute__((always_inline)) {
        constexpr int jj = decltype(JJ)::value, p = jj >> 3, i = jj & 7, j = 16 + jj, slot = j & 3;
        constexpr int q = ML_PQ[p], qg = ML_PG[p], bb = q & 1;
        GeluState gs;
        f32x4 gb;
        if constexpr (qg >= 0) gb = *reinterpret_cast<const f32x4*>(vec + V_BF1 + 256 * qg + 32 * wave + 8 * (i & 3) + 4 * kh);
        if constexpr (ML_PT[p] == 1) {
            // ---- linear1 item: two k-steps of the quarter's 32 columns for both token tiles (8 MFMAs behind ONE batch of 8 fragment
            //      reads: with a batch per k-step (4 MFMAs) neither wave of a SIMD covers the other's LDS round trip) ----------------
            bf16x8 af[2][2][2];
#pragma unroll
            for (int kl = 0; kl < 2; ++kl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        af[kl][ks][mt] = *reinterpret_cast<const bf16x8*>(smem + (2 * i + kl) * 4096 + a_off_y[ks] + mt * 2048);
            asm volatile("" ::: "memory");
            advance(j);
            static_for<8>([&](auto C) __attribute__((always_inline)) {
                constexpr int kl = decltype(C)::value >> 2, ks = (decltype(C)::value >> 1) & 1, mt = decltype(C)::value & 1;
                if constexpr (i == 0 && kl == 0 && ks == 0)
                    acc1[bb][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[slot][kl][ks], af[kl][ks][mt], zero16, 0, 0, 0);
                else
                    acc1[bb][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[slot][kl][ks], af[kl][ks][mt], acc1[bb][mt], 0, 0, 0);
#ifndef ML_EXP_NOGELU
                if constexpr (qg >= 0)
                    gelu_chunk(gs, gb, std::integral_constant<int, (qg >= 0 ? qg : 0)>{}, std::integral_constant<int, i>{}, C);
#endif
#ifndef ML_EXP_NOINIT
                if constexpr (p == 0 && (decltype(C)::value & 3) == 3) {
                    // linear2's accumulator starts from its bias + the residual h' (own elements of Y): 16 groups over the 16 half-items
                    constexpr int gi = 2 * i + kl, nt = gi >> 3, i4 = (gi >> 1) & 3, mt2 = gi & 1;
                    init_group(acc2, *reinterpret_cast<const f32x4*>(vec + V_BF2 + col4(nt, i4)), smem + ML_Y, nt, i4, mt2);
                }
#endif
            });
        } else {
            // ---- linear2 item: hidden k-block 8 q + i (k-block i of image q & 1) for the wave's 64 output columns ----------------
            const char* sb = smem + ML_X + bb * 32768 + i * 4096;
            bf16x8 af[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) af[ks][mt] = *reinterpret_cast<const bf16x8*>(sb + a_off[ks] + mt * 2048);
            asm volatile("" ::: "memory");
            advance(j);
            static_for<8>([&](auto C) __attribute__((always_inline)) {
                constexpr int ks = decltype(C)::value >> 2, nt = (decltype(C)::value >> 1) & 1, mt = decltype(C)::value & 1;
                acc2[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[slot][nt][ks], af[ks][mt], acc2[nt][mt], 0, 0, 0);
#ifndef ML_EXP_NOGELU
                if constexpr (qg >= 0)
                    gelu_chunk(gs, gb, std::integral_constant<int, (qg >= 0 ? qg : 0)>{}, std::integral_constant<int, i>{}, C);
#endif
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (qg >= 0 && i == 7) {                            // the quarter's image is complete (and the previous user of the buffer long done)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#ifdef RGN_ML_PROF
        if constexpr (i == 7) { RGN_MT(6 + p) }
#endif
    });
    RGN_MT(4)

    // =============== stage 3: norm3 -> output planes ========================================================================
    layernorm(acc2, vec + V_G3, vec + V_B3, nullptr, std::integral_constant<int, 1>{});            // (its barrier also fences the last reads of X)
    RGN_MT(17)
    store_img(acc2, smem + ML_X);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RGN_MT(18)
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave * 8 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
            const int m = m0 + r;
            if (m < g.M) {
                const int off = blk * 4096 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                *reinterpret_cast<bf16x8*>(g.out + ((size_t)blk * g.rows + m) * 32 + c * 8) = *reinterpret_cast<const bf16x8*>(smem + ML_X + off);
            }
        }
    }
    RGN_MT(5)
}

#ifdef RGN_ML_PROF
void ml_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ml_prof), sizeof(long long) * 32); }
#endif

bool mlp_supported(int d, int ff, int Tq) { return d == ML_D && ff == 2 * ML_D && 63 / Tq + 2 <= ML_NSAMP; }   // samples a 64-row tile can touch
hipError_t configure_mlp() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp), hipFuncAttributeMaxDynamicSharedMemorySize, ML_LDS);
}
hipError_t launch_mlp(const MlpArgs& g, hipStream_t s) {
    hipLaunchKernelGGL(k_mlp, dim3((g.M + ML_BM - 1) / ML_BM), dim3(ML_NT), ML_LDS, s, g);
    return hipGetLastError();
}

}  // namespace rgn
