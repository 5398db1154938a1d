// Row-persistent decoder-layer tail for the plain-bf16 phase: for a tile of 64 complete token rows ONE workgroup runs
//
//   h' = LN2( LN1( att . Wo^T + bo + h ) + call_time[step] + call_cond[sample] )        out_proj, norm1, folded cross-attn, norm2
//   y  = LN3( gelu( h' . W1^T + b1 ) . W2^T + b2 + h' )                                 linear1, GELU, linear2, norm3
//
// (nn.TransformerDecoderLayer post-norm blocks constructed at model/cmdm.py:75-81, called at :227) with every intermediate
// resident on chip. It replaces three k_rowgemm launches (out_proj+LN, linear1+GELU, linear2+LN): their prologues, the
// [M, 1024] hidden tensor and the h' round trip through memory disappear, and the LayerNorms run on the accumulators.
//
// Structure (8 waves, wave w = output columns [64 w, 64 w + 64) of whichever GEMM is running, accumulators transposed:
// lane = token, registers = columns, exactly as in k_rowgemm):
//   LDS X (64 KiB): att tile image (A operand of out_proj) -> GELU(hidden half) image (A operand of linear2) -> output image
//   LDS Y (64 KiB): h tile image (residual of norm1) -> h' image (A operand of linear1, residual of norm3), updated in place
//   stage 1  out_proj (K = 512) from X, + bias + residual (Y), LN1, + vectors, LN2 -> bf16 h' into Y
//   stage 2  for each half c of the 1024 hidden columns: linear1 columns [512 c, 512 c + 512) from Y -> GELU -> bf16 into X;
//            linear2 accumulates its k-blocks [16 c, 16 c + 16) from X into the SAME 64 x 512 accumulator
//   stage 3  + bias + residual (h' from Y), LN3 -> bf16 image in X -> contiguous 1 KiB wave-stores into the residual planes
// Weights stream straight into a 4-deep register ring from the fragment-ordered planes (see rgn_rowgemm.hip); LayerNorm
// statistics: in-register partial sums over the lane's 32 columns, one lane^32 exchange, and a 2 KiB LDS exchange between
// the 8 column slabs (two-pass, like k_layernorm).
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
#ifndef RGN_ML_LD_AUX
#define RGN_ML_LD_AUX 0    // cache policy of the two input-tile DMAs (2 = nt: 55.5 -> 54.2 us in tools/mlp_bench, but -1.5 % in the sampling loop, where the tiles were just written)
#endif
#ifndef RGN_ML_ST_AUX
#define RGN_ML_ST_AUX 16   // cache policy of the output stores: 16 = sc1 (write-through); tools: -DRGN_ML_ST_AUX=0 for plain stores
#endif

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

constexpr int ML_BM = 64, ML_D = 512, ML_NT = 512, ML_PF = 3;   // PF = weight prefetch distance (ring of PF + 1)
// LDS map: X | Y | reduction scratch (2 x 2 KiB) | per-column vectors: bo g1 b1 g2 b2 bf2 g3 b3 (8 x 512) + bf1 (1024) floats |
// sv + pv of the (at most ML_NSAMP) samples a 64-row tile touches (Tq >= 22)  = 160 KiB
constexpr int ML_NSAMP = 4;
constexpr int ML_X = 0, ML_Y = 64 * 1024, ML_RED = 128 * 1024, ML_VEC = ML_RED + 2 * 2048, ML_SPV = ML_VEC + (8 * 512 + 1024) * 4,
              ML_LDS = ML_SPV + ML_NSAMP * 512 * 4;
static_assert(ML_LDS <= 160 * 1024, "LDS map");
enum { V_BO = 0, V_G1 = 512, V_B1 = 1024, V_G2 = 1536, V_B2 = 2048, V_BF2 = 2560, V_G3 = 3072, V_B3 = 3584, V_BF1 = 4096 };

#ifdef RGN_ML_PROF
__device__ long long g_ml_prof[16];
#define RGN_MT(i) if (blockIdx.x == RGN_ML_PROF && threadIdx.x == 0) g_ml_prof[i] = __builtin_readcyclecounter();
#else
#define RGN_MT(i)
#endif

// GELU (erf form): x (0.5 + 0.5 erf(x / sqrt 2)) with 0.5 erf(x / sqrt 2) = t Q(t^2), t = clamp(x, +-3.2 sqrt 2): rgn_rowgemm.hip's
// odd degree-15 polynomial of erf (max abs error 1.6e-4) with the 1/sqrt 2, the 1/2^k of u^2 = x^2 / 2 and the 0.5 folded into
// the coefficients: 12 instructions per pair of values instead of 14 (the epilogues are VALU-issue-bound)
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
    // ---- per-column vectors -> LDS (coalesced; the epilogues read them back as broadcast float4: 64 dependent global loads
    //      per lane cost ~20 k cycles per LayerNorm pair before this)
    {
        const float* src[8] = {g.bo, g.g1, g.b1, g.g2, g.b2, g.bf2, g.g3, g.b3};
#pragma unroll
        for (int v = 0; v < 8; ++v) vec[v * 512 + tid] = src[v][tid];
        vec[V_BF1 + tid] = g.bf1[tid];
        vec[V_BF1 + 512 + tid] = g.bf1[512 + tid];
        const float sv = g.stepvec ? g.stepvec[(size_t)(*g.d_step) * g.ldstep + tid] : 0.f;
        const int s0 = m0 / g.Tq, slast = (g.M - 1) / g.Tq;
#pragma unroll
        for (int j = 0; j < ML_NSAMP; ++j) {
            const int sidx = s0 + j < slast ? s0 + j : slast;
            spv[j * 512 + tid] = sv + g.b1[tid] + (g.pervec ? g.pervec[(size_t)sidx * g.ldper + tid] : 0.f);   // norm1's beta folded in
        }
    }

    // ---- both input tiles -> LDS: att (A operand of out_proj) into X, h (residual) into Y; 16 k-blocks x 4 pieces of 1 KiB
    //      each, wave w issues the pieces p = w, w + 8, ... of both images
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave + 8 * j, kb = p >> 2, r = (p & 3) * 16 + r16;
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            const size_t src = ((size_t)kb * g.rows + m) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.att + src), (RGN_AS3 void*)(smem + ML_X + p * 1024), 16, 0, RGN_ML_LD_AUX);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.h + src), (RGN_AS3 void*)(smem + ML_Y + p * 1024), 16, 0, RGN_ML_LD_AUX);
        }
    }
    int a_off[2][2];                                                  // [mt][ks]: B-operand fragment of token 32 mt + l31 inside a k-block
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int rr = 32 * mt + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[mt][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    // the same inside Y: ds_read offsets are 16 bits, so reads of the second 64 KiB want their own base registers (one v_add per
    // fragment read otherwise: 128 VALU in the linear1 loops)
    int a_off_y[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off_y[mt][ks] = a_off[mt][ks] + ML_Y;
    // One GEMM pass: acc[nt][mt] += A_image(16 k-blocks) . W[column blocks cb0 + {0, 1}, k-blocks kt0 .. kt0 + 15]^T
    // W: fragment-ordered plane [K/32][nb_all][2][64][8]. first = the pass right behind the tile DMA (the compiler drains
    // vmcnt completely at the first ds_read behind a direct-to-LDS DMA, so the weight prefetch of that pass starts after it).
    bf16x8 wf[ML_PF + 1][2][2];
    const unsigned lane8 = (unsigned)lane * 8u;
    auto load_w = [&](const __bf16* W, int nb_all, int cb0, int kt, int slot) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const __bf16* base = W + ((size_t)kt * nb_all + cb0 + nt) * 1024;   // wave-uniform: scalar base + the lane's 32-bit offset
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[slot][ks][nt] = *reinterpret_cast<const bf16x8*>(base + ks * 512 + lane8);
        }
    };
    // the first PF k-steps' fragments of a pass: issued ahead of whatever precedes the pass (tile DMA wait, an epilogue)
    auto gemm_prefetch = [&](const __bf16* W, int nb_all, int cb0, int kt0) {
#pragma unroll
        for (int s = 0; s < ML_PF; ++s) load_w(W, nb_all, cb0, kt0 + s, s);
    };
    auto gemm16 = [&](f32x16 (&acc)[2][2], const int (&aoff)[2][2], const __bf16* W, int nb_all, int cb0, int kt0) {
        __builtin_amdgcn_sched_barrier(0);                            // keep epilogue loads out of the k-loop (register pressure -> spills)
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            const char* sb = smem + kt * 4096;
            bf16x8 af[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) af[ks][mt] = *reinterpret_cast<const bf16x8*>(sb + aoff[mt][ks]);
            asm volatile("" ::: "memory");
            if (kt + ML_PF < 16) {
                load_w(W, nb_all, cb0, kt0 + kt + ML_PF, (kt + ML_PF) & 3);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // this step's fragments are in; the next three steps' stay in flight
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kt & 3][ks][nt], af[ks][mt], acc[nt][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // accumulators start from the bias of their columns (vec: the LDS copy of the per-column vectors)
    auto init_bias = [&](f32x16 (&acc)[2][2], const float* bias) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 64 * wave + 32 * nt + 8 * i4 + 4 * kh);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] = b[e];
            }
    };
    // element (token 32 mt + l31, column 64 wave + 32 nt + 8 i4 + 4 kh + e) <-> register acc[nt][mt][4 i4 + e]; its 8-byte
    // run inside a [16 column blocks][64 rows][64 B] swizzled image:
    auto img_off = [&](int nt, int i4, int mt) {
        const int r = 32 * mt + l31;
        return (2 * wave + nt) * 4096 + r * 64 + ((i4 ^ ((r >> 2) & 3)) << 4) + 8 * kh;
    };
    auto col4 = [&](int nt, int i4) { return 64 * wave + 32 * nt + 8 * i4 + 4 * kh; };
    // sum over all 512 columns of a per-token partial (sum[mt] = this lane's 32 columns of token 32 mt + l31)
    int red_slot = 0;
    auto row_sum = [&](float (&v)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) v[mt] += __shfl_xor(v[mt], 32, 64);
        float* buf = red + (red_slot & 1) * 512;   // two alternating buffers suffice: a barrier separates each write from its reads
        ++red_slot;
        if (kh == 0) {                                                // [token][wave]: a token's 8 partials are two float4
            buf[l31 * 8 + wave] = v[0];
            buf[(32 + l31) * 8 + wave] = v[1];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(buf + l31 * 8), a1 = *reinterpret_cast<const f32x4*>(buf + l31 * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(buf + (32 + l31) * 8), b1 = *reinterpret_cast<const f32x4*>(buf + (32 + l31) * 8 + 4);
        const float s0 = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a1[0] + a1[1]) + (a1[2] + a1[3]));
        const float s1 = ((b0[0] + b0[1]) + (b0[2] + b0[3])) + ((b1[0] + b1[1]) + (b1[2] + b1[3]));
        v[0] = s0;
        v[1] = s1;
    };
    const float invn = 1.0f / (float)ML_D;
    // (has_beta: a compile-time tag - a run-time `bet != nullptr` on an LDS-derived pointer costs two v_cndmask + a wasted add per pair)
    auto layernorm = [&](f32x16 (&acc)[2][2], const float* gam, const float* bet, auto has_beta) {   // two-pass, in place; gam / bet in LDS
        // packed fp32 throughout (v_pk_add / v_pk_fma: two columns per instruction): the epilogues are VALU-issue-bound with the
        // matrix cores idle, so every instruction here is on the kernel's critical path
        f32x2 s2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; i += 2) s2[mt] += f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]};
        float s[2] = {s2[0][0] + s2[0][1], s2[1][0] + s2[1][1]};
        row_sum(s);
        const f32x2 nmean[2] = {f32x2{-s[0] * invn, -s[0] * invn}, f32x2{-s[1] * invn, -s[1] * invn}};
        f32x2 q2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 dlt = f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]} + nmean[mt];
                    acc[nt][mt][i] = dlt[0];
                    acc[nt][mt][i + 1] = dlt[1];
                    q2[mt] = __builtin_elementwise_fma(dlt, dlt, q2[mt]);
                }
        float q[2] = {q2[0][0] + q2[0][1], q2[1][0] + q2[1][1]};
        row_sum(q);
        const float rstd[2] = {__builtin_amdgcn_rsqf(q[0] * invn + 1e-5f), __builtin_amdgcn_rsqf(q[1] * invn + 1e-5f)};
        f32x4 ga[2][4], be[2][4];                                     // all the LDS reads first, then the arithmetic
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                ga[nt][i4] = *reinterpret_cast<const f32x4*>(gam + col4(nt, i4));
                if constexpr (decltype(has_beta)::value) be[nt][i4] = *reinterpret_cast<const f32x4*>(bet + col4(nt, i4));
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 rg = f32x2{ga[nt][i4][e], ga[nt][i4][e + 1]} * f32x2{rstd[mt], rstd[mt]};
                        const f32x2 v = f32x2{acc[nt][mt][4 * i4 + e], acc[nt][mt][4 * i4 + e + 1]};
                        f32x2 o;
                        if constexpr (decltype(has_beta)::value) o = __builtin_elementwise_fma(v, rg, f32x2{be[nt][i4][e], be[nt][i4][e + 1]});
                        else o = v * rg;
                        acc[nt][mt][4 * i4 + e] = o[0];
                        acc[nt][mt][4 * i4 + e + 1] = o[1];
                    }
    };
    // acc += bf16 image value (the residual); all sixteen 8-byte reads first, then the adds
    auto add_resid = [&](f32x16 (&acc)[2][2], const char* img) {
        bf16x4 r[2][4][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) r[nt][i4][mt] = *reinterpret_cast<const bf16x4*>(img + img_off(nt, i4, mt));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] += (float)r[nt][i4][mt][e];
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

    // =============== stage 1: out_proj + residual + norm1 + folded cross-attention + norm2 -> h' (Y) ====================
    f32x16 acc[2][2];
    gemm_prefetch(g.Wo, 16, 2 * wave, 0);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                // in order: both tile images landed, the weight prefetch may still fly
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // ... and this thread's share of the vectors is written
    __builtin_amdgcn_s_barrier();
    init_bias(acc, vec + V_BO);
    RGN_MT(1)
    gemm16(acc, a_off, g.Wo, 16, 2 * wave, 0);
    RGN_MT(2)
    add_resid(acc, smem + ML_Y);
    layernorm(acc, vec + V_G1, nullptr, std::false_type{});                              // beta of norm1 rides in the per-sample vector below
    {   // + norm1.beta + call_time[step] + call_cond[sample of the token] (pre-summed per sample in LDS)
        const int s0 = m0 / g.Tq;
        int sj[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + 32 * mt + l31;
            sj[mt] = ((m < g.M ? m : g.M - 1) / g.Tq - s0) * 512;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int n = col4(nt, i4);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(spv + sj[mt] + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][mt][4 * i4 + e] += a[e];
                }
            }
    }
    layernorm(acc, vec + V_G2, vec + V_B2, std::true_type{});
    store_img(acc, smem + ML_Y);                                      // h' replaces h element by element (each lane read its own first)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RGN_MT(3)

    // =============== stage 2: linear1 + GELU + linear2, the hidden 1024 columns in two halves ==============================
    f32x16 acc2[2][2];
    init_bias(acc2, vec + V_BF2);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        init_bias(acc, vec + V_BF1 + 512 * c);
        gemm_prefetch(g.W1, 32, 16 * c + 2 * wave, 0);
        gemm16(acc, a_off_y, g.W1, 32, 16 * c + 2 * wave, 0);    // hidden columns [512 c, 512 c + 512)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 gl = ml_gelu2(f32x2{acc[nt][mt][i], acc[nt][mt][i + 1]});
                    acc[nt][mt][i] = gl[0];
                    acc[nt][mt][i + 1] = gl[1];
                }
        if (c == 1) __builtin_amdgcn_s_barrier();                     // every wave is done reading the first half's image
        store_img(acc, smem + ML_X);                                  // (c == 0: X still holds the att tile, dead since stage 1)
        gemm_prefetch(g.W2, 16, 2 * wave, 16 * c);                   // flies while the barrier passes (acc is dead: no extra registers)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        gemm16(acc2, a_off, g.W2, 16, 2 * wave, 16 * c);       // linear2 over hidden k-blocks [16 c, 16 c + 16)
    }
    RGN_MT(4)

    // =============== stage 3: + bias + residual h' + norm3 -> output planes ===============================================
    add_resid(acc2, smem + ML_Y);
    layernorm(acc2, vec + V_G3, vec + V_B3, std::true_type{});                                      // (its barriers also fence the last reads of X)
    store_img(acc2, smem + ML_X);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        // WRITE-THROUGH stores (sc1): nothing of the 15 MB this launch writes is left dirty in the XCD L2s for the end-of-kernel
        // write-back (the next kernel reads it through the fabric either way: L2 contents do not survive a kernel boundary)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)((size_t)g.rows * 512 * 2), 0x00020000);
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave * 8 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
            const int m = m0 + r;
            if (m < g.M) {
                const int off = blk * 4096 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(smem + ML_X + off), o_rs,
                                                       (int)((((size_t)blk * g.rows + m) * 32 + c * 8) * 2), 0, RGN_ML_ST_AUX);
            }
        }
    }
    RGN_MT(5)
}

#ifdef RGN_ML_PROF
void ml_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ml_prof), sizeof(long long) * 16); }
#endif

bool mlp_supported(int d, int ff, int Tq) { return d == ML_D && ff == 2 * ML_D && 63 / Tq + 2 <= ML_NSAMP; }   // samples a 64-row tile can touch
// REGENNET_MLP_KERNEL: 1 = this file's kernel, 2 = rgn_mlp2.hip with 64-row tiles, 3 = rgn_mlp2.hip with 32-row tiles
static int mlp_kernel() {
    static const int r = [] {
        const char* e = getenv("REGENNET_MLP_KERNEL");
        return e ? atoi(e) : 2;
    }();
    return r;
}
hipError_t configure_mlp() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp), hipFuncAttributeMaxDynamicSharedMemorySize, ML_LDS);
    return e != hipSuccess ? e : configure_mlp2();
}
hipError_t launch_mlp(const MlpArgs& g, hipStream_t s) {
    const int k = mlp_kernel();
    if (k == 3 && mlp2_supported(32, ML_D, 2 * ML_D, g.Tq)) return launch_mlp2(32, g, s);
    if (k == 2 && mlp2_supported(64, ML_D, 2 * ML_D, g.Tq)) return launch_mlp2(64, g, s);
    hipLaunchKernelGGL(k_mlp, dim3((g.M + ML_BM - 1) / ML_BM), dim3(ML_NT), ML_LDS, s, g);
    return hipGetLastError();
}

}  // namespace rgn
