// Row-persistent decoder-layer tail for the plain-bf16 phase: for a tile of 64 complete token rows ONE workgroup runs
//
//   h' = LN2( LN1( att . Wo^T + bo + h ) + call_time[step] + call_cond[sample] )        out_proj, norm1, folded cross-attn, norm2
//   y  = LN3( gelu( h' . W1^T + b1 ) . W2^T + b2 + h' )                                 linear1, GELU, linear2, norm3
//
// (nn.TransformerDecoderLayer post-norm blocks constructed at model/cmdm.py:75-81, called at :227) with every intermediate
// resident on chip: the [M, 1024] hidden tensor and the h' round trip never reach memory, the LayerNorms run on the accumulators.
//
// Structure: SIXTEEN waves (1024 threads, four per SIMD at 128 registers); wave w owns the output columns [32 w, 32 w + 32) of
// whichever GEMM is running for all 64 rows, accumulators transposed (lane = token, registers = columns: two 16-register
// tiles). Round 2 ran this kernel with 8 waves x 64 columns at 250 registers: its five GEMM passes were at the matrix pipe's
// floor, but two thirds of the kernel were LayerNorm / GELU / exchange / store phases in which two waves per SIMD could not
// cover each other's LDS, barrier and dependent-issue latencies (phase stamps: 58 k of 88 k ticks with the matrix cores idle,
// for 13 k ticks worth of VALU issue). Interleaving that VALU work between the MFMAs of the same wave does not hide it either
// (measured: GELU issued in the MFMA shadow costs what it costs alone - the shadow is already full of the loop's own LDS /
// memory / scalar instructions, and packed fp32 next to MFMAs is priced at +11 cycles each). What hides latency on this
// machine is occupancy: four waves per SIMD, at the price of one LDS fragment read per MFMA instead of one per two.
//   LDS Y (64 KiB): h tile image (residual of norm1: consumed when the accumulators are initialised) -> h' image (A operand of
//                   linear1; residual of norm3: consumed when linear2's accumulator is initialised)
//   LDS X (64 KiB): att tile image (A operand of out_proj) -> GELU(hidden half) image (A operand of linear2) -> output image
//   stage 1  out_proj (K = 512) from X on accumulators that start at bo + h; LN1; + vectors; LN2 -> bf16 h' into Y
//   stage 2  for each half c of the 1024 hidden columns: linear1 columns [512 c, 512 c + 512) from Y -> GELU -> bf16 into X;
//            linear2 accumulates its k-blocks [16 c, 16 c + 16) from X into ONE 64 x 512 accumulator (which started at b2 + h')
//   stage 3  LN3 -> bf16 image in X -> contiguous 1 KiB wave-stores into the residual planes
// ONE weight stream for the whole kernel: 80 k-steps of two 1 KiB fragments per wave, fragment-ordered planes (rgn_rowgemm.hip)
// -> a 4-slot register ring, always three k-steps ahead, across pass boundaries, barriers and the LayerNorm / GELU phases
// (weights depend on nothing). LayerNorm statistics: in-register partial sums over the lane's 16 columns, one lane^32
// exchange, and a 4 KiB LDS exchange between the 16 column slabs (two-pass, like k_layernorm).
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <type_traits>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

constexpr int ML_BM = 64, ML_D = 512, ML_NT = 1024;
// LDS map: X | Y | statistics exchange (2 x 4 KiB) | per-column vectors g1 g2 b2 bf2 g3 b3 (6 x 512) + bf1 (1024) floats |
// sv + pv of the (at most ML_NSAMP) samples a 64-row tile touches (Tq >= 22)  = 160 KiB
constexpr int ML_NSAMP = 4;
constexpr int ML_X = 0, ML_Y = 64 * 1024, ML_RED = 128 * 1024, ML_VEC = ML_RED + 2 * 4096, ML_SPV = ML_VEC + (6 * 512 + 1024) * 4,
              ML_LDS = ML_SPV + ML_NSAMP * 512 * 4;
static_assert(ML_LDS <= 160 * 1024, "LDS map");
enum { V_G1 = 0, V_G2 = 512, V_B2 = 1024, V_BF2 = 1536, V_G3 = 2048, V_B3 = 2560, V_BF1 = 3072 };
constexpr int ML_NITEM = 5 * 16;                                      // k-steps of the weight stream

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

__global__ __launch_bounds__(ML_NT, 4) void k_mlp(MlpArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = xcd_affine(blockIdx.x, gridDim.x) * ML_BM;
    float* red = reinterpret_cast<float*>(smem + ML_RED);
    float* vec = reinterpret_cast<float*>(smem + ML_VEC);
    float* spv = reinterpret_cast<float*>(smem + ML_SPV);
    RGN_MT(0)
    auto col4 = [&](int i4) { return 32 * wave + 8 * i4 + 4 * kh; };
    // ---- prologue, in the order the counted waits rely on (vmcnt retires in order):
    //   (1) h tile -> Y by DMA   (2) per-column vectors + out_proj's bias -> registers   (3) att tile -> X by DMA   (4) weight k-steps 0-2
    // then the vectors go to LDS (their loads are in => so is Y), the accumulators are initialised from Y + bo while X lands.
    // 16 k-blocks x 4 pieces of 1 KiB per image, wave w issues the pieces p = w, w + 16, ... (coalesced 1 KiB runs of the planes)
    auto tile_dma = [&](const __bf16* src_plane, int dst) {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = wave + 16 * j, kb = p >> 2, r = (p & 3) * 16 + r16;
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
    f32x4 bo4[4];                                                     // out_proj's bias stays in registers (used once)
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) bo4[i4] = *reinterpret_cast<const f32x4*>(g.bo + col4(i4));
    // staging of the vectors: thread (half, col) = (tid >> 9, tid & 511); half 0: g1 g2 b2 bf1[col], samples 0, 1; half 1: bf2 g3 b3
    // bf1[512 + col], samples 2, 3   (no arithmetic on the loaded values before the att tile is requested)
    const int half = tid >> 9, scol = tid & 511;
    float vv[4], pv[2], sv, b1v;
    {
        vv[0] = (half ? g.bf2 : g.g1)[scol];
        vv[1] = (half ? g.g3 : g.g2)[scol];
        vv[2] = (half ? g.b3 : g.b2)[scol];
        vv[3] = g.bf1[tid];
        sv = g.stepvec ? g.stepvec[(size_t)step * g.ldstep + scol] : 0.f;
        b1v = g.b1[scol];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int sidx = s0 + 2 * half + j < slast ? s0 + 2 * half + j : slast;
            pv[j] = g.pervec ? g.pervec[(size_t)sidx * g.ldper + scol] : 0.f;
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
        // opaque to the optimiser: it would otherwise fold ML_Y back out, find that 65536 + kt * 4096 does not fit the 16-bit
        // immediate, and keep (and spill) one base register per unrolled k-step
        asm volatile("" : "+v"(a_off_y[ks]));
    }
    // ---- the weight stream: k-step j of the kernel (pass j >> 4, k-step kt = j & 15) -> ring slot j & 3 (two fragments = 8
    //      registers), requested three k-steps ahead; column block w of
    //        pass 0: Wo [16 k][16 cb]     pass 1 / 3: W1 [16 k][32 cb], blocks w / 16 + w     pass 2 / 4: W2 [32 k][16 cb], k-blocks kt / 16 + kt
    // buffer loads: descriptor (base advanced to the wave's column block) + compile-time byte offset in an SGPR, ONE address VGPR per
    // lane for the whole stream (with flat addresses the compiler keeps a 64-bit address pair per unrolled k-step and spills)
    bf16x8 wf[4][2];
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Wo) + (size_t)wave * 1024, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.W1) + (size_t)wave * 1024, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.W2) + (size_t)wave * 1024, 0, -1, 0x00020000);
    const int lane16 = lane * 16;
    auto load_item = [&](int j) {
        const int slot = j & 3, pass = j >> 4, kt = j & 15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (pass == 0)
                wf[slot][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_o, lane16, ((kt * 16) * 1024 + ks * 512) * 2, 0));
            else if (pass == 1 || pass == 3)
                wf[slot][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_1, lane16, ((kt * 32 + 8 * (pass - 1)) * 1024 + ks * 512) * 2, 0));
            else
                wf[slot][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_2, lane16, (((8 * (pass - 2) + kt) * 16) * 1024 + ks * 512) * 2, 0));
        }
    };
    // k-step j is about to run: request k-step j + 3, then wait until k-step j's fragments are in (the later ones stay in flight)
    auto advance = [&](int j) {
        if (j + 3 < ML_NITEM) {
            load_item(j + 3);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (j + 3 == ML_NITEM) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (j + 2 == ML_NITEM) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // one GEMM pass: acc[mt] (+)= A image (16 k-blocks at abase + kt * 4096) . W(pass)^T for the wave's 32 columns
    auto gemm_pass = [&](f32x16 (&acc)[2], int pass, const int (&aoff)[2], int abase) {
        __builtin_amdgcn_sched_barrier(0);                            // keep epilogue code out of the k-loop (register pressure -> spills)
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            const char* sb = smem + abase + kt * 4096;
            bf16x8 af[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) af[ks][mt] = *reinterpret_cast<const bf16x8*>(sb + aoff[ks] + mt * 2048);
            asm volatile("" ::: "memory");
            advance(16 * pass + kt);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kt & 3][ks], af[ks][mt], acc[mt], 0, 0, 0);
            // pin both accumulator chains to their k-step: without the use the compiler sinks ALL of one chain's MFMAs of the first
            // pass behind the loop and parks their operands in scratch (observed: 255 spilled registers)
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // element (token 32 mt + l31, column 32 wave + 8 i4 + 4 kh + e) <-> register acc[mt][4 i4 + e]; its 8-byte run inside a
    // [16 column blocks][64 rows][64 B] swizzled image: i_off[i4] + mt * 2048 (the latter in the immediate offset)
    int i_off[4];
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) i_off[i4] = wave * 4096 + l31 * 64 + ((i4 ^ ((l31 >> 2) & 3)) << 4) + 8 * kh;
    // sum over all 512 columns of a per-token partial (v[mt] = this lane's 16 columns of token 32 mt + l31)
    int red_slot = 0;
    auto row_sum = [&](float (&v)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) v[mt] += __shfl_xor(v[mt], 32, 64);
        float* buf = red + (red_slot & 1) * 1024;   // two alternating buffers suffice: a barrier separates each write from its reads
        ++red_slot;
        if (kh == 0) {                                                // [token][wave]: a token's 16 partials are four float4
            buf[l31 * 16 + wave] = v[0];
            buf[(32 + l31) * 16 + wave] = v[1];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const f32x4*>(buf + (32 * mt + l31) * 16 + 4 * k);
            const f32x4 s = (a[0] + a[1]) + (a[2] + a[3]);
            v[mt] = (s[0] + s[1]) + (s[2] + s[3]);
        }
    };
    const float invn = 1.0f / (float)ML_D;
    // two-pass LayerNorm, in place; gam / bet in LDS. beta_kind 1: bet[col], 2: bet[bsel[mt] + col] (per-sample vector)
    auto layernorm = [&](f32x16 (&acc)[2], const float* gam, const float* bet, const int* bsel, auto beta_kind) {
        constexpr int BK = decltype(beta_kind)::value;
        f32x2 s2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; i += 2) s2[mt] += f32x2{acc[mt][i], acc[mt][i + 1]};
        float s[2] = {s2[0][0] + s2[0][1], s2[1][0] + s2[1][1]};
        RGN_MT(20 + 3 * (red_slot >> 1))
        row_sum(s);
        RGN_MT(21 + 3 * ((red_slot - 1) >> 1))
        f32x2 q2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x2 nmean = f32x2{-s[mt] * invn, -s[mt] * invn};
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 dlt = f32x2{acc[mt][i], acc[mt][i + 1]} + nmean;
                acc[mt][i] = dlt[0];
                acc[mt][i + 1] = dlt[1];
                q2[mt] = __builtin_elementwise_fma(dlt, dlt, q2[mt]);
            }
        }
        float q[2] = {q2[0][0] + q2[0][1], q2[1][0] + q2[1][1]};
        row_sum(q);
        RGN_MT(22 + 3 * ((red_slot - 1) >> 1))
        const float rstd[2] = {__builtin_amdgcn_rsqf(q[0] * invn + 1e-5f), __builtin_amdgcn_rsqf(q[1] * invn + 1e-5f)};
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + col4(i4));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4 be;
                if constexpr (BK == 1) be = *reinterpret_cast<const f32x4*>(bet + col4(i4));
                if constexpr (BK == 2) be = *reinterpret_cast<const f32x4*>(bet + bsel[mt] + col4(i4));
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const f32x2 rg = f32x2{ga[e], ga[e + 1]} * f32x2{rstd[mt], rstd[mt]};
                    const f32x2 o = __builtin_elementwise_fma(f32x2{acc[mt][4 * i4 + e], acc[mt][4 * i4 + e + 1]}, rg, f32x2{be[e], be[e + 1]});
                    acc[mt][4 * i4 + e] = o[0];
                    acc[mt][4 * i4 + e + 1] = o[1];
                }
            }
        }
    };
    auto store_img = [&](const f32x16 (&acc)[2], char* img) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (__bf16)acc[mt][4 * i4 + e];
                *reinterpret_cast<bf16x4*>(img + i_off[i4] + mt * 2048) = h;
            }
    };
    // acc = bias + bf16 image values (the residual: this lane's own elements)
    auto init_acc = [&](f32x16 (&acc)[2], const f32x4 (&b)[4], const char* img) {
        bf16x4 r[4][2];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) r[i4][mt] = *reinterpret_cast<const bf16x4*>(img + i_off[i4] + mt * 2048);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][4 * i4 + e] = b[i4][e] + (float)r[i4][mt][e];
    };

    // =============== stage 1: out_proj + residual + norm1 + folded cross-attention + norm2 -> h' (Y) ====================
    f32x16 acc[2];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 3; ++j) load_item(j);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                // in order: the h tile and the vectors are in; att and the weights may still fly
    vec[(half ? V_BF2 : V_G1) + scol] = vv[0];
    vec[(half ? V_G3 : V_G2) + scol] = vv[1];
    vec[(half ? V_B3 : V_B2) + scol] = vv[2];
    vec[V_BF1 + tid] = vv[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) spv[(2 * half + j) * 512 + scol] = sv + b1v + pv[j];   // norm1's beta folded in
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    init_acc(acc, bo4, smem + ML_Y);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                 // the att tile landed
    __builtin_amdgcn_s_barrier();
    RGN_MT(1)
    gemm_pass(acc, 0, a_off, ML_X);
    RGN_MT(2)
    int sj[2];                                                        // the token's sample inside the tile -> its row of spv
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + 32 * mt + l31;
        int rel = 0;
#pragma unroll
        for (int k = 1; k < ML_NSAMP; ++k) rel += (m >= (s0 + k) * g.Tq) ? 1 : 0;
        sj[mt] = (rel < slast - s0 ? rel : slast - s0) * 512;          // (rows past M repeat the last sample's vector)
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

    // =============== stage 2: linear1 + GELU + linear2, the hidden 1024 columns in two halves ==============================
    f32x16 acc2[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {                              // linear1 starts from its bias
            const f32x4 b = *reinterpret_cast<const f32x4*>(vec + V_BF1 + 512 * c + col4(i4));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][4 * i4 + e] = b[e];
        }
        gemm_pass(acc, 1 + 2 * c, a_off_y, 0);                        // hidden columns [512 c, 512 c + 512) (a_off_y carries ML_Y)
#ifdef RGN_ML_PROF
        if (c == 0) { RGN_MT(7) }
#endif
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 gl = ml_gelu2(f32x2{acc[mt][i], acc[mt][i + 1]});
                acc[mt][i] = gl[0];
                acc[mt][i + 1] = gl[1];
            }
#ifdef RGN_ML_PROF
        if (c == 0) { RGN_MT(8) }
#endif
        if (c == 1) __builtin_amdgcn_s_barrier();                     // every wave is done reading the first half's image
        store_img(acc, smem + ML_X);                                  // (c == 0: X still holds the att tile, dead since stage 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef RGN_ML_PROF
        if (c == 0) { RGN_MT(9) }
#endif
        if (c == 0) {                                                 // linear2 starts from its bias + the residual h' (this lane's own elements of Y)
            f32x4 b2v[4];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) b2v[i4] = *reinterpret_cast<const f32x4*>(vec + V_BF2 + col4(i4));
            init_acc(acc2, b2v, smem + ML_Y);
        }
        gemm_pass(acc2, 2 + 2 * c, a_off, ML_X);                      // linear2 over hidden k-blocks [16 c, 16 c + 16)
#ifdef RGN_ML_PROF
        if (c == 0) { RGN_MT(6) }
#endif
    }
    RGN_MT(4)

    // =============== stage 3: norm3 -> output planes ========================================================================
    layernorm(acc2, vec + V_G3, vec + V_B3, nullptr, std::integral_constant<int, 1>{});            // (its barriers also fence the last reads of X)
    RGN_MT(17)
    store_img(acc2, smem + ML_X);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RGN_MT(18)
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = wave * 4 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
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
