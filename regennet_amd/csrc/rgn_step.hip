// Step boundary of the plain-bf16 phase as ONE kernel (d = 512; GUIDED = classifier-free guidance, see below):
//
//   x0   = h . Wout^T + bout                       output projection of the evaluation just finished (OutputProcess, cmdm.py:353)
//   x'   = sampler(x, x0, eps)                     p_sample / ddim_sample update in place (gaussian_diffusion.py:265-276, 508-560,
//                                                  419-423, 744-794), same device arithmetic and Philox stream as k_update
//   h'   = x' . Wx'^T + c0                         input embedding of the NEXT evaluation (InputProcess / fuse / positional part
//                                                  hoisted into c0, cmdm.py:201-218) -> residual-stream planes
//
// for one tile of 64 token rows per workgroup. Replaces three launches per chain and step (k_gemm_x3 out, k_update,
// k_gemm_x3 in) and their round trips: the fp32 x0 rows (20.6 MB written + read at B=256), the token-major x' planes
// (21.6 MB written, 10.8 MB read), 31 MB of h planes. Both GEMMs use k_mlp's machinery (activation image in LDS, weights
// streamed from the fragment-ordered planes into a 4-slot register ring).
//
// Phases / LDS:  A  h tile -> image (64 KiB, DMA) ; GEMM 1 (K = 512, N = F <= 352: waves 0-5)
//                B  x0 = acc + bias -> fp32 tile [64][XLD] (over the dead image)
//                C  lanes = rows (consecutive frames of a sample: coalesced x accesses), waves stride the features:
//                   sampler update in place, x' as bf16 into the K32-blocked image of GEMM 2's A operand (44 KiB)
//                D  GEMM 2 (K = 352, N = 512) -> bf16 image (64 KiB, over the dead tile) -> + c0 (a bf16 copy of the condition rows:
//                   half the bytes of the largest operand of the step) in the coalesced copy-out
//                   (the sum is rounded to bf16 twice, c0 once: plain-bf16 phase only)
// GUIDED (cfg_sampler.py:22-31): a token's conditional / unconditional evaluations are rows m / m + half of the planes. One
// launch over the conditional rows after the chains joined: A stages BOTH tiles (2 x 64 KiB) and runs the two output projections
// in one pass over Wout (every weight fragment feeds both accumulator sets), B forms x0 = u + scale_b (c - u) with k_update's
// rounding, C is unchanged, D embeds x' once and writes it to both halves with their own c0 rows.
// The last workgroup to finish (over all launches of the step) moves the device-side loop index on, like k_update.
#include "rgn_internal.h"
#include "rgn_philox.h"

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))
#ifndef RGN_ST_ST_AUX
#define RGN_ST_ST_AUX 16   // cache policy of the h-plane stores: 16 = sc1 (write-through)
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}) ... (the update loop's body is too large for `#pragma unroll` to be honoured,
// and its prefetched operands must live in registers, i.e. be indexed by constants)
template <int... Is, class F>
__device__ __forceinline__ void st_static_for_seq(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void st_static_for(F&& f) { st_static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

namespace {
constexpr int ST_BM = 64, ST_NT = 512, ST_PF = 3;
constexpr int ST_XLD = 356;                                          // fp32 tile row stride (floats): F <= 352, 16-byte aligned rows
constexpr int ST_TILE = 0, ST_XIMG = 92160;                          // fp32 tile 64 x 356 x 4 = 91136 B; x' image behind it
constexpr int ST_LDS = ST_XIMG + 11 * 4096;
static_assert(ST_BM * ST_XLD * 4 <= ST_XIMG && 16 * 4096 <= ST_XIMG && 2 * 65536 <= ST_LDS, "tile / images (guided: two 64 KiB input images)");
}  // namespace

template <int NKX, bool GUIDED, bool F16 = false>
__global__ __launch_bounds__(ST_NT, 2) void k_step(StepArgs g) {
    using OP = OpFmt<F16>;                // bf16 or fp16 operands (rgn_internal.h): h planes in and out, Wout / Wx, the x' image, c0
    using op_t = typename OP::t;
    using op8 = typename OP::v8;
    using op4 = typename OP::v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = xcd_affine(blockIdx.x, gridDim.x) * ST_BM;
    const SampleParams sp = *g.sp;
    const int step = *g.d_step;
    const StepCoef k = g.tab[step];

    // ---- A: the h tile -> LDS image [16 k-blocks][64 rows][64 B] (16-byte chunk c of row r at c ^ ((r >> 2) & 3))
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave + 8 * j, kb = p >> 2, r = (p & 3) * 16 + r16;
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;
            const size_t src = ((size_t)kb * g.rows + m) * 32 + ((c ^ ((r >> 2) & 3)) << 3);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.h + src), (RGN_AS3 void*)(smem + p * 1024), 16, 0, 0);
            if constexpr (GUIDED)   // the unconditional evaluation's rows of the same tokens (second half of the planes) -> a second image
                __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.h + src + (size_t)g.half * 32), (RGN_AS3 void*)(smem + 65536 + p * 1024), 16, 0, 0);
        }
    }
    int a_off[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int rr = 32 * mt + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[mt][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    op8 wf[ST_PF + 1][2][2];
    const unsigned lane8 = (unsigned)lane * 8u;
    auto load_w = [&](const __bf16* W, int nb_all, int cb0, int kt, int slot) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int cb = cb0 + nt < nb_all ? cb0 + nt : nb_all - 1;
            const __bf16* base = W + ((size_t)kt * nb_all + cb) * 1024;   // wave-uniform: scalar base + the lane's 32-bit offset
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[slot][ks][nt] = *reinterpret_cast<const op8*>(base + ks * 512 + lane8);
        }
    };
    auto prefetch = [&](const __bf16* W, int nb_all, int cb0) {
#pragma unroll
        for (int s = 0; s < ST_PF; ++s) load_w(W, nb_all, cb0, s, s);
    };
    auto gemm = [&](f32x16 (&acc)[2][2], const char* img, const __bf16* W, int nb_all, int cb0, auto nk_c) {
        constexpr int NK = decltype(nk_c)::value;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            const char* sb = img + kt * 4096;
            op8 af[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) af[ks][mt] = *reinterpret_cast<const op8*>(sb + a_off[mt][ks]);
            asm volatile("" ::: "memory");
            if (kt + ST_PF < NK) {
                load_w(W, nb_all, cb0, kt + ST_PF, (kt + ST_PF) & 3);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // this step's fragments are in; the next three steps' stay in flight
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = OP::mfma(wf[kt & 3][ks][nt], af[ks][mt], acc[nt][mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto col4 = [&](int nt, int i4) { return 64 * wave + 32 * nt + 8 * i4 + 4 * kh; };

    // ---- GEMM 1: x0 = h . Wout^T (+ bias below)
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    f32x16 accu[GUIDED ? 2 : 1][2];                                   // guided: the unconditional evaluation's x0
    if constexpr (GUIDED) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) accu[a][b][i] = 0.f;
    }
    prefetch(g.Wout, g.nb_out, 2 * wave);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                // in order: the tile(s) landed, the weight prefetch may still fly
    __builtin_amdgcn_s_barrier();
    if constexpr (!GUIDED) {
        gemm(acc, smem, g.Wout, g.nb_out, 2 * wave, std::integral_constant<int, 16>{});
    } else {
        // both evaluations in ONE pass over the weights: every fragment feeds the conditional and the unconditional tile
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            const char* sb = smem + kt * 4096;
            op8 afc[2][2], afu[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    afc[ks][mt] = *reinterpret_cast<const op8*>(sb + a_off[mt][ks]);
                    afu[ks][mt] = *reinterpret_cast<const op8*>(sb + 65536 + a_off[mt][ks]);
                }
            asm volatile("" ::: "memory");
            if (kt + ST_PF < 16) {
                load_w(g.Wout, g.nb_out, 2 * wave, kt + ST_PF, (kt + ST_PF) & 3);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[nt][mt] = OP::mfma(wf[kt & 3][ks][nt], afc[ks][mt], acc[nt][mt]);
                        accu[nt][mt] = OP::mfma(wf[kt & 3][ks][nt], afu[ks][mt], accu[nt][mt]);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();                                     // every wave is done reading the image(s)

    // ---- B: x0 -> fp32 tile
    float* tile = reinterpret_cast<float*>(smem + ST_TILE);
    float scl[2] = {0.f, 0.f};                                        // guidance scale of the sample of token 32 mt + l31
    if constexpr (GUIDED) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int mm = m0 + 32 * mt + l31;
            scl[mt] = g.scale[g.s0 + (mm < g.M ? mm : g.M - 1) / g.T];
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int n = col4(nt, i4);
            if (n < 352) {                                            // (waves 6, 7 hold no column of the tile; F % 4 == 0: whole runs)
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (n < g.F) b = *reinterpret_cast<const f32x4*>(g.bout + n);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 v = {acc[nt][mt][4 * i4] + b[0], acc[nt][mt][4 * i4 + 1] + b[1], acc[nt][mt][4 * i4 + 2] + b[2], acc[nt][mt][4 * i4 + 3] + b[3]};
                    if constexpr (GUIDED) {   // x0 = x0_u + scale_b (x0_c - x0_u), cfg_sampler.py:31, rounded like k_update
                        const float sc = scl[mt];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float u = accu[nt][mt][4 * i4 + e] + b[e];
                            v[e] = __fadd_rn(u, __fmul_rn(sc, __fsub_rn(v[e], u)));
                        }
                    }
                    *reinterpret_cast<f32x4*>(tile + (32 * mt + l31) * ST_XLD + n) = v;
                }
            }
        }
    prefetch(g.Wx, 16, 2 * wave);                                    // GEMM 2's first fragments fly under the update phase
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- C: sampler update. lane = row of the tile (consecutive frames of a sample), the waves stride the features.
    //      Noise: one Philox4x32-10 call yields the four normals of frames 4j .. 4j+3 of a (sample, step, feature). When the
    //      rows of every quad of lanes are exactly such a run (60 frames: always, but for the surplus rows of the last tile),
    //      lane q of a quad draws feature f + q for the quad's four frames and the quad transposes (DPP): a quarter of the
    //      Philox rounds, half of the Box-Muller work; the per-element form (bit-identical values) covers everything else.
    {
        const int m = m0 + lane;
        const bool valid = m < g.M;
        const int bl = (valid ? m : g.M - 1) / g.T, t = (valid ? m : g.M - 1) - bl * g.T, b = g.s0 + bl;
        const size_t FT = (size_t)g.F * g.T;
        const int bn = sp.const_noise ? 0 : b;                          // const_noise: motion 0's draw for every motion
        char* ximg = smem + ST_XIMG;
        const int q = lane & 3, tq = t - q;                            // frame of the quad's first lane, if the quad is a run
        const bool run4 = valid && (m0 + (lane | 3)) < g.M && tq >= 0 && (tq & 3) == 0 && tq + 3 < g.T;
        const bool quads = !sp.noise && !g.no_quads && __all(run4);
        // the sampler state of this wave's NKX x 4 features, requested in ONE batch: inside the loop below every group of four paid
        // its own memory round trip (11 dependent latencies per wave; without a noise tape nothing else in the loop reads memory)
        float xpre[NKX][4];
#pragma unroll
        for (int it = 0; it < NKX; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 4 * (wave + 8 * it) + j;
                xpre[it][j] = (valid && f < g.F) ? sp.x[(size_t)b * FT + (size_t)f * g.T + t] : 0.f;
            }
        auto update = [&](int f, float eps_in, float xv) {
            float nv = 0.f;
            if (valid && f < g.F) {
                float x0 = tile[lane * ST_XLD + f];
                if (sp.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                const size_t o = (size_t)b * FT + (size_t)f * g.T + t;
                if (sp.x0_out) sp.x0_out[o] = x0;
                float eps = eps_in;
                if (sp.noise)
                    eps = sp.noise[(size_t)(sp.first_index - step) * g.B * FT + (size_t)bn * FT + (size_t)f * g.T + t];
                else if (!quads)
                    eps = philox_normal(sp.seed, sp.sample_offset + bn, (uint32_t)step, (uint32_t)(f * 4096 + t));
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
            // x' (0 in the K padding columns and the surplus rows) -> the K32-blocked image of GEMM 2's A operand
            const int r = lane, chunk = (f & 31) >> 3;
            *reinterpret_cast<op_t*>(ximg + (f >> 5) * 4096 + r * 64 + ((chunk ^ ((r >> 2) & 3)) << 4) + (f & 7) * 2) = (op_t)nv;
        };
        st_static_for<NKX>([&](auto IT) __attribute__((always_inline)) {   // groups of 4 features
            constexpr int it = decltype(IT)::value;
            const int fg = wave + 8 * it;
            float eps4[4] = {0.f, 0.f, 0.f, 0.f};
            if (quads) {                                               // wave-uniform
                // this lane: the four normals of (feature 4 fg + q, frames tq .. tq + 3), exactly philox_normal's arithmetic
                const uint32_t elem = (uint32_t)((4 * fg + q) * 4096 + tq);
                const unsigned long long sample = sp.sample_offset + bn;
                uint32_t r[4];
                philox4x32_10(elem >> 2, (uint32_t)step, (uint32_t)sample, (uint32_t)(sample >> 32), (uint32_t)sp.seed, (uint32_t)(sp.seed >> 32), r);
                float n4[4];
#pragma unroll
                for (int pair = 0; pair < 2; ++pair) box_muller(r[2 * pair], r[2 * pair + 1], n4[2 * pair], n4[2 * pair + 1]);
                // transpose inside the quad: this lane (frame tq + q) needs, for feature 4 fg + j, element q of lane j's n4
                auto pick = [&](auto jc) {                              // element q of lane jc's n4, broadcast inside the quad
                    constexpr int J = decltype(jc)::value;
                    float v = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bc = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, n4[e]), J * 0x55, 0xf, 0xf, true));   // quad_perm [J, J, J, J]
                        v = q == e ? bc : v;
                    }
                    return v;
                };
                eps4[0] = pick(std::integral_constant<int, 0>{});
                eps4[1] = pick(std::integral_constant<int, 1>{});
                eps4[2] = pick(std::integral_constant<int, 2>{});
                eps4[3] = pick(std::integral_constant<int, 3>{});
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) update(4 * fg + j, eps4[j], xpre[it][j]);
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- D: h' = x' . Wx'^T (+ c0 in the copy-out). The condition rows of the copy-out are requested NOW, ahead of the GEMM
    //      (in the copy-out loop each piece waited for its own two loads)
    op8 c0v[8];
    {
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave * 8 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
            const int m = m0 + r < g.M ? m0 + r : g.M - 1;
            c0v[j] = *reinterpret_cast<const op8*>(g.c0 + (size_t)m * 512 + blk * 32 + c * 8);
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    gemm(acc, smem + ST_XIMG, g.Wx, 16, 2 * wave, std::integral_constant<int, NKX>{});
    // (the fp32 tile is dead since the barrier above: the output image goes over it; the x' image is still being read by
    //  slower waves, but it lies behind the 64 KiB the output image takes)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int r = 32 * mt + l31;
                op4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (op_t)acc[nt][mt][4 * i4 + e];
                *reinterpret_cast<op4*>(smem + (2 * wave + nt) * 4096 + r * 64 + ((i4 ^ ((r >> 2) & 3)) << 4) + 8 * kh) = hv;
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        // (write-through stores, as k_mlp's: the 15 / 31 MB of h planes are not left dirty in L2 for the end-of-kernel write-back)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(g.hout, 0, (int)((size_t)g.rows * 512 * 2), 0x00020000);
        const int r16 = lane >> 2, c = lane & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int p = wave * 8 + j, blk = p >> 2, r = (p & 3) * 16 + r16;
            const int m = m0 + r;
            if (m < g.M) {
                const op8 v = *reinterpret_cast<const op8*>(smem + blk * 4096 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4));
                const __bf16* cp = g.c0 + (size_t)m * 512 + blk * 32 + c * 8;
                op8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (op_t)((float)v[e] + (float)c0v[j][e]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), h_rs, (int)((((size_t)blk * g.rows + m) * 32 + c * 8) * 2), 0, RGN_ST_ST_AUX);
                if constexpr (GUIDED) {   // the unconditional evaluation sees the same x', with its own condition part
                    const op8 cu = *reinterpret_cast<const op8*>(cp + (size_t)g.half * 512);
                    op8 ou;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ou[e] = (op_t)((float)v[e] + (float)cu[e]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ou), h_rs, (int)((((size_t)blk * g.rows + m + g.half) * 32 + c * 8) * 2), 0, RGN_ST_ST_AUX);
                }
            }
        }
    }
    if (tid == 0) {   // ticket: the last tile of the step moves the loop index on (see k_update)
        int* tick = g.d_step + 4;
        if (atomicAdd(tick, 1) == g.total_tiles - 1) {
            tick[0] = 0;
            g.d_step[0] = step - 1;
        }
    }
}

bool step_fused_supported(int d, int F, int Kpx) { return d == 512 && F % 4 == 0 && F <= 352 && Kpx == 352; }
hipError_t configure_step() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<11, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<11, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<11, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_step<11, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
}
hipError_t launch_step(const StepArgs& g, hipStream_t s) {
    if (g.nkx != 11 || g.M <= 0) return hipErrorInvalidValue;
    const dim3 grid((g.M + ST_BM - 1) / ST_BM);
    if (g.scale) {   // guided: M = token rows of the conditional half; half = row distance to the unconditional half
        if (g.f16) hipLaunchKernelGGL((k_step<11, true, true>), grid, dim3(ST_NT), ST_LDS, s, g);
        else hipLaunchKernelGGL((k_step<11, true>), grid, dim3(ST_NT), ST_LDS, s, g);
    } else {
        if (g.f16) hipLaunchKernelGGL((k_step<11, false, true>), grid, dim3(ST_NT), ST_LDS, s, g);
        else hipLaunchKernelGGL((k_step<11, false>), grid, dim3(ST_NT), ST_LDS, s, g);
    }
    return hipGetLastError();
}

}  // namespace rgn
