// Fused in_proj GEMM + causal self-attention (Tq <= 64 tokens, head dim 128): q, k, v never leave the register file.
//
// Replaces, per decoder layer, the packed in_proj Linear of nn.MultiheadAttention and the attention itself
// (model/cmdm.py:227 -> TransformerDecoderLayer._sa_block with generate_square_subsequent_mask :168-171).
// A workgroup owns TWO samples and H/2 heads (grid = ceil(Bm/2) x 2; one head each for small launches). For each of its
// heads it computes [q_h | k_h | v_h] = x . W_h^T for both samples at once (128 x 384, split-bf16 MFMA, operands by
// direct-to-LDS DMA from the K32-blocked planes exactly as in k_gemm_x3) and then runs the attention straight from the
// accumulators: an MFMA contraction does not care in which order the reduction index is fed as long as both operands
// agree, and the C/D layouts of q^T, k^T (accumulated transposed: W fragment as the MFMA A operand), v and the
// softmaxed scores agree by construction, so S^T = K.Q^T and O^T = V^T.P^T take the accumulator registers as operands
// with no LDS staging, no transposes and no fragment reads. The only exchange is the sum of the four dh-tile partials
// of S^T through an fp32 LDS buffer that aliases the dead pipeline stages. The head's output is stored as split planes
// in the K32-blocked layout the out_proj GEMM consumes.
//
// Why two samples x half the heads instead of one sample x all heads: the kernel is bound by the ~20 B/clk a CU pulls
// from L2 into LDS (see rgn_gemm_x3.hip), and almost all of that is the in_proj weight stream. Pairing samples halves
// the weight bytes per sample (2.0 MiB of DMA per workgroup instead of 3.5 MiB) at the same workgroup count.
//
// 8 waves; wave = (sample wm, dh tile wn): wave tile 64 x 96 = both token tiles x the 32 columns wn of each of q, k, v;
// two 64 KiB LDS stages [A 128x32 | W_h 384x32] x {hi, lo}; tile rows 0-63 / 64-127 are the tokens of sample 0 / 1
// (padding rows replicate the last token and are masked).
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace rgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))
#ifndef RGN_QA_ST_AUX
#define RGN_QA_ST_AUX 16   // cache policy of the plain-bf16 build's output stores: 16 = sc1 (write-through)
#endif

constexpr int QA_ROWS = 64, QA_NS = 2, QA_DH = 128, QA_WROWS = 3 * QA_DH, QA_NT = 512;

#ifdef RGN_QA_LIFE
__device__ long long g_qa_life[2048 * 3];   // tools only: wall_clock64() (10 ns) at start / first MFMA operands landed / end of every workgroup
#define RGN_QL(i) if (tid == 0) g_qa_life[(blockIdx.y * gridDim.x + blockIdx.x) * 3 + i] = wall_clock64();
#else
#define RGN_QL(i)
#endif
#ifdef RGN_QA_PROF
__device__ long long g_qa_prof[64];   // tools only: phase cycle stamps of one workgroup (wave 0)
#define RGN_QT(i) if (blockIdx.x == RGN_QA_PROF && blockIdx.y == 0 && tid == 0) g_qa_prof[i] = clock64();
#else
#define RGN_QT(i)
#endif

// Attention straight from the in_proj accumulators of one head (see the file header); shared by the DMA-fed and the
// register-streamed GEMM phases. acc[token tile][q | k | v]; smem: the 96 KiB fp32 exchange buffer for the S^T partials.
template <bool X3, bool F16 = false>
__device__ __forceinline__ void qa_attention(f32x16 (&acc)[2][3], const QkvAttnArgs& g, char* smem, const float* bias_h, int hd, int hslot,
                                             int wm, int wn, int nsamp, int b0, int lane, int tid) {
    static_assert(!(X3 && F16), "the split form is bf16 (hi, lo) pairs");
    using OP = OpFmt<F16>;                // plain form: bf16 or fp16 operands (rgn_internal.h)
    using op_t = typename OP::t;
    using op8 = typename OP::v8;
    using op4 = typename OP::v4;
    const int l31 = lane & 31, kh = lane >> 5, Tq = g.Tq;
    (void)tid; (void)hslot;
    // ---------------- attention straight from the accumulators: q, k, v never leave the register file --------------
    // Wave (wm, wn) holds, for its sample and the 32 dh columns of tile wn: q and k transposed (lane = token, the 16
    // registers = dh (i&3) + 8(i>>2) + 4*kh) and v plain (lane = dh, registers = tokens in the same pattern), for both
    // token tiles. An MFMA contraction does not care in which order the reduction index is fed as long as both
    // operands agree, and here they do by construction: register i of lane half kh means the same dh in q and in k,
    // and the same token in v and in p. So
    //   S^T[keys, queries] (partial over this wave's 32 dh) = K-regs (A operand) x Q-regs (B operand),
    //   O^T[dh tile, queries] = V-regs (A) x P^T-regs (B, the softmaxed S^T accumulators)
    // need no LDS staging, no transposes and no ds_reads. The only exchange is the sum of the four dh-tile partials of
    // S^T, done through an fp32 LDS buffer that aliases the dead pipeline stages (96 KiB: 2 samples x 3 causal tiles
    // x 4 waves x 4 KiB). Both samples proceed at the same time on their own four waves.
    const bool live = wm < nsamp;                               // (an odd batch: the second sample of the last pair is a dummy)
    const size_t row0 = (size_t)(b0 + (live ? wm : 0)) * Tq;
        op8 qh[2][2], ql[2][2], kfh[2][2], kfl[2][2], vh[2][2], vl[2][2];   // [token tile][16-slice of the register index]
    {
        const float bv = bias_h[2 * QA_DH + l31];
        const float qs2 = g.qscale * 1.44269504088896340736f;   // scores in log2 units: softmax = exp2(s - max), one mul less per score
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const f32x4 bq0 = *reinterpret_cast<const f32x4*>(bias_h + 16 * sl + 4 * kh);
                const f32x4 bq1 = *reinterpret_cast<const f32x4*>(bias_h + 16 * sl + 8 + 4 * kh);
                const f32x4 bk0 = *reinterpret_cast<const f32x4*>(bias_h + QA_DH + 16 * sl + 4 * kh);
                const f32x4 bk1 = *reinterpret_cast<const f32x4*>(bias_h + QA_DH + 16 * sl + 8 + 4 * kh);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * sl + j;
                    const float xq = (acc[ta][0][i] + (j < 4 ? bq0[j] : bq1[j - 4])) * qs2;
                    const float xk = acc[ta][1][i] + (j < 4 ? bk0[j] : bk1[j - 4]);
                    const float xv = acc[ta][2][i] + bv;
                    qh[ta][sl][j] = (op_t)xq;
                    kfh[ta][sl][j] = (op_t)xk;
                    vh[ta][sl][j] = (op_t)xv;
                    if (X3) {
                        ql[ta][sl][j] = (op_t)(xq - (float)qh[ta][sl][j]);
                        kfl[ta][sl][j] = (op_t)(xk - (float)kfh[ta][sl][j]);
                        vl[ta][sl][j] = (op_t)(xv - (float)vh[ta][sl][j]);
                    }
                }
            }
    }
    // causal tiles of S^T: 0 = (keys 0-31, queries 0-31), 1 = (keys 0-31, queries 32-63), 2 = (keys 32-63, queries 32-63)
    f32x16 st[3];
#pragma unroll
    for (int tl = 0; tl < 3; ++tl) {
        const int kj = tl >> 1, qtile = tl ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[tl][i] = 0.f;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (X3) {
                st[tl] = OP::mfma(kfl[kj][sl], qh[qtile][sl], st[tl]);
                st[tl] = OP::mfma(kfh[kj][sl], ql[qtile][sl], st[tl]);
            }
            st[tl] = OP::mfma(kfh[kj][sl], qh[qtile][sl], st[tl]);
        }
    }
    // sum the partials of the four dh tiles: [sample][tile][wave wn][i/4][lane] float4
    f32x4* sred = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int tl = 0; tl < 3; ++tl)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 v = {st[tl][4 * i4], st[tl][4 * i4 + 1], st[tl][4 * i4 + 2], st[tl][4 * i4 + 3]};
            sred[(((wm * 3 + tl) * 4 + wn) * 4 + i4) * 64 + lane] = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RGN_QT(hslot * 8 + 2)
#pragma unroll
    for (int tl = 0; tl < 3; ++tl)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            f32x4 v = sred[(((wm * 3 + tl) * 4 + 0) * 4 + i4) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 u = sred[(((wm * 3 + tl) * 4 + w) * 4 + i4) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += u[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) st[tl][4 * i4 + e] = v[e];
        }
    RGN_QT(hslot * 8 + 4)
    // softmax over keys for the lane's two queries (l31 and 32 + l31); every wave of the sample does the same work
    float inv[2];
#pragma unroll
    for (int qtile = 0; qtile < 2; ++qtile) {
        const int q = 32 * qtile + l31;
        float mx = -INFINITY;
#pragma unroll
        for (int tl = qtile; tl <= 2 * qtile; ++tl) {             // tiles {0} for query tile 0, {1, 2} for query tile 1
            const int kj = tl >> 1;
            // key of register i = 32 kj + 4 kh + c_i, c_i = (i & 3) + 8 (i >> 2); visible iff key <= min(q, Tq - 1) (tile 1 - keys 0-31,
            // queries 32-63 - lies below the diagonal: only key < Tq). ONE per-lane limit, made opaque here: written as 48 compares of
            // loop-invariant values they are hoisted out of the head loop and their lane masks live in ~70 SGPRs spilled to VGPR lanes
            int lim = (tl == 1 ? Tq - 1 : (q < Tq - 1 ? q : Tq - 1)) - 32 * kj - 4 * kh;
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                st[tl][i] = ((i & 3) + 8 * (i >> 2) <= lim) ? st[tl][i] : -INFINITY;
                mx = fmaxf(mx, st[tl][i]);
            }
        }
        mx = half_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int tl = qtile; tl <= 2 * qtile; ++tl)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __builtin_amdgcn_exp2f(st[tl][i] - mx);
                st[tl][i] = e;
                sum += e;
            }
        sum = half_sum(sum);
        inv[qtile] = 1.0f / sum;
    }
    RGN_QT(hslot * 8 + 5)
    // O^T[dh tile wn, queries] = V (A operand, registers = keys) x P^T (B operand, registers = keys)
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
                op8 ph, pl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = st[tl][8 * sl + j];
                    ph[j] = (op_t)x;
                    pl[j] = (op_t)(x - (float)ph[j]);
                }
                if (X3) {
                    oa = OP::mfma(vl[kj][sl], ph, oa);
                    oa = OP::mfma(vh[kj][sl], pl, oa);
                }
                oa = OP::mfma(vh[kj][sl], ph, oa);
            }
        }
        // O^T tile: lane = query (column), registers = 16 dh indices -> 4 runs of 4 consecutive dh = 8-byte plane
        // stores; the 32 x 32 tile is one contiguous 2 KiB run of the K32-blocked plane
        const int q = 32 * qtile + l31;
        if constexpr (!X3) {
            // plain-bf16 phase (hi plane only): the lane pair (l31, kh = 0 / 1) holds the two 8-byte halves of every 16-byte chunk of the
            // row - one v_permlane32_swap per register pairs them up, and the row goes out as two 16-byte WRITE-THROUGH stores per lane
            // (sc1: nothing of the plane stays dirty in L2 for the end-of-kernel write-back; an 8-byte sc1 store costs 2.7x per byte)
            u32x2 run[4];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                op4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (op_t)(oa[4 * i4 + e] * inv[qtile]);
                run[i4] = __builtin_bit_cast(u32x2, hv);
            }
            const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(g.out.hi, 0, (int)((size_t)g.out.rows * g.d * 2), 0x00020000);
            const size_t o = ((size_t)(hd * (QA_DH / 32) + wn) * g.out.rows + row0 + q) * 32 + 8 * kh;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {          // runs (2 pr, 2 pr + 1) = elements 16 pr + 4 kh + {0..3} and 16 pr + 8 + 4 kh + {0..3}
                u32x4 v;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    // after the swap: lanes kh = 0 hold (own even run, partner's even run), lanes kh = 1 (partner's odd run, own odd run)
                    const auto sw = __builtin_amdgcn_permlane32_swap(run[2 * pr][w], run[2 * pr + 1][w], false, false);
                    v[w] = sw[0];
                    v[2 + w] = sw[1];
                }
                if (live && q < Tq) __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, (int)((o + 16 * pr) * 2), 0, RGN_QA_ST_AUX);
            }
        } else if (live && q < Tq) {
            const size_t o = ((size_t)(hd * (QA_DH / 32) + wn) * g.out.rows + row0 + q) * 32 + 4 * kh;
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                op4 hv, lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = oa[4 * i4 + e] * inv[qtile];
                    hv[e] = (op_t)x;
                    lv[e] = (op_t)(x - (float)hv[e]);
                }
                *reinterpret_cast<op4*>(reinterpret_cast<op_t*>(g.out.hi) + o + 8 * i4) = hv;
                if (g.out.lo) *reinterpret_cast<op4*>(reinterpret_cast<op_t*>(g.out.lo) + o + 8 * i4) = lv;
            }
        }
    }
}

template <bool X3>
__global__ __launch_bounds__(QA_NT, 1) void k_qkv_attn(QkvAttnArgs g) {
    constexpr int NPL = X3 ? 2 : 1;
    constexpr int A_BYTES = QA_NS * QA_ROWS * 64, W_BYTES = QA_WROWS * 64;
    constexpr int STAGE = NPL * (A_BYTES + W_BYTES);                 // 64 KiB (x3), 32 KiB (plain bf16)
    // pipeline depth: the same 128 KiB hold two split-bf16 stages or four plain-bf16 ones; what a CU pulls from L2 is set by
    // the bytes it keeps in flight (tools/l2_paths_bench), so the plain-bf16 build prefetches three tiles ahead
    constexpr int NSTG = X3 ? 2 : 4;
    constexpr int W_IT = QA_WROWS * 4 / QA_NT;                       // 3
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                         // GEMM phase roles: wm = sample of the pair
    const int l31 = lane & 31, kh = lane >> 5;
    const int b0 = xcd_affine(blockIdx.x, gridDim.x) * QA_NS, Tq = g.Tq, d = g.d;   // (gridDim.x = sample pairs; id = y * gridDim.x + x)
    const int hpb = g.H / (int)gridDim.y, hd0 = blockIdx.y * hpb;    // heads of this workgroup
    const int nsamp = g.Bm - b0 < QA_NS ? g.Bm - b0 : QA_NS;         // an odd batch leaves the last pair half empty

    size_t a_src;
    {
        const int r = tid >> 2, c = (tid & 3) ^ ((r >> 2) & 3);      // tile row r: sample r / 64, token r % 64
        const int sm = (r >> 6) < nsamp ? (r >> 6) : 0, tk = r & 63;
        const int rr = tk < Tq ? tk : Tq - 1;                        // padding rows replicate the last token (masked later)
        a_src = ((size_t)(b0 + sm) * Tq + rr) * 32 + c * 8;
    }
    int a_off[2][2], w_off[3][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rr = wm * 64 + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int rr = t * QA_DH + wn * 32 + l31;                    // t = 0 / 1 / 2: the wave's 32 columns of q / k / v
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    const int nk = g.Kp / 32;
    float* bias_s = reinterpret_cast<float*>(smem + 4 * (A_BYTES + W_BYTES));   // [hpb][3][128] in_proj biases, past the stages / operand buffers
    for (int i = tid; i < hpb * QA_WROWS; i += QA_NT) {
        const int hh = i / QA_WROWS, r = i - hh * QA_WROWS;
        bias_s[i] = g.bias[(r >> 7) * d + (hd0 + hh) * QA_DH + (r & 127)];
    }
    // (visible to every wave after the first barrier of the k-loop)

    for (int hd = hd0; hd < hd0 + hpb; ++hd) {
        // ---------------- GEMM: [128 x Kp] . W_h[384 x Kp]^T ----------------------------------------------------
        size_t w_src[W_IT];
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int q = it * QA_NT + tid, r = q >> 2, c = (q & 3) ^ ((r >> 2) & 3);
            const int which = r >> 7, j = r & 127;
            w_src[it] = ((size_t)which * d + hd * QA_DH + j) * 32 + c * 8;
        }
        // DMA piece idx (compile-time after unrolling) of tile kt into stage buffer sb: A hi/lo, then the 3 W pieces hi/lo
        auto piece = [&](int idx, int kt, char* sb) {
            const size_t ka = (size_t)kt * g.a_rows * 32, kw = (size_t)kt * 3 * d * 32;
            if (idx < NPL) {
                const int lo = (tid & ~63) * 16;
                __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)((idx ? g.Alo : g.Ahi) + a_src + ka), (RGN_AS3 void*)(sb + idx * A_BYTES + lo), 16, 0, 0);
            } else {
                const int j = idx - NPL, it = j / NPL, pl = j % NPL;
                const int lo = (it * QA_NT + (tid & ~63)) * 16;
                __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)((pl ? g.Wlo : g.Whi) + w_src[it] + kw),
                                                 (RGN_AS3 void*)(sb + NPL * A_BYTES + pl * W_BYTES + lo), 16, 0, 0);
            }
        };
        constexpr int LPT = NPL * (1 + W_IT);                        // 8 (x3)
        f32x16 acc[2][3];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ta][t][i] = 0.f;
        RGN_QT((hd - hd0) * 8 + 0)
#pragma unroll
        for (int t0 = 0; t0 < NSTG - 1; ++t0)
#pragma unroll
            for (int idx = 0; idx < LPT; ++idx) piece(idx, t0, smem + t0 * STAGE);   // nk >= NSTG - 1 (host-checked)
        // One barrier per k-step: tile kt+1's pieces are issued two at a time behind the MFMA groups of the first K half of
        // tile kt (a back-to-back burst would stall the in-order wave for the whole queue of the CU's vector-memory
        // path), into the stage every wave finished reading before this step's barrier; fragments are fetched one MFMA
        // group (one W tile x both A tiles) ahead.
        for (int kt = 0; kt < nk; ++kt) {
            // my pieces of tile kt landed; tiles kt+1 .. kt+NSTG-2 stay in flight (fewer near the end of the loop)
            if (NSTG == 2 || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (kt + 2 >= nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // LPT = 4 in the plain-bf16 build
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // ... everyone's did, and everyone left step kt-1
            const char* sb = smem + (kt % NSTG) * STAGE;
            char* nb = smem + ((kt + NSTG - 1) % NSTG) * STAGE;
            const bool more = kt + NSTG - 1 < nk;
            bf16x8 ah[2][2], al[2][2], wh[2], wl[2];
            auto fetch = [&](int grp) {                             // grp = ks * 3 + t
                const int ks = grp / 3, t = grp % 3;
                if (t == 0) {
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta) {
                        ah[ks][ta] = *reinterpret_cast<const bf16x8*>(sb + a_off[ta][ks]);
                        if (X3) al[ks][ta] = *reinterpret_cast<const bf16x8*>(sb + A_BYTES + a_off[ta][ks]);
                    }
                }
                wh[grp & 1] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + w_off[t][ks]);
                if (X3) wl[grp & 1] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + W_BYTES + w_off[t][ks]);
            };
            fetch(0);
#pragma unroll
            for (int grp = 0; grp < 6; ++grp) {
                if (grp + 1 < 6) fetch(grp + 1);
                const int ks = grp / 3, t = grp % 3;
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    if (t < 2) {   // q, k tiles transposed (lane = token, registers = dh): 8-byte LDS writes below
                        if (X3) {
                            acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[grp & 1], al[ks][ta], acc[ta][t], 0, 0, 0);
                            acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[grp & 1], ah[ks][ta], acc[ta][t], 0, 0, 0);
                        }
                        acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[grp & 1], ah[ks][ta], acc[ta][t], 0, 0, 0);
                    } else {       // v tile: lane = dh, registers = tokens (v goes to LDS transposed)
                        if (X3) {
                            acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][ta], wh[grp & 1], acc[ta][t], 0, 0, 0);
                            acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][ta], wl[grp & 1], acc[ta][t], 0, 0, 0);
                        }
                        acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][ta], wh[grp & 1], acc[ta][t], 0, 0, 0);
                    }
                }
                if (more && grp < 4) {
#pragma unroll
                    for (int q = 0; q < LPT / 4; ++q) piece(grp * (LPT / 4) + q, kt + NSTG - 1, nb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave is done reading the last stage
        RGN_QT((hd - hd0) * 8 + 1)
        qa_attention<X3>(acc, g, smem, bias_s + (hd - hd0) * QA_WROWS + wn * 32, hd, hd - hd0, wm, wn, nsamp, b0, lane, tid);
        RGN_QT((hd - hd0) * 8 + 6)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // the next head's DMA overwrites the reduction buffer
        RGN_QT((hd - hd0) * 8 + 3)
    }
}

// ---- plain-bf16 phase, d = 512: the same kernel with the in_proj weights streamed into REGISTERS -------------------------
// The DMA-fed loop above moves 112 KiB through LDS per k-step (32 KiB of DMA writes + 80 KiB of fragment reads, of which
// 48 KiB are the W fragments) against the 128 B/clk the LDS delivers: ~1400 clk per k-step for 768 clk of MFMA work
// (tools/qkv_attn_bench -DRGN_QA_PROF). Here every wave loads its own q/k/v weight fragments straight from the
// fragment-ordered plane ([Kp/32][3d/32][2 ks][64 lanes][8], one contiguous 1 KiB run per wave-load, as in k_rowgemm)
// into a 4-slot register ring, 3 k-steps ahead; only the activation tile (8 KiB per k-step, loaded to registers 3 steps
// ahead as well and written to a 2-stage LDS ring one step ahead) still passes through LDS. No direct-to-LDS DMA in the
// loop, and the 16 k-steps are fully unrolled: the compiler's own s_waitcnt bookkeeping then stays exact (it drains
// vmcnt at loop back-edges and before the first LDS read behind a DMA).
constexpr int QR_NK = 16, QR_WQ = 18;                          // weight fragment ring (6 fragments per k-step)
constexpr int QR_DA = 4, QR_ARING = QR_DA + 1;                // activation pieces: QR_DA ahead (they must be in LDS one step early)
// NS = samples per workgroup (4 waves each). LDS: [exchange buffer NS x 48 KiB | activation ring [2][NS x 64 rows][64 B] | biases].
// NS = 1 is the build for full launches: two INDEPENDENT 4-wave workgroups per CU (one wave each per SIMD) instead of one
// 8-wave workgroup whose two waves per SIMD move through the phases in lockstep - the attention phase and the ring fill of one
// workgroup (no MFMA work, no weight requests) then run under the other's k-loop.
template <int NS> constexpr int qr_abuf() { return NS * 48 * 1024; }
template <int NS> constexpr int qr_lds() { return qr_abuf<NS>() + 2 * NS * QA_ROWS * 64 + 8 * QA_WROWS * 4; }
template <int NS, bool F16 = false>
__global__ __launch_bounds__(NS * 256, 2 / NS) void k_qkv_attn_rs(QkvAttnArgs g, const __bf16* __restrict__ Wfr) {
    using OP = OpFmt<F16>;                // bf16 or fp16 operands (rgn_internal.h): input plane, weight plane, q / k / v / p, output plane
    using op8 = typename OP::v8;
    constexpr int NT = NS * 256, STAGE = NS * QA_ROWS * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, kh = lane >> 5;
    const int b0 = xcd_affine(blockIdx.x, gridDim.x) * NS, Tq = g.Tq, d = g.d;   // (gridDim.x = sample groups; id = y * gridDim.x + x)
    const int hpb = g.H / (int)gridDim.y, hd0 = blockIdx.y * hpb;
    const int nsamp = g.Bm - b0 < NS ? g.Bm - b0 : NS;
    RGN_QL(0)
    constexpr int nb_all = 3 * 512 / 32;                             // d = 512 (launch_qkv_attn): compile-time weight offsets - as run-time
                                                                     // values the 96 k-step offsets of a head live in SGPRs and spill to VGPR lanes

    unsigned a_voff;                                                 // this thread's 16 bytes of every activation k-block (elements)
    {
        const int r = tid >> 2, c = (tid & 3) ^ ((r >> 2) & 3);      // tile row r: sample r / 64, token r % 64
        const int sm = (r >> 6) < nsamp ? (r >> 6) : 0, tk = r & 63;
        const int rr = tk < Tq ? tk : Tq - 1;                        // padding rows replicate the last token (masked later)
        a_voff = (unsigned)((b0 + sm) * Tq + rr) * 32u + c * 8;
    }
    // buffer loads: descriptor + uniform k-step offset in SGPRs, one 32-bit VGPR per lane address (with flat addresses the
    // compiler materialises a 64-bit address pair per unrolled k-step and spills)
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Ahi), 0, (int)((size_t)g.a_rows * g.Kp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wfr), 0, 3 * d * g.Kp * 2, 0x00020000);
    const unsigned a_kbytes = (unsigned)g.a_rows * 64u;
    int a_off[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rr = wm * 64 + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    char* abuf = smem + qr_abuf<NS>();
    float* bias_s = reinterpret_cast<float*>(abuf + 2 * STAGE);
    for (int i = tid; i < hpb * QA_WROWS; i += NT) {
        const int hh = i / QA_WROWS, r = i - hh * QA_WROWS;
        bias_s[i] = g.bias[(r >> 7) * d + (hd0 + hh) * QA_DH + (r & 127)];
    }

    for (int hd = hd0; hd < hd0 + hpb; ++hd) {
        unsigned wofs[3];                                            // fragment block of this wave's 32 columns of q / k / v (elements)
#pragma unroll
        for (int t = 0; t < 3; ++t) wofs[t] = (unsigned)((t * d + hd * QA_DH) / 32 + wn) * 1024u + lane * 8;
        f32x16 acc[2][3];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ta][t][i] = 0.f;
        RGN_QT((hd - hd0) * 8 + 0)
        // Weight fragments: a ring of QR_WQ = 18 fragments (3 k-steps' worth) recycled ONE AT A TIME: fragment q = 6 kt + 3 ks + t
        // is consumed by two MFMAs and its registers immediately take fragment q + 18 - the same 72 VGPRs as three whole-step
        // slots, but every load is issued three k-steps (not two) ahead of its use and the loads are spread between the MFMAs.
        op8 wq[QR_WQ];
        u32x4 areg[QR_ARING];
        auto issue_a = [&](int kt) { areg[kt % QR_ARING] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_voff * 2, kt * a_kbytes, 0)); };
        auto issue_q = [&](int q) {                                  // q compile-time after unrolling
            const int kt = q / 6, ks = (q % 6) / 3, t = q % 3;
            wq[q % QR_WQ] = __builtin_bit_cast(op8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, wofs[t] * 2, (kt * nb_all * 1024 + ks * 512) * 2, 0));
        };
#pragma unroll
        for (int kt = 0; kt < QR_DA; ++kt) issue_a(kt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < QR_WQ; ++q) issue_q(q);
        __builtin_amdgcn_sched_barrier(0);
        *reinterpret_cast<u32x4*>(abuf + tid * 16) = areg[0];        // stage 0 <- k-block 0
#pragma unroll
        for (int kt = 0; kt < QR_NK; ++kt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // k-block kt is in its stage; everyone left stage (kt + 1) % 2
            const char* sb = abuf + (kt & 1) * STAGE;
            op8 af[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) af[ks][ta] = *reinterpret_cast<const op8*>(sb + a_off[ta][ks]);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + QR_DA < QR_NK) issue_a(kt + QR_DA);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int grp = 0; grp < 6; ++grp) {
                const int ks = grp / 3, t = grp % 3, q = kt * 6 + grp;
#ifdef RGN_QA_LIFE
                if (kt == 0 && grp == 1 && hd == hd0) { RGN_QL(1) }
#endif
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    if (t < 2)     // q, k tiles transposed (lane = token, registers = dh)
                        acc[ta][t] = OP::mfma(wq[q % QR_WQ], af[ks][ta], acc[ta][t]);
                    else           // v tile: lane = dh, registers = tokens
                        acc[ta][t] = OP::mfma(af[ks][ta], wq[q % QR_WQ], acc[ta][t]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (q + QR_WQ < 6 * QR_NK) issue_q(q + QR_WQ);
                if (grp == 2 && kt + 1 < QR_NK)                      // next k-block of the activation tile -> the other stage
                    *reinterpret_cast<u32x4*>(abuf + ((kt + 1) & 1) * STAGE + tid * 16) = areg[(kt + 1) % QR_ARING];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        RGN_QT((hd - hd0) * 8 + 1)
        qa_attention<false, F16>(acc, g, smem, bias_s + (hd - hd0) * QA_WROWS + wn * 32, hd, hd - hd0, wm, wn, nsamp, b0, lane, tid);
        RGN_QT((hd - hd0) * 8 + 6)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RGN_QT((hd - hd0) * 8 + 3)
    }
    RGN_QL(2)
}

// ---- split-bf16 phase, d = 512: the register-streamed form of the SPLIT arithmetic (round 6) ---------------------------------------
// k_qkv_attn<true> above moves 64 KiB of operands per k-step through the ~20 B/clk direct-to-LDS path against 2304 cycles of MFMA (3400 measured):
// the same weight stream through the vector-memory path into register rings, hi and lo fragment planes side by side, leaves only the activation tile
// (hi | lo, 16 KiB per k-step pair of stages) in LDS. One sample (4 waves) per workgroup, two workgroups per CU like k_qkv_attn_rs<1>. Every accumulator
// sees the MFMAs of k_qkv_attn<true> in the same order (k-block, k half; lo x hi, hi x lo, hi x hi): the two forms agree bit for bit
// (tests/test_hip_parity.py), "QKV_X3_DMA" = 1 keeps the direct-to-LDS form. The kernel itself runs 53 us where the direct-to-LDS form ran 73 (rocprofv3, the
// evaluation schedule at B = 256) - and the CALL gains 1 - 3 %: the split steps run three kernel chains side by side and are short of L2 bandwidth chip-wide
// (k_mlp_x3 streams 5.2 MB of weight fragments per 32-row tile), so time a chain's in_proj gives back is taken by its neighbours' layer tails.
constexpr int QX_WQ = 9, QX_DA = 2, QX_ARING = QX_DA + 1;        // ring depths: 1.5 k-steps of (hi, lo) weight fragment pairs, activation pieces 2 ahead
template <int NS> constexpr int qx_lds() { return qr_abuf<NS>() + 2 * 2 * NS * QA_ROWS * 64 + 8 * QA_WROWS * 4; }
// NS = samples per workgroup (4 waves each): 2 = one 8-wave workgroup per CU whose sample halves request the SAME weight fragments at the same time (one L2
// read serves both: half the L2 weight traffic per sample, which is what a full chip of concurrent chains is short of); 1 = two independent workgroups per CU.
template <int NS>
__global__ __launch_bounds__(NS * 256, 2 / NS) void k_qkv_attn_rs_x3(QkvAttnArgs g) {
    constexpr int NT = NS * 256, STAGE = NS * QA_ROWS * 64;          // one plane of one stage; a stage = hi | lo
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, kh = lane >> 5;
    const int b0 = xcd_affine(blockIdx.x, gridDim.x) * NS, Tq = g.Tq, d = g.d;
    const int hpb = g.H / (int)gridDim.y, hd0 = blockIdx.y * hpb;
    const int nsamp = g.Bm - b0 < NS ? g.Bm - b0 : NS;
    constexpr int nb_all = 3 * 512 / 32;
    unsigned a_voff;
    {
        const int r = tid >> 2, c = (tid & 3) ^ ((r >> 2) & 3);      // tile row r: sample r / 64, token r % 64
        const int sm = (r >> 6) < nsamp ? (r >> 6) : 0, tk = r & 63;
        const int rr = tk < Tq ? tk : Tq - 1;                        // padding rows replicate the last token (masked later)
        a_voff = (unsigned)((b0 + sm) * Tq + rr) * 32u + c * 8;
    }
    const __amdgpu_buffer_rsrc_t ah_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Ahi), 0, (int)((size_t)g.a_rows * g.Kp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t al_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Alo), 0, (int)((size_t)g.a_rows * g.Kp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wh_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Wfr), 0, 3 * d * g.Kp * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t wl_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Wfr_lo), 0, 3 * d * g.Kp * 2, 0x00020000);
    const unsigned a_kbytes = (unsigned)g.a_rows * 64u;
    int a_off[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rr = wm * 64 + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    char* abuf = smem + qr_abuf<NS>();                               // [stage 2][plane 2][NS x 64 rows][64 B]
    float* bias_s = reinterpret_cast<float*>(abuf + 4 * STAGE);
    for (int i = tid; i < hpb * QA_WROWS; i += NT) {
        const int hh = i / QA_WROWS, r = i - hh * QA_WROWS;
        bias_s[i] = g.bias[(r >> 7) * d + (hd0 + hh) * QA_DH + (r & 127)];
    }
    for (int hd = hd0; hd < hd0 + hpb; ++hd) {
        unsigned wofs[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) wofs[t] = (unsigned)((t * d + hd * QA_DH) / 32 + wn) * 1024u + lane * 8;
        f32x16 acc[2][3];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ta][t][i] = 0.f;
        bf16x8 wqh[QX_WQ], wql[QX_WQ];
        u32x4 arh[QX_ARING], arl[QX_ARING];
        auto issue_a = [&](int kt) {
            arh[kt % QX_ARING] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ah_rs, a_voff * 2, kt * a_kbytes, 0));
            arl[kt % QX_ARING] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(al_rs, a_voff * 2, kt * a_kbytes, 0));
        };
        auto issue_q = [&](int q) {                                  // q compile-time after unrolling
            const int kt = q / 6, ks = (q % 6) / 3, t = q % 3;
            wqh[q % QX_WQ] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wh_rs, wofs[t] * 2, (kt * nb_all * 1024 + ks * 512) * 2, 0));
            wql[q % QX_WQ] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wl_rs, wofs[t] * 2, (kt * nb_all * 1024 + ks * 512) * 2, 0));
        };
#pragma unroll
        for (int kt = 0; kt < QX_DA; ++kt) issue_a(kt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < QX_WQ; ++q) issue_q(q);
        __builtin_amdgcn_sched_barrier(0);
        *reinterpret_cast<u32x4*>(abuf + tid * 16) = arh[0];         // stage 0 <- k-block 0
        *reinterpret_cast<u32x4*>(abuf + STAGE + tid * 16) = arl[0];
#pragma unroll
        for (int kt = 0; kt < QR_NK; ++kt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // k-block kt is in its stage; everyone left stage (kt + 1) % 2
            const char* sb = abuf + (kt & 1) * 2 * STAGE;
            bf16x8 ah[2][2], al[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    ah[ks][ta] = *reinterpret_cast<const bf16x8*>(sb + a_off[ta][ks]);
                    al[ks][ta] = *reinterpret_cast<const bf16x8*>(sb + STAGE + a_off[ta][ks]);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (kt + QX_DA < QR_NK) issue_a(kt + QX_DA);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int grp = 0; grp < 6; ++grp) {
                const int ks = grp / 3, t = grp % 3, q = kt * 6 + grp;
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    if (t < 2) {   // q, k tiles transposed (lane = token, registers = dh)
                        acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wqh[q % QX_WQ], al[ks][ta], acc[ta][t], 0, 0, 0);
                        acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wql[q % QX_WQ], ah[ks][ta], acc[ta][t], 0, 0, 0);
                        acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wqh[q % QX_WQ], ah[ks][ta], acc[ta][t], 0, 0, 0);
                    } else {       // v tile: lane = dh, registers = tokens
                        acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][ta], wqh[q % QX_WQ], acc[ta][t], 0, 0, 0);
                        acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][ta], wql[q % QX_WQ], acc[ta][t], 0, 0, 0);
                        acc[ta][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][ta], wqh[q % QX_WQ], acc[ta][t], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (q + QX_WQ < 6 * QR_NK) issue_q(q + QX_WQ);
                if (grp == 2 && kt + 1 < QR_NK) {                    // next k-block of the activation tile -> the other stage
                    *reinterpret_cast<u32x4*>(abuf + ((kt + 1) & 1) * 2 * STAGE + tid * 16) = arh[(kt + 1) % QX_ARING];
                    *reinterpret_cast<u32x4*>(abuf + ((kt + 1) & 1) * 2 * STAGE + STAGE + tid * 16) = arl[(kt + 1) % QX_ARING];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // (the exchange buffer does not alias the ring, but every wave must have left the k-loop's last stage reads)
        qa_attention<true>(acc, g, smem, bias_s + (hd - hd0) * QA_WROWS + wn * 32, hd, hd - hd0, wm, wn, nsamp, b0, lane, tid);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

bool qkv_attn_supported(int Tq, int dh, int d) { return Tq <= QA_ROWS && dh == QA_DH && d % 32 == 0 && d / dh <= 8 && d >= 128; }
static int qa_lds(bool x3) { return 2 * (x3 ? 2 : 1) * (QA_NS * QA_ROWS * 64 + QA_WROWS * 64) + 8 * QA_WROWS * 4 /* biases of <= 8 heads (odd H: one workgroup runs them all) */; }
hipError_t configure_qkv_attn() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, qa_lds(true));
    if (e != hipSuccess) return e;
    // (the plain-bf16 build is given the same allocation: its operand buffers, one plane each, alias its 64 KiB of stages)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, qa_lds(true));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn_rs<2>), hipFuncAttributeMaxDynamicSharedMemorySize, qr_lds<2>());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn_rs<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, qr_lds<1>());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn_rs_x3<1>), hipFuncAttributeMaxDynamicSharedMemorySize, qx_lds<1>());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn_rs_x3<2>), hipFuncAttributeMaxDynamicSharedMemorySize, qx_lds<2>());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn_rs<1>), hipFuncAttributeMaxDynamicSharedMemorySize, qr_lds<1>());
}
hipError_t launch_qkv_attn(const QkvAttnArgs& g, bool x3, hipStream_t s) {
    if (!x3 && g.Wfr && g.Kp == 32 * QR_NK && g.d == 512 && (size_t)g.a_rows * g.Kp * 2 < (1ull << 31)) {   // plain-bf16 phase: weights streamed to registers (32-bit buffer offsets)
        // One sample (4 waves) per workgroup, two workgroups per CU: same results bit for bit as the two-sample workgroup, 36.4 -> 35.4 us
        // alone at B = 256, 28 -> 21 us at B = 128, and +3 - 4.5 % on the whole step (tools/qkv_attn_bench; REGENNET_QKV_NS=2 keeps the
        // two-sample build for that comparison).
        static const bool two = getenv("REGENNET_QKV_NS") && atoi(getenv("REGENNET_QKV_NS")) == 2;
        static const int hs_env = getenv("REGENNET_QKV_HSPLIT") ? atoi(getenv("REGENNET_QKV_HSPLIT")) : 0;   // tools
        const int pairs = (g.Bm + QA_NS - 1) / QA_NS;
        // Heads per workgroup: two (half the in_proj weight requests per sample) once the evaluation alone fills the chip with 4-wave
        // workgroups - >= 256 samples over all chains: 512 workgroups, two per CU - and ONE head per workgroup below that (B = 64 / 128 /
        // 192 at 60 frames: 116.9 / 131.0 / 154.1 vs 135.6 / 139.7 / 158.0 ms per 250-step call; B = 256: 360 vs 369 motions/s the other way)
        const int beval = g.Bm_eval > 0 ? g.Bm_eval : g.Bm;
        const int hsplit = (hs_env > 0 && g.H % hs_env == 0) ? hs_env : (beval < 256 || g.H % 2) ? g.H : 2;
        if (g.f16)     // fp16 operands (the schedule's fp16 sub-phase): the one-sample form
            hipLaunchKernelGGL((k_qkv_attn_rs<1, true>), dim3(g.Bm, hsplit), dim3(256), qr_lds<1>(), s, g, g.Wfr);
        else if (!two)
            hipLaunchKernelGGL(k_qkv_attn_rs<1>, dim3(g.Bm, hsplit), dim3(256), qr_lds<1>(), s, g, g.Wfr);
        else
            hipLaunchKernelGGL(k_qkv_attn_rs<2>, dim3(pairs, hsplit), dim3(512), qr_lds<2>(), s, g, g.Wfr);
        return hipGetLastError();
    }
    // heads per workgroup: half of them (the weight stream per sample is what bounds the kernel), but one head each while
    // the launch is small (<= 64 workgroups): a small batch is latency-bound and the heads of a workgroup run back to back
    if (g.f16) return hipErrorInvalidValue;                         // (only the register-streamed plain form has the fp16 instantiation)
    if (x3 && g.Wfr && g.Wfr_lo && g.Alo && g.Kp == 32 * QR_NK && g.d == 512 && (size_t)g.a_rows * g.Kp * 2 < (1ull << 31)) {   // split phase, weights streamed to registers
        const int beval = g.Bm_eval > 0 ? g.Bm_eval : g.Bm;
        const int hsplit = (beval < 256 || g.H % 2) ? g.H : 2;      // (the plain form's rule)
        // one sample per workgroup: 5.71 ms per ddim5 call at B = 256 against 5.79 with two (and 5.78 with the direct-to-LDS form), 3.04 / 3.06 / 3.14 at B = 64
        // (tools/ab_qkv_x3.sh, profiles/r06_qkv_x3_forms.txt; REGENNET_QKV_X3_NS=2 keeps the two-sample build for that comparison)
        static const bool two = getenv("REGENNET_QKV_X3_NS") && atoi(getenv("REGENNET_QKV_X3_NS")) == 2;
        if (!two) hipLaunchKernelGGL(k_qkv_attn_rs_x3<1>, dim3(g.Bm, hsplit), dim3(256), qx_lds<1>(), s, g);
        else hipLaunchKernelGGL(k_qkv_attn_rs_x3<2>, dim3((g.Bm + 1) / 2, hsplit), dim3(512), qx_lds<2>(), s, g);
        return hipGetLastError();
    }
    const int pairs = (g.Bm + QA_NS - 1) / QA_NS;
    const int hsplit = (pairs * g.H <= 64) ? g.H : (g.H % 2 == 0 ? 2 : 1);
    const dim3 grid(pairs, hsplit);
    if (x3)
        hipLaunchKernelGGL((k_qkv_attn<true>), grid, dim3(QA_NT), qa_lds(true), s, g);
    else
        hipLaunchKernelGGL((k_qkv_attn<false>), grid, dim3(QA_NT), qa_lds(true), s, g);
    return hipGetLastError();
}

#ifdef RGN_QA_LIFE
void qa_life_read(long long* out, int n) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qa_life), sizeof(long long) * n); }
#endif
#ifdef RGN_QA_PROF
void qa_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qa_prof), sizeof(long long) * 64); }
#endif

}  // namespace rgn
