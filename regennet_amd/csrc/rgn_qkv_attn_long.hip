// Fused in_proj GEMM + causal self-attention for LONG sequences (65 .. 160 tokens: Chi3D / text-conditioned, 150), head dim
// 128, d = 512, plain-bf16 phase of the precision schedule.
//
// Replaces, per decoder layer, the pair k_rowgemm<act = 2> (packed in_proj with the attention-ready scatter) + k_attn_x3:
// q, k and v of a head no longer make a round trip through HBM (94 MB read + 3 x 31 MB written per layer at B=128), and the
// in_proj weight stream drops from 1.5 MB per 64-row tile to 0.39 MB per 150-row (sample, head).
//
// One workgroup per (sample, head), 12 waves:
//   GEMM   [160 x 512] . W_h[384 x 512]^T      wave c owns the 32 columns c of [q_h | k_h | v_h] for all five 32-token tiles
//          (5 accumulator tiles). Its weight fragments come straight from the fragment-ordered plane into a 4-slot register
//          ring three k-steps ahead (2 wave-loads per k-step, no duplication between waves); the activation tile (10 KiB per
//          k-step) travels registers -> 2-stage LDS ring as in k_qkv_attn_rs. 16 k-steps, fully unrolled, buffer loads.
//   q, k   accumulated transposed (W as the MFMA A operand: lane = token, registers = dh) and written as [token][dh] slabs;
//   v      accumulated the same way and scattered (2-byte LDS writes) into the transposed slab V^T[dh][token]:
//          the layouts k_attn_x3 stages through LDS, produced in place (Q 42.5 + K 42.5 + V^T 41 KiB).
//   attention (waves 0-4, one 32-query tile each; rgn_attn_x3.hip's scheme): S^T = K . Q^T from the slabs, softmax in
//          registers, O^T = V^T . P^T with P from the S^T accumulators, result transposed through the dead Q/K slabs and stored as
//          the hi plane of the K32-blocked layout the out_proj GEMM consumes.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

namespace rgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int QL_NW = 12, QL_NT = 64 * QL_NW;                        // waves / threads
constexpr int QL_TT = 5, QL_TQP = 32 * QL_TT, QL_DH = 128, QL_NK = 16;
constexpr int QL_KLD = QL_DH + 8, QL_VLD = QL_TQP + 4;               // slab row strides (bf16): conflict-free ds_read_b128 / b64
constexpr int QL_Q = 0, QL_K = QL_Q + QL_TQP * QL_KLD * 2, QL_V = QL_K + QL_TQP * QL_KLD * 2;
constexpr int QL_A = QL_V + QL_DH * QL_VLD * 2;                      // activation ring: 2 stages x [160 rows][64 B]
constexpr int QL_ASTAGE = QL_NT * 16;                                // (every thread moves 16 bytes per k-block: rows 160 .. 191 are never read)
constexpr int QL_BIAS = QL_A + 2 * QL_ASTAGE;                        // [3][128] floats
constexpr int QL_LDS = QL_BIAS + 3 * QL_DH * 4;
constexpr int QL_D = 3, QL_RING = QL_D + 1, QL_DA = 4, QL_ARING = QL_DA + 1;
constexpr int QL_OLD = QL_DH + 4;                                    // fp32 output patch row stride
static_assert(QL_TT * 32 * QL_OLD * 4 <= QL_V, "output patches alias the Q and K slabs");
static_assert(QL_LDS <= 160 * 1024, "LDS");
}  // namespace

#ifdef RGN_QL_PROF
__device__ long long g_ql_prof[16 * 8];   // tools only: phase cycle stamps of the waves of workgroup RGN_QL_PROF
#define RGN_LT(i) if (blockIdx.x == RGN_QL_PROF && lane == 0) g_ql_prof[wave * 8 + (i)] = clock64();
void ql_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ql_prof), sizeof(long long) * 16 * 8); }
#else
#define RGN_LT(i)
#endif

template <bool F16>
__global__ __launch_bounds__(QL_NT, 1) void k_qkv_attn_long(QkvAttnArgs g, const __bf16* __restrict__ Wfr) {
    using OP = OpFmt<F16>;                // bf16 or fp16 operands (rgn_internal.h): input plane, weight plane, q / k / v slabs, p, output plane
    using op_t = typename OP::t;
    using op8 = typename OP::v8;
    using op4 = typename OP::v4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int vid = xcd_affine(blockIdx.x, gridDim.x);
    const int b = vid / g.H, hd = vid - b * g.H;
    const int Tq = g.Tq, d = g.d;
    const int which = wave >> 2, wn = wave & 3;                      // GEMM role: 32 columns wn of q (0) / k (1) / v (2)
    const int nb_all = 3 * d / 32;

    RGN_LT(0)
    float* bias_s = reinterpret_cast<float*>(smem + QL_BIAS);
    if (tid < 3 * QL_DH) bias_s[tid] = g.bias[(tid >> 7) * d + hd * QL_DH + (tid & 127)];

    // ---- GEMM phase ---------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(g.Ahi), 0, (int)((size_t)g.a_rows * g.Kp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wfr), 0, 3 * d * g.Kp * 2, 0x00020000);
    const unsigned a_kbytes = (unsigned)g.a_rows * 64u;
    unsigned a_voff;                                                 // this thread's 16 bytes of every activation k-block
    {
        const int r = tid >> 2, c = (tid & 3) ^ ((r >> 2) & 3);
        const int rr = r < Tq ? r : Tq - 1;                          // padding rows replicate the last token (masked as keys, never stored as queries)
        a_voff = ((unsigned)(b * Tq + rr) * 32u + c * 8) * 2u;
    }
    const unsigned w_voff = ((unsigned)((which * d + hd * QL_DH) / 32 + wn) * 1024u + lane * 8) * 2u;
    int a_off[QL_TT][2];
#pragma unroll
    for (int t = 0; t < QL_TT; ++t) {
        const int rr = t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    f32x16 acc[QL_TT];
#pragma unroll
    for (int t = 0; t < QL_TT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    op8 wf[QL_RING][2];
    u32x4 areg[QL_ARING];
    auto issue_a = [&](int kt) {
        areg[kt % QL_ARING] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_voff, kt * a_kbytes, 0));
    };
    auto issue_w = [&](int kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wf[kt % QL_RING][ks] = __builtin_bit_cast(op8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_voff, (kt * nb_all * 1024 + ks * 512) * 2, 0));
    };
    char* abuf = smem + QL_A;
#pragma unroll
    for (int kt = 0; kt < QL_DA; ++kt) issue_a(kt);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < QL_D; ++kt) issue_w(kt);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<u32x4*>(abuf + tid * 16) = areg[0];
#pragma unroll
    for (int kt = 0; kt < QL_NK; ++kt) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // k-block kt is in its stage; everyone left the other stage
        const char* sb = abuf + (kt & 1) * QL_ASTAGE;
        __builtin_amdgcn_sched_barrier(0);
        if (kt + QL_DA < QL_NK) issue_a(kt + QL_DA);
        if (kt + QL_D < QL_NK) issue_w(kt + QL_D);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < QL_TT; ++t) {
                const op8 af = *reinterpret_cast<const op8*>(sb + a_off[t][ks]);
                acc[t] = OP::mfma(wf[kt % QL_RING][ks], af, acc[t]);   // transposed: lane = token, registers = columns
            }
        if (kt + 1 < QL_NK) *reinterpret_cast<u32x4*>(abuf + ((kt + 1) & 1) * QL_ASTAGE + tid * 16) = areg[(kt + 1) % QL_ARING];
        __builtin_amdgcn_sched_barrier(0);
    }

    RGN_LT(1)
    // ---- accumulators -> the attention slabs (bf16) ---------------------------------------------------------------------
    op_t* sQ = reinterpret_cast<op_t*>(smem + QL_Q);
    op_t* sK = reinterpret_cast<op_t*>(smem + QL_K);
    op_t* sV = reinterpret_cast<op_t*>(smem + QL_V);
    // (q is stored UNSCALED: 1 / sqrt(dh) goes into the softmax's exponent, one FMA where the subtraction was. k is stored WITHOUT its bias:
    //  q . (k + b_k) = q . k + q . b_k adds the same number to every score of a query's row, which the softmax removes - exact in real arithmetic)
    if (which == 0) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_s + wn * 32 + 8 * i4 + 4 * kh);
#pragma unroll
            for (int t = 0; t < QL_TT; ++t) {
                op4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (op_t)(acc[t][4 * i4 + e] + bq[e]);
                *reinterpret_cast<op4*>(sQ + (t * 32 + l31) * QL_KLD + wn * 32 + 8 * i4 + 4 * kh) = h;
            }
        }
    } else if (which == 1) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
            for (int t = 0; t < QL_TT; ++t) {
                op4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (op_t)acc[t][4 * i4 + e];
                *reinterpret_cast<op4*>(sK + (t * 32 + l31) * QL_KLD + wn * 32 + 8 * i4 + 4 * kh) = h;
            }
    } else {   // v -> V^T[dh][token]: 2-byte writes, a wave's 32 lanes (consecutive tokens) fill one 64-byte run
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + 2 * QL_DH + wn * 32 + 8 * i4 + 4 * kh);
#pragma unroll
            for (int t = 0; t < QL_TT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) sV[(wn * 32 + 8 * i4 + 4 * kh + e) * QL_VLD + t * 32 + l31] = (op_t)(acc[t][4 * i4 + e] + bv[e]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    RGN_LT(2)
    // ---- attention. Query tile t needs key tiles 0 .. t (causal): one wave per query tile would leave the wave of tile 4 five key
    //      tiles deep (9.1 k cycles, stamped) with seven waves idle. Nine waves take (query tile, key-tile range) units of at most
    //      two key tiles instead - primaries 0-4 and secondaries 5-8 -
    //          wave   0    1      2      3      4    5    6      7      8
    //          tile   0    1      2      3      4    2    3      4      4
    //          keys   0   0-1    1-2    2-3     4    0   0-1    0-1    2-3
    //      each with its own running maximum / sum (flash-style); a secondary dumps its unnormalised O^T accumulators (a register
    //      image: the primary's lanes hold the same elements) + (max, sum) into the dead V^T slab / activation ring / bias area, and
    //      the primary of the tile merges: O = sum_u e^(m_u - m) O_u / sum_u e^(m_u - m) l_u.
    constexpr int NS = QL_DH / 16, ND = QL_DH / 32;
    constexpr int DUMP = ND * 16 * 64 * 4 + 64 * 8;                  // 16 KiB of accumulators + (m, l) per lane
    static_assert(QL_V + 4 * DUMP <= QL_LDS && QL_TT * 32 * QL_OLD * 4 <= QL_V, "dumps behind the output patches");
    const int w = wave;
    const int qt = w < 5 ? w : (w == 5 ? 2 : (w == 6 ? 3 : 4));
    const int k0 = w < 2 ? 0 : (w < 4 ? w - 1 : (w == 4 ? 4 : (w == 8 ? 2 : 0)));
    const int k1 = w < 5 ? w : (w == 5 ? 0 : (w == 8 ? 3 : 1));
    const int qrow = 32 * qt + l31;
    const float qs2 = g.qscale * 1.44269504088896340736f;          // 1 / sqrt(dh) in log2 units: the scores stay unscaled, exp2(qs2 s - qs2 max)
    f32x16 oa[ND];
    float inv = 0.f, m_run = -INFINITY, l_run = 0.f;
    if (w < 9) {
        op8 qh[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) qh[s] = *reinterpret_cast<const op8*>(sQ + qrow * QL_KLD + 16 * s + 8 * kh);
        // key tiles one at a time with a running maximum / sum (flash-style): one S^T tile of registers instead of five
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oa[dt][i] = 0.f;
#pragma unroll
        for (int kj = 0; kj < QL_TT; ++kj) {
            if (kj >= k0 && kj <= k1) {
                f32x16 st;
#pragma unroll
                for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const op8 kf = *reinterpret_cast<const op8*>(sK + (32 * kj + l31) * QL_KLD + 16 * s + 8 * kh);
                    st = OP::mfma(kf, qh[s], st);
                }
                // key = 32 kj + (i&3) + 8 (i>>2) + 4 kh, query = qrow (a unit's first key tile holds a valid key for every real
                // query - key 0, or the keys below the query's own tile, or the diagonal - so the maximum is finite)
                float mt = -INFINITY;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kj + (i & 3) + 8 * (i >> 2) + 4 * kh;
                    const bool ok = (key <= qrow) && (key < Tq);
                    st[i] = ok ? st[i] : -INFINITY;
                    mt = fmaxf(mt, st[i]);
                }
                mt = half_max(mt);                                    // (one v_permlane32_swap instead of a ds_bpermute round trip: rgn_internal.h)
                const float m_new = fmaxf(fmaxf(m_run, mt), -1e30f);   // (a unit with no valid key at all - rows beyond Tq only - stays finite)
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * qs2);   // (first tile: exp2(-inf) = 0)
                const float nm = -m_new * qs2;
                float ls = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    st[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[i], qs2, nm));
                    ls += st[i];
                }
                l_run = l_run * alpha + ls;
                m_run = m_new;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oa[dt][i] *= alpha;
#pragma unroll
                for (int step = 0; step < 2; ++step) {
                    op8 ph;     // B operand: this lane's 8 keys = registers 8*step .. 8*step+7
#pragma unroll
                    for (int j = 0; j < 8; ++j) ph[j] = (op_t)st[8 * step + j];
                    const int kb = 32 * kj + 16 * step + 4 * kh;      // keys kb..kb+3 and kb+8..kb+11
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt) {
                        const int o = (32 * dt + l31) * QL_VLD + kb;
                        u32x4 vh;
                        vh.lo = *reinterpret_cast<const u32x2*>(sV + o);
                        vh.hi = *reinterpret_cast<const u32x2*>(sV + o + 8);
                        oa[dt] = OP::mfma(__builtin_bit_cast(op8, vh), ph, oa[dt]);
                    }
                }
            }
        }
        l_run = half_sum(l_run);
    }
    RGN_LT(3)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave is done with Q / K / V^T: patches over Q and K, dumps behind them
    if (w >= 5 && w < 9) {
        char* dump = smem + QL_V + (w - 5) * DUMP;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                *reinterpret_cast<f32x4*>(dump + ((dt * 4 + i4) * 64 + lane) * 16) = f32x4{oa[dt][4 * i4], oa[dt][4 * i4 + 1], oa[dt][4 * i4 + 2], oa[dt][4 * i4 + 3]};
        *reinterpret_cast<float2*>(dump + ND * 16 * 64 * 4 + lane * 8) = float2{m_run, l_run};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (w < QL_TT) {
        if (w >= 2) {                                                // tiles 2, 3: one secondary (dumps 0, 1); tile 4: two (dumps 2, 3)
            const int d0 = w == 4 ? 2 : w - 2, nd = w == 4 ? 2 : 1;
            float ms[2], lsec[2], m = m_run;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float2 st2 = *reinterpret_cast<const float2*>(smem + QL_V + (d0 + (u < nd ? u : 0)) * DUMP + ND * 16 * 64 * 4 + lane * 8);
                ms[u] = st2.x;
                lsec[u] = st2.y;
                if (u < nd) m = fmaxf(m, ms[u]);
            }
            const float fp = __builtin_amdgcn_exp2f((m_run - m) * qs2);
            l_run *= fp;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) oa[dt][i] *= fp;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u < nd) {                                        // wave-uniform
                    const float fs = __builtin_amdgcn_exp2f((ms[u] - m) * qs2);
                    l_run = __builtin_fmaf(lsec[u], fs, l_run);
                    const char* dump = smem + QL_V + (d0 + u) * DUMP;
#pragma unroll
                    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(dump + ((dt * 4 + i4) * 64 + lane) * 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) oa[dt][4 * i4 + e] = __builtin_fmaf(v[e], fs, oa[dt][4 * i4 + e]);
                        }
                }
        }
        inv = 1.0f / l_run;
        float* patch = reinterpret_cast<float*>(smem) + w * (32 * QL_OLD);
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) patch[l31 * QL_OLD + 32 * dt + (i & 3) + 8 * (i >> 2) + 4 * kh] = oa[dt][i] * inv;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // 16-byte WRITE-THROUGH stores (sc1): the plane is not left dirty in the XCD L2s for the end-of-kernel write-back
        constexpr int C8 = QL_DH / 8;
        const size_t row0 = (size_t)b * Tq;
        const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(g.out.hi, 0, (int)((size_t)g.out.rows * g.d * 2), 0x00020000);
#pragma unroll
        for (int it = 0; it < 32 * C8 / 64; ++it) {
            const int idx = lane + 64 * it, r = idx / C8, c = (idx - r * C8) * 8;
            const int q = 32 * w + r;
            if (q < Tq) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&patch[r * QL_OLD + c]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&patch[r * QL_OLD + c + 4]);
                const int col = hd * QL_DH + c;
                const size_t o = ((size_t)(col >> 5) * g.out.rows + row0 + q) * 32 + (col & 31);
                op8 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = (op_t)v0[e]; h[4 + e] = (op_t)v1[e]; }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), o_rs, (int)(o * 2), 0, 16);
            }
        }
    }
    RGN_LT(4)
}

bool qkv_attn_long_supported(int Tq, int dh, int d) { return Tq > 64 && Tq <= QL_TQP && dh == QL_DH && d == 32 * QL_NK; }
hipError_t configure_qkv_attn_long() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn_long<false>), hipFuncAttributeMaxDynamicSharedMemorySize, QL_LDS);
    return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn_long<true>), hipFuncAttributeMaxDynamicSharedMemorySize, QL_LDS);
}
hipError_t launch_qkv_attn_long(const QkvAttnArgs& g, hipStream_t s) {
    if (!g.Wfr || g.Kp != 32 * QL_NK || g.out.lo || (size_t)g.a_rows * g.Kp * 2 >= (1ull << 31)) return hipErrorInvalidValue;   // (32-bit buffer offsets)
    if (g.f16) hipLaunchKernelGGL(k_qkv_attn_long<true>, dim3(g.Bm * g.H), dim3(QL_NT), QL_LDS, s, g, g.Wfr);
    else hipLaunchKernelGGL(k_qkv_attn_long<false>, dim3(g.Bm * g.H), dim3(QL_NT), QL_LDS, s, g, g.Wfr);
    return hipGetLastError();
}

}  // namespace rgn
