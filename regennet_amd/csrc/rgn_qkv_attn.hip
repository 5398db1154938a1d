// Fused in_proj GEMM + causal self-attention (Tq <= 64 tokens, head dim 128): q, k, v never go to HBM.
//
// Replaces, per decoder layer, the packed in_proj Linear of nn.MultiheadAttention and the attention itself
// (model/cmdm.py:227 -> TransformerDecoderLayer._sa_block with generate_square_subsequent_mask :168-171).
// A workgroup owns TWO samples and H/2 heads (grid = ceil(Bm/2) x 2). For each of its heads it computes
// [q_h | k_h | v_h] = x . W_h^T for both samples at once (128 x 384, split-bf16 MFMA, operands by direct-to-LDS DMA from
// the K32-blocked planes exactly as in k_gemm_x3), then per sample converts the accumulators to hi/lo bf16 straight into
// LDS (q, k token-major with padded rows; v transposed), runs the transposed attention of k_attn_x3 on them
// (S^T = K.Q^T, in-register softmax, O^T = V^T.P^T) and stores the head's output as split planes in the K32-blocked
// layout the out_proj GEMM consumes.
//
// Why two samples x half the heads instead of one sample x all heads: the kernel is bound by the ~20 B/clk a CU pulls
// from L2 into LDS (see rgn_gemm_x3.hip), and almost all of that is the in_proj weight stream. Pairing samples halves
// the weight bytes per sample (2.0 MiB of DMA per workgroup instead of 3.5 MiB) at the same workgroup count.
//
// 8 waves. GEMM phase: wave = (sample, column tile wn of each of q, k, v), wave tile 64 x 96 (2 x 3 MFMA tiles; the q and k
// tiles are accumulated transposed - W fragment as the MFMA A operand - so their conversion to LDS is 8-byte writes), two 64 KiB LDS stages
// [A 128x32 | W_h 384x32] x {hi, lo}; tile rows 0-63 / 64-127 are the tokens of sample 0 / 1 (padding rows replicate
// the last token and are masked). Attention phase (once per sample): wave = (query tile, dh tile); the three LDS operand
// buffers (104 KiB) alias the dead pipeline stages.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

namespace rgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

constexpr int QA_ROWS = 64, QA_NS = 2, QA_DH = 128, QA_WROWS = 3 * QA_DH, QA_NT = 512;
constexpr int QA_KLD = QA_DH + 8;      // q / k row stride in LDS (bf16): conflict-free ds_read_b128
constexpr int QA_VLD = QA_ROWS + 4;    // v^T row stride (bf16): conflict-free ds_read_b64

#ifdef RGN_QA_PROF
__device__ long long g_qa_prof[64];   // tools only: phase cycle stamps of one workgroup (wave 0)
#define RGN_QT(i) if (blockIdx.x == RGN_QA_PROF && blockIdx.y == 0 && tid == 0) g_qa_prof[i] = clock64();
#else
#define RGN_QT(i)
#endif

template <bool X3>
__global__ __launch_bounds__(QA_NT, 1) void k_qkv_attn(QkvAttnArgs g) {
    constexpr int NPL = X3 ? 2 : 1;
    constexpr int A_BYTES = QA_NS * QA_ROWS * 64, W_BYTES = QA_WROWS * 64;
    constexpr int STAGE = NPL * (A_BYTES + W_BYTES);                 // 64 KiB (x3)
    constexpr int W_IT = QA_WROWS * 4 / QA_NT;                       // 3
    constexpr int QK_PLANE = QA_ROWS * QA_KLD, VT_PLANE = QA_DH * QA_VLD;   // elements
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Qs = reinterpret_cast<__bf16*>(smem);                    // [NPL][64][136]
    __bf16* Ks = Qs + 2 * QK_PLANE;                                  // [NPL][64][136]
    __bf16* Vt = Ks + 2 * QK_PLANE;                                  // [NPL][128][68]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                         // GEMM phase roles: wm = sample of the pair
    // attention phase roles (query tile, dh tile); the causal query tile 1 does twice the score MFMAs of tile 0, so the
    // two waves sharing a SIMD (wave, wave + 4) get one tile of each
    const int qt = (wave ^ (wave >> 2)) & 1, dt = wave >> 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int b0 = blockIdx.x * QA_NS, Tq = g.Tq, d = g.d;
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
    const int qrow = qt * 32 + l31;                                  // this lane's query in the attention phase
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
        for (int idx = 0; idx < LPT; ++idx) piece(idx, 0, smem);
        // One barrier per k-step: tile kt+1's pieces are issued two at a time behind the MFMA groups of the first K half of
        // tile kt (a back-to-back burst would stall the in-order wave for the whole queue of the CU's vector-memory
        // path), into the stage every wave finished reading before this step's barrier; fragments are fetched one MFMA
        // group (one W tile x both A tiles) ahead.
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my pieces of tile kt landed
            __builtin_amdgcn_s_barrier();                          // ... everyone's did, and everyone left step kt-1
            const char* sb = smem + (kt & 1) * STAGE;
            char* nb = smem + ((kt + 1) & 1) * STAGE;
            const bool more = kt + 1 < nk;
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
                    for (int q = 0; q < LPT / 4; ++q) piece(grp * (LPT / 4) + q, kt + 1, nb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave is done reading the last stage
        RGN_QT((hd - hd0) * 8 + 1)
#pragma unroll 1
        for (int sm = 0; sm < nsamp; ++sm) {
        const size_t row0 = (size_t)(b0 + sm) * Tq;
        if (wm == sm) {
#pragma unroll
            for (int ta = 0; ta < 2; ++ta) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {                        // q (pre-scaled), k: lane = token, register quads = 4 dh
                    __bf16* dst = t == 0 ? Qs : Ks;
                    const float sc = t == 0 ? g.qscale : 1.0f;
                    const int tok = ta * 32 + l31;
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const int dhc = wn * 32 + 8 * i4 + 4 * kh;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + (hd - hd0) * QA_WROWS + t * QA_DH + dhc);
                        bf16x4 hv, lv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = (acc[ta][t][4 * i4 + e] + b4[e]) * sc;
                            hv[e] = (__bf16)x;
                            lv[e] = (__bf16)(x - (float)hv[e]);
                        }
                        *reinterpret_cast<bf16x4*>(&dst[tok * QA_KLD + dhc]) = hv;
                        if (X3) *reinterpret_cast<bf16x4*>(&dst[QK_PLANE + tok * QA_KLD + dhc]) = lv;
                    }
                }
                {                                                    // v: lane = dh, register quads = 4 tokens -> v^T rows
                    const int dhc = wn * 32 + l31;
                    const float bias = bias_s[(hd - hd0) * QA_WROWS + 2 * QA_DH + dhc];
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const int tok = ta * 32 + 8 * i4 + 4 * kh;
                        bf16x4 hv, lv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = acc[ta][2][4 * i4 + e] + bias;
                            hv[e] = (__bf16)x;
                            lv[e] = (__bf16)(x - (float)hv[e]);
                        }
                        *reinterpret_cast<bf16x4*>(&Vt[dhc * QA_VLD + tok]) = hv;
                        if (X3) *reinterpret_cast<bf16x4*>(&Vt[VT_PLANE + dhc * QA_VLD + tok]) = lv;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RGN_QT((hd - hd0) * 8 + 2 + 2 * sm)
        // ---------------- attention for this head: wave = (query tile qt, dh tile dt) -----------------------------
        f32x16 st[2];
#pragma unroll
        for (int kj = 0; kj < 2; ++kj) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kj][i] = 0.f;
            if (kj <= qt) {
#pragma unroll
                for (int s = 0; s < QA_DH / 16; ++s) {
                    const int qo = qrow * QA_KLD + 16 * s + 8 * kh, ko = (32 * kj + l31) * QA_KLD + 16 * s + 8 * kh;
                    const bf16x8 qh = *reinterpret_cast<const bf16x8*>(&Qs[qo]);
                    const bf16x8 kfh = *reinterpret_cast<const bf16x8*>(&Ks[ko]);
                    if (X3) {
                        const bf16x8 ql = *reinterpret_cast<const bf16x8*>(&Qs[QK_PLANE + qo]);
                        const bf16x8 kfl = *reinterpret_cast<const bf16x8*>(&Ks[QK_PLANE + ko]);
                        st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qh, st[kj], 0, 0, 0);
                        st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, ql, st[kj], 0, 0, 0);
                    }
                    st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qh, st[kj], 0, 0, 0);
                }
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kj = 0; kj < 2; ++kj)
            if (kj <= qt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kj + (i & 3) + 8 * (i >> 2) + 4 * kh;
                    const bool ok = (key <= qrow) && (key < Tq);
                    st[kj][i] = ok ? st[kj][i] : -INFINITY;
                    mx = fmaxf(mx, st[kj][i]);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kj = 0; kj < 2; ++kj)
            if (kj <= qt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __expf(st[kj][i] - mx);
                    st[kj][i] = e;
                    sum += e;
                }
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        f32x16 oa;
#pragma unroll
        for (int i = 0; i < 16; ++i) oa[i] = 0.f;
#pragma unroll
        for (int kj = 0; kj < 2; ++kj) {
            if (kj <= qt) {
#pragma unroll
                for (int step = 0; step < 2; ++step) {
                    bf16x8 ph, pl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = st[kj][8 * step + j];
                        ph[j] = (__bf16)x;
                        pl[j] = (__bf16)(x - (float)ph[j]);
                    }
                    const int o = (32 * dt + l31) * QA_VLD + 32 * kj + 16 * step + 4 * kh;   // keys o..o+3 and o+8..o+11
                    u32x4 vh;
                    vh.lo = *reinterpret_cast<const u32x2*>(&Vt[o]);
                    vh.hi = *reinterpret_cast<const u32x2*>(&Vt[o + 8]);
                    const bf16x8 vfh = __builtin_bit_cast(bf16x8, vh);
                    if (X3) {
                        u32x4 vl;
                        vl.lo = *reinterpret_cast<const u32x2*>(&Vt[VT_PLANE + o]);
                        vl.hi = *reinterpret_cast<const u32x2*>(&Vt[VT_PLANE + o + 8]);
                        const bf16x8 vfl = __builtin_bit_cast(bf16x8, vl);
                        oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, oa, 0, 0, 0);
                        oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl, oa, 0, 0, 0);
                    }
                    oa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph, oa, 0, 0, 0);
                }
            }
        }
        // O^T tile: lane = query (column), registers = 16 dh indices -> 4 runs of 4 consecutive dh = 8-byte plane stores;
        // the 32 x 32 tile is one contiguous 2 KiB run of the K32-blocked plane
        if (qrow < Tq) {
            const size_t o = ((size_t)(hd * (QA_DH / 32) + dt) * g.out.rows + row0 + qrow) * 32 + 4 * kh;
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                bf16x4 hv, lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = oa[4 * i4 + e] * inv;
                    hv[e] = (__bf16)x;
                    lv[e] = (__bf16)(x - (float)hv[e]);
                }
                *reinterpret_cast<bf16x4*>(g.out.hi + o + 8 * i4) = hv;
                if (g.out.lo) *reinterpret_cast<bf16x4*>(g.out.lo + o + 8 * i4) = lv;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // the other sample's operands / the next head's DMA overwrite the buffers
        RGN_QT((hd - hd0) * 8 + 3 + 2 * sm)
        }
    }
}

bool qkv_attn_supported(int Tq, int dh, int d) { return Tq <= QA_ROWS && dh == QA_DH && d % 32 == 0 && d / dh <= 8; }
static int qa_lds(bool x3) { return 2 * (x3 ? 2 : 1) * (QA_NS * QA_ROWS * 64 + QA_WROWS * 64) + 8 * QA_WROWS * 4 /* biases of <= 8 heads (odd H: one workgroup runs them all) */; }
hipError_t configure_qkv_attn() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, qa_lds(true));
    if (e != hipSuccess) return e;
    // (the plain-bf16 build is given the same allocation: its operand buffers, one plane each, alias its 64 KiB of stages)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, qa_lds(true));
}
hipError_t launch_qkv_attn(const QkvAttnArgs& g, bool x3, hipStream_t s) {
    // heads per workgroup: half of them (the weight stream per sample is what bounds the kernel), but one head each while
    // the launch is small (<= 64 workgroups): a small batch is latency-bound and the heads of a workgroup run back to back
    const int pairs = (g.Bm + QA_NS - 1) / QA_NS;
    const int hsplit = (pairs * g.H <= 64) ? g.H : (g.H % 2 == 0 ? 2 : 1);
    const dim3 grid(pairs, hsplit);
    if (x3)
        hipLaunchKernelGGL((k_qkv_attn<true>), grid, dim3(QA_NT), qa_lds(true), s, g);
    else
        hipLaunchKernelGGL((k_qkv_attn<false>), grid, dim3(QA_NT), qa_lds(true), s, g);
    return hipGetLastError();
}

#ifdef RGN_QA_PROF
void qa_prof_read(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qa_prof), sizeof(long long) * 64); }
#endif

}  // namespace rgn
