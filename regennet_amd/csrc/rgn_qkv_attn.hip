// Fused in_proj GEMM + causal self-attention, one workgroup per sample (Tq <= 64 tokens, head dim 128).
//
// Replaces, per decoder layer, the packed in_proj Linear of nn.MultiheadAttention and the attention itself
// (model/cmdm.py:227 -> TransformerDecoderLayer._sa_block with generate_square_subsequent_mask :168-171) WITHOUT ever
// writing q, k, v to HBM: for each head the workgroup computes [q_h | k_h | v_h] = x . W_h^T (64 x 384, split-bf16 MFMA,
// operands by direct-to-LDS DMA from the K32-blocked planes exactly as in k_gemm_x3), converts the accumulators to
// hi/lo bf16 straight into LDS (q, k token-major with padded rows; v transposed), runs the transposed attention of
// k_attn_x3 on them (S^T = K.Q^T, in-register softmax, O^T = V^T.P^T) and stores the head's output as split planes in
// the K32-blocked layout the out_proj GEMM consumes. HBM traffic per layer drops from
// (x planes re-read per column tile + 4 B/elt q,k,v write + read) to (x planes once + weights from L2).
//
// 8 waves. GEMM phase: 2 (M) x 4 (N) waves, wave tile 32 x 96 (3 MFMA tiles), two 56 KiB LDS stages
// [A 64x32 | W_h 384x32] x {hi, lo}. Attention phase: wave = (query tile, dh tile); the three LDS operand
// buffers (104 KiB) alias the dead pipeline stages.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

namespace rgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

constexpr int QA_ROWS = 64, QA_DH = 128, QA_WROWS = 3 * QA_DH, QA_NT = 512;
constexpr int QA_KLD = QA_DH + 8;      // q / k row stride in LDS (bf16): conflict-free ds_read_b128
constexpr int QA_VLD = QA_ROWS + 4;    // v^T row stride (bf16): conflict-free ds_read_b64

template <bool X3>
__global__ __launch_bounds__(QA_NT, 2) void k_qkv_attn(QkvAttnArgs g) {
    constexpr int NPL = X3 ? 2 : 1;
    constexpr int A_BYTES = QA_ROWS * 64, W_BYTES = QA_WROWS * 64;
    constexpr int STAGE = NPL * (A_BYTES + W_BYTES);                 // 56 KiB (x3)
    constexpr int W_IT = QA_WROWS * 4 / QA_NT;                       // 3
    constexpr int QK_PLANE = QA_ROWS * QA_KLD, VT_PLANE = QA_DH * QA_VLD;   // elements
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Qs = reinterpret_cast<__bf16*>(smem);                    // [NPL][64][136]
    __bf16* Ks = Qs + 2 * QK_PLANE;                                  // [NPL][64][136]
    __bf16* Vt = Ks + 2 * QK_PLANE;                                  // [NPL][128][68]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                         // GEMM phase roles
    const int qt = wave & 1, dt = wave >> 1;                         // attention phase roles
    const int l31 = lane & 31, kh = lane >> 5;
    const int b = blockIdx.x, Tq = g.Tq, d = g.d;
    const size_t row0 = (size_t)b * Tq;
    const bool a_loader = wave < 4;

    size_t a_src;
    {
        const int r = tid >> 2 & 63, c = (tid & 3) ^ ((r >> 2) & 3);
        const int rr = r < Tq ? r : Tq - 1;                          // padding rows replicate the last token (masked later)
        a_src = (row0 + rr) * 32 + c * 8;
    }
    int a_off[2], w_off[3][2];
    {
        const int rr = wm * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a_off[ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int rr = wn * 96 + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    const int nk = g.Kp / 32;
    const int qrow = qt * 32 + l31;                                  // this lane's query in the attention phase

    for (int hd = 0; hd < g.H; ++hd) {
        // ---------------- GEMM: [64 x Kp] . W_h[384 x Kp]^T ----------------------------------------------------
        size_t w_src[W_IT];
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int q = it * QA_NT + tid, r = q >> 2, c = (q & 3) ^ ((r >> 2) & 3);
            const int which = r >> 7, j = r & 127;
            w_src[it] = ((size_t)which * d + hd * QA_DH + j) * 32 + c * 8;
        }
        auto issue = [&](int kt, int stage) {
            char* sb = smem + stage * STAGE;
            const size_t ka = (size_t)kt * g.a_rows * 32, kw = (size_t)kt * 3 * d * 32;
            if (a_loader) {
                const int lo = (tid & ~63) * 16;
                __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.Ahi + a_src + ka), (RGN_AS3 void*)(sb + lo), 16, 0, 0);
                if (X3)
                    __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.Alo + a_src + ka), (RGN_AS3 void*)(sb + A_BYTES + lo), 16, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < W_IT; ++it) {
                const int lo = (it * QA_NT + (tid & ~63)) * 16;
                __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.Whi + w_src[it] + kw), (RGN_AS3 void*)(sb + NPL * A_BYTES + lo), 16, 0, 0);
                if (X3)
                    __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(g.Wlo + w_src[it] + kw), (RGN_AS3 void*)(sb + NPL * A_BYTES + W_BYTES + lo), 16, 0, 0);
            }
        };
        f32x16 acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        issue(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                issue(kt + 1, (kt + 1) & 1);
                if (a_loader) {
                    if (X3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    if (X3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            const char* sb = smem + (kt & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ah, al, bh[3], bl[3];
                ah = *reinterpret_cast<const bf16x8*>(sb + a_off[ks]);
                if (X3) al = *reinterpret_cast<const bf16x8*>(sb + A_BYTES + a_off[ks]);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    bh[t] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + w_off[t][ks]);
                    if (X3) bl[t] = *reinterpret_cast<const bf16x8*>(sb + NPL * A_BYTES + W_BYTES + w_off[t][ks]);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (X3) {
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[t], 0, 0, 0);
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // ---------------- accumulators -> LDS operand buffers (hi/lo bf16); the stages are dead --------------------
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int ct = wn * 3 + t, which = ct >> 2, dhc = (ct & 3) * 32 + l31;
            const float bias = g.bias[which * d + hd * QA_DH + dhc];
            const float sc = which == 0 ? g.qscale : 1.0f;
            if (which < 2) {
                __bf16* dst = which == 0 ? Qs : Ks;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int tok = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
                    const float x = (acc[t][i] + bias) * sc;
                    const __bf16 h = (__bf16)x;
                    dst[tok * QA_KLD + dhc] = h;
                    if (X3) dst[QK_PLANE + tok * QA_KLD + dhc] = (__bf16)(x - (float)h);
                }
            } else {
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const int tok = wm * 32 + 8 * i4 + 4 * kh;      // 4 consecutive tokens of one dh column
                    bf16x4 hv, lv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc[t][4 * i4 + e] + bias;
                        hv[e] = (__bf16)x;
                        lv[e] = (__bf16)(x - (float)hv[e]);
                    }
                    *reinterpret_cast<bf16x4*>(&Vt[dhc * QA_VLD + tok]) = hv;
                    if (X3) *reinterpret_cast<bf16x4*>(&Vt[VT_PLANE + dhc * QA_VLD + tok]) = lv;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
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
        __builtin_amdgcn_s_barrier();      // the next head's DMA overwrites the operand buffers
    }
}

bool qkv_attn_supported(int Tq, int dh, int d) { return Tq <= QA_ROWS && dh == QA_DH && d % 32 == 0; }
static int qa_lds(bool x3) { return 2 * (x3 ? 2 : 1) * (QA_ROWS * 64 + QA_WROWS * 64); }
hipError_t configure_qkv_attn() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, qa_lds(true));
    if (e != hipSuccess) return e;
    // the plain-bf16 build still needs room for the operand buffers (one plane each: 52 KiB) next to its 56 KiB of stages
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_attn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, qa_lds(true));
}
hipError_t launch_qkv_attn(const QkvAttnArgs& g, bool x3, hipStream_t s) {
    if (x3)
        hipLaunchKernelGGL((k_qkv_attn<true>), dim3(g.Bm), dim3(QA_NT), qa_lds(true), s, g);
    else
        hipLaunchKernelGGL((k_qkv_attn<false>), dim3(g.Bm), dim3(QA_NT), qa_lds(true), s, g);
    return hipGetLastError();
}

}  // namespace rgn
