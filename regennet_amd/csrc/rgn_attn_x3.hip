// Split-bf16 ("bf16x3") causal self-attention on attention-ready planes emitted by the in_proj GEMM epilogue.
//
// One workgroup per (sample, head), one wave per 32-query tile, v_mfma_f32_32x32x16_bf16 with fp32 accumulate;
// every product of two fp32 quantities a.b is formed as ah.bh + ah.bl + al.bh (hi/lo bf16 planes), like the GEMMs.
// Everything is computed transposed (see k_attn_mfma in rgn_kernels.hip for the fp32 variant of the same scheme):
//   S^T[key, query] = K . Q^T       A = K tile (LDS, rows padded to dh+8 bf16: conflict-free ds_read_b128),
//                                   B = Q fragments loaded once from HBM into registers (pre-scaled by 1/sqrt(dh))
//   softmax over keys               in-register max/sum + one lane^32 exchange
//   O^T[dh, query]  = V^T . P^T     B = P from the S^T accumulator registers (split into hi/lo bf16 in place: the
//                                   C/D layout is the B layout up to a permutation of the key index), A = V^T tile:
//                                   v is transposed on its way into LDS (two keys packed per 4-byte ds_write, lanes
//                                   along keys: conflict-free), so a lane's 8 keys are two 8-byte LDS reads
//                                   (rows padded to Tqp+4 bf16: conflict-free ds_read_b64)
// K and V^T time-share the LDS slab. The result is transposed through the dead slab and written as split planes in
// the K32-blocked layout the out_proj GEMM consumes.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

namespace rgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int NT, int DH, bool X3>
__global__ __launch_bounds__(64 * NT) void k_attn_x3(AttnX3Args a) {
    constexpr int DP = DH < 32 ? 32 : DH;          // dh padded to a multiple of 32 for the PV tiles
    constexpr int NS = DH / 16;                    // k16 steps of S^T (DH is 16, 32, 64 or 128)
    constexpr int ND = DP / 32;                    // 32-wide dh tiles of O^T
    constexpr int TQP = 32 * NT;
    constexpr int KLD = DH + 8;                    // K slab row stride (bf16)
    constexpr int VLD = TQP + 4;                   // V^T slab row stride (bf16)
    constexpr int PLANE = (TQP * KLD > DP * VLD ? TQP * KLD : DP * VLD);   // elements per plane of the shared slab
    constexpr int OLD = DP + 4;                    // fp32 output patch row stride
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* sh = reinterpret_cast<__bf16*>(smem);  // [2 planes][PLANE]
    // XCD-affine mapping (block id -> XCD id%8): every XCD gets one contiguous range of samples, matching the row ranges
    // the in_proj / out_proj GEMM tiles occupy on that XCD
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int b = vid / a.H, hd = vid - b * a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, l31 = lane & 31;
    const int Tq = a.Tq;
    const size_t slab = (size_t)b * a.H + hd;
    const __bf16* gK[2] = {a.Khi + slab * a.Tqp * DH, a.Klo + (X3 ? slab * a.Tqp * DH : 0)};
    const __bf16* gV[2] = {a.Vthi + slab * a.Tqp * DH, a.Vtlo + (X3 ? slab * a.Tqp * DH : 0)};
    constexpr int NPL = X3 ? 2 : 1;

    // ---- K -> LDS (rows >= Tq read as zero)
    for (int p = 0; p < NPL; ++p)
        for (int idx = tid; idx < TQP * (DH / 8); idx += 64 * NT) {
            const int r = idx / (DH / 8), c = (idx - r * (DH / 8)) * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r < Tq) v = *reinterpret_cast<const u32x4*>(gK[p] + (size_t)r * DH + c);
            *reinterpret_cast<u32x4*>(&sh[p * PLANE + r * KLD + c]) = v;
        }
    // ---- Q fragments: lane (query, kh) holds q[query][16 s + 8 kh .. +7] for every k16 step s
    const int qrow = 32 * w + l31;
    bf16x8 qh[NS], ql[NS];
    {
        const __bf16* gq = a.Qhi + (slab * a.Tqp + qrow) * DH + 8 * kh;
        const __bf16* gl = a.Qlo + (X3 ? (slab * a.Tqp + qrow) * DH + 8 * kh : 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            u32x4 z = {0u, 0u, 0u, 0u};
            u32x4 vh = z, vl = z;
            if (qrow < Tq) {
                vh = *reinterpret_cast<const u32x4*>(gq + 16 * s);
                if (X3) vl = *reinterpret_cast<const u32x4*>(gl + 16 * s);
            }
            qh[s] = __builtin_bit_cast(bf16x8, vh);
            ql[s] = __builtin_bit_cast(bf16x8, vl);
        }
    }
    // ---- V rows -> registers now (two keys x 8 dh per item): their global round trip overlaps the K staging and the S^T
    //      phase instead of following the softmax; they go to LDS (transposed) once every wave is done with K
    //      (sequences of up to 64 tokens only: the latency-bound small launches; at 150 tokens the kernel is HBM-bound and
    //      the 32-64 extra live VGPRs cost more occupancy than the overlap returns: 27.3 -> 28.8 us at B=128)
    constexpr int V_IT = ((TQP / 2) * (DH / 8) + 64 * NT - 1) / (64 * NT);
    constexpr bool V_EARLY = NT <= 2;
    u32x4 vreg[NPL][V_IT][2];
    auto load_v = [&] {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = tid + it * 64 * NT;
            const int kp = idx % (TQP / 2), c = (idx / (TQP / 2)) * 8, k0 = 2 * kp;
            const bool in = idx < (TQP / 2) * (DH / 8);
            u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
            if (in && k0 < Tq) v0 = *reinterpret_cast<const u32x4*>(gV[p] + (size_t)k0 * DH + c);
            if (in && k0 + 1 < Tq) v1 = *reinterpret_cast<const u32x4*>(gV[p] + (size_t)(k0 + 1) * DH + c);
            vreg[p][it][0] = v0;
            vreg[p][it][1] = v1;
        }
    };
    if constexpr (V_EARLY) load_v();
    __syncthreads();
    f32x16 st[NT];
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
#pragma unroll
        for (int i = 0; i < 16; ++i) st[kj][i] = 0.f;
        if (kj <= w) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int o = (32 * kj + l31) * KLD + 16 * s + 8 * kh;
                const bf16x8 kfh = *reinterpret_cast<const bf16x8*>(&sh[o]);
                if (X3) {
                    const bf16x8 kfl = *reinterpret_cast<const bf16x8*>(&sh[PLANE + o]);
                    st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qh[s], st[kj], 0, 0, 0);
                    st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, ql[s], st[kj], 0, 0, 0);
                }
                st[kj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qh[s], st[kj], 0, 0, 0);
            }
        }
    }
    // ---- softmax over keys: key = 32 kj + (i&3) + 8 (i>>2) + 4 kh, query = qrow
    float mx = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
        if (kj <= w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 32 * kj + (i & 3) + 8 * (i >> 2) + 4 * kh;
                const bool ok = (key <= qrow) && (key < Tq);
                st[kj][i] = ok ? st[kj][i] : -INFINITY;
                mx = fmaxf(mx, st[kj][i]);
            }
        }
    mx = half_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
        if (kj <= w) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float e = __expf(st[kj][i] - mx);
                st[kj][i] = e;
                sum += e;
            }
        }
    sum = half_sum(sum);
    const float inv = 1.0f / sum;
    __syncthreads();          // every wave is done with K
    // ---- V -> LDS transposed: Vt[dh][key] (row stride VLD); lanes run along key PAIRS so each ds_write_b32 packs
    //      (key, key+1) of one dh column and a wave's writes fall on consecutive banks. Keys >= Tq read as zero.
    if (DP > DH) {   // zero the dh padding rows (DH = 16 only)
        for (int idx = tid; idx < NPL * (DP - DH) * VLD / 2; idx += 64 * NT) {
            const int p = idx / ((DP - DH) * VLD / 2), o = idx - p * ((DP - DH) * VLD / 2);
            reinterpret_cast<unsigned int*>(&sh[p * PLANE + DH * VLD])[o] = 0u;
        }
    }
    if constexpr (!V_EARLY) load_v();
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = tid + it * 64 * NT;
            if (idx >= (TQP / 2) * (DH / 8)) continue;
            const int kp = idx % (TQP / 2), c = (idx / (TQP / 2)) * 8;
            const int k0 = 2 * kp;
            const u32x4 v0 = vreg[p][it][0], v1 = vreg[p][it][1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned int e0 = (v0[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                const unsigned int e1 = (v1[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                *reinterpret_cast<unsigned int*>(&sh[p * PLANE + (c + j) * VLD + k0]) = e0 | (e1 << 16);
            }
        }
    __syncthreads();
    f32x16 oa[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oa[dt][i] = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
        if (kj <= w) {
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                bf16x8 ph, pl;     // B operand: this lane's 8 keys = registers 8*step .. 8*step+7
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = st[kj][8 * step + j];
                    ph[j] = (__bf16)x;
                    pl[j] = (__bf16)(x - (float)ph[j]);
                }
                const int kb = 32 * kj + 16 * step + 4 * kh;      // keys kb..kb+3 and kb+8..kb+11
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const int o = (32 * dt + l31) * VLD + kb;
                    u32x4 vh;
                    vh.lo = *reinterpret_cast<const u32x2*>(&sh[o]);
                    vh.hi = *reinterpret_cast<const u32x2*>(&sh[o + 8]);
                    const bf16x8 vfh = __builtin_bit_cast(bf16x8, vh);
                    if (X3) {
                        u32x4 vl;
                        vl.lo = *reinterpret_cast<const u32x2*>(&sh[PLANE + o]);
                        vl.hi = *reinterpret_cast<const u32x2*>(&sh[PLANE + o + 8]);
                        const bf16x8 vfl = __builtin_bit_cast(bf16x8, vl);
                        oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, oa[dt], 0, 0, 0);
                        oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl, oa[dt], 0, 0, 0);
                    }
                    oa[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph, oa[dt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();          // every wave is done with V^T: reuse the slab as this wave's private fp32 patch
    float* patch = reinterpret_cast<float*>(smem) + w * (32 * OLD);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) patch[l31 * OLD + 32 * dt + (i & 3) + 8 * (i >> 2) + 4 * kh] = oa[dt][i] * inv;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int C4 = DH / 4;
    const size_t row0 = (size_t)b * Tq;
    for (int idx = lane; idx < 32 * C4; idx += 64) {
        const int r = idx / C4, c = (idx - r * C4) * 4;
        const int q = 32 * w + r;
        if (q < Tq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&patch[r * OLD + c]);
            const int col = hd * DH + c;
            const size_t o = ((size_t)(col >> 5) * a.out.rows + row0 + q) * 32 + (col & 31);
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (__bf16)v[e];
                l[e] = (__bf16)(v[e] - (float)h[e]);
            }
            *reinterpret_cast<bf16x4*>(a.out.hi + o) = h;
            if (a.out.lo) *reinterpret_cast<bf16x4*>(a.out.lo + o) = l;
        }
    }
}

template <int NT, int DH>
static size_t ax3_lds() {
    constexpr int DP = DH < 32 ? 32 : DH, TQP = 32 * NT, KLD = DH + 8, VLD = TQP + 4;
    constexpr size_t plane = (size_t)(TQP * KLD > DP * VLD ? TQP * KLD : DP * VLD);
    const size_t slab = 2 * plane * 2, patch = (size_t)NT * 32 * (DP + 4) * 4;
    return slab > patch ? slab : patch;
}
template <int NT, int DH>
static hipError_t ax3_go(const AttnX3Args& a, hipStream_t s, bool cfg) {
    const size_t lds = ax3_lds<NT, DH>();
    if (cfg) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_x3<NT, DH, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_x3<NT, DH, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (a.x3)
        hipLaunchKernelGGL((k_attn_x3<NT, DH, true>), dim3(a.Bm * a.H), dim3(64 * NT), lds, s, a);
    else
        hipLaunchKernelGGL((k_attn_x3<NT, DH, false>), dim3(a.Bm * a.H), dim3(64 * NT), lds, s, a);
    return hipGetLastError();
}
template <int DH>
static hipError_t ax3_nt(int nt, const AttnX3Args& a, hipStream_t s, bool cfg) {
    switch (nt) {
        case 1: return ax3_go<1, DH>(a, s, cfg);
        case 2: return ax3_go<2, DH>(a, s, cfg);
        case 3: return ax3_go<3, DH>(a, s, cfg);
        case 4: return ax3_go<4, DH>(a, s, cfg);
        case 5: return ax3_go<5, DH>(a, s, cfg);
    }
    return hipErrorInvalidValue;
}
static hipError_t ax3(const AttnX3Args& a, hipStream_t s, bool cfg) {
    const int nt = (a.Tq + 31) / 32;
    switch (a.dh) {
        case 16: return ax3_nt<16>(nt, a, s, cfg);
        case 32: return ax3_nt<32>(nt, a, s, cfg);
        case 64: return ax3_nt<64>(nt, a, s, cfg);
        case 128: return ax3_nt<128>(nt, a, s, cfg);
    }
    return hipErrorInvalidValue;
}
bool attn_x3_supported(int Tq, int dh) { return Tq <= 160 && (dh == 16 || dh == 32 || dh == 64 || dh == 128); }
hipError_t configure_attn_x3(int Tq, int dh) {
    AttnX3Args a{};
    a.Tq = Tq;
    a.dh = dh;
    return ax3(a, nullptr, true);
}
hipError_t launch_attn_x3(const AttnX3Args& a, hipStream_t s) { return ax3(a, s, false); }

}  // namespace rgn
