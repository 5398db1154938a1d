// The ST-GCN recogniser's fused kernels (host side: rgn_stgcn.hip; design: DESIGN.md section 8b). All of them are split-bf16 MFMA GEMMs on the
// K32-blocked operand planes of rgn_gemm_x3.hip - [C/32][rows][32] bf16, hi and lo - with rows = (sequence, frame, vertex):
//   k_sg_gcn        graph aggregation + 1 x 1 convolution: the aggregated operand is formed in registers, in MFMA fragment layout
//   k_sg_tconv      9 x 1 temporal convolution of a stride-1 block: the activation window of a tile resident in LDS, nine taps V rows apart
//   k_sg_tconv_s2   ... of a stride-2 block at the output rate on polyphase planes, + the block's convolved shortcut
//   sg_epilogue     their common epilogue: bias (per column or per vertex), identity residual, ReLU, fp32 or split-plane output
// LDS images, DMA (global_load_lds_dwordx4, 16 rows x 64 B per wave-instruction) and the 16-byte chunk swizzle c ^ ((row >> 2) & 3) are those of
// k_gemm_x3 (rgn_gemm_x3.hip's header). All three kernels are PERSISTENT over output tiles.
#include "rgn_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

namespace rgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// F16 (every kernel of this file): the SINGLE-PLANE fp16 form - the hi planes of activations, residual, weights and output hold IEEE fp16, the lo planes
// are neither read nor written, one v_mfma_f32_32x32x16_f16 per product instead of three bf16 ones: 2^-12 per operand instead of ~2^-16 per product
// (rgn_stgcn.hip: SG_F16). The LDS images, the DMA schedule and every hazard stay those of the split form: what was the lo plane of channel block cb
// (window and weight tile) is the fp16 plane of channel block cb + 1 - a k-step is 64 channels deep (two MFMAs per fragment pair instead of three),
// there are half as many of them, and the per-step costs that bound these kernels (DMA wait, barrier, DMA issue) are paid half as often. Channel
// block counts are even (64 / 128 / 256 channels).
__device__ __forceinline__ float f16_bits_to_float(unsigned b) { return (float)__builtin_bit_cast(_Float16, (unsigned short)b); }

#define RGN_AS1 __attribute__((address_space(1)))
#define RGN_AS3 __attribute__((address_space(3)))

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // at most N vector-memory operations of this wave still outstanding (the six-bit counter saturates at 63)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
}

// ---- the ST-GCN kernels' epilogue: out = act(acc + bias + addend), as fp32 [M, ldc] or as split planes. MODE (compile time):
//   SGE_VERTEX_BIAS  the bias is a per-vertex row add[(row % add_mod)][n] (the graph convolution's, rgn_stgcn.hip); else bias[n]
//   SGE_RES_PLANES   + the residual Rhi + Rlo (split planes [N/32][r_rows][32], the block's input: identity shortcut)
//   SGE_RELU | SGE_PLANES (output as split planes, else fp32)
//   SGE_POLY         the planes are written POLYPHASE for a stride-2 consumer: input row (sequence, frame t < T of T + 4, vertex) -> region t & 1 (poly_region
//                    rows each), frame t >> 1 of ceil(T / 2) + 4; the input's pad frames are not written (the consumer's pads are zeroed by the host's k_sg_zero)
// The callers' counted wait behind an epilogue (`stores_behind`: vmcnt <= the stores of one tile) is an upper bound on what may stay outstanding, not the
// guarantee that the next k-step's DMA has landed: a wave whose 32 rows are all pad rows SKIPS its polyphase stores (s_cbranch_execz in the ISA). The guarantee
// there is the in-order retirement of vmcnt: SGE_POLY (the only mode with conditional stores on the counted path) always comes with SGE_RES_PLANES, whose residual
// loads are issued AFTER that DMA and consumed BEFORE the first store - the DMA has retired by then whatever the number of stores. The other modes issue
// exactly the counted stores on the interior path (CHECK = false); edge tiles (CHECK = true) are followed by a full wait.
// What a 32 x 32 tile needs from memory is requested one tile AHEAD of its use: vmcnt retires in order, so a load queued behind the previous tile's
// stores would wait for their acknowledgements - 2 TM TN round trips to memory per workgroup tile in x3_epilogue's order.
enum { SGE_VERTEX_BIAS = 1, SGE_RELU = 2, SGE_PLANES = 4, SGE_RES_PLANES = 8, SGE_POLY = 16 };
template <int TM, int TN, bool CHECK, int MODE, bool F16 = false>
__device__ __forceinline__ void sg_epilogue(const GemmX3Args& g, f32x16 (&acc)[TM][TN], int mw, int nw, int lane) {
    const int l31 = lane & 31, kh = lane >> 5;
    const bool odd = lane & 1;
    constexpr int NF = (MODE & (SGE_VERTEX_BIAS | SGE_RES_PLANES)) ? 16 : 1;
    auto fetch = [&](int idx, unsigned (&f)[NF]) {
        const int ta = idx / TN, tb = idx - ta * TN;                   // (row tile outermost: what depends on the rows only is shared by its TN column tiles)
        const int n = nw + tb * 32 + l31, mb = mw + ta * 32 + 4 * kh;
        const bool n_ok = !CHECK || n < g.N;
        if constexpr ((MODE & SGE_VERTEX_BIAS) != 0) {
            const int base = (int)((unsigned)mb % (unsigned)g.add_mod);
            const float* ap = g.add + n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int rr = base + (i & 3) + 8 * (i >> 2);
                rr = rr >= g.add_mod ? rr - g.add_mod : rr;
                f[i] = n_ok ? __builtin_bit_cast(unsigned, ap[rr * g.ldadd]) : 0u;
            }
        } else if constexpr ((MODE & SGE_RES_PLANES) != 0) {
            // column pairs (n & ~1, + 1) as one 4-byte load: even lanes the rows of registers 0..7, odd lanes those of registers 8..15; f[0..7] hi, f[8..15] lo
            const size_t o = ((size_t)(n >> 5) * g.r_rows + mb) * 32 + (n & 30);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ii = odd ? i + 8 : i, ro = (ii & 3) + 8 * (ii >> 2);
                const bool ok = !CHECK || (n_ok && mb + ro < g.M);
                f[i] = ok ? *reinterpret_cast<const unsigned*>(g.Rhi + o + ro * 32) : 0u;
                if constexpr (!F16) f[8 + i] = ok ? *reinterpret_cast<const unsigned*>(g.Rlo + o + ro * 32) : 0u;
            }
        } else {
            f[0] = (g.bias && n_ok) ? __builtin_bit_cast(unsigned, g.bias[n]) : 0u;
        }
    };
    unsigned nxt[NF];
    fetch(0, nxt);
#pragma unroll
    for (int idx = 0; idx < TM * TN; ++idx) {
        const int ta = idx / TN, tb = idx - ta * TN;                   // (row tile outermost: what depends on the rows only is shared by its TN column tiles)
        const int n = nw + tb * 32 + l31, mb = mw + ta * 32 + 4 * kh;
        const bool n_ok = !CHECK || n < g.N;
        float r[16];
        if constexpr ((MODE & SGE_VERTEX_BIAS) != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = acc[ta][tb][i] + __builtin_bit_cast(float, nxt[i]);
        } else if constexpr ((MODE & SGE_RES_PLANES) != 0) {
            const float b = (g.bias && n_ok) ? g.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // the partner lane (lane ^ 1) holds this column's values for the other eight rows
                float mine, other;
                if constexpr (F16) {
                    const unsigned ph = nxt[i], qh = (unsigned)__builtin_amdgcn_mov_dpp((int)ph, 0xB1, 0xf, 0xf, true);
                    mine = f16_bits_to_float(odd ? (ph >> 16) : (ph & 0xffffu));
                    other = f16_bits_to_float(odd ? (qh >> 16) : (qh & 0xffffu));
                } else {
                    const unsigned ph = nxt[i], pl = nxt[8 + i];
                    const unsigned qh = (unsigned)__builtin_amdgcn_mov_dpp((int)ph, 0xB1, 0xf, 0xf, true), ql = (unsigned)__builtin_amdgcn_mov_dpp((int)pl, 0xB1, 0xf, 0xf, true);
                    const unsigned mh = odd ? (ph & 0xffff0000u) : (ph << 16), ml = odd ? (pl & 0xffff0000u) : (pl << 16);   // own rows: register ii = odd ? i + 8 : i
                    const unsigned oh = odd ? (qh & 0xffff0000u) : (qh << 16), ol = odd ? (ql & 0xffff0000u) : (ql << 16);   // the partner's rows
                    mine = __builtin_bit_cast(float, mh) + __builtin_bit_cast(float, ml);
                    other = __builtin_bit_cast(float, oh) + __builtin_bit_cast(float, ol);
                }
                r[i] = (acc[ta][tb][i] + b) + (odd ? other : mine);          // (static register indices: a lane-dependent index is a 16-way select chain)
                r[i + 8] = (acc[ta][tb][i + 8] + b) + (odd ? mine : other);
            }
        } else {
            const float b = __builtin_bit_cast(float, nxt[0]);
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = acc[ta][tb][i] + b;
        }
        if (idx + 1 < TM * TN) fetch(idx + 1, nxt);
        if constexpr ((MODE & SGE_RELU) != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = fmaxf(r[i], 0.f);
        }
        if constexpr ((MODE & SGE_PLANES) == 0) {
            float* cp = g.C + (size_t)mb * g.ldc + n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ro = (i & 3) + 8 * (i >> 2);
                if (!CHECK || (n_ok && mb + ro < g.M)) cp[(size_t)ro * g.ldc] = r[i];
            }
        } else if constexpr ((MODE & SGE_POLY) != 0) {
            // rows mb + ro, ro < 32 <= V: at most one step into the next frame, which may be the next sequence's first
            const int Vv = g.poly_V, Tr = g.poly_T, Tp = Tr + 4, Tpo = ((Tr + 1) >> 1) + 4;
            const unsigned f0 = (unsigned)mb / (unsigned)Vv, nm0 = f0 / (unsigned)Tp;
            const int v0 = mb - (int)f0 * Vv, t0 = (int)(f0 - nm0 * (unsigned)Tp);
            const size_t cb = (size_t)(n >> 5) * g.c_rows;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float mine = odd ? r[i + 8] : r[i], give = odd ? r[i] : r[i + 8];
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                const float c0 = odd ? got : mine, c1 = odd ? mine : got;
                const int ii = odd ? i + 8 : i, ro = (ii & 3) + 8 * (ii >> 2);
                int v = v0 + ro, t = t0, nm = (int)nm0;
                if (v >= Vv) { v -= Vv; ++t; }
                if (t >= Tp) { t -= Tp; ++nm; }
                if (t < Tr && (!CHECK || mb + ro < g.M)) {
                    const size_t o = (cb + (size_t)(t & 1) * g.poly_region + ((size_t)nm * Tpo + (t >> 1)) * Vv + v) * 32 + (n & 30);
                    if constexpr (F16) {
                        f16x2 hv = {(_Float16)c0, (_Float16)c1};
                        *reinterpret_cast<f16x2*>(g.Chi + o) = hv;
                    } else {
                        const __bf16 h0 = (__bf16)c0, h1 = (__bf16)c1;
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        bf16x2 hv = {h0, h1}, lv = {(__bf16)(c0 - (float)h0), (__bf16)(c1 - (float)h1)};
                        *reinterpret_cast<bf16x2*>(g.Chi + o) = hv;
                        *reinterpret_cast<bf16x2*>(g.Clo + o) = lv;
                    }
                }
            }
        } else if constexpr (!CHECK) {   // K32-blocked planes [N/32][c_rows][32]: a 32-column tile is one contiguous run of rows
            // adjacent columns paired across lane ^ 1 -> packed bf16x2 stores: even lanes rows of registers 0..7, odd lanes those of registers 8..15
            const size_t o = ((size_t)(n >> 5) * g.c_rows + mb) * 32 + (n & 30);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float mine = odd ? r[i + 8] : r[i], give = odd ? r[i] : r[i + 8];
                const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                const float c0 = odd ? got : mine, c1 = odd ? mine : got;
                const int ii = odd ? i + 8 : i, ro = (ii & 3) + 8 * (ii >> 2);
                if constexpr (F16) {
                    f16x2 hv = {(_Float16)c0, (_Float16)c1};
                    *reinterpret_cast<f16x2*>(g.Chi + o + ro * 32) = hv;
                } else {
                    const __bf16 h0 = (__bf16)c0, h1 = (__bf16)c1;
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    bf16x2 hv = {h0, h1}, lv = {(__bf16)(c0 - (float)h0), (__bf16)(c1 - (float)h1)};
                    *reinterpret_cast<bf16x2*>(g.Chi + o + ro * 32) = hv;
                    *reinterpret_cast<bf16x2*>(g.Clo + o + ro * 32) = lv;
                }
            }
        } else {
            const size_t o = ((size_t)(n >> 5) * g.c_rows + mb) * 32 + (n & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ro = (i & 3) + 8 * (i >> 2);
                if (n_ok && mb + ro < g.M) {
                    if constexpr (F16) reinterpret_cast<_Float16*>(g.Chi)[o + ro * 32] = (_Float16)r[i];
                    else {
                        const __bf16 h = (__bf16)r[i];
                        g.Chi[o + ro * 32] = h;
                        g.Clo[o + ro * 32] = (__bf16)(r[i] - (float)h);
                    }
                }
            }
        }
    }
}
#ifdef RGN_SG_PROF
// tools only (tools/r05_tconv_stamps.sh, tools/sg_stamps.py): cycle stamps of workgroup 0's first 64 k-steps in k_sg_tconv<256, 256>: [step][wave][5]
__device__ long long g_sg_prof[64 * 8 * 5];
#define SG_STAMP(i) if (PROF && blockIdx.x == 0 && lane == 0 && gstep < 64) g_sg_prof[(gstep * 8 + wave) * 5 + (i)] = __builtin_readcyclecounter();
#else
#define SG_STAMP(i)
#endif
static int sg_cu_count() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
}

// ---- the 9 x 1 temporal convolution of a stride-1 ST-GCN block with the activation WINDOW resident in LDS (rgn_stgcn.hip) ---------------------
// As a row-shifted GEMM (above) every tap DMAs its own 256-row slice of the activation - 9 x 32 KB per channel block through the ~20 B/clk L2 -> LDS
// path beside 9 x 16 KB of weights: 48 KB per k-step against 2304 cycles of MFMA (PMC: matrix pipe 0.35 busy, and NOT because of the fabric - with the
// taps innermost the L2 hit rate rose 0.61 -> 0.76 and the HBM fetch halved while the kernel got 4 % slower). Here the 256 + 8 V rows a tile needs of
// one channel block sit in LDS ONCE ([row][64 B] x {hi, lo}, 88 KiB at V = 56) and the nine taps read their fragments V rows apart: per k-step the
// path carries the weight tile + V rows of the NEXT window (the next channel block's - or the next tile's first), which overwrite the V rows the
// finished tap no longer needs - tap dt reads window rows [dt V, dt V + 256), so after it only rows >= (dt + 1) V are live; the top 256 rows of a
// window are fetched during its own taps 0-3 (first needed by tap 4). K order: k-block kt = (channel block kt / 9, tap kt % 9).
// PERSISTENT over tiles (slot, + gridDim, ...) like k_sg_gcn: no launch / first-window / store-acknowledgement gap between tiles. The epilogue (MODE,
// sg_epilogue) writes the fp32 convolution, or - the block's whole tail in place - relu(conv + b2' [+ identity residual]) as the next block's planes.
template <int BM, int BN, int WM, int MODE, bool F16 = false>
__global__ __launch_bounds__(512, 1) void k_sg_tconv(GemmX3Args g, int nbx, int ntiles, int V) {
    constexpr int NT = 512, TAPS = 9;
    [[maybe_unused]] constexpr bool PROF = !F16 && BM == 256 && BN == 256 && MODE == (SGE_RELU | SGE_PLANES | SGE_RES_PLANES);   // (tools: SG_STAMP)
    constexpr int ST_PER = (F16 && (MODE & SGE_PLANES)) ? 8 : 16;   // stores per 32 x 32 tile of the epilogue (planes: hi [+ lo] of 8 column pairs)
    constexpr int WN = 8 / WM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int W_BYTES = BN * 64, W_STAGE = 2 * W_BYTES;      // one plane tile; a stage = hi | lo
    constexpr int W_IT = BN * 8 / NT;                            // DMA instructions per thread per weight tile (hi + lo): 1 / 2 / 4
    static_assert(TM >= 1 && TN >= 1 && BN * 8 % NT == 0 && (BN * 4) % 64 == 0 && BM % (32 * WM) == 0 && BM % 64 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int G = gridDim.x, bid = blockIdx.x;
    const int q8 = G >> 3, r8 = G & 7, xcd = bid & 7;
    const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);   // workgroups of one XCD take neighbouring tiles (they share windows)
    // BM = 256: the window is BM + 8 V rows in place. BM = 512 (> 8 V): CIRCULAR - a window needs its rows [0, BM) at tap 0 while the previous window's
    // tap 8 still reads its rows [8 V, 8 V + BM), so the buffer holds 2 BM = 1024 rows and every window starts SP = BM - 8 V rows in front of the last:
    // its first SP rows land in rows nobody uses, rows [SP, BM) follow the previous window's dying strips, the top 8 V rows come under its own taps 0-3.
    constexpr bool CIRC = BM > 256;
    constexpr int RMASK = 2 * BM - 1;
    const int WR = BM + 8 * V, A_PLANE = (CIRC ? 2 * BM : WR) * 64;   // window rows (a multiple of 16: V even); bytes per plane
    const int SP = CIRC ? BM - 8 * V : 0;
    char* const wst = smem + 2 * A_PLANE;                        // weight stages behind the two window planes
    constexpr int CBS = F16 ? 2 : 1;                             // channel blocks per k-step (F16: the "lo" images hold the next channel block)
    const int ncb = g.Kp / (32 * TAPS * CBS);
    const long long row_lo = -4LL * V, row_hi = (long long)g.M + 4LL * V - 1;   // the planes' zero guard rows bound what a window may touch
    const char* const a_pl[2] = {reinterpret_cast<const char*>(g.Ahi), F16 ? reinterpret_cast<const char*>(g.Ahi) + (long long)g.a_rows * 64 : reinterpret_cast<const char*>(g.Alo)};
    const char* const w_pl[2] = {reinterpret_cast<const char*>(g.Whi), F16 ? reinterpret_cast<const char*>(g.Whi) + (size_t)TAPS * g.N * 64 : reinterpret_cast<const char*>(g.Wlo)};
    auto wk = [&](int cb, int dt) { return CBS * cb * TAPS + dt; };   // weight k-block of (channel block [pair], tap)

    // one 16-row piece (1 KiB per plane) of the window of (tile rows m0t, channel block cb), rows [r0, r0 + 16) limited to rows < rend, into its place in plane pl
    // ws = the window's first physical row (0 unless CIRC)
    auto a_piece = [&](int m0t, int cb, int pl, int r0, int rend, int ws) {
        const int r = r0 + (lane >> 2);
        if (r < rend) {
            long long gr = (long long)m0t - 4LL * V + r;
            gr = gr < row_lo ? row_lo : (gr > row_hi ? row_hi : gr);
            const int p0 = CIRC ? ((ws + r0) & RMASK) : r0, pr = p0 + (lane >> 2);   // (pieces start on multiples of 16: they never wrap)
            const char* src = a_pl[pl] + ((long long)(CBS * cb) * g.a_rows + gr) * 64 + (((lane & 3) ^ ((pr >> 2) & 3)) << 4);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)src, (RGN_AS3 void*)(smem + pl * A_PLANE + p0 * 64), 16, 0, 0);
        }
    };
    unsigned w_lane[W_IT];                                       // this thread's 16 bytes of a weight tile (N is a multiple of BN: no column clamp)
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int q = it * NT + tid, pl = (it * NT + (tid & ~63)) / (BN * 4), qq = q - pl * (BN * 4), r = qq >> 2, c = (qq & 3) ^ ((r >> 2) & 3);
        w_lane[it] = (unsigned)r * 64u + c * 16u;
    }
    auto w_tile = [&](int n0t, int kt, char* stage) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int pl = (it * NT + (tid & ~63)) / (BN * 4);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(w_pl[pl] + ((size_t)kt * g.N + n0t) * 64 + w_lane[it]),
                                             (RGN_AS3 void*)(stage + pl * W_BYTES + (it * NT + (tid & ~63) - pl * (BN * 4)) * 16), 16, 0, 0);
        }
    };
    int w_off[TN][2];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int rr = wn * (BN / WN) + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    const int arow0 = wm * (BM / WM) + l31;
    auto mma3 = [&](f32x16& c, const bf16x8& a_h, const bf16x8& a_l, const bf16x8& w_h, const bf16x8& w_l) {
        if constexpr (F16) {                                     // (a_l, w_l: the next channel block's fragments)
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_h), __builtin_bit_cast(f16x8, w_h), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_l), __builtin_bit_cast(f16x8, w_l), c, 0, 0, 0);
        } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, w_h, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, w_l, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, w_h, c, 0, 0, 0);
        }
    };

    int tile = slot;
    int m0 = (tile / nbx) * BM, n0 = (tile % nbx) * BN;
    // prologue: rows [0, 8 V + SP) of the first window (its top rows follow under taps 0-3 like every window's) and the first weight tile
    {
        const int np = (8 * V + SP) / 16;                        // pieces per plane
        for (int q = wave; q < 2 * np; q += 8) a_piece(m0, 0, q / np, (q % np) * 16, 8 * V + SP, 0);
        w_tile(n0, 0, wst);
    }
    unsigned gstep = 0;                                          // k-steps done: weight stage gstep & 1
    int ws = 0;                                                  // first physical row of the current window
    bool stores_behind = false;
    while (true) {
        const int tnext = tile + G;
        const bool more = tnext < ntiles;
        const int m0n = (tnext / nbx) * BM, n0n = (tnext % nbx) * BN;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
        for (int cb = 0; cb < ncb; ++cb) {
            const bool wlast = cb + 1 == ncb;
#pragma unroll 1
            for (int dt = 0; dt < TAPS; ++dt, ++gstep) {
                SG_STAMP(0)
                if (stores_behind) wait_vmcnt<ST_PER * TM * TN>();   // this k-step's DMA is older than the 16 (F16: 8) TM TN stores of the tile just written
                else wait_vmcnt<0>();                            // everything this thread requested up to the last k-step has landed ...
                stores_behind = false;
                SG_STAMP(1)
                __builtin_amdgcn_s_barrier();                    // ... everyone's has, and everyone is done reading the previous k-step
                SG_STAMP(2)
                const char* wsb = wst + (gstep & 1) * W_STAGE;
                // fragments: the tap's rows start dt V further down the window (the 16-byte chunk swizzle follows the LDS row)
                bf16x8 ah[2][TM], al[2][TM], wh[2], wl[2];
                auto fetch = [&](int grp) {                      // grp = ks * TN + tb
                    const int ks = grp / TN, tb = grp % TN;
                    if (tb == 0) {
#pragma unroll
                        for (int t = 0; t < TM; ++t) {
                            const int rl = arow0 + t * 32 + dt * V, rr = CIRC ? ((ws + rl) & RMASK) : rl;
                            const int o = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
                            ah[ks][t] = *reinterpret_cast<const bf16x8*>(smem + o);
                            al[ks][t] = *reinterpret_cast<const bf16x8*>(smem + A_PLANE + o);
                        }
                    }
                    wh[grp & 1] = *reinterpret_cast<const bf16x8*>(wsb + w_off[tb][ks]);
                    wl[grp & 1] = *reinterpret_cast<const bf16x8*>(wsb + W_BYTES + w_off[tb][ks]);
                };
                fetch(0);
#pragma unroll
                for (int grp = 0; grp < 2 * TN; ++grp) {
                    if (grp + 1 < 2 * TN) fetch(grp + 1);
#pragma unroll
                    for (int ta = 0; ta < TM; ++ta) mma3(acc[ta][grp % TN], ah[grp / TN][ta], al[grp / TN][ta], wh[grp & 1], wl[grp & 1]);
                    if (grp == 0) {
                        // the path's load for this k-step, behind the first MFMA group: next weight tile, the dead strip's successor, the window's top quarter
                        const bool klast = wlast && dt == TAPS - 1;
                        if (!klast) w_tile(n0, dt + 1 < TAPS ? wk(cb, dt + 1) : wk(cb + 1, 0), wst + ((gstep + 1) & 1) * W_STAGE);
                        else if (more) w_tile(n0n, 0, wst + ((gstep + 1) & 1) * W_STAGE);
                        if (!wlast || more) {
                            const int m0w = wlast ? m0n : m0, cbw = wlast ? 0 : cb + 1;
                            if constexpr (!CIRC) {
                                if (dt >= 1) a_piece(m0w, cbw, wave >> 2, (dt - 1) * V + 16 * (wave & 3), dt * V, 0);   // V <= 64 rows: four pieces per plane
                            } else {
                                // the next window's piece p (rows [16 p, 16 p + 16)) lies on this window's rows [16 p - SP, ...): free once tap dt has passed them
                                const int c0 = dt == 0 ? 0 : ((dt - 1) * V + SP) >> 4, c1 = (dt * V + SP) >> 4;
                                for (int p = c0 + (wave & 3); p < c1; p += 4) a_piece(m0w, cbw, wave >> 2, 16 * p, BM, (ws - SP) & RMASK);
                            }
                        }
                        SG_STAMP(3)
                        if (dt < 4) {
                            if constexpr (!CIRC) {
                                constexpr int PPT = BM / 64;             // 16-row pieces per plane per tap: a quarter of the window top
#pragma unroll
                                for (int j = 0; j < (2 * PPT + 7) / 8; ++j) {
                                    const int p = wave + 8 * j;
                                    if (p < 2 * PPT) a_piece(m0, cb, p / PPT, 8 * V + (BM / 4) * dt + 16 * (p % PPT), WR, 0);
                                }
                            } else {
                                // the top 8 V rows, 2 V per tap (tap dt + 1 reads rows below (dt + 1) V + BM <= BM + 2 V (dt + 1))
                                for (int p = wave & 3; 16 * p < 2 * V; p += 4) a_piece(m0, cb, wave >> 2, BM + 2 * V * dt + 16 * p, BM + 2 * V * (dt + 1), ws);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                SG_STAMP(4)
            }
            if constexpr (CIRC) ws = (ws - SP) & RMASK;
        }
        const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
        if (interior) {
            sg_epilogue<TM, TN, false, MODE, F16>(g, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
            stores_behind = true;
        } else sg_epilogue<TM, TN, true, MODE, F16>(g, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
        if (!more) break;
        tile = tnext; m0 = m0n; n0 = n0n;
    }
}
static int tconv_lds_bytes(int BM, int BN, int V) { return 2 * (BM > 256 ? 2 * BM : BM + 8 * V) * 64 + 2 * 2 * BN * 64; }
template <int BM, int BN, int WM, int MODE, bool F16 = false>
static hipError_t tconv_launch(const GemmX3Args& g, int V, hipStream_t s, bool configure_only) {
    if (configure_only)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_tconv<BM, BN, WM, MODE, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, std::min(160 * 1024, tconv_lds_bytes(BM, BN, 64)));
    const int nbx = (g.N + BN - 1) / BN, ntiles = nbx * ((g.M + BM - 1) / BM);
    hipLaunchKernelGGL((k_sg_tconv<BM, BN, WM, MODE, F16>), dim3(std::min(ntiles, sg_cu_count())), dim3(512), tconv_lds_bytes(BM, BN, V), s, g, nbx, ntiles, V);
    return hipGetLastError();
}
// (8 V >= 256: tap 0 reads window rows [0, 256), and only the first 8 V rows of a window are in place before its own taps run)
bool sg_tconv_supported(int N, int Kp, int V) { return (N == 64 || N == 128 || N == 256) && Kp == 9 * N && V % 4 == 0 && V >= 32 && V <= 64; }
// tail 0: C = conv + bias (fp32);  1: planes relu(conv + bias);  2: planes relu(conv + bias + (Rhi + Rlo));  3: as 2, written polyphase (poly_*)
template <int BM, int BN, int WM>
static hipError_t tconv_dispatch(const GemmX3Args& g, int V, int tail, hipStream_t s, bool configure_only) {
    if (configure_only) {
        hipError_t e = tconv_launch<BM, BN, WM, 0>(g, V, s, true);
        if (e != hipSuccess) return e;
        e = tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES>(g, V, s, true);
        if (e == hipSuccess) e = tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES | SGE_RES_PLANES>(g, V, s, true);
        if (e == hipSuccess) e = tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES | SGE_RES_PLANES | SGE_POLY>(g, V, s, true);
        if (e == hipSuccess) e = tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES, true>(g, V, s, true);
        if (e == hipSuccess) e = tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES | SGE_RES_PLANES, true>(g, V, s, true);
        return e != hipSuccess ? e : tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES | SGE_RES_PLANES | SGE_POLY, true>(g, V, s, true);
    }
    if (g.f16) {                                                 // the fp16 form exists for the plane tails only (rgn_stgcn.hip refuses the rest)
        if (tail == 1) return tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES, true>(g, V, s, false);
        if (tail == 2) return tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES | SGE_RES_PLANES, true>(g, V, s, false);
        if (tail == 3) return tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES | SGE_RES_PLANES | SGE_POLY, true>(g, V, s, false);
        return hipErrorInvalidValue;
    }
    if (tail == 0) return tconv_launch<BM, BN, WM, 0>(g, V, s, false);
    if (tail == 1) return tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES>(g, V, s, false);
    if (tail == 2) return tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES | SGE_RES_PLANES>(g, V, s, false);
    return tconv_launch<BM, BN, WM, SGE_RELU | SGE_PLANES | SGE_RES_PLANES | SGE_POLY>(g, V, s, false);
}
// Tile shapes. Per k-step the L2 -> LDS path (~20 B/clk) carries the weight tile + V rows + (taps 0-3) a quarter of the window top against
// TM TN x 6 MFMAs per wave: 256 x 64 tiles (768 MFMA cycles per SIMD, 21 KB) and 256 x 128 (1536, 29 KB) sit at or past the path's rate; a 256-wide tile
// for the 256-channel blocks halves the window bytes per MFMA. Taller tiles would halve the weight bytes, but a tile taller than 8 V rows needs its rows
// [8 V, BM) at tap 0 of a window while the previous window's tap 8 still reads them (512 rows, run without regard for that: -16 ... -24 % on the 64- and
// 128-channel kernels), and 384-row tiles (4 x 2 waves of 96 rows) measured 5 - 15 % SLOWER than 256 (profiles/r05/stgcn_tconv_shapes.txt).
hipError_t launch_sg_tconv(const GemmX3Args& g, int V, int tail, bool small, hipStream_t s) {   // small: 256-row tiles, <= 128 wide (tools / tests)
    const bool tall = !small && V % 8 == 0 && 8 * V <= 512;         // (the circular window: 2 V-row pieces, SP = 512 - 8 V >= 0)
    if (g.N == 64) return tall ? tconv_dispatch<512, 64, 8>(g, V, tail, s, false) : tconv_dispatch<256, 64, 8>(g, V, tail, s, false);
    if (g.N == 128 && tall) return tconv_dispatch<512, 128, 8>(g, V, tail, s, false);
    // (256 x 256 on 4 x 2 waves - 64 x 128 per wave, a third fewer fragment reads from LDS than 8 x 1 - measured the same within 0.3 % in both arithmetics:
    //  profiles/r06_recogniser_experiments.txt)
    if (g.N == 256 && !small && tconv_lds_bytes(256, 256, V) <= 160 * 1024) return tconv_dispatch<256, 256, 8>(g, V, tail, s, false);
    return tconv_dispatch<256, 128, 4>(g, V, tail, s, false);
}
template <int MODE, bool F16 = false>
__global__ __launch_bounds__(512, 1) void k_sg_tconv_s2(GemmX3Args g, int nbx, int ntiles, int V, long long o_rows);
hipError_t configure_sg_tconv() {
    GemmX3Args g{};
    hipError_t e = tconv_dispatch<256, 64, 8>(g, 0, 0, nullptr, true);
    if (e == hipSuccess) e = tconv_dispatch<256, 128, 4>(g, 0, 0, nullptr, true);
    if (e == hipSuccess) e = tconv_dispatch<256, 256, 8>(g, 0, 0, nullptr, true);
    if (e == hipSuccess) e = tconv_dispatch<512, 64, 8>(g, 0, 0, nullptr, true);
    if (e == hipSuccess) e = tconv_dispatch<512, 128, 8>(g, 0, 0, nullptr, true);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_tconv_s2<SGE_RELU | SGE_PLANES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_tconv_s2<SGE_RELU | SGE_PLANES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

// ---- the 9 x 1 temporal convolution of a STRIDE-2 block, at the output rate, on polyphase planes (region E: even frames, region O: odd frames, same
// geometry, O starts o_rows rows behind E) + the block's convolved shortcut + its tail. Output row m = sum over the even taps dt = 2 j of
// g_E[m + (j - 2) V] W_dt + over the odd taps dt = 2 j + 1 of g_O[m + (j - 2) V] W_dt: two resident windows per channel block, E (256 + 4 V rows,
// taps j = 0..4) and O (256 + 3 V rows, j = 0..3), walked E first. Each window's top 256 rows arrive while the OTHER window is being read (E top
// of the next channel block under the O taps, O top under the E taps), its low rows strip by strip behind the taps that are done with them - per
// k-step the L2 -> LDS path carries the weight tile + V rows + 64 rows (31 KB; 48 KB as a row-shifted GEMM). The strided 1 x 1 shortcut (BN folded)
// is k2 more k-steps on the same accumulators: its operand fragments come straight from the block's input planes (region E = the even frames)
// into registers, requested one k-step ahead. Epilogue: planes relu(acc + bias), bias = b2' + br'. Persistent over tiles like k_sg_tconv.
template <int MODE, bool F16>
__global__ __launch_bounds__(512, 1) void k_sg_tconv_s2(GemmX3Args g, int nbx, int ntiles, int V, long long o_rows) {
    constexpr int BM = 256, BN = 128, NT = 512, WM = 4, WN = 2, TM = 2, TN = 2;
    constexpr int ST_PER = (F16 && (MODE & SGE_PLANES)) ? 8 : 16;
    constexpr int W_BYTES = BN * 64, W_STAGE = 2 * W_BYTES, W_IT = BN * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int G = gridDim.x, bid = blockIdx.x;
    const int q8 = G >> 3, r8 = G & 7, xcd = bid & 7;
    const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int WE = BM + 4 * V, WO = BM + 3 * V, EP = WE * 64, OP = WO * 64;   // window rows, bytes per plane
    char* const ebuf = smem;                                     // E hi | E lo
    char* const obuf = smem + 2 * EP;                            // O hi | O lo
    char* const wst = obuf + 2 * OP;
    constexpr int CBS = F16 ? 2 : 1;                             // channel blocks per k-step (F16: the "lo" images hold the next channel block)
    const int ncb = g.Kp / (32 * 9 * CBS), k2 = g.k2 / CBS;
    const long long row_lo = -4LL * V, row_hi = o_rows + (long long)g.M + 4LL * V - 1;   // guard rows in front of E and behind O
    const char* const a_pl[2] = {reinterpret_cast<const char*>(g.Ahi), F16 ? reinterpret_cast<const char*>(g.Ahi) + (long long)g.a_rows * 64 : reinterpret_cast<const char*>(g.Alo)};
    const __bf16* const w1lo = F16 ? g.Whi + (size_t)9 * g.N * 32 : g.Wlo;      // (the same tap of the next channel block: 9 k-blocks of N x 32 further on)
    const __bf16* const w2lo = F16 ? g.W2hi + (size_t)g.N * 32 : g.W2lo;
    const __bf16* const a2lo = F16 ? g.A2hi + (size_t)g.a2_rows * 32 : g.A2lo;
    // one 16-row piece of a window: `first` = the global row of window row 0, rows [r0, r0 + 16) below rend, plane pl of the buffer at `buf` (`pb` bytes per plane)
    auto a_piece = [&](long long first, int cb, int pl, int r0, int rend, char* buf, int pb) {
        const int r = r0 + (lane >> 2);
        if (r < rend) {
            long long gr = first + r;
            gr = gr < row_lo ? row_lo : (gr > row_hi ? row_hi : gr);
            const char* src = a_pl[pl] + ((long long)(CBS * cb) * g.a_rows + gr) * 64 + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)src, (RGN_AS3 void*)(buf + pl * pb + r0 * 64), 16, 0, 0);
        }
    };
    unsigned w_lane[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int q = it * NT + tid, pl = (it * NT + (tid & ~63)) / (BN * 4), qq = q - pl * (BN * 4), r = qq >> 2, c = (qq & 3) ^ ((r >> 2) & 3);
        w_lane[it] = (unsigned)r * 64u + c * 16u;
    }
    auto w_tile = [&](const __bf16* whi, const __bf16* wlo, int n0t, int kt, char* stage) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int pl = (it * NT + (tid & ~63)) / (BN * 4);
            const char* base = reinterpret_cast<const char*>(pl ? wlo : whi);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(base + ((size_t)kt * g.N + n0t) * 64 + w_lane[it]),
                                             (RGN_AS3 void*)(stage + pl * W_BYTES + (it * NT + (tid & ~63) - pl * (BN * 4)) * 16), 16, 0, 0);
        }
    };
    int w_off[TN][2];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int rr = wn * (BN / WN) + t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w_off[t][ks] = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
    }
    const int arow0 = wm * (BM / WM) + l31;
    auto mma3 = [&](f32x16& c, const bf16x8& a_h, const bf16x8& a_l, const bf16x8& w_h, const bf16x8& w_l) {
        if constexpr (F16) {                                     // (a_l, w_l: the next channel block's fragments)
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_h), __builtin_bit_cast(f16x8, w_h), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_l), __builtin_bit_cast(f16x8, w_l), c, 0, 0, 0);
        } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, w_h, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, w_l, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, w_h, c, 0, 0, 0);
        }
    };
    // the shortcut's operand fragments of this lane's rows, k-block cr (straight from the input planes, region E)
    bf16x8 sh[2][TM], sl[2][TM];
    auto shortcut_fetch = [&](int m0t, int cr) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            long long row = (long long)m0t + arow0 + t * 32;
            row = row > row_hi ? row_hi : row;
            const size_t o = ((size_t)(CBS * cr) * g.a2_rows + (size_t)row) * 32 + 8 * kh;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                sh[ks][t] = *reinterpret_cast<const bf16x8*>(g.A2hi + o + 16 * ks);
                sl[ks][t] = *reinterpret_cast<const bf16x8*>(a2lo + o + 16 * ks);
            }
        }
    };

    int tile = slot;
    int m0 = (tile / nbx) * BM, n0 = (tile % nbx) * BN;
    {   // prologue: all of E and the low 3 V rows of O of the first window (O's top follows under the E taps like every window's), the first weight tile
        const int ne = (WE + 15) / 16, no = (3 * V + 15) / 16;
        for (int q = wave; q < 2 * ne; q += 8) a_piece((long long)m0 - 2 * V, 0, q / ne, (q % ne) * 16, WE, ebuf, EP);
        for (int q = wave; q < 2 * no; q += 8) a_piece(o_rows + m0 - 2 * V, 0, q / no, (q % no) * 16, 3 * V, obuf, OP);
        w_tile(g.Whi, w1lo, n0, 0, wst);
    }
    unsigned gstep = 0;
    bool stores_behind = false;
    while (true) {
        const int tnext = tile + G;
        const bool more = tnext < ntiles;
        const int m0n = (tnext / nbx) * BM, n0n = (tnext % nbx) * BN;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
        const int nsteps = ncb * 9 + k2;
        for (int st = 0; st < nsteps; ++st, ++gstep) {
            const bool shortcut = st >= ncb * 9;
            const int cb = shortcut ? ncb - 1 : st / 9, i9 = shortcut ? 9 : st - cb * 9;      // i9: 0..4 E taps, 5..8 O taps
            const bool isE = i9 < 5;
            const int j = isE ? i9 : i9 - 5;
            if (stores_behind) wait_vmcnt<ST_PER * TM * TN>();
            else wait_vmcnt<0>();
            stores_behind = false;
            __builtin_amdgcn_s_barrier();
            const char* wsb = wst + (gstep & 1) * W_STAGE;
            const char* abuf = isE ? ebuf : obuf;
            const int apl = isE ? EP : OP;
            bf16x8 ah[2][TM], al[2][TM], wh[2], wl[2];
            if (shortcut) {                                      // (both k halves now: the next k-block's fragments are requested into sh / sl under this step)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        ah[ks][t] = sh[ks][t];
                        al[ks][t] = sl[ks][t];
                    }
            }
            auto fetch = [&](int grp) {                          // grp = ks * TN + tb
                const int ks = grp / TN, tb = grp % TN;
                if (tb == 0 && !shortcut) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) {
                        const int rr = arow0 + t * 32 + j * V;
                        const int o = rr * 64 + (((2 * ks + kh) ^ ((rr >> 2) & 3)) << 4);
                        ah[ks][t] = *reinterpret_cast<const bf16x8*>(abuf + o);
                        al[ks][t] = *reinterpret_cast<const bf16x8*>(abuf + apl + o);
                    }
                }
                wh[grp & 1] = *reinterpret_cast<const bf16x8*>(wsb + w_off[tb][ks]);
                wl[grp & 1] = *reinterpret_cast<const bf16x8*>(wsb + W_BYTES + w_off[tb][ks]);
            };
            fetch(0);
#pragma unroll
            for (int grp = 0; grp < 2 * TN; ++grp) {
                if (grp + 1 < 2 * TN) fetch(grp + 1);
#pragma unroll
                for (int ta = 0; ta < TM; ++ta) mma3(acc[ta][grp % TN], ah[grp / TN][ta], al[grp / TN][ta], wh[grp & 1], wl[grp & 1]);
                if (grp == 0) {
                    char* nstage = wst + ((gstep + 1) & 1) * W_STAGE;
                    // the next k-step's weight tile (tap order E 0, 2, 4, 6, 8 then O 1, 3, 5, 7: step i of a channel block is tap i < 5 ? 2 i : 2 (i - 5) + 1)
                    if (st + 1 < nsteps) {
                        const int s1 = st + 1;
                        if (s1 < ncb * 9) {
                            const int c1 = s1 / 9, i1 = s1 - c1 * 9;
                            w_tile(g.Whi, w1lo, n0, CBS * c1 * 9 + (i1 < 5 ? 2 * i1 : 2 * (i1 - 5) + 1), nstage);
                        } else w_tile(g.W2hi, w2lo, n0, CBS * (s1 - ncb * 9), nstage);
                    } else if (more) w_tile(g.Whi, w1lo, n0n, 0, nstage);
                    if (!shortcut) {
                        const bool wlast = cb + 1 == ncb;
                        const bool nextw = !wlast || more;       // a next pair of windows exists: (this tile, cb + 1) or (next tile, 0)
                        const long long firstn = (long long)(wlast ? m0n : m0) - 2 * V;
                        const int cbn = wlast ? 0 : cb + 1;
                        if (isE) {
                            if (j >= 1 && nextw) a_piece(firstn, cbn, wave >> 2, (j - 1) * V + 16 * (wave & 3), j * V, ebuf, EP);       // E strip of the next window
                            if (j < 4) a_piece(o_rows + m0 - 2 * V, cb, wave >> 2, 3 * V + 64 * j + 16 * (wave & 3), WO, obuf, OP);       // this window's O top
                        } else {
                            if (j >= 1 && nextw) a_piece(o_rows + firstn, cbn, wave >> 2, (j - 1) * V + 16 * (wave & 3), j * V, obuf, OP);   // O strip of the next window
                            if (nextw) a_piece(firstn, cbn, wave >> 2, 4 * V + 64 * j + 16 * (wave & 3), WE, ebuf, EP);                  // the next window's E top
                        }
                    }
                    // the shortcut's fragments for the next k-step
                    if (st + 1 >= ncb * 9 && st + 1 < nsteps) shortcut_fetch(m0, st + 1 - ncb * 9);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
        if (interior) {
            sg_epilogue<TM, TN, false, MODE, F16>(g, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
            stores_behind = true;
        } else sg_epilogue<TM, TN, true, MODE, F16>(g, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane);
        if (!more) break;
        tile = tnext; m0 = m0n; n0 = n0n;
    }
}
static int tconv_s2_lds_bytes(int V) { return 2 * (256 + 4 * V) * 64 + 2 * (256 + 3 * V) * 64 + 2 * 2 * 128 * 64; }
bool sg_tconv_s2_supported(int N, int Kp, int V) { return (N == 128 || N == 256) && Kp == 9 * N && V % 4 == 0 && V >= 16 && V <= 64 && tconv_s2_lds_bytes(V) <= 160 * 1024; }
hipError_t launch_sg_tconv_s2(const GemmX3Args& g, int V, long long o_rows, hipStream_t s) {
    const int nbx = g.N / 128, ntiles = nbx * ((g.M + 255) / 256);
    if (g.f16) hipLaunchKernelGGL((k_sg_tconv_s2<SGE_RELU | SGE_PLANES, true>), dim3(std::min(ntiles, sg_cu_count())), dim3(512), tconv_s2_lds_bytes(V), s, g, nbx, ntiles, V, o_rows);
    else hipLaunchKernelGGL((k_sg_tconv_s2<SGE_RELU | SGE_PLANES>), dim3(std::min(ntiles, sg_cu_count())), dim3(512), tconv_s2_lds_bytes(V), s, g, nbx, ntiles, V, o_rows);
    return hipGetLastError();
}

// ---- graph aggregation + 1 x 1 convolution of an ST-GCN block as ONE kernel (rgn_stgcn.hip) -------------------------------------------------------
// z[(frame, w), (k, ci)] = sum_v A'_k[v, w] x[(frame, v), ci] followed by z . W1'^T was two launches with z (3 x the activation) written and read back:
// 24 of the 52 bytes per activation element a block moved, both kernels fabric-bound (the 64-channel GEMM kept its matrix pipe 13 % busy). Here a
// workgroup holds the rows [m0 - V, m0 + 256 + V) of ONE 32-channel block of x in LDS (every frame a tile row belongs to lies inside) and each wave
// builds the operand fragments of its own 32 rows in registers, in the fragment layout itself: lane (row, k half) sums a_j x[frame base + v_j] over
// the nonzeros of A'_k[:, w(row)] for its 8 channels in fp32, splits the sum into bf16 hi / lo and feeds three MFMAs per weight fragment - the
// arithmetic of k_sg_agg + k_gemm_x3 to the bit, z never exists. K order: (channel block, k); the weight k-block of (cb, k) is k C/32 + cb. Eight waves
// of 32 rows x BN columns (no wave repeats another's aggregation); the next channel block's window arrives a third per k-step beside the weight tile.
// PERSISTENT: a workgroup walks tiles blockIdx, + gridDim, ... as one k-step stream - the next tile's first window and weight tile are in flight under
// the last k-steps of this one, and its first wait leaves this tile's stores outstanding (a tile is only 6 - 24 k-steps: as one workgroup per tile,
// launch + first window + store acknowledgements were 25 - 45 % of the kernel with the matrix pipe and the fabric taking turns idling).
// ALLK: one barrier per CHANNEL BLOCK - a weight stage holds the tiles of all KP <= 4 partitions (narrow tiles only: 3 x 8 KB at BN = 64): the k-step's fixed
// costs (DMA wait, barrier, DMA issue; ~1950 cycles in k_sg_tconv's stamps) are paid once per 96-deep step instead of once per 32.
template <int BN, bool ALLK, bool F16 = false>
__global__ __launch_bounds__(512, 1) void k_sg_gcn(GemmX3Args g, int nbx, int ntiles, int V, int KP, unsigned slot_k, const int* __restrict__ sl_v, const float* __restrict__ sl_a) {
    constexpr int BM = 256, NT = 512, TN = BN / 32, NS = 8;
    constexpr int ST_PER = F16 ? 8 : 16;                         // plane stores per 32 x 32 tile of the epilogue
    constexpr int W_BYTES = BN * 64, W_STAGE = 2 * W_BYTES * (ALLK ? 4 : 1);
    constexpr int W_IT = BN * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int WRX = BM + 2 * V, XP = WRX * 64, XW = 2 * XP;      // window rows, bytes per plane, bytes per window (hi | lo)
    char* const wst = smem + 2 * XW;
    int* const t_v = reinterpret_cast<int*>(wst + 2 * W_STAGE);   // the slot tables [V][8]
    float* const t_a = reinterpret_cast<float*>(t_v + V * NS);
    for (int i = tid; i < V * NS; i += NT) {
        t_v[i] = sl_v[i];
        t_a[i] = sl_a[i];
    }
    __syncthreads();
    constexpr int CBS = F16 ? 2 : 1;                             // channel blocks per window (F16: the "lo" images hold the next channel block)
    const int ncbf = g.Kp / (32 * KP), ncb = ncbf / CBS;         // channel blocks, windows
    const long long row_hi = (long long)g.M + 4LL * V - 1;       // (the planes carry 4 V guard rows at both ends)
    const char* const a_pl[2] = {reinterpret_cast<const char*>(g.Ahi), F16 ? reinterpret_cast<const char*>(g.Ahi) + (long long)g.a_rows * 64 : reinterpret_cast<const char*>(g.Alo)};
    const char* const w_pl[2] = {reinterpret_cast<const char*>(g.Whi), F16 ? reinterpret_cast<const char*>(g.Whi) + (size_t)g.N * 64 : reinterpret_cast<const char*>(g.Wlo)};
    auto wkb = [&](int cb, int k) { return k * ncbf + CBS * cb; };   // weight k-block of (window, partition)
    const int npc = (WRX + 15) / 16, npieces = 2 * npc;          // 16-row pieces of a window: hi plane, then lo plane
    const int ppk = (npieces + 8 * KP - 1) / (8 * KP);           // pieces per wave per k-step
    auto x_piece = [&](int m0t, int cb, int p, int buf) {        // piece p of the window of (tile rows m0t, channel block cb) into window buffer buf
        const int pl = p >= npc ? 1 : 0, r0 = (p - pl * npc) * 16, r = r0 + (lane >> 2);
        if (r < WRX) {
            long long gr = (long long)m0t - V + r;
            gr = gr > row_hi ? row_hi : gr;
            const char* src = a_pl[pl] + ((long long)(CBS * cb) * g.a_rows + gr) * 64 + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)src, (RGN_AS3 void*)(smem + buf * XW + pl * XP + r0 * 64), 16, 0, 0);
        }
    };
    auto w_tile = [&](int n0t, int kb, char* stage) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int q = it * NT + tid, pl = (it * NT + (tid & ~63)) / (BN * 4), qq = q - pl * (BN * 4), r = qq >> 2, c = (qq & 3) ^ ((r >> 2) & 3);
            int n = n0t + r;
            n = n < g.N ? n : g.N - 1;
            __builtin_amdgcn_global_load_lds((const RGN_AS1 void*)(w_pl[pl] + ((size_t)kb * g.N + n) * 64 + c * 16),
                                             (RGN_AS3 void*)(stage + pl * W_BYTES + (it * NT + (tid & ~63) - pl * (BN * 4)) * 16), 16, 0, 0);
        }
    };
    int w_off[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int rr = t * 32 + l31;
        w_off[t] = rr * 64 + ((kh ^ ((rr >> 2) & 3)) << 4);      // k half ks = 0; ks = 1 is the chunk two further on: offset ^ 32
    }
    const int r = wave * 32 + l31;

    // workgroups of one XCD take neighbouring tiles: a tile's window overlaps its neighbours' by V rows on either side, which then come from that XCD's L2
    const int G = gridDim.x, q8 = G >> 3, r8 = G & 7, xcd = blockIdx.x & 7;
    int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
    int m0 = (tile / nbx) * BM, n0 = (tile % nbx) * BN;
    for (int p = wave; p < npieces; p += 8) x_piece(m0, 0, p, 0);
    if constexpr (ALLK) {
        for (int kk = 0; kk < KP; ++kk) w_tile(n0, wkb(0, kk), wst + kk * 2 * W_BYTES);
    } else w_tile(n0, 0, wst);
    unsigned gstep = 0, widx = 0;                                // k-steps / windows consumed so far: stage gstep & 1, window buffer widx & 1
    bool stores_behind = false;                                  // the previous tile's stores were issued after everything the next wait is for
    while (true) {
        // This lane's row keeps its vertex w for the whole tile, so its lists do too: NS slots of (source row in the window, coefficient), slot s serving
        // partition (slot_k >> 4 s) & 15 - a partition owns as many slots as its longest list; shorter lists are padded with (own row, 0).
        const int wv = (int)((unsigned)(m0 + r) % (unsigned)V);
        const int fbase = V + r - wv;                            // window row of vertex 0 of this row's frame
        int so[NS];
        float sa[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int src = fbase + t_v[wv * NS + s];
            so[s] = src * 64 + ((kh ^ ((src >> 2) & 3)) << 4);
            sa[s] = t_a[wv * NS + s];
        }
        unsigned live = 0;                                       // slots that carry a coefficient for at least one row of this wave (a hub vertex's long list
#pragma unroll                                                   //  pads everyone else's: most waves skip most of its slots)
        for (int s = 0; s < NS; ++s) live |= (__builtin_amdgcn_ballot_w64(sa[s] != 0.f) != 0ull ? 1u : 0u) << s;
        const int tnext = tile + (int)gridDim.x;
        const bool more = tnext < ntiles;
        const int m0n = (tnext / nbx) * BM, n0n = (tnext % nbx) * BN;
        f32x16 acc[1][TN];
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][b][i] = 0.f;

        for (int cb = 0; cb < ncb; ++cb, ++widx) {
            const char* xw = smem + (widx & 1) * XW;
            const bool wlast = cb + 1 == ncb;
            for (int k = 0; k < KP; ++k) {
                if (!ALLK || k == 0) {
                    if (stores_behind) wait_vmcnt<ST_PER * TN>();    // the DMA of this step is older than the 16 (F16: 8) TN stores of the tile just written
                    else wait_vmcnt<0>();
                    stores_behind = false;
                    __builtin_amdgcn_s_barrier();                // the tile(s) of this step (and, at k = 0, the window) are in LDS; the previous step is read out
                    if constexpr (!ALLK) {
                        const bool klast = k + 1 == KP;
                        if (!(klast && wlast)) w_tile(n0, klast ? wkb(cb + 1, 0) : wkb(cb, k + 1), wst + ((gstep + 1) & 1) * W_STAGE);
                        else if (more) w_tile(n0n, 0, wst + ((gstep + 1) & 1) * W_STAGE);
                        if (!wlast || more)
                            for (int i = 0; i < ppk; ++i) {
                                const int p = (k * ppk + i) * 8 + wave;
                                if (p < npieces) x_piece(wlast ? m0n : m0, wlast ? 0 : cb + 1, p, (widx + 1) & 1);
                            }
                    } else if (!wlast || more) {                 // every partition's tile of the next channel block, and its whole window
                        for (int kk = 0; kk < KP; ++kk) w_tile(wlast ? n0n : n0, wkb(wlast ? 0 : cb + 1, kk), wst + ((gstep + 1) & 1) * W_STAGE + kk * 2 * W_BYTES);
                        for (int p = wave; p < npieces; p += 8) x_piece(wlast ? m0n : m0, wlast ? 0 : cb + 1, p, (widx + 1) & 1);
                    }
                }
                const char* wsb = wst + (gstep & 1) * W_STAGE + (ALLK ? k * 2 * W_BYTES : 0);
                float z[2][8];
                // F16: the sums of the window's two channel blocks, formed in PACKED fp16 (v_pk_fma_f16: 4 instructions per slot and fragment where the fp32
                // sums take 8 conversions + 8 fmas - this aggregation, not the MFMAs, bounds the narrow tiles): <= 6 terms with |coefficient| <= 1, one
                // fp16 rounding per term where the fp32 form rounds once at the end (stated with SG_F16's bound, tests/test_eval_gpu.py)
                [[maybe_unused]] f16x8 zf[2], zf2[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        z[ks][e] = 0.f;
                        zf[ks][e] = zf2[ks][e] = (_Float16)0.f;
                    }
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if ((int)((slot_k >> (4 * s)) & 15u) == k && ((live >> s) & 1u)) {     // (uniform)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const int o = so[s] ^ (32 * ks);
                            if constexpr (F16) {
                                const f16x8 h = *reinterpret_cast<const f16x8*>(xw + o), h2 = *reinterpret_cast<const f16x8*>(xw + XP + o);
                                const _Float16 c16 = (_Float16)sa[s];
                                const f16x8 cv = {c16, c16, c16, c16, c16, c16, c16, c16};
                                zf[ks] = __builtin_elementwise_fma(cv, h, zf[ks]);
                                zf2[ks] = __builtin_elementwise_fma(cv, h2, zf2[ks]);
                            } else {
                                const bf16x8 h = *reinterpret_cast<const bf16x8*>(xw + o), l = *reinterpret_cast<const bf16x8*>(xw + XP + o);
#pragma unroll
                                for (int e = 0; e < 8; ++e) z[ks][e] = fmaf(sa[s], (float)h[e] + (float)l[e], z[ks][e]);
                            }
                        }
                    }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if constexpr (F16) {
                        const f16x8 af = zf[ks], af2 = zf2[ks];
#pragma unroll
                        for (int tb = 0; tb < TN; ++tb) {
                            const int o = w_off[tb] ^ (32 * ks);
                            const f16x8 wf = *reinterpret_cast<const f16x8*>(wsb + o), wf2 = *reinterpret_cast<const f16x8*>(wsb + W_BYTES + o);
                            acc[0][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, wf, acc[0][tb], 0, 0, 0);
                            acc[0][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af2, wf2, acc[0][tb], 0, 0, 0);
                        }
                    } else {
                        bf16x8 ah, al;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            ah[e] = (__bf16)z[ks][e];
                            al[e] = (__bf16)(z[ks][e] - (float)ah[e]);
                        }
#pragma unroll
                        for (int tb = 0; tb < TN; ++tb) {
                            const int o = w_off[tb] ^ (32 * ks);
                            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wsb + o), wl = *reinterpret_cast<const bf16x8*>(wsb + W_BYTES + o);
                            acc[0][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc[0][tb], 0, 0, 0);
                            acc[0][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc[0][tb], 0, 0, 0);
                            acc[0][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc[0][tb], 0, 0, 0);
                        }
                    }
                }
                if (!ALLK || k + 1 == KP) ++gstep;
            }
        }
        const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
        if (interior) {
            sg_epilogue<1, TN, false, SGE_VERTEX_BIAS | SGE_RELU | SGE_PLANES, F16>(g, acc, m0 + wave * 32, n0, lane);
            stores_behind = true;                                // (exactly 16 TN stores, all issued after the next k-step's DMA)
        } else sg_epilogue<1, TN, true, SGE_VERTEX_BIAS | SGE_RELU | SGE_PLANES, F16>(g, acc, m0 + wave * 32, n0, lane);
        if (!more) break;
        tile = tnext; m0 = m0n; n0 = n0n;
    }
}
static int gcn_lds_bytes(int BN, int V, bool allk = false) { return 2 * 2 * (256 + 2 * V) * 64 + 2 * 2 * BN * 64 * (allk ? 4 : 1) + 2 * 4 * 8 * V; }
bool sg_gcn_supported(int N, int Kp, int V, int KP) {
    return (N == 64 || N == 128 || N == 256) && KP >= 1 && KP <= 8 && Kp % (32 * KP) == 0 && V % 4 == 0 && V >= 16 && V <= 64 && gcn_lds_bytes(64, V) <= 160 * 1024;
}
template <int BN, bool ALLK>
static hipError_t gcn_launch(const GemmX3Args& g, int V, int KP, unsigned slot_k, const int* sl_v, const float* sl_a, hipStream_t s) {
    const int nbx = (g.N + BN - 1) / BN, ntiles = nbx * ((g.M + 255) / 256);
    if (g.f16) hipLaunchKernelGGL((k_sg_gcn<BN, ALLK, true>), dim3(std::min(ntiles, sg_cu_count())), dim3(512), gcn_lds_bytes(BN, V, ALLK), s, g, nbx, ntiles, V, KP, slot_k, sl_v, sl_a);
    else hipLaunchKernelGGL((k_sg_gcn<BN, ALLK>), dim3(std::min(ntiles, sg_cu_count())), dim3(512), gcn_lds_bytes(BN, V, ALLK), s, g, nbx, ntiles, V, KP, slot_k, sl_v, sl_a);
    return hipGetLastError();
}
hipError_t launch_sg_gcn(const GemmX3Args& g, int V, int KP, unsigned slot_k, const int* sl_v, const float* sl_a, int cap, bool per_block_barrier, hipStream_t s) {   // cap: widest tile (tools / tests)
    if (g.N >= 256 && cap >= 256 && gcn_lds_bytes(256, V) <= 160 * 1024) return gcn_launch<256, false>(g, V, KP, slot_k, sl_v, sl_a, s);
    if (g.N >= 128 && cap >= 128 && gcn_lds_bytes(128, V) <= 160 * 1024) return gcn_launch<128, false>(g, V, KP, slot_k, sl_v, sl_a, s);
    if (!per_block_barrier && KP <= 4 && gcn_lds_bytes(64, V, true) <= 160 * 1024) return gcn_launch<64, true>(g, V, KP, slot_k, sl_v, sl_a, s);
    return gcn_launch<64, false>(g, V, KP, slot_k, sl_v, sl_a, s);
}
hipError_t configure_sg_gcn() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_gcn<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_gcn<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_gcn<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_gcn<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_gcn<64, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_gcn<64, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_gcn<128, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sg_gcn<256, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

}  // namespace rgn
#ifdef RGN_SG_PROF
extern "C" __attribute__((visibility("default"))) int rgn_debug_sg_prof(long long* out) {   // (tools build only: not part of the C-ABI)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rgn::g_sg_prof), sizeof(long long) * 64 * 8 * 5);
}
#endif
