// What would a decoder layer cost with G CUs of one XCD per sample (the batch gap, 16 <= B <= 128 at 60 frames)? One workgroup per
// sample (k_layers) is a latency chain whose floor on ONE CU is the weight stream - 4.2 MB per layer through one vector-memory path. A column / head
// split gives every CU 1 / G of every weight matrix; what it adds are exchanges through the XCD's L2. This tool runs the SKELETON of that kernel on
// the hardware: the real weight stream (fragment-ordered planes, buffer loads into a register ring), the real MFMA count on resident 64-row images,
// the real exchange volumes behind group barriers, and calibrated stand-ins (dependent FMA chains) for the VALU phases - no arithmetic that means
// anything. G = 1 is the control: it must land near k_layers' stamped 100 - 112 k cycles per layer.
//
//   per layer and CU (member m of G):   in_proj of heads {m H/G ..}   (1536 / G columns, K = 512)  + attention stand-in / G
//                                       X1: publish the attention output slice (64 x 512 / G bf16), read the other G - 1 slices      [G > 1]
//                                       out_proj, ALL 512 columns (redundant on every member: norm1 / norm2 stay local) + LayerNorm stand-in
//                                       linear1: 1024 / G hidden columns + GELU stand-in / G;  linear2 over those K rows -> fp32 partial [64][512]
//                                       X2: all-reduce of the partials (128 KiB written, (G - 1) x 128 KiB read per CU)                [G > 1]
//                                       norm3 stand-in (redundant)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gcu_layer_bench.hip -o tools/bin/gcu_layer_bench ; run: tools/bin/gcu_layer_bench [samples] [layers x steps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                            \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int NTH = 512, KB = 4096, LDS_BYTES = 160 * 1024, RD = 8;
struct Ctrl {
    int bar[256];          // one monotonic barrier counter per group
    int timeouts;
    long long cyc[256];    // per workgroup: cycles of the whole run
};
struct Args {
    const __bf16* W;       // per layer: Wqkv [16][48] | Wo [16][16] | W1 [16][32] | W2 [32][16] blocks of 2 KiB (fragment order [kb][cb][2][64][8])
    char* xbuf;            // exchange scratch: per group 4 x 128 KiB
    Ctrl* c;
    int layers;            // layers x steps to run
    int valu_attn, valu_ln12, valu_gelu, valu_ln3;   // dependent-FMA chain lengths of the stand-ins (per wave)
    int xmask;             // bit 0: the attention-output exchange, bit 1: the all-reduce of the FFN partials (timing builds: parts removed)
};
constexpr size_t LAYER_BLOCKS = 16 * 48 + 16 * 16 + 16 * 32 + 32 * 16;   // 2 KiB each = 4 MiB

__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int... Is, class F>
__device__ __forceinline__ void static_for_seq(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }

// A dependent chain of n FMAs on 8 registers per lane: the VALU phases' stand-in (4 cycles per instruction and wave; two waves per SIMD)
__device__ __forceinline__ float valu_chain(int n, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    for (int i = 0; i < n; i += 8) {
        a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
        a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f); a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int G>
__global__ __launch_bounds__(NTH, 2) void k_gcu(Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5, swz = (l31 >> 2) & 3;
    // the hardware places workgroup id b on XCD b % 8: the G members of a group share an XCD (ids b, b + 8, ...)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = (slot / G) * 8 + xcd, mem = slot % G;
    int a_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_off[ks] = l31 * 64 + (((2 * ks + kh) ^ swz) << 4);
    const int lane16 = lane * 16;
    for (int i = tid; i < LDS_BYTES / 16; i += NTH) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();

    bf16x8 wf[RD][2];
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // one GEMM pass of this wave: NT column blocks (cb0, cb0 + 1) x 64 rows over NKB k-blocks from kb0, weights W [..][nb_all] blocks of 2 KiB, A operand
    // from the image at LDS byte offset img. The ring holds RD - 1 granules (half k-steps) ahead; drained at the end of a pass (the real kernel chains
    // passes - a refinement this skeleton leaves out, in G = 1's disfavour as much as in G = 4's).
    auto pass = [&](const __bf16* W, int nb_all, int cb0, int kb0, auto nkb_c, auto nt_c, int img) {
        constexpr int NG = 2 * decltype(nkb_c)::value, NT = decltype(nt_c)::value, AH = RD - 1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W) + ((size_t)kb0 * nb_all + cb0) * 1024, 0, 0x7fffffff, 0x00020000);
        const int kstride = nb_all * 2048;
        auto load_g = [&](int hs, int slot_) {
            const int soff = (hs >> 1) * kstride + (hs & 1) * 1024;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[slot_][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, soff + nt * 2048, 0));
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for_seq(std::make_integer_sequence<int, (AH < NG ? AH : NG)>{}, [&](auto S) { load_g(decltype(S)::value, decltype(S)::value); });
        static_for_seq(std::make_integer_sequence<int, NG>{}, [&](auto HS) {
            constexpr int hs = decltype(HS)::value;
            bf16x8 af[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = *reinterpret_cast<const bf16x8*>(smem + img + ((hs >> 1) & 15) * KB + a_off[hs & 1] + mt * 2048);
            if constexpr (hs + AH < NG) {
                load_g(hs + AH, (hs + AH) % RD);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT * AH) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT * (NG - 1 - hs)) : "memory");
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[hs % RD][nt], af[mt], acc[nt][mt], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto group_barrier = [&](int& epoch) {
        __syncthreads();
        if (G > 1 && tid == 0) {
            ++epoch;
            __hip_atomic_fetch_add(&g.c->bar[grp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (ld_sc1(&g.c->bar[grp]) < epoch * G) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { atomicAdd(&g.c->timeouts, 1); break; }     // (never hang the GPU over a benchmark)
            }
        }
        __syncthreads();
    };
    // exchange: every member stores `bytes` of its LDS (from offset src) to its slot, barrier, loads the other members' slots into LDS (sc1: served by
    // the XCD's L2, past this CU's L1), barrier (the slots may be overwritten by the next exchange)
    auto exchange = [&](int bytes, int src, int dst, int& epoch, bool accumulate) {
        if (G == 1) return;
        char* base = g.xbuf + (size_t)grp * (4 * 131072);
        const int n = bytes / (NTH * 16);                                  // 16-byte pieces per thread: 16 for 128 KiB
        for (int j2 = 0; j2 < n; ++j2) {
            const int o = (j2 * NTH + tid) * 16;
            *reinterpret_cast<u32x4*>(base + mem * 131072 + o) = *reinterpret_cast<const u32x4*>(smem + src + (o & 65535));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        group_barrier(epoch);
        for (int m2 = 1; m2 < G; ++m2) {
            const char* sp = base + ((mem + m2) % G) * 131072;
            u32x4 v[16];
#pragma unroll
            for (int j2 = 0; j2 < 16; ++j2)
                if (j2 < n) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j2]) : "v"(sp + (j2 * NTH + tid) * 16) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j2 = 0; j2 < 16; ++j2)
                if (j2 < n) {
                    u32x4* d = reinterpret_cast<u32x4*>(smem + dst + (((j2 * NTH + tid) * 16) & 65535));
                    if (accumulate) {
                        const f32x4 a = __builtin_bit_cast(f32x4, *d), b = __builtin_bit_cast(f32x4, v[j2]);
                        *d = __builtin_bit_cast(u32x4, f32x4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]});
                    } else *d = v[j2];
                }
        }
        group_barrier(epoch);
    };

    int epoch = 0;
    float sink = 0.f;
    group_barrier(epoch);
    const long long t0 = __builtin_readcyclecounter();
    for (int l = 0; l < g.layers; ++l) {
        const __bf16* Wqkv = g.W + (size_t)(l & 7) * LAYER_BLOCKS * 1024;
        const __bf16* Wo = Wqkv + (size_t)16 * 48 * 1024;
        const __bf16* W1 = Wo + (size_t)16 * 16 * 1024;
        const __bf16* W2 = W1 + (size_t)16 * 32 * 1024;
        using K16 = std::integral_constant<int, 16>;
        using N2 = std::integral_constant<int, 2>;
        using N1 = std::integral_constant<int, 1>;
        // ---- in_proj: 48 / G column blocks of this member (its heads' q | k | v), 8 waves
        if constexpr (G == 1) {
            for (int p = 0; p < 3; ++p) pass(Wqkv, 48, 16 * p + 2 * wave, 0, K16{}, N2{}, 0);
        } else if constexpr (G == 2) {
            pass(Wqkv, 48, 24 * mem + 2 * wave, 0, K16{}, N2{}, 0);                           // 16 of its 24 blocks
            pass(Wqkv, 48, 24 * mem + 16 + wave, 0, K16{}, N1{}, 0);                          // the other 8: one per wave
        } else {
            pass(Wqkv, 48, 12 * mem + wave, 0, K16{}, N1{}, 0);                               // 8 of its 12 blocks
            if (wave < 4) pass(Wqkv, 48, 12 * mem + 8 + wave, 0, K16{}, N1{}, 0);             // the other 4
        }
        sink += valu_chain(g.valu_attn / G, sink);
        __syncthreads();
        if (g.xmask & 1) exchange(65536 / G, 65536, 65536, epoch, false);                     // attention output slice -> image Y of every member
        // ---- out_proj (all 512 columns, redundant) + residual / norm1 / norm2
        pass(Wo, 16, 2 * wave, 0, K16{}, N2{}, 65536);
        sink += valu_chain(g.valu_ln12, sink);
        __syncthreads();
        // ---- FFN: linear1 over 32 / G column blocks, GELU, linear2 over the matching 32 / G k-blocks -> partial sums
        if constexpr (G == 1) {
            for (int c = 0; c < 2; ++c) {
                pass(W1, 32, 16 * c + 2 * wave, 0, K16{}, N2{}, 0);
                sink += valu_chain(g.valu_gelu / 2, sink);
                __syncthreads();
                pass(W2, 16, 2 * wave, 16 * c, K16{}, N2{}, 65536);
            }
        } else if constexpr (G == 2) {
            pass(W1, 32, 16 * mem + 2 * wave, 0, K16{}, N2{}, 0);
            sink += valu_chain(g.valu_gelu / 2, sink);
            __syncthreads();
            pass(W2, 16, 2 * wave, 16 * mem, K16{}, N2{}, 65536);
        } else {
            pass(W1, 32, 8 * mem + wave, 0, K16{}, N1{}, 0);
            sink += valu_chain(g.valu_gelu / 4, sink);
            __syncthreads();
            pass(W2, 16, 2 * wave, 8 * mem, std::integral_constant<int, 8>{}, N2{}, 65536);
        }
        __syncthreads();
        if (g.xmask & 2) exchange(131072, 0, 0, epoch, true);                                 // all-reduce of the fp32 partials [64][512]
        sink += valu_chain(g.valu_ln3, sink);
        __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) g.c->cyc[blockIdx.x] = t1 - t0;
    float s = sink;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) s += acc[a][b][0];
    if (s == 12345.678f) g.c->timeouts = -1;
}

template <int G>
void run(int samples, int layers, const Args& a0, Ctrl* c, int xmask = 3, bool valu = true) {
    Args a = a0;
    a.layers = layers;
    a.xmask = xmask;
    if (!valu) a.valu_attn = a.valu_ln12 = a.valu_gelu = a.valu_ln3 = 0;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gcu<G>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int wgs = samples * G;
    if (wgs > 256 || wgs % (8 * G)) { printf("G = %d: %d samples -> %d workgroups: skipped (needs a multiple of %d, <= 256)\n", G, samples, wgs, 8 * G); return; }
    double best = 1e30, worst = 0;
    int to = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(c, 0, sizeof(Ctrl)));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gcu<G>, dim3(wgs), dim3(NTH), LDS_BYTES, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        Ctrl h;
        CHECK(hipMemcpy(&h, c, sizeof(Ctrl), hipMemcpyDeviceToHost));
        long long mx = 0;
        for (int i = 0; i < wgs; ++i) mx = h.cyc[i] > mx ? h.cyc[i] : mx;
        const double us = 1e3 * ms / layers;
        if (us < best) best = us;
        if ((double)mx / layers > worst) worst = (double)mx / layers;
        to += h.timeouts;
    }
    printf("G = %d  %3d samples on %3d CUs [%s%s%s]: %7.2f us per layer (launch / layers; slowest workgroup %6.1f k counter ticks) -> x 8 + 30 us boundary = %6.1f us per step%s\n",
           G, samples, wgs, (xmask & 1) ? "X1 " : "", (xmask & 2) ? "X2 " : "", valu ? "VALU" : "", best, worst / 1e3, 8 * best + 30.0, to ? "   [BARRIER TIMEOUTS]" : "");
}

int main(int argc, char** argv) {
    const int samples = argc > 1 ? atoi(argv[1]) : 64, layers = argc > 2 ? atoi(argv[2]) : 400;
    const size_t wbytes = (size_t)8 * LAYER_BLOCKS * 2048;
    __bf16* W;
    char* xbuf;
    Ctrl* c;
    CHECK(hipMalloc(&W, wbytes));
    CHECK(hipMemset(W, 0x3c, wbytes));                     // small positive bf16 values
    CHECK(hipMalloc(&xbuf, (size_t)256 * 4 * 131072));
    CHECK(hipMalloc(&c, sizeof(Ctrl)));
    // stand-ins, per wave and layer (k_layers' stamps, DESIGN.md 4.0d: VALU phases ~41 k of a layer's 112 k cycles with two waves per SIMD in the same
    // phase: 8 cycles of SIMD time per instruction pair): attention 2 x ~5 k, norm1 + norm2 + images 11.4 k, GELU ~11 k, norm3 6.3 k
    Args a{W, xbuf, c, 0, 10000 / 8, 11400 / 8, 11000 / 8, 6300 / 8, 3};
    printf("G-CUs-per-sample layer skeleton: %d samples, %d layers per launch (weights: 8 layers x 4 MiB, L2-resident per XCD like the real stream)\n", samples, layers);
    for (int smp : {samples, 32}) {
        run<1>(smp, layers, a, c);
        run<1>(smp, layers, a, c, 3, false);
        run<2>(smp, layers, a, c);
        run<2>(smp, layers, a, c, 0);
        run<4>(smp, layers, a, c);
        run<4>(smp, layers, a, c, 1);
        run<4>(smp, layers, a, c, 2);
        run<4>(smp, layers, a, c, 0);
        run<4>(smp, layers, a, c, 0, false);
    }
    return 0;
}
