// What would one PHASE of a persistent single-XCD small-batch kernel cost? (the B = 1 path is 43 dependent launches of 4.7 - 8.3 us per
// sampler step; the alternative on the table is ONE persistent launch on the 32 CUs of one XCD - one L2 for every hand-off - with XCD-local
// barriers where the kernel boundaries are.) This tool measures that floor on the hardware, placement-independently:
//   census     every workgroup (one per CU: 160 KiB of LDS) reads HW_REG_XCC_ID and takes a slot in its XCD's counter; after a chip-wide
//              rendezvous the workgroups of the XCD that block 0 landed on stay (P of them), the others exit
//   mode 0     P workgroups x `phases` barriers: monotonic counter in L2 (relaxed agent-scope atomics), sc1-load poll + s_sleep
//   mode 1     ... + fence(acquire, "agent") (buffer_inv sc1: what a consumer needs before plain loads of other CUs' data)
//   mode 2     hand-off by sc1 loads, no fence: every workgroup stores its 1/P slice of a 64 KiB activation tile (plain stores + vmcnt(0)),
//              barrier, then reads the WHOLE tile (the next column-split GEMM's A operand) with sc1 loads and checks every word
//   mode 3     the same hand-off with plain loads behind an acquire fence
//   mode 4     mode 2 + a 48 KiB weight slice per workgroup and phase streamed from a 54 MB buffer (L2-cold), requested one phase ahead
//   chip modes 10 / 12: modes 0 / 2 with ALL 256 workgroups (release fence + acquire fence: different XCDs, different L2s)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcd_phase_bench.hip -o tools/bin/xcd_phase_bench ; run: tools/bin/xcd_phase_bench [phases]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct Ctrl {
    int count[16];        // census: workgroups per XCC
    int arrived;          // chip-wide rendezvous
    int target;           // XCC of block 0
    int bar;              // barrier counter (monotonic)
    int bar2;             // second barrier of a hand-off phase (readers done)
    int errors;
    long long t0, t1;
};

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                            \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ __launch_bounds__(256) void k_phase(Ctrl* c, int* tile, const int4* weights, size_t wvec, int phases, int* sink) {
    extern __shared__ int lds[];
    constexpr bool CHIP = MODE >= 10;
    constexpr int M = MODE % 10;
    const int tid = threadIdx.x;
    __shared__ int s_rank, s_P;
    if (tid == 0) {
        int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        const int slot = atomicAdd(&c->count[xcc], 1);
        if (blockIdx.x == 0) __hip_atomic_store(&c->target, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        atomicAdd(&c->arrived, 1);
        while (ld_sc1(&c->arrived) < (int)gridDim.x) __builtin_amdgcn_s_sleep(2);
        __threadfence();
        const int tg = ld_sc1(&c->target);
        if (CHIP) {
            s_rank = blockIdx.x;
            s_P = gridDim.x;
        } else {
            s_rank = xcc == tg ? slot : -1;
            s_P = ld_sc1(&c->count[tg]);
        }
    }
    __syncthreads();
    const int rank = s_rank, P = s_P;
    if (rank < 0) return;
    const int words = 16384 / P;                       // this workgroup's slice of the 64 KiB tile, in ints
    int acc = 0, err = 0;
    int4 wreg[12];                                     // 48 KiB per workgroup and phase = 12 x 16 B per thread
    size_t wpos = (size_t)rank * 3072;
    if (M == 4) {
#pragma unroll
        for (int j = 0; j < 12; ++j) wreg[j] = weights[(wpos + (size_t)j * 256 + tid) % wvec];
    }
    if (rank == 0 && tid == 0) c->t0 = wall_clock64();
    for (int ph = 0; ph < phases; ++ph) {
        if (M >= 2) {
            for (int i = 4 * tid; i < words; i += 1024) {
                const int b0 = ph * 131 + rank * words + i;
                *reinterpret_cast<int4*>(tile + rank * words + i) = int4{b0, b0 + 1, b0 + 2, b0 + 3};
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (M == 4) {                                  // consume last phase's slice, request the next one (another 48 KiB further on: L2-cold)
#pragma unroll
            for (int j = 0; j < 12; ++j) acc += wreg[j].x ^ wreg[j].w;
            wpos += (size_t)P * 3072;
#pragma unroll
            for (int j = 0; j < 12; ++j) wreg[j] = weights[(wpos + (size_t)j * 256 + tid) % wvec];
        }
        __syncthreads();
        if (tid == 0) {
            if (CHIP) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(&c->bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int want = (ph + 1) * P;
            while (ld_sc1(&c->bar) < want) __builtin_amdgcn_s_sleep(1);
            if (M == 1 || M == 3 || CHIP) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (M >= 2) {                                  // every workgroup reads the whole tile
            const int4* t4 = reinterpret_cast<const int4*>(tile);      // 16-byte loads, 16 per thread in flight
            int4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (M == 3 || CHIP) v[j] = t4[tid + 256 * j];
                else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(t4 + tid + 256 * j) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = 4 * (tid + 256 * j), b0 = ph * 131 + i;
                err += (v[j].x != b0) + (v[j].y != b0 + 1) + (v[j].z != b0 + 2) + (v[j].w != b0 + 3);
            }
            __syncthreads();                           // (everyone is done reading before anyone overwrites: second barrier of a real phase
            if (tid == 0) {                            //  would be the next phase's own barrier on a second buffer; kept simple: counted here)
                __hip_atomic_fetch_add(&c->bar2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int want = (ph + 1) * P;
                while (ld_sc1(&c->bar2) < want) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
        }
    }
    if (rank == 0 && tid == 0) c->t1 = wall_clock64();
    if (err) atomicAdd(&c->errors, err);
    if (acc == 0x12345678) sink[0] = acc;
}

template <int MODE>
void run(const char* what, int phases, int* tile, int4* w, size_t wvec, int* sink) {
    Ctrl* c;
    CHECK(hipMalloc(&c, sizeof(Ctrl) + 64));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_phase<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    double best = 1e30;
    int P = 0, errs = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(c, 0, sizeof(Ctrl) + 64));
        hipLaunchKernelGGL(k_phase<MODE>, dim3(256), dim3(256), 150 * 1024, 0, c, tile, w, wvec, phases, sink);
        CHECK(hipDeviceSynchronize());
        Ctrl h;
        CHECK(hipMemcpy(&h, c, sizeof(Ctrl), hipMemcpyDeviceToHost));
        const double us = (double)(h.t1 - h.t0) / 100.0 / phases;      // wall_clock64: 100 MHz
        if (us < best) best = us;
        P = MODE >= 10 ? 256 : h.count[h.target];
        errs += h.errors;
    }
    printf("%-92s P = %3d  %7.2f us per phase  (stale words: %d)\n", what, P, best, errs);
    CHECK(hipFree(c));
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 2000;
    int *tile, *sink;
    int4* w;
    const size_t wbytes = (size_t)54 << 20, wvec = wbytes / 16;
    CHECK(hipMalloc(&tile, 65536));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&w, wbytes));
    CHECK(hipMemset(w, 1, wbytes));
    printf("persistent-phase floor, %d phases, 256 workgroups launched (one per CU), wall_clock64\n", phases);
    run<0>("one XCD: barrier only (counter in the XCD's L2, sc1 poll)", phases, tile, w, wvec, sink);
    run<1>("one XCD: barrier + fence(acquire, agent) [buffer_inv sc1]", phases, tile, w, wvec, sink);
    run<2>("one XCD: 64 KiB tile handed over (plain stores + vmcnt(0) | barrier | sc1 loads of the whole tile | barrier)", phases, tile, w, wvec, sink);
    run<3>("one XCD: 64 KiB tile handed over (plain stores | barrier + acquire fence | plain loads | barrier)", phases, tile, w, wvec, sink);
    run<4>("one XCD: mode 2 + a 48 KiB weight slice per workgroup and phase, L2-cold, requested a phase ahead", phases, tile, w, wvec, sink);
    run<10>("whole chip: barrier with release + acquire fences", phases, tile, w, wvec, sink);
    run<12>("whole chip: 64 KiB tile handed over (release fence | barrier | acquire fence | plain loads | barrier)", phases, tile, w, wvec, sink);
    return 0;
}
