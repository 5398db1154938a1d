// Micro-benchmark: how many bytes per clock one CU can pull from L2 (a) into LDS by direct-to-LDS DMA
// (global_load_lds_dwordx4), (b) into VGPRs (global_load_dwordx4), (c) both at once. Decides how the plain-bf16
// phase GEMM should feed its operands (the split-bf16 GEMM measured ~20 B/clk/CU of LDS-DMA).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_paths_bench.hip -o tools/bin/l2_paths_bench
// Every workgroup streams the same L2-resident buffer (operand-like: all CUs of an XCD re-read one weight matrix)
// from a workgroup-specific start offset, in 1 KiB wave-instructions.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mode bit 0: waves with (wave & 1) == 0 (or all, if mode == 1) do LDS-DMA; bit 1: the others (or all, mode == 2) load to VGPRs
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_stream(const char* __restrict__ buf, size_t buf_bytes, int iters, int mode, unsigned* sink,
                                                   long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NW x 8 KiB ring slots
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool dma = mode == 1 || (mode == 3 && (wave & 1) == 0);
    // wave-private stream position: start staggered per workgroup and wave, stride 1 KiB per instruction
    size_t pos = ((size_t)blockIdx.x * 131072 + (size_t)wave * 16384) % buf_bytes;
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (dma) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const char* src = buf + ((pos + (size_t)j * 1024) % buf_bytes) + lane * 16;
                __builtin_amdgcn_global_load_lds((const AS1 void*)src, (AS3 void*)(smem + wave * 8192 + j * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // keep half the burst in flight
        } else {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4*>(buf + ((pos + (size_t)j * 1024) % buf_bytes) + lane * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j];
        }
        pos = (pos + 8192) % buf_bytes;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    if (dma) acc[0] ^= *reinterpret_cast<const unsigned*>(smem + wave * 8192 + lane * 4);
    if (acc[0] == 0x12345678u && acc[1] == 7u) sink[0] = acc[2] ^ acc[3];
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NW>
static void run(const char* d, size_t bytes, int mode, int wgs_per_cu, unsigned* sink, long long* dcyc) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, NW * 8192));
    hipLaunchKernelGGL(k_stream<NW>, dim3(grid), dim3(64 * NW), NW * 8192, 0, d, bytes, 50, mode, sink, dcyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_stream<NW>, dim3(grid), dim3(64 * NW), NW * 8192, 0, d, bytes, iters, mode, sink, dcyc);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<long long> cyc(grid);
    CK(hipMemcpy(cyc.data(), dcyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= grid;
    const double total = (double)grid * NW * iters * 8192.0;
    const double per_cu_Bclk = (double)wgs_per_cu * NW * iters * 8192.0 / mean;   // bytes per shader clock per CU
    printf("  mode=%s waves/WG=%2d WG/CU=%d buf=%5.1f MB : %7.1f GB/s chip, %6.1f GB/s per CU, %5.1f B/clk/CU (clock64), kernel %.3f ms\n",
           mode == 1 ? "lds-dma " : mode == 2 ? "vgpr    " : "dma+vgpr", NW, wgs_per_cu, bytes / 1e6, total / ms / 1e6, total / ms / 1e6 / 256, per_cu_Bclk, ms);
}

int main() {
    unsigned* sink; long long* dcyc;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&dcyc, 4096 * sizeof(long long)));
    for (size_t mb : {1, 2, 16, 128}) {   // 1-2 MB: L2-resident per XCD; 16 MB: all-XCD L2 / MALL; 128 MB: MALL / HBM
        const size_t bytes = mb << 20;
        char* d;
        CK(hipMalloc(&d, bytes + 65536));
        CK(hipMemset(d, 1, bytes + 65536));
        printf("buffer %zu MB\n", mb);
        for (int mode : {1, 2, 3}) {
            run<4>(d, bytes, mode, 1, sink, dcyc);
            run<4>(d, bytes, mode, 2, sink, dcyc);
            run<8>(d, bytes, mode, 1, sink, dcyc);
            run<8>(d, bytes, mode, 2, sink, dcyc);
            run<16>(d, bytes, mode, 1, sink, dcyc);
        }
        CK(hipFree(d));
    }
    return 0;
}
