// Do clean lines in an XCD's L2 survive a kernel boundary? (Decides whether a small-batch kernel can pre-touch the NEXT kernel's weight slice: DESIGN.md 11.2.)
// Kernel A (256 workgroups, one per CU) reads workgroup-private 32 KiB slices of a buffer; kernel B, launched behind it on the same stream with the same
// block -> slice map (block b runs on XCD b % 8 in both), times a dependent pointer-chase through its slice with s_memtime. Compared: B behind A on the
// SAME slices (L2-warm if lines survive), B behind an A that read OTHER slices of the same 8 MB set (L2-cold, MALL-warm), and B reading its slice twice
// (second pass: L2-hit latency as the reference point).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_across_kernels.hip -o tools/bin/l2_across_kernels
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));             \
            exit(1);                                                   \
        }                                                              \
    } while (0)

constexpr int SLICE_LINES = 256;   // 32 KiB per workgroup = 256 lines of 128 B

__global__ void k_touch(const int* buf, int shift, int* sink) {
    const int b = (blockIdx.x + shift) % gridDim.x;
    const int* p = buf + (size_t)b * SLICE_LINES * 32;
    int acc = 0;
    for (int i = threadIdx.x; i < SLICE_LINES * 32; i += blockDim.x) acc += p[i];
    if (acc == 0x7fffffff) sink[0] = acc;
}
// one lane chases line -> line (each line's first word holds the index of the next line of the slice); cycles per hop
__global__ void k_chase(const int* buf, long long* cyc, int passes, int* sink) {
    if (threadIdx.x != 0) return;
    const int* p = buf + (size_t)blockIdx.x * SLICE_LINES * 32;
    int idx = 0;
    for (int ps = 0; ps < passes; ++ps) {
        const long long t0 = __builtin_readcyclecounter();
        for (int h = 0; h < SLICE_LINES; ++h) idx = p[idx * 32];
        const long long t1 = __builtin_readcyclecounter();
        cyc[blockIdx.x * 2 + (ps ? 1 : 0)] = t1 - t0;
    }
    if (idx == 0x7fffffff) sink[0] = idx;
}

int main() {
    const int G = 256;
    const size_t n = (size_t)G * SLICE_LINES * 32;
    std::vector<int> h(n, 0);
    for (int b = 0; b < G; ++b)
        for (int l = 0; l < SLICE_LINES; ++l) h[((size_t)b * SLICE_LINES + l) * 32] = (l * 97 + 31) % SLICE_LINES;   // a permutation walk (97 coprime to 256)
    int *buf, *sink;
    long long* cyc;
    CHECK(hipMalloc(&buf, n * 4));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&cyc, G * 2 * 8));
    CHECK(hipMemcpy(buf, h.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<long long> c(G * 2);
    auto report = [&](const char* what, int slot) {
        CHECK(hipMemcpy(c.data(), cyc, G * 2 * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (int b = 0; b < G; ++b) s += (double)c[b * 2 + slot] / SLICE_LINES;
        printf("%-86s %7.0f cycles per dependent line load\n", what, s / G);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_touch, dim3(G), dim3(256), 0, 0, buf, 0, sink);
        hipLaunchKernelGGL(k_chase, dim3(G), dim3(64), 0, 0, buf, cyc, 2, sink);
        CHECK(hipDeviceSynchronize());
        report("kernel B behind a kernel A that read the SAME slices on the same XCDs (first pass)", 0);
        report("  ... second pass inside kernel B (L2 / L1-hit reference)", 1);
        hipLaunchKernelGGL(k_touch, dim3(G), dim3(256), 0, 0, buf, 3, sink);     // block b reads slice b + 3: another XCD's slices
        hipLaunchKernelGGL(k_chase, dim3(G), dim3(64), 0, 0, buf, cyc, 1, sink);
        CHECK(hipDeviceSynchronize());
        report("kernel B behind a kernel A that read OTHER slices (this XCD's L2 cold, MALL warm)", 0);
    }
    return 0;
}
