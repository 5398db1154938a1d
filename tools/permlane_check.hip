// v_permlane32_swap_b32 through inline asm (the builtin's second result is miscompiled by ROCm 7.2's clang: it adds vdst to itself):
// checks [a.lo | b.lo], [a.hi | b.hi] semantics back to back with VALU producers / consumers, 1 M times.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__global__ void k(int* bad, int iters) {
    const int lane = threadIdx.x & 63;
    int nb = 0;
    for (int it = 0; it < iters; ++it) {
        float s = lane + it * 3.0f, q = 1000.f + lane + it * 7.0f;
        s = s * 1.0f + 0.0f;                   // VALU producer right in front
        float a = s, b = q;
        swap32(a, b);
        const float sum = a + b;               // VALU consumer right behind: kh = 0 lanes: s.lo + s.hi, kh = 1 lanes: q.lo + q.hi
        const int l31 = lane & 31;
        const float es = (l31 + it * 3.0f) + (l31 + 32 + it * 3.0f), eq = (1000.f + l31 + it * 7.0f) + (1000.f + l31 + 32 + it * 7.0f);
        if (sum != (lane < 32 ? es : eq)) ++nb;
    }
    atomicAdd(bad, nb);
}
int main() {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 4096);
    int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("permlane32_swap mismatches: %d\n", h);
    return h != 0;
}
