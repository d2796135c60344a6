// Cross-wave co-issue probe: waves with even id run only v_mad_u64_u32, odd waves only plain VALU ops.
// If the integer-multiply pipe runs beside the plain VALU pipe for DIFFERENT waves, the mixed launch takes ~max, not ~sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
#define MAD4 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3\n\t"
#define ADD8 "v_add_u32 %0, %4, %0\n\tv_add_u32 %1, %4, %1\n\tv_add_u32 %2, %4, %2\n\tv_add_u32 %3, %4, %3\n\tv_xor_b32 %0, %5, %0\n\tv_xor_b32 %1, %5, %1\n\tv_xor_b32 %2, %5, %2\n\tv_xor_b32 %3, %5, %3\n\t"
// mode 0: all waves mads; 1: all waves adds; 2: even waves mads, odd waves adds (same per-wave work as in 0 / 1)
__global__ __launch_bounds__(256) void k(int mode, int iters, unsigned* sink) {
    unsigned tid = blockIdx.x * 256 + threadIdx.x, a = tid * 2654435761u, b = tid | 1u;
    unsigned wave = threadIdx.x >> 6;
    bool do_mad = mode == 0 || (mode == 2 && (wave & 1) == 0);
    unsigned long long c0 = tid, c1 = tid + 1, c2 = tid + 2, c3 = tid + 3; unsigned d0 = tid, d1 = tid + 5, d2 = tid + 6, d3 = tid + 7;
    if (do_mad) { for (int it = 0; it < iters; it++) asm volatile(R8(MAD4) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc"); }
    else        { for (int it = 0; it < iters; it++) asm volatile(R8(ADD8) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b)); }
    unsigned out = (unsigned)(c0 ^ c1 ^ c2 ^ c3) ^ d0 ^ d1 ^ d2 ^ d3;
    if (out == 0x12345678u) sink[0] = out;
}
int main() {
    unsigned* sink; (void)hipMalloc(&sink, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[3] = {"all waves: 32 mads/iter", "all waves: 64 plain/iter", "even waves mads, odd waves plain"};
    for (int wpb = 0; wpb < 2; wpb++) {
        int blocks = wpb == 0 ? 256 * 8 : 256 * 2;      // 8 or 2 waves per SIMD
        printf("--- %d waves per SIMD\n", wpb == 0 ? 8 : 2);
        for (int rep = 0; rep < 2; rep++) for (int m = 0; m < 3; m++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, m, iters, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-36s %8.3f ms\n", names[m], ms);
        }
    }
    return 0;
}
