// Does the integer-multiply pipe (v_mad_u64_u32) co-execute with plain VALU ops?  Times N mads alone, M adds alone and
// both interleaved (same wave), with 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(X) X X X X
#define MADS "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
#define ADDS "v_add_u32 %4, %8, %4\n\tv_add_u32 %5, %8, %5\n\tv_add_u32 %6, %8, %6\n\tv_add_u32 %7, %8, %7\n\tv_xor_b32 %4, %9, %4\n\tv_xor_b32 %5, %9, %5\n\tv_xor_b32 %6, %9, %6\n\tv_xor_b32 %7, %9, %7\n\t"
#define MIX  "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_add_u32 %4, %8, %4\n\tv_add_u32 %5, %8, %5\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_add_u32 %6, %8, %6\n\tv_add_u32 %7, %8, %7\n\t" \
             "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_xor_b32 %4, %9, %4\n\tv_xor_b32 %5, %9, %5\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_xor_b32 %6, %9, %6\n\tv_xor_b32 %7, %9, %7\n\t"
template <int MODE> __global__ __launch_bounds__(256) void k(int iters, unsigned* sink) {
    unsigned tid = blockIdx.x * 256 + threadIdx.x, a = tid * 2654435761u, b = tid | 1u;
    unsigned long long c0 = tid, c1 = tid + 1, c2 = tid + 2, c3 = tid + 3; unsigned d0 = tid, d1 = tid + 5, d2 = tid + 6, d3 = tid + 7;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) asm volatile(R4(MADS) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");
        if (MODE == 1) asm volatile(R4(ADDS) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");
        if (MODE == 2) asm volatile(R4(MIX) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "vcc");
    }
    unsigned out = (unsigned)(c0 ^ c1 ^ c2 ^ c3) ^ d0 ^ d1 ^ d2 ^ d3;
    if (out == 0x12345678u) sink[0] = out;
}
int main() {
    unsigned* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    const char* names[3] = {"16 mads / iter", "32 simple / iter", "16 mads + 32 simple interleaved"};
    for (int rep = 0; rep < 2; rep++)
    for (int m = 0; m < 3; m++) {
        hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, iters, sink);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, iters, sink);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double waves_per_simd = blocks * 4.0 / (256 * 4);
        printf("%-36s %8.3f ms   -> %.2f ns per iteration per SIMD-wave-slot\n", names[m], ms, ms * 1e6 / iters / waves_per_simd);
    }
    return 0;
}
