#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "plume_fe_asm.h"
using namespace plume;
__global__ void kcheck(uint32_t* out, const uint32_t* in, int n) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    fe a, b, r1, r2, r3, r4;
    for (int i = 0; i < 8; i++) { a.v[i] = in[id * 16 + i]; b.v[i] = in[id * 16 + 8 + i]; }
    fe_mul(r1, a, b); fe_mul_asm(r2, a, b); fe_sqr(r3, a); fe_sqr_asm(r4, a);
    for (int i = 0; i < 8; i++) { out[id * 32 + i] = r1.v[i]; out[id * 32 + 8 + i] = r2.v[i]; out[id * 32 + 16 + i] = r3.v[i]; out[id * 32 + 24 + i] = r4.v[i]; }
}
template <int V> __global__ __launch_bounds__(256) void kbench(uint32_t* out, const uint32_t* in, int iters) {
    unsigned tid = blockIdx.x * 256 + threadIdx.x;
    fe x, y; for (int i = 0; i < 8; i++) { x.v[i] = in[(tid * 16 + i) & 1023]; y.v[i] = in[(tid * 16 + 8 + i) & 1023]; }
    for (int it = 0; it < iters; it++) {
        if (V == 0) { fe_mul(x, x, y); fe_mul(y, y, x); }
        if (V == 1) { fe_mul_asm(x, x, y); fe_mul_asm(y, y, x); }
        if (V == 2) { fe_sqr(x, x); fe_sqr(y, y); }
        if (V == 3) { fe_sqr_asm(x, x); fe_sqr_asm(y, y); }
    }
    uint32_t acc = 0; for (int i = 0; i < 8; i++) acc ^= x.v[i] ^ y.v[i];
    out[tid] = acc;
}
int main() {
    const int n = 8192;
    static uint32_t in[n * 16], out[n * 32];
    srand(7);
    for (int i = 0; i < n * 16; i++) { int k = rand() % 8; in[i] = k == 0 ? 0u : k == 1 ? 0xFFFFFFFFu : ((uint32_t)rand() << 16) ^ rand(); }
    // a few fully extreme operands
    for (int i = 0; i < 16; i++) in[i] = 0xFFFFFFFFu;
    for (int i = 16; i < 32; i++) in[i] = 0;
    uint32_t pw[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    memcpy(in + 32, pw, 32); memcpy(in + 40, pw, 32);
    uint32_t *di, *dout; hipMalloc(&di, sizeof in); hipMalloc(&dout, sizeof out); hipMemcpy(di, in, sizeof in, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kcheck, dim3(n / 64), dim3(64), 0, 0, dout, di, n);
    hipError_t e = hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    int badm = 0, bads = 0, badhost = 0;
    for (int i = 0; i < n; i++) {
        if (memcmp(out + 32 * i, out + 32 * i + 8, 32)) { if (!badm) { printf("first mul mismatch at %d\n", i); for (int j = 0; j < 8; j++) printf("  [%d] c++ %08x asm %08x\n", j, out[32*i+j], out[32*i+8+j]); } badm++; }
        if (memcmp(out + 32 * i + 16, out + 32 * i + 24, 32)) { if (!bads) printf("first sqr mismatch at %d\n", i); bads++; }
        fe a, b, r; memcpy(a.v, in + 16 * i, 32); memcpy(b.v, in + 16 * i + 8, 32); fe_mul(r, a, b);
        if (memcmp(r.v, out + 32 * i, 32)) badhost++;
    }
    printf("hip: %s  mul mismatches %d, sqr mismatches %d, device-c++ vs host mismatches %d (of %d)\n", hipGetErrorString(e), badm, bads, badhost, n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* nm[4] = {"fe_mul c++", "fe_mul asm", "fe_sqr c++", "fe_sqr asm"};
    uint32_t* o2; hipMalloc(&o2, 4 * 256 * 2048);
    for (int rep = 0; rep < 2; rep++) for (int v = 0; v < 4; v++) {
        int iters = 8192;
        hipEventRecord(e0);
        switch (v) { case 0: hipLaunchKernelGGL(kbench<0>, dim3(2048), dim3(256), 0, 0, o2, di, iters); break; case 1: hipLaunchKernelGGL(kbench<1>, dim3(2048), dim3(256), 0, 0, o2, di, iters); break;
                     case 2: hipLaunchKernelGGL(kbench<2>, dim3(2048), dim3(256), 0, 0, o2, di, iters); break; case 3: hipLaunchKernelGGL(kbench<3>, dim3(2048), dim3(256), 0, 0, o2, di, iters); break; }
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-12s %8.3f ms  %.3e ops/s\n", nm[v], ms, 2.0 * iters * 2048 * 256 / (ms * 1e-3));
    }
    return (badm || bads) ? 1 : 0;
}
