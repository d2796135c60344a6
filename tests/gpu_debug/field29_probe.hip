#include <hip/hip_runtime.h>
#include <cstdio>
#include "plume_field29.h"
#include "plume_fe29_cols.h"
using namespace plume;
template <int V> __global__ __launch_bounds__(256) void k(uint32_t* out, const uint32_t* in, int iters) {
    unsigned tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    if (V == 0) { fe x, y; for (int i = 0; i < 8; i++) { x.v[i] = in[(tid * 16 + i) & 1023]; y.v[i] = in[(tid * 16 + 8 + i) & 1023]; }
        for (int it = 0; it < iters; it++) { fe_mul(x, x, y); fe_mul(y, y, x); } for (int i = 0; i < 8; i++) acc ^= x.v[i] ^ y.v[i]; }
    if (V == 1) { fe29 x, y; for (int i = 0; i < 9; i++) { x.v[i] = in[(tid * 18 + i) & 1023] & 0x1FFFFFFF; y.v[i] = in[(tid * 18 + 9 + i) & 1023] & 0x1FFFFFFF; } x.v[8] &= 0xFFFFFF; y.v[8] &= 0xFFFFFF;
        for (int it = 0; it < iters; it++) { fe29_mul(x, x, y); fe29_mul(y, y, x); } for (int i = 0; i < 9; i++) acc ^= x.v[i] ^ y.v[i]; }
    if (V == 2) { fe x, y; for (int i = 0; i < 8; i++) { x.v[i] = in[(tid * 16 + i) & 1023]; y.v[i] = in[(tid * 16 + 8 + i) & 1023]; }
        for (int it = 0; it < iters; it++) { fe_sqr(x, x); fe_sqr(y, y); } for (int i = 0; i < 8; i++) acc ^= x.v[i] ^ y.v[i]; }
    if (V == 3) { fe29 x, y; for (int i = 0; i < 9; i++) { x.v[i] = in[(tid * 18 + i) & 1023] & 0x1FFFFFFF; y.v[i] = in[(tid * 18 + 9 + i) & 1023] & 0x1FFFFFFF; } x.v[8] &= 0xFFFFFF; y.v[8] &= 0xFFFFFF;
        for (int it = 0; it < iters; it++) { fe29_sqr(x, x); fe29_sqr(y, y); } for (int i = 0; i < 9; i++) acc ^= x.v[i] ^ y.v[i]; }
    if (V == 4) { fe x, y, z; for (int i = 0; i < 8; i++) { x.v[i] = in[(tid * 16 + i) & 1023]; y.v[i] = in[(tid * 16 + 8 + i) & 1023]; z.v[i] = i; }
        for (int it = 0; it < iters; it++) { fe_sub(z, x, y); fe_add(x, z, y); fe_sub(y, x, z); fe_add(z, z, x); } for (int i = 0; i < 8; i++) acc ^= x.v[i] ^ y.v[i] ^ z.v[i]; }
    if (V == 5) { fe29 x, y, z; for (int i = 0; i < 9; i++) { x.v[i] = in[(tid * 18 + i) & 1023] & 0x1FFFFFFF; y.v[i] = in[(tid * 18 + 9 + i) & 1023] & 0x1FFFFFFF; z.v[i] = i; } x.v[8] &= 0xFFFFFF; y.v[8] &= 0xFFFFFF;
        for (int it = 0; it < iters; it++) { fe29_sub(z, x, y); fe29_add(x, z, y); fe29_carry(x); fe29_sub(y, x, z); fe29_add(z, z, x); fe29_carry(z); } for (int i = 0; i < 9; i++) acc ^= x.v[i] ^ y.v[i] ^ z.v[i]; }
    if (V == 6) { fe29 x, y; for (int i = 0; i < 9; i++) { x.v[i] = in[(tid * 18 + i) & 1023] & 0x1FFFFFFF; y.v[i] = in[(tid * 18 + 9 + i) & 1023] & 0x1FFFFFFF; } x.v[8] &= 0xFFFFFF; y.v[8] &= 0xFFFFFF;
        for (int it = 0; it < iters; it++) { fe29_mul2(x, x, y); fe29_mul2(y, y, x); } for (int i = 0; i < 9; i++) acc ^= x.v[i] ^ y.v[i]; }
    if (V == 7) { fe29 x, y; for (int i = 0; i < 9; i++) { x.v[i] = in[(tid * 18 + i) & 1023] & 0x1FFFFFFF; y.v[i] = in[(tid * 18 + 9 + i) & 1023] & 0x1FFFFFFF; } x.v[8] &= 0xFFFFFF; y.v[8] &= 0xFFFFFF;
        for (int it = 0; it < iters; it++) { fe29_sqr2(x, x); fe29_sqr2(y, y); } for (int i = 0; i < 9; i++) acc ^= x.v[i] ^ y.v[i]; }
    if (V == 8) { fe29 x, y; for (int i = 0; i < 9; i++) { x.v[i] = in[(tid * 18 + i) & 1023] & 0x1FFFFFFF; y.v[i] = in[(tid * 18 + 9 + i) & 1023] & 0x1FFFFFFF; } x.v[8] &= 0xFFFFFF; y.v[8] &= 0xFFFFFF;
        for (int it = 0; it < iters; it++) { fe29_mul4(x, x, y); fe29_mul4(y, y, x); } for (int i = 0; i < 9; i++) acc ^= x.v[i] ^ y.v[i]; }
    if (V == 9) { fe29 x, y; for (int i = 0; i < 9; i++) { x.v[i] = in[(tid * 18 + i) & 1023] & 0x1FFFFFFF; y.v[i] = in[(tid * 18 + 9 + i) & 1023] & 0x1FFFFFFF; } x.v[8] &= 0xFFFFFF; y.v[8] &= 0xFFFFFF;
        for (int it = 0; it < iters; it++) { fe29_sqr4(x, x); fe29_sqr4(y, y); } for (int i = 0; i < 9; i++) acc ^= x.v[i] ^ y.v[i]; }
    out[tid] = acc;
}
int main() {
    uint32_t *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 2048);
    uint32_t h[1024]; for (int i = 0; i < 1024; i++) h[i] = i * 2654435761u + 12345; hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* nm[10] = {"fe_mul 8x32", "fe_mul 9x29", "fe_sqr 8x32", "fe_sqr 9x29", "2add+2sub 8x32", "2add+2sub(+carry) 9x29", "fe_mul 9x29 v2", "fe_sqr 9x29 v2", "fe_mul 9x29 v4", "fe_sqr 9x29 v4"};
    for (int rep = 0; rep < 2; rep++) for (int v = 0; v < 10; v++) {
        int iters = (v < 4 || v > 5) ? 8192 : 65536;
        hipEventRecord(e0);
        switch (v) { case 0: hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, out, in, iters); break; case 1: hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, out, in, iters); break;
                     case 2: hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, out, in, iters); break; case 3: hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, out, in, iters); break;
                     case 4: hipLaunchKernelGGL(k<4>, dim3(2048), dim3(256), 0, 0, out, in, iters); break; case 5: hipLaunchKernelGGL(k<5>, dim3(2048), dim3(256), 0, 0, out, in, iters); break;
                     case 6: hipLaunchKernelGGL(k<6>, dim3(2048), dim3(256), 0, 0, out, in, iters); break; case 7: hipLaunchKernelGGL(k<7>, dim3(2048), dim3(256), 0, 0, out, in, iters); break;
                     case 8: hipLaunchKernelGGL(k<8>, dim3(2048), dim3(256), 0, 0, out, in, iters); break; case 9: hipLaunchKernelGGL(k<9>, dim3(2048), dim3(256), 0, 0, out, in, iters); break; }
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-26s %8.3f ms  %.3e ops/s\n", nm[v], ms, 2.0 * iters * 2048 * 256 / (ms * 1e-3) * ((v < 4 || v > 5) ? 1 : 2));
    }
}
