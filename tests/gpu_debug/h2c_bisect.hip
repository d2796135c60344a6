// GPU-vs-host bisect of hash_to_curve stages: the same header code runs on the device and on the host; prints the
// first stage whose outputs differ.  Test tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "plume_stages.h"
using namespace plume;

struct Dump { uint32_t b0[8]; uint32_t uni[24]; fe u0, u1; fe xn, xd, y; jac q0, q1, h; uint32_t sha_abc[8]; fe c1; fe sq; fe ms; };

__host__ __device__ void run(Dump& d, const uint8_t* msg, uint32_t mlen, const fe& pkx, uint32_t tag) {
    xmd_b0(d.b0, msg, mlen, pkx, tag, PLUME_ENC_POINT);
    uint32_t x[8];
    xmd_bi(d.uni, d.b0, 1);
    for (int i = 0; i < 8; i++) x[i] = d.b0[i] ^ d.uni[i];
    xmd_bi(d.uni + 8, x, 2);
    for (int i = 0; i < 8; i++) x[i] = d.b0[i] ^ d.uni[8 + i];
    xmd_bi(d.uni + 16, x, 3);
    fe_from_be48_words(d.u0, d.uni); fe_from_be48_words(d.u1, d.uni + 12);
    sswu_frac(d.xn, d.xd, d.y, d.u0);
    iso3_frac_to_jac(d.q0, d.xn, d.xd, d.y);
    fe xn, xd, y;
    sswu_frac(xn, xd, y, d.u1);
    iso3_frac_to_jac(d.q1, xn, xd, y);
    d.h = d.q0; jac_add(d.h, d.q1);
    sha256_init(d.sha_abc);
    sha256_absorb_pad(d.sha_abc, 0u, mlen, [&](uint32_t pos) -> uint32_t { return msg[pos]; });
    fe_pow_c1(d.c1, d.u0);
    fe_sqr(d.sq, d.u0);
    fe_mul_small(d.ms, d.u0, 1771);
}
__global__ void k(Dump* d, const uint8_t* msg, uint32_t mlen, fe pkx, uint32_t tag) { if (threadIdx.x == 0) run(*d, msg, mlen, pkx, tag); }

#define CMP(field) if (memcmp(&hd.field, &dd.field, sizeof hd.field)) { printf("MISMATCH at %s\n", #field); const uint32_t* a=(const uint32_t*)&hd.field; const uint32_t* b=(const uint32_t*)&dd.field; for (size_t i=0;i<sizeof(hd.field)/4;i++) printf("  [%zu] host %08x dev %08x\n", i, a[i], b[i]); bad++; }

int main() {
    const char* m = "An example app message string";
    uint32_t mlen = 29;
    fe pkx = fe_set(0x0cec028eu, 0xe08d09e0u, 0x2672a683u, 0x10814354u, 0xf9eabfffu, 0x0de6daccu, 0x1cd3a774u, 0x496076aeu);
    Dump hd; memset(&hd, 0, sizeof hd);
    run(hd, (const uint8_t*)m, mlen, pkx, 3);
    Dump* d_d; uint8_t* d_m;
    hipMalloc(&d_d, sizeof(Dump)); hipMalloc(&d_m, 64); hipMemcpy(d_m, m, mlen, hipMemcpyHostToDevice); hipMemset(d_d, 0, sizeof(Dump));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_d, d_m, mlen, pkx, 3u);
    Dump dd; hipError_t e = hipMemcpy(&dd, d_d, sizeof dd, hipMemcpyDeviceToHost);
    printf("hip: %s\n", hipGetErrorString(e));
    int bad = 0;
    CMP(sha_abc) CMP(sq) CMP(ms) CMP(c1) CMP(b0) CMP(uni) CMP(u0) CMP(u1) CMP(xn) CMP(xd) CMP(y) CMP(q0) CMP(q1) CMP(h)
    printf("h.x host: "); for (int i = 7; i >= 0; i--) printf("%08x", hd.h.x.v[i]); printf("\n");
    printf(bad ? "FAILED (%d fields differ)\n" : "ALL EQUAL\n", bad);
    return bad != 0;
}
