// Device-vs-host differential test of the field layer: the same header compiled for gfx950 (asm chains) and for the host (plain C++).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "plume_ec.h"
using namespace plume;
#define NOPS 22
// in: 2 x 8 words per item; out: NOPS x 8 words per item
__host__ __device__ inline void run_item(const uint32_t* in, uint32_t* out) {
    fe a, b, r;
    fe_from_words(a, in); fe_from_words(b, in + 8);
    int k = 0;
    auto put = [&](fe x) { fe_normalize(x); fe_to_words(out + 8 * k, x); k++; };
    fe_mul(r, a, b); put(r);
    fe_sqr(r, a); put(r);
    fe_mul_k(r, fe_beta(), a); put(r);
    fe_add(r, a, b); put(r);
    fe_sub(r, a, b); put(r);
    fe_neg(r, a); put(r);
    { fe s, d; fe_add_lazy(s, a, b); fe_sub_lazy<2>(d, a, b); fe_mul(r, d, s); put(r); }
    fe_mul_small(r, a, 1771); put(r);
    r = a; put(r);
    fe_inv_fermat(r, a); put(r);
    fe_inv_gcd(r, a); put(r);
    { fe t1, t2; fe_inv_gcd(t1, a); fe_mul(t2, t1, a); put(t2); }
    { uint32_t t[16]; for (int i = 0; i < 8; i++) { t[i] = in[i]; t[8 + i] = i < 4 ? in[8 + i] : 0; } fe_from_words16(r, t); put(r); }
    {   // the opening of sswu_frac, step by step
        const fe A = fe_set(0x3F8731ABu, 0xDD661ADCu, 0xA08A5558u, 0xF0F5D272u, 0xE953D363u, 0xCB6F0E5Du, 0x405447C0u, 0x1A444533u);
        fe tv1, tv2, tv3, tv4;
        fe_sqr(tv1, a); put(tv1);
        fe_mul_small(tv1, tv1, 11); put(tv1);
        fe_neg(tv1, tv1); put(tv1);
        fe_sqr(tv2, tv1); fe_add(tv2, tv2, tv1); put(tv2);
        fe one = fe_small(1); fe_add(tv3, tv2, one); fe_mul_small(tv3, tv3, 1771); put(tv3);
        if (fe_is_zero(tv2)) { tv4 = fe_small(11); fe_neg(tv4, tv4); } else { fe_neg(tv4, tv2); }
        put(tv4);
        fe_mul_k(tv4, A, tv4); put(tv4);
    }
    { fe t, n; fe_sub_lazy<2>(t, a, b); fe_neg_lazy(n, b); fe_muladd(r, a, t, n, b); put(r); }     // a(a-b) - b*b with one fold
    { jac p; p.x = fe_gx(); p.y = fe_gy(); p.z = a; p.inf = 0; fe z2, z3; fe_sqr(z2, a); fe_mul(z3, z2, a); fe_mul(p.x, p.x, z2); fe_mul(p.y, p.y, z3); jac_dbl(p); jac_madd(p, fe_gx(), fe_gy());
      fe zi, zi2; fe_inv(zi, p.z); fe_sqr(zi2, zi); fe_mul(r, p.x, zi2); put(r); }
}
__global__ void k(const uint32_t* in, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) run_item(in + 16 * i, out + 8 * NOPS * i);
}
int main() {
    const int n = 4096;
    std::vector<uint32_t> in(16 * n), hout(8 * NOPS * n), dout(8 * NOPS * n);
    srand(7);
    for (auto& w : in) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    for (int i = 0; i < 16; i++) { in[i] = 0xFFFFFFFFu; }                     // all-ones operands
    for (int i = 16; i < 32; i++) in[i] = 0;                                  // zeros
    for (int i = 0; i < n; i++) run_item(in.data() + 16 * i, hout.data() + 8 * NOPS * i);
    uint32_t *din, *dd; hipMalloc(&din, in.size() * 4); hipMalloc(&dd, dout.size() * 4);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 63) / 64), dim3(64), 0, 0, din, dd, n);
    hipMemcpy(dout.data(), dd, dout.size() * 4, hipMemcpyDeviceToHost);
    const char* names[NOPS] = {"mul", "sqr", "mul_k(beta)", "add", "sub", "neg", "lazy (a+b)(a-b)", "mul_small", "normalize", "inv fermat", "inv divsteps", "a * inv(a)", "from_words16", "s1 sqr", "s2 *11", "s3 neg", "s4 tv2", "s5 tv3", "s6 tv4 sel", "s7 A*tv4", "muladd", "dbl+madd x"};
    int bad[NOPS] = {0};
    for (int i = 0; i < n; i++) for (int o = 0; o < NOPS; o++) if (memcmp(&hout[8 * (NOPS * i + o)], &dout[8 * (NOPS * i + o)], 32)) { if (!bad[o]) printf("first mismatch op %s item %d\n", names[o], i); bad[o]++; }
    int tot = 0; for (int o = 0; o < NOPS; o++) { printf("%-18s mismatches %d / %d\n", names[o], bad[o], n); tot += bad[o]; }
    printf(tot ? "DIFFER\n" : "ALL EQUAL\n");
    return tot != 0;
}
