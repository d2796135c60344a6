// Device-vs-host differential test of the larger per-lane bodies: hash_to_curve, table_build + msm_run, normalize_points.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "plume_stages.h"
using namespace plume;
#define NOUT 12
__host__ __device__ inline void to_aff(uint32_t* ox, uint32_t* oy, const jac& p) {
    fe x = fe_zero(), y = fe_zero();
    if (!p.inf) { fe zi, zi2; fe_inv(zi, p.z); fe_sqr(zi2, zi); fe_mul(x, p.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(y, p.y, zi2); }
    fe_normalize(x); fe_normalize(y); fe_to_words(ox, x); fe_to_words(oy, y);
}
// scratch per item: bases 27 words, tab PLUME_TAB_WORDS, table scratch 8 entries, dig 66 bytes
__host__ __device__ inline void run_item(const uint32_t* in, uint32_t* out, uint32_t* bases, uint32_t* tab, uint32_t* scr, int8_t* dig) {
    // A: h2c(msg = 32 bytes of in, pk = G)
    uint8_t msg[32]; for (int i = 0; i < 32; i++) msg[i] = (uint8_t)(in[i >> 2] >> (8 * (i & 3)));
    jac h; hash_to_curve_jac(h, msg, 32, fe_gx(), 2u, PLUME_ENC_POINT);
    to_aff(out, out + 8, h);
    // B: k * H via table + msm
    h.inf = 0;
    st_base(bases, 0, h);
    uint8_t flag = 0;
    table_build(tab, bases, &flag, 1, 0, 1, scr, 1, 0);
    sc k; for (int i = 0; i < 8; i++) k.v[i] = in[8 + i]; k.v[7] &= 0x7FFFFFFFu;
    glv_half h1, h2; glv_split(h1, h2, k);
    booth_store(dig, 1, h1, false); booth_store(dig + PLUME_NDIG, 1, h2, false);
    jac acc; msm_run(acc, tab, nullptr, 2, dig, 1);
    to_aff(out + 16, out + 24, acc);
    {   // D: h2c internals
        fe u[2]; hash_to_field2(u[0], u[1], msg, 32, fe_gx(), 2u, PLUME_ENC_POINT);
        fe t = u[0]; fe_normalize(t); fe_to_words(out + 48, t);
        t = u[1]; fe_normalize(t); fe_to_words(out + 56, t);
        fe xn, xd, y; sswu_frac(xn, xd, y, u[0]);
        t = xn; fe_normalize(t); fe_to_words(out + 64, t);
        t = xd; fe_normalize(t); fe_to_words(out + 72, t);
        t = y; fe_normalize(t); fe_to_words(out + 80, t);
        jac q; iso3_frac_to_jac(q, xn, xd, y);
        uint32_t qy[8]; to_aff(out + 88, qy, q);
    }
    // C: table entry 7 (8H) x and beta*x
    fe e, ey; ld_tab_xy(e, ey, tab + 7 * PLUME_TAB_ENTRY_WORDS, false); fe_normalize(e); fe_to_words(out + 32, e);
    ld_tab_xy(e, ey, tab + 7 * PLUME_TAB_ENTRY_WORDS, true); fe_normalize(e); fe_to_words(out + 40, e);
}
__global__ void k(const uint32_t* in, uint32_t* out, uint32_t* bases, uint32_t* tab, uint32_t* scr, int8_t* dig, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) run_item(in + 16 * i, out + 8 * NOUT * i, bases + PLUME_JAC_WORDS * i, tab + (size_t)PLUME_TAB_WORDS * i, scr + (size_t)PLUME_TAB_ENTRIES * PLUME_TAB_SCR_WORDS * i, dig + 66 * i);
}
int main() {
    const int n = 512;
    std::vector<uint32_t> in(16 * n), hout(8 * NOUT * n), dout(8 * NOUT * n), hb(PLUME_JAC_WORDS * n), ht((size_t)PLUME_TAB_WORDS * n), hs((size_t)PLUME_TAB_ENTRIES * PLUME_TAB_SCR_WORDS * n);
    std::vector<int8_t> hd(66 * n);
    srand(11);
    for (auto& w : in) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    for (int i = 0; i < n; i++) run_item(in.data() + 16 * i, hout.data() + 8 * NOUT * i, hb.data() + PLUME_JAC_WORDS * i, ht.data() + (size_t)PLUME_TAB_WORDS * i, hs.data() + (size_t)PLUME_TAB_ENTRIES * PLUME_TAB_SCR_WORDS * i, hd.data() + 66 * i);
    uint32_t *din, *dd, *db, *dt, *ds; int8_t* ddig;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dd, dout.size() * 4); hipMalloc(&db, hb.size() * 4); hipMalloc(&dt, ht.size() * 4); hipMalloc(&ds, hs.size() * 4); hipMalloc(&ddig, hd.size());
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 63) / 64), dim3(64), 0, 0, din, dd, db, dt, ds, ddig, n);
    hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 2; }
    hipMemcpy(dout.data(), dd, dout.size() * 4, hipMemcpyDeviceToHost);
    const char* names[NOUT] = {"h2c x", "h2c y", "k*H x", "k*H y", "8H x", "beta*8H x", "u0", "u1", "sswu xn", "sswu xd", "sswu y", "iso q0.x"};
    int bad[NOUT] = {0};
    for (int i = 0; i < n; i++) for (int o = 0; o < NOUT; o++) if (memcmp(&hout[8 * (NOUT * i + o)], &dout[8 * (NOUT * i + o)], 32)) { if (!bad[o]) printf("first mismatch %s item %d\n", names[o], i); bad[o]++; }
    int tot = 0; for (int o = 0; o < NOUT; o++) { printf("%-12s mismatches %d / %d\n", names[o], bad[o], n); tot += bad[o]; }
    printf(tot ? "DIFFER\n" : "ALL EQUAL\n");
    return tot != 0;
}
