// Hazard probe: VALU writes VCC (carry-out) -> filler -> VALU reads VCC as carry-in.  Each variant computes
// r = carry_out(a + b) via v_addc 0+0+vcc after a chain that first sets vcc to the OPPOSITE value two writes earlier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define SEQ(NAME, FILLER)                                                                                     \
    __global__ void NAME(uint32_t* out, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* junk) { \
        int id = blockIdx.x * blockDim.x + threadIdx.x;                                                       \
        uint32_t x = a[id], y = b[id], z = c[id], r, t, u = id;                                               \
        uint32_t* jp = junk + id;                                                                             \
        asm volatile(                                                                                         \
            "v_add_co_u32 %1, vcc, %3, %4\n\t" /* vcc = carry(x+y) */                                         \
            "s_nop 4\n\t"                                                                                     \
            "v_addc_co_u32 %1, vcc, %5, %5, vcc\n\t" /* vcc = carry(z+z+c1): second write */                  \
            FILLER                                                                                            \
            "v_addc_co_u32 %0, vcc, 0, 0, vcc\n\t" /* r = vcc (must be carry of second write) */              \
            "s_nop 4\n\t"                                                                                     \
            : "=&v"(r), "=&v"(t), "+v"(u) : "v"(x), "v"(y), "v"(z), "v"(jp) : "vcc", "memory");               \
        out[id] = r + (u & 0);                                                                                \
    }
SEQ(k_nop1, "s_nop 1\n\t")
SEQ(k_nop0, "s_nop 0\n\t")
SEQ(k_none, "")
SEQ(k_mov_nop0, "v_mov_b32 %2, 0x3d1\n\ts_nop 0\n\t")
SEQ(k_add_nop0, "v_add_u32 %2, %2, %3\n\ts_nop 0\n\t")
SEQ(k_mov_mov, "v_mov_b32 %2, 0x3d1\n\tv_mov_b32 %2, 0x3d2\n\t")
SEQ(k_store_nop0, "global_store_dword %6, %3, off\n\ts_nop 0\n\t")
SEQ(k_mad_nop0, "v_mad_u64_u32 v[100:101], s[10:11], %3, %4, 0\n\ts_nop 0\n\t")
SEQ(k_nop3, "s_nop 3\n\t")
int main() {
    const int n = 1 << 16;
    uint32_t *ha = new uint32_t[n], *hb = new uint32_t[n], *hc = new uint32_t[n], *ho = new uint32_t[n];
    srand(3);
    for (int i = 0; i < n; i++) { ha[i] = (rand() & 1) ? 0xFFFFFFFFu : 1u; hb[i] = (rand() & 1) ? 0xFFFFFFFFu : 0u; hc[i] = (rand() & 1) ? 0x80000000u : 0x7FFFFFFFu; }
    uint32_t *a, *b, *c, *o, *j;
    hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n); hipMalloc(&c, 4 * n); hipMalloc(&o, 4 * n); hipMalloc(&j, 4 * n);
    hipMemcpy(a, ha, 4 * n, hipMemcpyHostToDevice); hipMemcpy(b, hb, 4 * n, hipMemcpyHostToDevice); hipMemcpy(c, hc, 4 * n, hipMemcpyHostToDevice);
    typedef void (*K)(uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t*);
    struct { const char* name; K k; } ks[] = {{"s_nop 1", k_nop1}, {"s_nop 0", k_nop0}, {"(none)", k_none}, {"v_mov; s_nop 0", k_mov_nop0}, {"v_add_u32; s_nop 0", k_add_nop0},
                                              {"v_mov; v_mov", k_mov_mov}, {"global_store; s_nop 0", k_store_nop0}, {"v_mad_u64_u32; s_nop 0", k_mad_nop0}, {"s_nop 3", k_nop3}};
    for (auto& e : ks) {
        hipMemset(o, 0xEE, 4 * n);
        hipLaunchKernelGGL(e.k, dim3(n / 256), dim3(256), 0, 0, o, a, b, c, j);
        hipMemcpy(ho, o, 4 * n, hipMemcpyDeviceToHost);
        int bad = 0, stale = 0;
        for (int i = 0; i < n; i++) {
            uint64_t s1 = (uint64_t)ha[i] + hb[i]; uint32_t c1 = (uint32_t)(s1 >> 32);
            uint64_t s2 = (uint64_t)hc[i] + hc[i] + c1; uint32_t c2 = (uint32_t)(s2 >> 32);
            if (ho[i] != c2) { bad++; if (ho[i] == c1) stale++; }
        }
        printf("%-28s wrong %6d / %d  (of which equal to the stale carry: %d)\n", e.name, bad, n, stale);
    }
    return 0;
}
