// secp256k1 base field, UNSATURATED representation for gfx950: 9 limbs of 29 bits in 32-bit VGPRs.
//
// Why: on CDNA4 every plain VALU op costs ~0.6 of a v_mad_u64_u32 (measured, tests/gpu_debug/microbench_rates.py),
// and the saturated 8x32 multiplication spends more issue time on carry chains, zero-extension moves and hazard
// nops (~156 plain ops) than on its 72 multiply-adds.  With 29-bit limbs a column of 9 products (< 2^60 each)
// accumulates in a 64-bit register pair by chained v_mad_u64_u32 with NO carries; carries are handled once per
// column, and additions / subtractions are 9 independent 32-bit ops with no carry chain at all.
//
// Representation: value = sum l[i] * 2^(29 i).  "Tight" (T): l[0..7] < 2^29 + 2^7, l[8] < 2^24 + 2^7 (what fe_mul /
// fe_sqr / fe_carry produce).  Lazy values carry an implicit magnitude bound; the rules used by the group law are
//     fe_mul / fe_sqr inputs : every limb <= 2^30.5-ish such that 9 * maxA * maxB < 2^64  (checked in host builds)
//     fe_add                 : limbwise, bounds add
//     fe_sub(a, b)           : a + (BIAS - b), BIAS = 8p arranged with every limb in [2^31, 2^32): needs b limbs <= 2^31
//                              result limbs < a + 2^32 would overflow, so fe_sub CARRIES its result (tight output)
// i.e. subtraction always returns a tight value, addition is lazy.  Host builds (tests/devsim) assert the bounds.
#pragma once
#include <stdint.h>

#include "plume_field.h"   // carry helpers, sc (scalar field stays saturated), opaque helpers

namespace plume {

#define PLUME_FE29_MASK 0x1FFFFFFFu

struct fe29 {
    uint32_t v[9];
};

#if !defined(__HIP_DEVICE_COMPILE__) && defined(PLUME_FE29_CHECK)
#include <assert.h>
#define PLUME_FE29_ASSERT(x) assert(x)
#else
#define PLUME_FE29_ASSERT(x) ((void)0)
#endif

// p = 2^256 - 2^32 - 977 in 29-bit limbs: 2^32 = 2^29 * 8
//   l0 = 2^29 - 977, l1 = 2^29 - 1 - 8 ... computed:  p = (2^256 - 1) - 2^32 - 976
PLUME_HD uint32_t fe29_p(int i) {
    return i == 0 ? 0x1FFFFC2Fu : i == 1 ? 0x1FFFFFF7u : i == 8 ? 0x00FFFFFFu : 0x1FFFFFFFu;
}

PLUME_HD fe29 fe29_zero() { fe29 r; PLUME_UNROLL for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
PLUME_HD fe29 fe29_small(uint32_t x) { fe29 r = fe29_zero(); r.v[0] = x & PLUME_FE29_MASK; r.v[1] = x >> 29; return r; }

// 8 x 32-bit little-endian words (value < 2^256) <-> 9 x 29
PLUME_HD void fe29_from_words(fe29& r, const uint32_t w[8]) {
    PLUME_UNROLL for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t lo = w[wi] >> sh;
        uint32_t hi = (sh > 3 && wi + 1 < 8) ? (w[wi + 1] << (32 - sh)) : 0u;
        r.v[i] = (lo | hi) & PLUME_FE29_MASK;
    }
}
// exact inverse for a TIGHT-canonical value (limbs < 2^29, top < 2^24)
PLUME_HD void fe29_to_words(uint32_t w[8], const fe29& a) {
    PLUME_UNROLL for (int k = 0; k < 8; k++) {
        // word k = bits [32k, 32k+32)
        const int bit = 32 * k, li = bit / 29, sh = bit - 29 * li;
        uint32_t x = a.v[li] >> sh;
        int have = 29 - sh;
        if (li + 1 < 9) x |= a.v[li + 1] << have;
        if (have + 29 < 32 && li + 2 < 9) x |= a.v[li + 2] << (have + 29);
        w[k] = x;
    }
}

// Weak carry pass: limbs -> tight.  Input limbs may be anything < 2^32; the part above bit 256 folds as *(2^32 + 977).
PLUME_HD void fe29_carry(fe29& a) {
    // top first: bits >= 256 live in l8 >> 24
    uint32_t t = a.v[8] >> 24;
    a.v[8] &= 0x00FFFFFFu;
    // t * (2^32 + 977): t*977 into l0 (t < 2^8 -> < 2^18), t*8 into l1
    uint32_t c[9];
    PLUME_UNROLL for (int i = 0; i < 8; i++) c[i] = a.v[i] >> 29;
    a.v[0] = (a.v[0] & PLUME_FE29_MASK) + t * 977u;
    a.v[1] = (a.v[1] & PLUME_FE29_MASK) + c[0] + (t << 3);
    PLUME_UNROLL for (int i = 2; i < 8; i++) a.v[i] = (a.v[i] & PLUME_FE29_MASK) + c[i - 1];
    a.v[8] = a.v[8] + c[7];
}

PLUME_HD void fe29_add(fe29& r, const fe29& a, const fe29& b) { PLUME_UNROLL for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; }
// BIAS = 4p, limbwise 4*p_i: every limb is ~2^31 (limb 8: ~2^26), so a + BIAS - b stays below 2^32 for a <= 2^30 and is
// non-negative for b <= 2^31 - 3908 (limb 8: b <= 2^26 - 4).
PLUME_HD uint32_t fe29_bias(int i) { return 4u * fe29_p(i); }
// r = a - b (mod p), b limbs must be <= min BIAS limb (>= 2^31); result is carried -> tight
PLUME_HD void fe29_sub(fe29& r, const fe29& a, const fe29& b) {
    PLUME_UNROLL for (int i = 0; i < 9; i++) {
        PLUME_FE29_ASSERT(b.v[i] <= fe29_bias(i));
        r.v[i] = a.v[i] + (fe29_bias(i) - b.v[i]);
    }
    fe29_carry(r);
}
PLUME_HD void fe29_neg(fe29& r, const fe29& a) {
    PLUME_UNROLL for (int i = 0; i < 9; i++) { PLUME_FE29_ASSERT(a.v[i] <= fe29_bias(i)); r.v[i] = fe29_bias(i) - a.v[i]; }
    fe29_carry(r);
}

// column sums -> tight result.  c[0..16] are the 17 product columns (each < 2^64 - slack).
PLUME_HD void fe29_reduce_cols(fe29& r, uint64_t c[17]) {
    // 1. carry-normalise the high columns 9..16 (plus the overflow limb h[8])
    uint32_t h[9];
    uint64_t carry = 0;
    PLUME_UNROLL for (int k = 9; k < 17; k++) {
        uint64_t t = c[k] + carry;
        h[k - 9] = (uint32_t)t & PLUME_FE29_MASK;
        carry = t >> 29;
    }
    h[8] = (uint32_t)carry;                         // < 2^(64-29) but in fact < 2^36; keep 32 bits: see bound note below
    PLUME_FE29_ASSERT((carry >> 32) == 0);
    // 2. fold: 2^261 = 2^37 + 31264 (mod p)  ->  column k-9 += h*31264, column k-8 += h << 8
    PLUME_UNROLL for (int k = 0; k < 9; k++) {
        c[k] += (uint64_t)h[k] * 31264u;
        if (k < 8) c[k + 1] += (uint64_t)h[k] << 8;
    }
    const uint64_t spill = (uint64_t)h[8] << 8;      // weight 2^261
    // 3. carry-normalise columns 0..8, then 9 (spill) folds once more
    carry = 0;
    uint32_t l[10];
    PLUME_UNROLL for (int k = 0; k < 9; k++) {
        uint64_t t = c[k] + carry;
        l[k] = (uint32_t)t & PLUME_FE29_MASK;
        carry = t >> 29;
    }
    uint64_t top = spill + carry;                    // weight 2^261
    // value = l[0..8] + top * 2^261;  also l[8] holds bits 232..260: bits >= 256 are l[8] >> 24
    uint64_t t256 = (top << 5) + (l[8] >> 24);       // everything at or above bit 256, as a multiple of 2^256
    l[8] &= 0x00FFFFFFu;
    PLUME_FE29_ASSERT(t256 < (1ull << 42));
    // t256 * (2^32 + 977):  *977 -> limb 0.. ; *2^32 = *8 * 2^29 -> limb 1..
    uint64_t x0 = (uint64_t)l[0] + t256 * 977u;
    uint64_t x1 = (uint64_t)l[1] + (t256 << 3) + (x0 >> 29);
    uint64_t x2 = (uint64_t)l[2] + (x1 >> 29);
    uint32_t x3 = l[3] + (uint32_t)(x2 >> 29);
    r.v[0] = (uint32_t)x0 & PLUME_FE29_MASK;
    r.v[1] = (uint32_t)x1 & PLUME_FE29_MASK;
    r.v[2] = (uint32_t)x2 & PLUME_FE29_MASK;
    r.v[3] = x3;                                      // < 2^29 + 2^(40+3-58)... tiny excess, stays tight
    PLUME_UNROLL for (int k = 4; k < 9; k++) r.v[k] = l[k];
}

PLUME_HD void fe29_mul(fe29& r, const fe29& a, const fe29& b) {
    uint64_t c[17];
    PLUME_UNROLL for (int k = 0; k < 17; k++) {
        uint64_t acc = 0;
        PLUME_UNROLL for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9) acc += (uint64_t)a.v[i] * b.v[j];
        }
        c[k] = acc;
    }
    fe29_reduce_cols(r, c);
}
PLUME_HD void fe29_sqr(fe29& r, const fe29& a) {
    uint64_t c[17];
    uint32_t d[9];
    PLUME_UNROLL for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;   // needs limbs < 2^31
    PLUME_UNROLL for (int k = 0; k < 17; k++) {
        uint64_t acc = 0;
        PLUME_UNROLL for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9 && i < j) acc += (uint64_t)a.v[i] * d[j];
            if (j == i) acc += (uint64_t)a.v[i] * a.v[i];
        }
        c[k] = acc;
    }
    fe29_reduce_cols(r, c);
}

// ---- v2: the carry of column k-1 is the 64-bit ADDEND of column k's first multiply-add, and the fold of the high half
// (2^261 = 2^37 + 31264 mod p) is two more multiply-adds per low column: no 64-bit shifts or adds remain, only
// alignbit / shift / mask on 32-bit halves.
PLUME_HD uint64_t shr29(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    return ((uint64_t)(opaque_u32(hi) >> 29) << 32) | __builtin_amdgcn_alignbit(hi, lo, 29);
#else
    return x >> 29;
#endif
}
// a * b + c as ONE v_mad_u64_u32 (the compiler otherwise strength-reduces small constants into 64-bit shift/add pairs)
#ifndef PLUME_F29_ASM_MAD
#define PLUME_F29_ASM_MAD 1
#endif
PLUME_HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && PLUME_F29_ASM_MAD
    uint64_t d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return (uint64_t)a * b + c;
#endif
}
PLUME_HD uint64_t mad64k(uint32_t a, uint32_t k, uint64_t c) {   // k: wave-uniform constant (SGPR)
#if defined(__HIP_DEVICE_COMPILE__) && PLUME_F29_ASM_MAD
    uint64_t d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(k), "v"(c));
    return d;
#elif defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)a * opaque_u32(k) + c;
#else
    return (uint64_t)a * k + c;
#endif
}
template <bool SQR>
PLUME_HD void fe29_mul2_impl(fe29& r, const fe29& a, const fe29& b) {
    uint32_t d[9];
    if (SQR) { PLUME_UNROLL for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1; }
    auto column = [&](uint64_t acc, int k) -> uint64_t {
        PLUME_UNROLL for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j >= 9) continue;
            if (!SQR) acc = mad64(a.v[i], b.v[j], acc);
            else if (i < j) acc = mad64(a.v[i], d[j], acc);
            else if (i == j) acc = mad64(a.v[i], a.v[i], acc);
        }
        return acc;
    };
    // high half: columns 9..16 -> h[0..7] (29 bits each) and h[8] (what is left)
    uint32_t h[9];
    uint64_t acc = 0;
    PLUME_UNROLL for (int k = 9; k < 17; k++) {
        acc = column(acc, k);
        h[k - 9] = (uint32_t)acc & PLUME_FE29_MASK;
        acc = shr29(acc);
    }
    h[8] = (uint32_t)acc;
    // low half with the fold: column k += h[k] * 31264 + h[k-1] * 2^8
    acc = 0;
    PLUME_UNROLL for (int k = 0; k < 9; k++) {
        acc = column(acc, k);
        acc = mad64k(h[k], 31264u, acc);
        if (k > 0) acc = mad64k(h[k - 1], 256u, acc);
        if (k == 8) acc = ((uint64_t)((uint32_t)(acc >> 32) + (h[8] << 5)) << 32) | (uint32_t)acc;          // h[8] * 2^8 belongs to column 9 = 2^29 * column 8 (h[8] < 2^21)
        if (k < 8) { r.v[k] = (uint32_t)acc & PLUME_FE29_MASK; acc = shr29(acc); }
    }
    // acc = column 8 (weight 2^232): bits >= 24 are multiples of 2^256 -> t = t0 + t1 * 2^29, times (2^32 + 977)
    const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    r.v[8] = lo & 0x00FFFFFFu;
    const uint32_t t0 = ((lo >> 24) | (hi << 8)) & PLUME_FE29_MASK;      // bits 24..52
    const uint32_t t1 = hi >> 21;                                       // bits 53..
    uint64_t x0 = mad64k(t0, 977u, r.v[0]);
    r.v[0] = (uint32_t)x0 & PLUME_FE29_MASK;
    uint64_t x1 = mad64k(t0, 8u, r.v[1] + (uint32_t)shr29(x0) + t1 * 977u);
    r.v[1] = (uint32_t)x1 & PLUME_FE29_MASK;
    r.v[2] += (uint32_t)shr29(x1) + (t1 << 3);
}
PLUME_HD void fe29_mul2(fe29& r, const fe29& a, const fe29& b) { fe29_mul2_impl<false>(r, a, b); }
PLUME_HD void fe29_sqr2(fe29& r, const fe29& a) { fe29_mul2_impl<true>(r, a, a); }


// ---- v3: plain C++ chains only (hipcc turns `acc = (u64)a*b + acc` into one v_mad_u64_u32 and `acc >>= 29` into one
// v_lshrrev_b64, with no hazard nops); the fold constants are made opaque so they stay multiply-adds.
// Column bookkeeping: h[k] (k = 0..8) has weight 2^(261 + 29k) = (2^37 + 31264) * 2^(29k):
//   column k   += h[k] * 31264,  column k+1 += h[k] * 2^8            for k = 0..7
//   h[8]: column 8 += h[8] * 31264; its 2^8 part lands on column 9 = 2^261 again: column 0 += h[8] * (31264 << 8), column 1 += h[8] << 16
PLUME_HD void pin64(uint64_t& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));        // no instruction: only stops the compiler from re-associating the accumulation chain across this point
#endif
}
template <bool SQR>
PLUME_HD void fe29_mul3_impl(fe29& r, const fe29& a, const fe29& b) {
    uint32_t d[9];
    if (SQR) { PLUME_UNROLL for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1; }
    auto column = [&](uint64_t acc, int k) -> uint64_t {
        PLUME_UNROLL for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j >= 9) continue;
            if (!SQR) { acc += (uint64_t)a.v[i] * b.v[j]; pin64(acc); }
            else if (i < j) { acc += (uint64_t)a.v[i] * d[j]; pin64(acc); }
            else if (i == j) { acc += (uint64_t)a.v[i] * a.v[i]; pin64(acc); }
        }
        return acc;
    };
    const uint32_t K0 = opaque_u32(31264u), K1 = opaque_u32(256u), K2 = opaque_u32(31264u << 8), K3 = opaque_u32(65536u), K4 = opaque_u32(977u), K5 = opaque_u32(8u);
    uint32_t h[9];
    uint64_t acc = 0;
    PLUME_UNROLL for (int k = 9; k < 17; k++) {
        acc = column(acc, k);
        h[k - 9] = (uint32_t)acc & PLUME_FE29_MASK;
        acc >>= 29;
    }
    h[8] = (uint32_t)acc;                                   // < 2^27 for top limbs < 2^27
    acc = 0;
    PLUME_UNROLL for (int k = 0; k < 9; k++) {
        acc = column(acc, k);
        acc += (uint64_t)h[k] * K0;
        pin64(acc);
        if (k > 0) { acc += (uint64_t)h[k - 1] * K1; pin64(acc); }
        if (k == 0) { acc += (uint64_t)h[8] * K2; pin64(acc); }
        if (k == 1) { acc += (uint64_t)h[8] * K3; pin64(acc); }
        if (k < 8) { r.v[k] = (uint32_t)acc & PLUME_FE29_MASK; acc >>= 29; }
    }
    // acc = column 8 (weight 2^232): bits >= 24 are multiples of 2^256 -> t = t0 + t1 * 2^29, times (2^32 + 977)
    r.v[8] = (uint32_t)acc & 0x00FFFFFFu;
    acc >>= 24;
    const uint32_t t0 = (uint32_t)acc & PLUME_FE29_MASK;
    const uint32_t t1 = (uint32_t)(acc >> 29);
    uint64_t x = (uint64_t)t0 * K4 + r.v[0];
    r.v[0] = (uint32_t)x & PLUME_FE29_MASK;
    x = (uint64_t)t0 * K5 + (r.v[1] + t1 * 977u) + (x >> 29);
    r.v[1] = (uint32_t)x & PLUME_FE29_MASK;
    r.v[2] += (uint32_t)(x >> 29) + (t1 << 3);
}
PLUME_HD void fe29_mul3(fe29& r, const fe29& a, const fe29& b) { fe29_mul3_impl<false>(r, a, b); }
PLUME_HD void fe29_sqr3(fe29& r, const fe29& a) { fe29_mul3_impl<true>(r, a, a); }

}  // namespace plume
