// secp256k1 base field Fp (p = 2^256 - 2^32 - 977) and scalar field Fn arithmetic for the PLUME hot path.
// Written for gfx950 (CDNA4): Fp elements are 9 x 29-bit limbs in VGPRs with products and their column sums through chains
// of v_mad_u64_u32 (32x32+64 -> 64) and NO carry chains; Fn elements (a few operations per item) are 8 x 32-bit limbs with
// v_add_co/v_addc_co chains (__builtin_addc).  No MFMA: this is integer VALU.
//
// The same header compiles as plain C++ for the host (tests/devsim) so that the exact device arithmetic is
// unit-tested on the CPU against the oracle; that build is test infrastructure and is never linked into the
// product library.
//
// Reference anchors: p rust-arkworks/src/secp256k1/fields/fq.rs:12, n fields/fr.rs:19 (the reference gets its
// arithmetic from the un-vendored k256 ~0.13.3 crate, rust-k256/Cargo.toml:18).
#pragma once
#include <stdint.h>
#if !defined(__HIP_DEVICE_COMPILE__) && defined(PLUME_FE_CHECK)
#include <assert.h>
#endif

#if defined(__HIPCC__) || defined(__HIP__)
#define PLUME_HD __host__ __device__ __forceinline__
#define PLUME_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))
#else
#define PLUME_HD inline
#define PLUME_HD_NOINLINE inline __attribute__((noinline))
#endif

#if defined(__clang__)
#define PLUME_UNROLL _Pragma("unroll")
#define PLUME_NOUNROLL _Pragma("unroll 1")
#else
#define PLUME_UNROLL
#define PLUME_NOUNROLL
#endif

namespace plume {

// ---------------------------------------------------------------------------------------------- carry helpers
PLUME_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t& c) {
#if defined(__clang__)
    unsigned co;
    uint32_t r = __builtin_addc(a, b, c, &co);
    c = co;
    return r;
#else
    uint64_t t = (uint64_t)a + b + c;
    c = (uint32_t)(t >> 32);
    return (uint32_t)t;
#endif
}
PLUME_HD uint32_t subb(uint32_t a, uint32_t b, uint32_t& bw) {
#if defined(__clang__)
    unsigned bo;
    uint32_t r = __builtin_subc(a, b, bw, &bo);
    bw = bo;
    return r;
#else
    uint64_t t = (uint64_t)a - b - bw;
    bw = (uint32_t)(t >> 63);
    return (uint32_t)t;
#endif
}
PLUME_HD uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// A zero the optimiser cannot see through.  ROCm 7.2's AMDGPU backend rewrites
//     uaddo_carry(add(x, y), 0, cin)  ->  uaddo_carry(x, y, cin)        (and the usubo_carry/sub twin)
// even when the carry-OUT is used; the merged instruction then also carries when the plain 32-bit add x + y
// wraps (observed on gfx950: SHA-256's final `state += a` folded into the first limb of the next carry chain,
// one spurious +1 in the following limb).  Every "propagate the carry through this limb" step therefore adds
// this opaque zero instead of the literal 0, which keeps the pattern from matching.
PLUME_HD uint32_t opaque_zero() {
    uint32_t z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+s"(z));
#endif
    return z;
}
// Values produced by WRAPPING 32-bit arithmetic (SHA-256 state words) must pass through this before they enter a
// carry chain, for the same reason: it hides the producing `add` from the combiner.
PLUME_HD uint32_t opaque_u32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));
#endif
    return x;
}
// Selects.  hipcc compiles `flag ? a : b` on 32-bit values to v_cndmask_b32 with the condition in VCC, and a RUN of those -- a multi-limb select, a select chain -- is the
// slowest thing the SIMD does: tests/gpu_debug/instr_rates_r03.txt, one compare + 7 v_cndmask on its VCC = 6.7 ns per instruction at ANY occupancy (a multiply-add: 1.8).
// A select through an opaque all-ones / all-zeros mask compiles to one v_bfi_b32 (1.8 ns) instead; the mask costs two instructions per condition.
PLUME_HD uint32_t sel_mask(bool flag) {
    uint32_t m = 0u - (uint32_t)flag;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(m));              // or the compiler turns the mask arithmetic back into a select
#endif
    return m;
}
PLUME_HD uint32_t sel32(uint32_t mask, uint32_t a, uint32_t b) {      // mask all ones: a, all zeros: b
#if defined(__HIP_DEVICE_COMPILE__)
    return (a & mask) | (b & ~mask);
#else
    return mask ? a : b;
#endif
}
PLUME_HD uint32_t addc0(uint32_t a, uint32_t& c) { return addc(a, opaque_zero(), c); }
PLUME_HD uint32_t subb0(uint32_t a, uint32_t& bw) { return subb(a, opaque_zero(), bw); }

// ------------------------------------------------------------------------------------------------------- Fp
// Representation: 9 limbs of 29 bits in 32-bit VGPRs, value = sum v[i] * 2^(29 i) (mod p), NOT necessarily reduced.
// Why not 8 x 32: on gfx950 a carry-chain step (v_add_co / v_addc_co) costs as much issue time as a 32x32+64
// multiply-add, and the saturated product is 73 multiply-adds + ~105 carry steps.  With 29-bit limbs a product column
// accumulates inside the 64-bit addend of chained v_mad_u64_u32, and additions are 9 independent plain adds
// (tests/gpu_debug/instr_rates_r01.txt, gen_fe_mul.py).
//
// Limb bounds ("magnitudes") are the caller's contract, checked by assertions in host builds with PLUME_FE_CHECK:
//   tight      limbs 0..7 <= 2^29 + 2^19, limb 8 <= 2^24 + 2^10        what fe_mul / fe_sqr / fe_carry / fe_add / fe_sub return
//   fe_mul, fe_sqr inputs: 9 * max(a[0..7]) * max(b[0..7]) < 2^64 - 2^50 AND limb 8 <= 2^26 on both sides.  In terms of lazy sums: both operands sums of two tight
//              values (limbs 0..7 up to ~1.3 * 2^30), or one a sum of THREE tight values (limbs 0..7 < 2^31, limb 8 <= 3 * (2^24 + 2^10) < 2^26) and the other tight.
//              A sum of FOUR tight values breaks the limb-8 bound (4 * (2^24 + 2^10) > 2^26) although its lower limbs would still pass: carry it first.  (Limb 8 is
//              tighter than the others since round 4: the product's top column adds h[8] << 8 on a 32-bit high word, gen_fe_mul.py.)  Host builds assert both bounds on
//              every product (PLUME_FE_CHECK); tests/test_devsim.py::test_group_law_compositions_keep_every_product_operand_in_bounds drives the group law's
//              compositions to the ends of the tight range.
//   fe_add_lazy / fe_sub_lazy<M> return unreduced sums: limb bounds add (a + M*p - b), no carry pass
//   fe_normalize gives the canonical representative (limbs < 2^29, value < p) where one is needed (comparisons, parity,
//              serialisation).
struct fe {
    uint32_t v[9];
};
#define PLUME_FE_MASK 0x1FFFFFFFu
#define PLUME_FE_WORDS 9          // 32-bit words per field element in HBM scratch (limb form)
#define PLUME_PC977 977u

#if !defined(__HIP_DEVICE_COMPILE__) && defined(PLUME_FE_CHECK)
#define PLUME_FE_ASSERT(x) assert(x)
#else
#define PLUME_FE_ASSERT(x) ((void)0)
#endif

// p = 2^256 - 2^32 - 977 as limbs: (2^29 - 977), (2^29 - 1 - 8), 2^29 - 1 (x6), 2^24 - 1
PLUME_HD constexpr uint32_t fe_p(int i) { return i == 0 ? 0x1FFFFC2Fu : i == 1 ? 0x1FFFFFF7u : i == 8 ? 0x00FFFFFFu : 0x1FFFFFFFu; }

PLUME_HD fe fe_zero() { fe r; PLUME_UNROLL for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
PLUME_HD fe fe_small(uint32_t x) { fe r = fe_zero(); r.v[0] = x & PLUME_FE_MASK; r.v[1] = x >> 29; return r; }
PLUME_HD bool fe_is_tight(const fe& a) {
    bool ok = a.v[8] <= (1u << 24) + (1u << 10);
    PLUME_UNROLL for (int i = 0; i < 8; i++) ok = ok && a.v[i] <= (1u << 29) + (1u << 19);
    return ok;
}
PLUME_HD bool fe_mul_inputs_ok(const fe& a, const fe& b) {
    uint64_t ma = 0, mb = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) { if (a.v[i] > ma) ma = a.v[i]; if (b.v[i] > mb) mb = b.v[i]; }
    if (a.v[8] > (1u << 26) || b.v[8] > (1u << 26) || ma >= (1ull << 31) || mb >= (1ull << 31)) return false;   // limb 8: the product's top column takes h[8] << 8 on its high word (gen_fe_mul.py)
    const uint64_t prod = ma * mb;                                     // < 2^62
    return prod < ((0xFFFFFFFFFFFFFFFFull - (1ull << 50)) / 9);
}

PLUME_HD bool fe_muladd_inputs_ok(const fe& a, const fe& b, const fe& c, const fe& e) {   // a*b + c*e through shared column sums
    uint64_t ma = 0, mb = 0, mc = 0, me = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) {
        if (a.v[i] > ma) ma = a.v[i];
        if (b.v[i] > mb) mb = b.v[i];
        if (c.v[i] > mc) mc = c.v[i];
        if (e.v[i] > me) me = e.v[i];
    }
    if (a.v[8] > (1u << 26) || b.v[8] > (1u << 26) || c.v[8] > (1u << 26) || e.v[8] > (1u << 26)) return false;
    const unsigned __int128 sum = (unsigned __int128)ma * mb + (unsigned __int128)mc * me;
    return sum < (unsigned __int128)((0xFFFFFFFFFFFFFFFFull - (1ull << 50)) / 9);
}

// 8 x 32-bit little-endian words (any 256-bit integer) <-> limbs
PLUME_HD void fe_from_words(fe& r, const uint32_t w[8]) {
    PLUME_UNROLL for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t x = w[wi] >> sh;
        if (sh > 3 && wi + 1 < 8) x |= w[wi + 1] << (32 - sh);
        r.v[i] = x & PLUME_FE_MASK;
    }
}
// exact inverse for a CANONICAL value (limbs < 2^29, top < 2^24)
PLUME_HD void fe_to_words(uint32_t w[8], const fe& a) {
    PLUME_UNROLL for (int k = 0; k < 8; k++) {
        const int bit = 32 * k, li = bit / 29, sh = bit - 29 * li;   // word k = bits [32k, 32k+32)
        uint32_t x = a.v[li] >> sh;
        const int have = 29 - sh;
        if (li + 1 < 9) x |= a.v[li + 1] << have;
        if (have + 29 < 32 && li + 2 < 9) x |= a.v[li + 2] << (have + 29);
        w[k] = x;
    }
}
PLUME_HD fe fe_set(uint32_t w7, uint32_t w6, uint32_t w5, uint32_t w4, uint32_t w3, uint32_t w2, uint32_t w1, uint32_t w0) {
    const uint32_t w[8] = {w0, w1, w2, w3, w4, w5, w6, w7};   // arguments in big-endian word order, as hex reads
    fe r; fe_from_words(r, w); return r;
}

// One parallel carry pass: any limbs < 2^32 -> tight.  The part above bit 256 folds as *(2^32 + 977) into limbs 0 and 1.
PLUME_HD void fe_carry(fe& a) {
    const uint32_t t = a.v[8] >> 24;
    uint32_t c[8];
    PLUME_UNROLL for (int i = 0; i < 8; i++) c[i] = a.v[i] >> 29;
    a.v[0] = (a.v[0] & PLUME_FE_MASK) + t * 977u;
    a.v[1] = (a.v[1] & PLUME_FE_MASK) + c[0] + (t << 3);
    PLUME_UNROLL for (int i = 2; i < 8; i++) a.v[i] = (a.v[i] & PLUME_FE_MASK) + c[i - 1];
    a.v[8] = (a.v[8] & 0x00FFFFFFu) + c[7];
}
PLUME_HD void fe_add_lazy(fe& r, const fe& a, const fe& b) {
    PLUME_UNROLL for (int i = 0; i < 9; i++) { PLUME_FE_ASSERT((uint64_t)a.v[i] + b.v[i] < (1ull << 32)); r.v[i] = a.v[i] + b.v[i]; }
}
// r = a + M*p - b limbwise: needs b[i] <= M*p[i] and a[i] + M*p[i] < 2^32.  M = 2 covers a tight b, M = 4 a sum of two.
template <int M>
PLUME_HD void fe_sub_lazy(fe& r, const fe& a, const fe& b) {
    PLUME_UNROLL for (int i = 0; i < 9; i++) {
        PLUME_FE_ASSERT(b.v[i] <= (uint32_t)M * fe_p(i) && (uint64_t)a.v[i] + (uint32_t)M * fe_p(i) < (1ull << 32));
        r.v[i] = a.v[i] + ((uint32_t)M * fe_p(i) - b.v[i]);
    }
}
// safe defaults: any operands with limbs <= 2^31 - 2^13 (sums of up to three tight values), tight result
PLUME_HD void fe_add(fe& r, const fe& a, const fe& b) { fe_add_lazy(r, a, b); fe_carry(r); }
PLUME_HD void fe_sub(fe& r, const fe& a, const fe& b) { fe_sub_lazy<4>(r, a, b); fe_carry(r); }
PLUME_HD void fe_neg(fe& r, const fe& a) { fe z = fe_zero(); fe_sub_lazy<4>(r, z, a); fe_carry(r); }
PLUME_HD void fe_dbl(fe& r, const fe& a) { fe_add_lazy(r, a, a); fe_carry(r); }

// canonical representative in [0, p): limbs < 2^29, top limb < 2^24
PLUME_HD void fe_ripple(fe& a) {   // sequential carry: limbs 0..7 < 2^29 afterwards, the excess collects in limb 8
    uint32_t c = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) { a.v[i] += c; c = a.v[i] >> 29; a.v[i] &= PLUME_FE_MASK; }
    a.v[8] += c;
}
PLUME_HD void fe_normalize(fe& a) {
    fe_carry(a);
    fe_ripple(a);
    const uint32_t t = a.v[8] >> 24;               // 0 or 1
    a.v[8] &= 0x00FFFFFFu;
    a.v[0] += t * 977u; a.v[1] += t << 3;
    fe_ripple(a);                                   // value < 2^256 now (if t was 1 the low part was tiny)
    // a >= p  <=>  a + (2^32 + 977) reaches bit 256; the sum with that bit cleared is a - p
    fe u = a;
    u.v[0] += 977u; u.v[1] += 8u;
    fe_ripple(u);
    const bool ge = (u.v[8] >> 24) != 0;
    u.v[8] &= 0x00FFFFFFu;
    const uint32_t gm = sel_mask(ge);
    PLUME_UNROLL for (int i = 0; i < 9; i++) a.v[i] = sel32(gm, u.v[i], a.v[i]);
}
PLUME_HD bool fe_is_zero(const fe& a) {  // a == 0 (mod p), any limbs < 2^32
    fe t = a;
    fe_carry(t);
    fe_ripple(t);                                   // unique limbs 0..7; value < 2^256 + 2^236 < 2p: zero is 0 or p
    uint32_t z = 0, pp = t.v[8] ^ fe_p(8);
    PLUME_UNROLL for (int i = 0; i < 9; i++) z |= t.v[i];
    PLUME_UNROLL for (int i = 0; i < 8; i++) pp |= t.v[i] ^ fe_p(i);
    return z == 0 || pp == 0;
}
PLUME_HD bool fe_eq(const fe& a, const fe& b) { fe d; fe_sub(d, a, b); return fe_is_zero(d); }
PLUME_HD bool fe_is_odd(const fe& a) { fe t = a; fe_normalize(t); return t.v[0] & 1; }
PLUME_HD void fe_cmov(fe& r, const fe& a, bool flag) { const uint32_t m = sel_mask(flag); PLUME_UNROLL for (int i = 0; i < 9; i++) r.v[i] = sel32(m, a.v[i], r.v[i]); }
// true iff the 256-bit integer in 8 little-endian words is < p (canonical encoding check for caller-supplied coordinates)
PLUME_HD bool words_lt_p(const uint32_t w[8]) {
    uint32_t c = 0;
    (void)addc(w[0], PLUME_PC977, c);
    (void)addc(w[1], 1u, c);
    PLUME_UNROLL for (int i = 2; i < 8; i++) (void)addc0(w[i], c);
    return c == 0;
}

// r = 2a limbwise, no carry pass.  hipcc turns a + a into v_lshlrev_b32, which the issue-rate probe puts at the multiply-add's cost on gfx950.  Naming v_add_u32 instead
// (plain-rate by the same probe) bought nothing either time it was tried: for every doubling (round 2: -0.9 %, the asm statement blocks the compiler's v_lshl_add_u32 /
// v_add3_u32 fusions) and for the squarings' doubled limbs only (round 4: multi-scalar kernel 15.505 vs 15.504 ms on one box).  LABNOTES.md.
PLUME_HD uint32_t u32_dbl(uint32_t x) { return x + x; }
PLUME_HD void fe_dbl_lazy(fe& r, const fe& a) {
    PLUME_UNROLL for (int i = 0; i < 9; i++) { PLUME_FE_ASSERT((uint64_t)a.v[i] * 2 < (1ull << 32)); r.v[i] = u32_dbl(a.v[i]); }
}

#include "plume_fe_mul.inc"

// r = a * k for a small k (< 2^20), any a with limbs < 2^32; tight result
PLUME_HD void fe_mul_small(fe& r, const fe& a, uint32_t k) {
    // k is hidden from the optimiser: with a literal 11 hipcc (ROCm 7.2) strength-reduces the 64-bit products into shift/add
    // chains and then merges their carries wrongly (the uaddo_carry combine described at opaque_zero(); caught by
    // tests/gpu_debug/fe_diff.hip) -- an opaque k keeps each product one v_mad_u64_u32
    k = opaque_u32(k);
    uint32_t l[9];
    uint64_t acc = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) { acc += (uint64_t)a.v[i] * k; l[i] = (uint32_t)acc & PLUME_FE_MASK; acc >>= 29; }
    acc += (uint64_t)a.v[8] * k;
    l[8] = (uint32_t)acc & 0x00FFFFFFu;
    const uint64_t t = acc >> 24;                  // < 2^29: multiples of 2^256 -> t * (2^32 + 977)
    acc = l[0] + t * 977u;
    l[0] = (uint32_t)acc & PLUME_FE_MASK; acc >>= 29;
    acc += l[1] + (t << 3);
    l[1] = (uint32_t)acc & PLUME_FE_MASK; acc >>= 29;
    l[2] += (uint32_t)acc;
    PLUME_UNROLL for (int i = 0; i < 9; i++) r.v[i] = l[i];
}
// lo + hi * 2^256 for two 256-bit integers given as words (hash_to_field's 48-byte OS2IP): hi * (2^32 + 977) + lo
PLUME_HD void fe_from_words16(fe& r, const uint32_t t[16]) {
    fe lo, hi, pc = fe_zero();
    fe_from_words(lo, t);
    fe_from_words(hi, t + 8);
    pc.v[0] = 977u; pc.v[1] = 8u;                  // 2^32 + 977
    fe_mul_k(hi, pc, hi);
    fe_add(r, lo, hi);
}
PLUME_HD void fe_sqr_n(fe& r, const fe& a, int n) {
    r = a;
    PLUME_NOUNROLL for (int i = 0; i < n; i++) fe_sqr(r, r);
}
// shared prefix of the two exponentiations: x2 = a^(2^2-1), x22 = a^(2^22-1), t = a^((2^223-1)*2^23 + 2^22-1)
PLUME_HD void fe_pow_prefix(fe& t, fe& x2, const fe& a) {
    fe x3, x6, x9, x11, x22, x44, x88, x176, x220, x223;
    fe_sqr(x2, a); fe_mul(x2, x2, a);
    fe_sqr(x3, x2); fe_mul(x3, x3, a);
    fe_sqr_n(x6, x3, 3); fe_mul(x6, x6, x3);
    fe_sqr_n(x9, x6, 3); fe_mul(x9, x9, x3);
    fe_sqr_n(x11, x9, 2); fe_mul(x11, x11, x2);
    fe_sqr_n(x22, x11, 11); fe_mul(x22, x22, x11);
    fe_sqr_n(x44, x22, 22); fe_mul(x44, x44, x22);
    fe_sqr_n(x88, x44, 44); fe_mul(x88, x88, x44);
    fe_sqr_n(x176, x88, 88); fe_mul(x176, x176, x88);
    fe_sqr_n(x220, x176, 44); fe_mul(x220, x220, x44);
    fe_sqr_n(x223, x220, 3); fe_mul(x223, x223, x3);
    fe_sqr_n(t, x223, 23); fe_mul(t, t, x22);
}
// a^(p-2): 255 squarings + 15 multiplications (kept as the cross-check of fe_inv below)
PLUME_HD void fe_inv_fermat(fe& r, const fe& a) {
    fe t, x2;
    fe_pow_prefix(t, x2, a);
    fe_sqr_n(t, t, 5); fe_mul(t, t, a);
    fe_sqr_n(t, t, 3); fe_mul(t, t, x2);
    fe_sqr_n(t, t, 2); fe_mul(r, t, a);
}
// ---- inversion by the Bernstein-Yang "safegcd" divsteps (constant iteration count: 20 batches of 30 divsteps cover 256-bit inputs,
// every lane of a wavefront runs the identical instruction stream).  About 3.5x cheaper than the 255-squaring Fermat chain above on
// gfx950 (measured through the table and affine-conversion kernels).  Internal form: 9 SIGNED limbs of 30 bits.
// The 2x2 transition matrix of a batch is scaled by 2^30; f, g shrink by exact division, d, e are kept mod p with the help of
// p^-1 mod 2^30.  (The algorithm is the published one -- Bernstein & Yang 2019, section 11; Wuille's "safegcd" notes -- restated
// here for this limb shape; checked against the Fermat chain and against Python's pow(x, -1, p) in tests/test_devsim.py.)
struct s30 {
    int32_t v[9];
};
#define PLUME_M30 0x3FFFFFFF
// p = -977 - 4*2^30 + 65536*2^240
PLUME_HD constexpr int32_t s30_p(int i) { return i == 0 ? -0x3D1 : i == 1 ? -4 : i == 8 ? 65536 : 0; }
#define PLUME_P_INV30 0x2DDACACFu     // p^-1 mod 2^30
struct trans30 {
    int32_t u, v, q, r;
};
// acc + a * b, signed 32 x 32 + 64 -> 64.  hipcc expands the C expression (sign extensions, unsigned multiply-add, two v_mul_lo corrections);
// the instruction exists (v_mad_i64_i32), so the matrix updates below ask for it by name: -37 % instructions per update.
PLUME_HD int64_t mad_i64(int64_t acc, int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t r; uint64_t cy_;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(r), "=&s"(cy_) : "v"(a), "v"(b), "v"(acc));
    return r;
#else
    return acc + (int64_t)a * b;
#endif
}
// 30 divsteps on the low limbs; zeta = -(delta + 1/2)
PLUME_HD int32_t divsteps30(int32_t zeta, uint32_t f0, uint32_t g0, trans30& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    PLUME_UNROLL for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(zeta >> 31);            // all ones if zeta < 0
        const uint32_t c2 = 0u - (g & 1u);               // all ones if g is odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // conditionally negated f, u, v
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;                                        // swap-and-negate case: zeta < 0 and g odd
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1;
        u += u; v += v;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
// (f, g) <- t * (f, g) / 2^30   (exact)
PLUME_HD void update_fg30(s30& f, s30& g, const trans30& t) {
    int64_t cf = mad_i64(mad_i64(0, t.u, f.v[0]), t.v, g.v[0]);
    int64_t cg = mad_i64(mad_i64(0, t.q, f.v[0]), t.r, g.v[0]);
    cf >>= 30; cg >>= 30;
    PLUME_UNROLL for (int i = 1; i < 9; i++) {
        const int32_t fi = f.v[i], gi = g.v[i];
        cf = mad_i64(mad_i64(cf, t.u, fi), t.v, gi);
        cg = mad_i64(mad_i64(cg, t.q, fi), t.r, gi);
        f.v[i - 1] = (int32_t)cf & PLUME_M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & PLUME_M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
}
// (d, e) <- t * (d, e) / 2^30  (mod p), both kept in (-2p, p)
PLUME_HD void update_de30(s30& d, s30& e, const trans30& t) {
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t cd = mad_i64(mad_i64(0, t.u, d.v[0]), t.v, e.v[0]);
    int64_t ce = mad_i64(mad_i64(0, t.q, d.v[0]), t.r, e.v[0]);
    md -= (int32_t)((PLUME_P_INV30 * (uint32_t)cd + (uint32_t)md) & PLUME_M30);
    me -= (int32_t)((PLUME_P_INV30 * (uint32_t)ce + (uint32_t)me) & PLUME_M30);
    cd += (int64_t)s30_p(0) * md; ce += (int64_t)s30_p(0) * me;
    cd >>= 30; ce >>= 30;
    PLUME_UNROLL for (int i = 1; i < 9; i++) {
        const int32_t di = d.v[i], ei = e.v[i];
        cd = mad_i64(mad_i64(cd, t.u, di), t.v, ei);
        ce = mad_i64(mad_i64(ce, t.q, di), t.r, ei);
        if (s30_p(i) != 0) { cd += (int64_t)s30_p(i) * md; ce += (int64_t)s30_p(i) * me; }
        d.v[i - 1] = (int32_t)cd & PLUME_M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & PLUME_M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
}
// r in (-2p, p) -> [0, p), negated first when sign < 0
PLUME_HD void normalize30(s30& r, int32_t sign) {
    int32_t cond_add = r.v[8] >> 31;
    const int32_t cond_neg = sign >> 31;
    PLUME_UNROLL for (int i = 0; i < 9; i++) { r.v[i] += s30_p(i) & cond_add; r.v[i] = (r.v[i] ^ cond_neg) - cond_neg; }
    PLUME_UNROLL for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= PLUME_M30; }
    cond_add = r.v[8] >> 31;
    PLUME_UNROLL for (int i = 0; i < 9; i++) r.v[i] += s30_p(i) & cond_add;
    PLUME_UNROLL for (int i = 0; i < 8; i++) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= PLUME_M30; }
}
// r = a^-1 mod p (0 for a = 0 mod p); a: any limbs < 2^32; r canonical
PLUME_HD void fe_inv_gcd(fe& r, const fe& a) {
    fe x = a;
    fe_normalize(x);
    uint32_t w[8];
    fe_to_words(w, x);
    s30 d, e, f, g;
    PLUME_UNROLL for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
        uint32_t lo = w[wi] >> sh;
        if (sh > 2 && wi + 1 < 8) lo |= w[wi + 1] << (32 - sh);
        g.v[i] = (int32_t)(lo & PLUME_M30);
        f.v[i] = s30_p(i); d.v[i] = 0; e.v[i] = i == 0 ? 1 : 0;
    }
    int32_t zeta = -1;
    PLUME_NOUNROLL for (int it = 0; it < 20; it++) {
        trans30 t;
        zeta = divsteps30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        update_de30(d, e, t);
        update_fg30(f, g, t);
    }
    normalize30(d, f.v[8]);                          // f = +-1 now (or +-p... for a = 0: d = 0)
    // 9 x 30 non-negative limbs -> 8 words -> 9 x 29
    PLUME_UNROLL for (int k = 0; k < 8; k++) {
        const int bit = 32 * k, li = bit / 30, sh = bit - 30 * li;
        uint32_t v = (uint32_t)d.v[li] >> sh;
        const int have = 30 - sh;
        if (li + 1 < 9) v |= (uint32_t)d.v[li + 1] << have;
        w[k] = v;
    }
    fe_from_words(r, w);
}
PLUME_HD void fe_inv(fe& r, const fe& a) { fe_inv_gcd(r, a); }

// a^((p-3)/4)  (RFC 9380 F.2.1.2 constant c1): 253 squarings + 14 multiplications
PLUME_HD void fe_pow_c1(fe& r, const fe& a) {
    fe t, x2;
    fe_pow_prefix(t, x2, a);
    fe_sqr_n(t, t, 5); fe_mul(t, t, a);
    fe_sqr_n(t, t, 3); fe_mul(r, t, x2);
}

// a^((p+1)/4): the square root of a when a is a quadratic residue (p = 3 mod 4); 253 squarings + 13 multiplications
PLUME_HD void fe_sqrt_candidate(fe& r, const fe& a) {
    fe t, x2;
    fe_pow_prefix(t, x2, a);
    fe_sqr_n(t, t, 6); fe_mul(t, t, x2);
    fe_sqr_n(r, t, 2);
}

// big-endian 32 bytes <-> little-endian words.  The pointers may be unaligned (caller arrays are byte arrays).
PLUME_HD void words_from_be(uint32_t w[8], const uint8_t* b) {
    PLUME_UNROLL for (int i = 0; i < 8; i++) {
        const uint8_t* q = b + 4 * (7 - i);
        w[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
}
PLUME_HD void words_to_be(uint8_t* b, const uint32_t w[8]) {
    PLUME_UNROLL for (int i = 0; i < 8; i++) {
        uint8_t* q = b + 4 * (7 - i);
        q[0] = (uint8_t)(w[i] >> 24); q[1] = (uint8_t)(w[i] >> 16); q[2] = (uint8_t)(w[i] >> 8); q[3] = (uint8_t)w[i];
    }
}
// 16-byte-aligned variants (caller arrays of 32/64-byte records in HBM are at least 16-byte aligned)
PLUME_HD void words_from_be_aligned(uint32_t w[8], const uint8_t* b) {
    const uint32_t* q = (const uint32_t*)b;
    PLUME_UNROLL for (int i = 0; i < 8; i++) w[i] = bswap32(q[7 - i]);
}
PLUME_HD void words_to_be_aligned(uint8_t* b, const uint32_t w[8]) {
    uint32_t* q = (uint32_t*)b;
    PLUME_UNROLL for (int i = 0; i < 8; i++) q[7 - i] = bswap32(w[i]);
}
PLUME_HD void fe_from_be(fe& r, const uint8_t* b) { uint32_t w[8]; words_from_be(w, b); fe_from_words(r, w); }
PLUME_HD void fe_from_be_aligned(fe& r, const uint8_t* b) { uint32_t w[8]; words_from_be_aligned(w, b); fe_from_words(r, w); }
PLUME_HD void fe_to_be(uint8_t* b, const fe& a) { uint32_t w[8]; fe_to_words(w, a); words_to_be(b, w); }                   // a must be canonical
PLUME_HD void fe_to_be_aligned(uint8_t* b, const fe& a) { uint32_t w[8]; fe_to_words(w, a); words_to_be_aligned(b, w); }   // a must be canonical

// ------------------------------------------------------------------------------------------------------- Fn
// Scalars mod n, 8 x 32-bit limbs, canonical (< n).  Used once or twice per item (range checks, GLV split,
// s = r + sk*c), so this is written for clarity, not speed.
struct sc {
    uint32_t v[8];
};
// n = FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFE BAAEDCE6 AF48A03B BFD25E8C D0364141 (fields/fr.rs:19)
PLUME_HD uint32_t sc_n(int i) {
    return i == 0 ? 0xD0364141u : i == 1 ? 0xBFD25E8Cu : i == 2 ? 0xAF48A03Bu : i == 3 ? 0xBAAEDCE6u : i == 4 ? 0xFFFFFFFEu : 0xFFFFFFFFu;
}
// NC = 2^256 - n = 1 45512319 50B75FC4 402DA173 2FC9BEBF (129 bits)
PLUME_HD uint32_t sc_nc(int i) { return i == 0 ? 0x2FC9BEBFu : i == 1 ? 0x402DA173u : i == 2 ? 0x50B75FC4u : i == 3 ? 0x45512319u : i == 4 ? 1u : 0u; }

PLUME_HD bool sc_is_zero(const sc& a) { uint32_t z = 0; PLUME_UNROLL for (int i = 0; i < 8; i++) z |= a.v[i]; return z == 0; }
PLUME_HD bool sc_lt_n(const sc& a) {  // a < n
    uint32_t bw = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) (void)subb(a.v[i], sc_n(i), bw);
    return bw != 0;
}
PLUME_HD void sc_cond_sub_n(sc& a) {  // a in [0, 2n) -> [0, n) ... (only one subtraction)
    sc t; uint32_t bw = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) t.v[i] = subb(a.v[i], sc_n(i), bw);
    const uint32_t bm = sel_mask(bw != 0);
    PLUME_UNROLL for (int i = 0; i < 8; i++) a.v[i] = sel32(bm, a.v[i], t.v[i]);
}
PLUME_HD void sc_from_be(sc& r, const uint8_t* b) {
    PLUME_UNROLL for (int i = 0; i < 8; i++) {
        const uint8_t* q = b + 4 * (7 - i);
        r.v[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
    }
}
PLUME_HD void sc_from_be_aligned(sc& r, const uint8_t* b) {
    const uint32_t* w = (const uint32_t*)b;
    PLUME_UNROLL for (int i = 0; i < 8; i++) r.v[i] = bswap32(w[7 - i]);
}
PLUME_HD void sc_to_be_aligned(uint8_t* b, const sc& a) {
    uint32_t* w = (uint32_t*)b;
    PLUME_UNROLL for (int i = 0; i < 8; i++) w[7 - i] = bswap32(a.v[i]);
}
PLUME_HD void sc_add(sc& r, const sc& a, const sc& b) {  // canonical inputs
    uint32_t c = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) r.v[i] = addc(a.v[i], b.v[i], c);
    // if carry or r >= n: subtract n  (r + NC mod 2^256 when carry)
    sc t; uint32_t bw = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) t.v[i] = subb(r.v[i], sc_n(i), bw);
    bool take = c || !bw;
    const uint32_t tm = sel_mask(take);
    PLUME_UNROLL for (int i = 0; i < 8; i++) r.v[i] = sel32(tm, t.v[i], r.v[i]);
}
PLUME_HD void sc_neg(sc& r, const sc& a) {  // n - a, 0 -> 0
    uint32_t bw = 0; bool z = sc_is_zero(a);
    PLUME_UNROLL for (int i = 0; i < 8; i++) r.v[i] = subb(sc_n(i), a.v[i], bw);
    const uint32_t zm = sel_mask(z);
    PLUME_UNROLL for (int i = 0; i < 8; i++) r.v[i] = r.v[i] & ~zm;
}
// generic NxM limb product (small helper for the scalar side)
template <int NA, int NB>
PLUME_HD void mul_limbs(uint32_t* t, const uint32_t* a, const uint32_t* b) {
    PLUME_UNROLL for (int i = 0; i < NA + NB; i++) t[i] = 0;
    PLUME_UNROLL for (int i = 0; i < NA; i++) {
        uint32_t carry = 0;
        PLUME_UNROLL for (int j = 0; j < NB; j++) {
            uint64_t p = (uint64_t)a[i] * b[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)p; carry = (uint32_t)(p >> 32);
        }
        t[i + NB] = carry;
    }
}
// 512-bit t -> t mod n.  Fold hi*NC three times (512 -> 385 -> 258 -> 256+), then at most two subtractions.
PLUME_HD void sc_reduce_wide(sc& r, const uint32_t t[16]) {
    uint32_t nc[5];
    PLUME_UNROLL for (int i = 0; i < 5; i++) nc[i] = sc_nc(i);
    // round 1: a = lo(8) + hi(8)*NC(5)  -> 13 limbs + carry -> 14 limbs
    uint32_t a[14], m[13];
    mul_limbs<8, 5>(m, t + 8, nc);
    uint32_t c = 0;
    PLUME_UNROLL for (int i = 0; i < 13; i++) a[i] = addc(m[i], i < 8 ? t[i] : opaque_zero(), c);
    a[13] = c;
    // round 2: b = a[0..8) + a[8..14)*NC -> 6+5 = 11 limbs (value < 2^(130+129)) + lo
    uint32_t m2[11], b[12];
    mul_limbs<6, 5>(m2, a + 8, nc);
    c = 0;
    PLUME_UNROLL for (int i = 0; i < 11; i++) b[i] = addc(m2[i], i < 8 ? a[i] : opaque_zero(), c);
    b[11] = c;
    // round 3: d = b[0..8) + b[8..12)*NC: b[8..] < 2^4 so the product < 2^133
    uint32_t m3[9], d[9];
    mul_limbs<4, 5>(m3, b + 8, nc);
    c = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) d[i] = addc(m3[i], b[i], c);
    d[8] = m3[8] + c;  // 0 or 1
    // d < 2^256 + 2^133: if d[8] then d - 2^256 + NC (< 2^134, no further carry)
    uint32_t c4 = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) r.v[i] = addc(d[i], d[8] ? sc_nc(i) : opaque_zero(), c4);
    sc_cond_sub_n(r);
    sc_cond_sub_n(r);
}
PLUME_HD void sc_mul(sc& r, const sc& a, const sc& b) {
    uint32_t t[16];
    mul_limbs<8, 8>(t, a.v, b.v);
    sc_reduce_wide(r, t);
}
// digest (32 bytes BE) -> scalar mod n; *canonical = digest in [1, n-1]  (Scalar::reduce, rust-k256/src/lib.rs:128;
// NonZeroScalar::from_repr, randomizedsigner.rs:90)
PLUME_HD void sc_from_digest_words(sc& r, const uint32_t h[8], bool& canonical) {
    PLUME_UNROLL for (int i = 0; i < 8; i++) r.v[i] = opaque_u32(h[7 - i]);
    canonical = sc_lt_n(r) && !sc_is_zero(r);
    sc_cond_sub_n(r);  // digest < 2^256 < 2n
}

}  // namespace plume
