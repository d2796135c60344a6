// secp256k1 group law, GLV split, fixed-position digit recoding, per-lane window tables and the multi-scalar
// loop for the PLUME hot path (the reference's `ProjectivePoint * Scalar` and point subtraction:
// rust-k256/src/lib.rs:101,109; randomizedsigner.rs:51,53,67,70 — there done by the k256 crate).
//
// Design (SIMT-first, not a port of k256/libsecp): one scalar multiplication "task" per lane; because every
// lane of a wavefront must execute the same instruction stream, recoding is by FIXED positions rather than wNAF
// or any other sparse form -- all lanes add at the same positions, a zero digit just idles its lane for one slot.
// The endomorphism split (k = k1 + k2*lambda, |k_i| < 2^128) halves the doubling chain to 128; since round 5 the
// pair (k1, k2) is recoded as ONE Eisenstein integer in base 4 (a digit per two doublings, from the residues of
// Z[w] / 4) and a table holds three AFFINE rows -- P, theta P = P - lambda P, 2P, each with beta*x -- whose unit
// multiples are free.  Tables are produced by the table passes with one field inversion per 8 lanes shared by
// all tables those lanes build (Montgomery's trick through HBM scratch).
#pragma once
#include "plume_field.h"

// host test builds count how often a multi-scalar chain had to be redone with checked additions (tests/devsim)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(PLUME_FE_CHECK)
namespace plume { inline unsigned long& fallback_counter() { static unsigned long c = 0; return c; } }
#define PLUME_COUNT_FALLBACK() (++plume::fallback_counter())
#else
#define PLUME_COUNT_FALLBACK() ((void)0)
#endif

namespace plume {

struct jac {
    fe x, y, z;
    uint32_t inf;
};

PLUME_HD fe fe_beta() { return fe_set(0x7AE96A2Bu, 0x657C0710u, 0x6E64479Eu, 0xAC3434E9u, 0x9CF04975u, 0x12F58995u, 0xC1396C28u, 0x719501EEu); }
PLUME_HD fe fe_gx() { return fe_set(0x79BE667Eu, 0xF9DCBBACu, 0x55A06295u, 0xCE870B07u, 0x029BFCDBu, 0x2DCE28D9u, 0x59F2815Bu, 0x16F81798u); }  // curves/mod.rs:52
PLUME_HD fe fe_gy() { return fe_set(0x483ADA77u, 0x26A3C465u, 0x5DA4FBFCu, 0x0E1108A8u, 0xFD17B448u, 0xA6855419u, 0x9C47D08Fu, 0xFB10D4B8u); }  // curves/mod.rs:57

// The uniform-schedule signer's accumulator offset (round 4): B = hash_to_curve("plume_hip uniform-schedule signer: accumulator offset", the RFC suite of the PLUME DST) --
// a point whose discrete logarithm to G and to any H nobody knows -- with -(2^128 B) and -B (computed once with the oracle, checked by the signer's parity tests: a wrong
// constant changes every output).  A chain that STARTS at B never holds the identity, so its additions need no "accumulator is the identity" case (which depends on
// the secret's leading digits); the offset comes off at the end with one checked addition.
PLUME_HD fe fe_off_x() { return fe_set(0xBE71E685u, 0x6986BD54u, 0xA5797708u, 0x9EF9F753u, 0x1397E48Bu, 0x08AA9EB1u, 0x128983CAu, 0x1D5B099Bu); }
PLUME_HD fe fe_off_y() { return fe_set(0x15AE5773u, 0x1FAF9233u, 0xF016B8C3u, 0x84D627FDu, 0x8CF7C500u, 0x22EAB6A1u, 0x27C3954Bu, 0x8A01ED04u); }
PLUME_HD fe fe_off_c128_x() { return fe_set(0x8141C548u, 0xF0861E56u, 0xAE562CECu, 0x00E54F8Bu, 0x7535C0B1u, 0x943A12D3u, 0x469080E1u, 0xA23EEBF9u); }   // -(2^128 B)
PLUME_HD fe fe_off_c128_y() { return fe_set(0xE429892Au, 0x9AB04F55u, 0xF3E86FE7u, 0xE225348Fu, 0xEF934040u, 0xDC585E87u, 0x1A3CC5BEu, 0x78194203u); }
PLUME_HD fe fe_off_c64_x() { return fe_set(0x8C789E12u, 0x56849B1Eu, 0x66BB27D0u, 0x5BE6CDC5u, 0x043289AFu, 0xB56465A3u, 0xAF7D81E9u, 0xD0746A2Fu); }   // -(2^64 B): the signer's half-length chains
PLUME_HD fe fe_off_c64_y() { return fe_set(0x951FAD83u, 0x6890A4BDu, 0x269416A6u, 0x5C972C30u, 0x9B5A23BDu, 0x5CBF67BDu, 0xB62C3867u, 0x3DBFF40Bu); }
PLUME_HD fe fe_off_c32_x() { return fe_set(0x04C05939u, 0x040E182Au, 0x8B808086u, 0x5E66F4DEu, 0x60F9C236u, 0xEB2BF058u, 0x0469B3DAu, 0x11F0FCAEu); }   // -(2^32 B): the signer's chains of 32 doublings (round 6)
PLUME_HD fe fe_off_c32_y() { return fe_set(0xF36744C1u, 0x473D51FAu, 0xF40D619Fu, 0xD4E1CF06u, 0xD5A12294u, 0x8CC33203u, 0x9C1D0738u, 0xDC827F44u); }
PLUME_HD fe fe_off_neg_y() { return fe_set(0xEA51A88Cu, 0xE0506DCCu, 0x0FE9473Cu, 0x7B29D802u, 0x73083AFFu, 0xDD15495Eu, 0xD83C6AB3u, 0x75FE0F2Bu); }    // -B = (x(B), this)

// y^2 == x^3 + 7 (curves/mod.rs:36-39)
PLUME_HD bool affine_on_curve(const fe& x, const fe& y) {
    fe l, r;
    fe_sqr(l, y);
    fe_sqr(r, x); fe_mul(r, r, x);
    fe seven = fe_small(7);
    fe_add(r, r, seven);
    return fe_eq(l, r);
}

// Limb-bound contract of a Jacobian point held in registers or HBM scratch (plume_field.h "tight"):
//   X, Y tight;  Z limbs <= 2^30 + 2^20 (a tight value or the unreduced double of one).
// The group-law formulas below keep additions and subtractions unreduced wherever the next multiplication tolerates it
// (fe_add_lazy / fe_sub_lazy<M>) and spend a carry pass only where a bound would otherwise be exceeded; host builds with
// PLUME_FE_CHECK assert every bound.
//
// unreduced negation of a tight value, limbs <= 2p (also a legal qy of jac_madd)
PLUME_HD void fe_neg_lazy(fe& r, const fe& a) { fe z = fe_zero(); fe_sub_lazy<2>(r, z, a); }
// 2P, a = 0:  E = 3X^2, B = 2Y^2, D = 4XY^2 = X * 2B, X' = E^2 - 2D, Y' = E (D - X') - 8Y^4 = E (D - X') + (-B) * 2B, Z' = 2YZ.
// Round 3: 3S + 2M + one two-product multiplication with a shared fold, NO carry pass and no separate 8Y^4: the subtraction of X' rides in the squaring's low
// columns (fe_sqr_sub2), 8Y^4 = B * 2B joins Y' as the second product of fe_muladd (81 multiply-adds more, a squaring + a subtraction + a carry pass less), and
// 4XY^2 comes out of its multiplication already doubled (the operand 2B): 1009 -> ~880 instructions per doubling.  (Rounds 1-2: 3M + 4S, 2 carry passes.)
// valid for every non-infinity point (the curve has no 2-torsion); the inf flag just rides along.
PLUME_HD void jac_dbl(jac& p) {
    fe B, dB, nB, E, D, t, dY;
    fe_sqr3(E, p.x);                                       // 3X^2, tight: the factor rides in the squaring's operands (no tripling, no carry pass)
    fe_sqr2_d(B, dY, p.y);                                 // 2Y^2 straight out of the squaring (cross products d_i d_j, diagonal a_i d_i);  dY = 2Y
    fe_mul(p.z, dY, p.z);                                  // Z' = 2YZ
    fe_dbl_lazy(dB, B);                                    // 4Y^2, limbs <= 2^30 + 2^20
    fe_mul(D, p.x, dB);                                    // 4XY^2, tight
    fe_sqr_sub2<2>(p.x, E, D);                             // X' = 9X^4 - 8XY^2, tight
    fe_sub_lazy<2>(t, D, p.x);                             // 4XY^2 - X', limbs <= 3 * 2^29 + 2^19
    fe_neg_lazy(nB, B);                                    // -2Y^2, unreduced (limbs <= 2p)
    fe_muladd(p.y, E, t, nB, dB);                          // Y' = E (D - X') - 8Y^4; bound: 9 (2^29 * 3 * 2^29 + 2^30 * 2^30) (1 + 2^-9) < 2^64 - 2^57
}
// -2P: the same doubling with the sign of Y' flipped -- Y'' = E (X' - D) + B * 2B = -Y' -- which needs no negated operand (jac_dbl spends nine subtractions on -B).
// (X', -Y', Z') is the point -2P, so an EVEN number of these in a row doubles that many times: the four doublings between two windows of the multi-scalar loops are
// two such pairs (round 4; 0.3 % of that kernel).
PLUME_HD void jac_dbl_neg(jac& p) {
    fe B, dB, E, D, t, dY;
    fe_sqr3(E, p.x);
    fe_sqr2_d(B, dY, p.y);
    fe_mul(p.z, dY, p.z);
    fe_dbl_lazy(dB, B);
    fe_mul(D, p.x, dB);
    fe_sqr_sub2<2>(p.x, E, D);                             // X' = 9X^4 - 8XY^2
    fe_sub_lazy<2>(t, p.x, D);                             // X' - 4XY^2, limbs <= 3 * 2^29 + 2^19
    fe_muladd(p.y, E, t, B, dB);                           // -Y' = E (X' - D) + 8Y^4
}
// cold path of the additions (P == Q).  Takes and returns BY VALUE through a local copy at the call site: passing the
// accumulator by reference would make its address escape and pin it in scratch memory for the whole hot loop
// (measured: ~900 scratch stores per lane and 21 GB of write traffic per 2^20 batch before this change).
PLUME_HD_NOINLINE void jac_dbl_cold_impl(jac* p) { jac_dbl(*p); }
PLUME_HD void jac_dbl_cold(jac& p) { jac tmp = p; jac_dbl_cold_impl(&tmp); p = tmp; }

// p += (qx, qy) affine, q != infinity; qx tight, qy tight or an unreduced negation (limbs <= 2p).  8M + 3S, no carry pass (round 3: the three subtractions ride in their products' folds).
// CHECKED = true handles every exceptional case (p infinite, p == q, p == -q).
// CHECKED = false is the hot-loop form: it handles p infinite but does NOT test for p == +-q.  In that case H = 0 (mod p)
// and Z' = Z*H = 0 (mod p), and every later doubling / addition keeps Z = 0 (mod p) (Z only ever gets multiplied), so
// the caller detects the event ONCE at the end (fe_is_zero(Z)) and recomputes that lane with CHECKED = true.  No other
// path reaches Z = 0: a legitimate identity is carried in the inf flag, and the curve has no point with Y = 0.
template <bool CHECKED = true>
PLUME_HD void jac_madd(jac& p, const fe& qx, const fe& qy) {
    if (p.inf) {
        p.x = qx; fe_carry(p.x); p.y = qy; fe_carry(p.y); p.z = fe_small(1); p.inf = 0;      // (qx: a table row's beta^2 x arrives unreduced, ld_tab_unit)
        return;
    }
    fe z1z1, s2, h, r, hh, hhh, v, t;
    fe_sqr(z1z1, p.z);
    fe_mul_sub<2>(h, qx, z1z1, p.x);                       // H = U2 - X1: the subtraction rides in the product's low columns (round 3; rounds 1-2: fe_sub_lazy + fe_carry, 47 instructions instead of 18)
    fe_mul(s2, p.z, z1z1);
    fe_mul_sub<2>(r, s2, qy, p.y);                         // r = S2 - Y1
    if (CHECKED) {
        if (fe_is_zero(h)) {
            if (fe_is_zero(r)) { jac_dbl_cold(p); } else { p.inf = 1; }
            return;
        }
    }
    fe_sqr(hh, h); fe_mul(hhh, hh, h); fe_mul(v, p.x, hh);
    fe_mul(p.z, p.z, h);
    fe_dbl_lazy(hh, v); fe_add_lazy(hh, hh, hhh);          // 2V + H^3, limbs <= 3 * (2^29 + 2^19) <= 4p
    fe_sqr_sub<4>(p.x, r, hh);                             // X' = r^2 - H^3 - 2V
    fe_sub_lazy<2>(t, v, p.x);                             // V - X'
    fe_neg_lazy(v, p.y);                                   // -Y1, unreduced
    fe_muladd(p.y, r, t, v, hhh);                          // Y' = r(V - X') - Y1*H^3: both products share one fold
}

// One table addition of the UNIFORM schedule (opt-in signer, round 4): the same instructions whatever the digit d is.  (qx, qy) is the row of |d| -- of 1 when d = 0 --,
// the sign is applied by a masked select, the addition always runs on a copy and a masked select keeps or drops it.  No branch depends on d; the row's ADDRESS still does.
// The accumulator is never the identity here (the chain starts at the offset point), so jac_madd's identity case is never taken.
template <bool CHECKED>
PLUME_HD void jac_madd_uniform(jac& p, const fe& qx, const fe& qy, int d) {
    fe y = qy, ny;
    fe_neg_lazy(ny, qy);
    fe_cmov(y, ny, d < 0);
    jac t = p;
    jac_madd<CHECKED>(t, qx, y);
    const bool take = d != 0;
    fe_cmov(p.x, t.x, take); fe_cmov(p.y, t.y, take); fe_cmov(p.z, t.z, take);
    if (CHECKED) p.inf = take ? t.inf : p.inf;      // (the checked form only runs in the redo of a chain that met p == +-q)
}

// p += q, both Jacobian.  12M + 4S; all exceptional cases handled.
PLUME_HD void jac_add(jac& p, const jac& q) {
    if (q.inf) return;
    if (p.inf) { p = q; return; }
    fe z1z1, z2z2, u1, s1, s2, h, r, hh, hhh, v, t;
    fe_sqr(z1z1, p.z); fe_sqr(z2z2, q.z);
    fe_mul(u1, p.x, z2z2);
    fe_mul(s1, q.z, z2z2); fe_mul(s1, s1, p.y);
    fe_mul_sub<2>(h, q.x, z1z1, u1);                       // H = U2 - U1 (the subtraction rides in the product's fold)
    fe_mul(s2, p.z, z1z1);
    fe_mul_sub<2>(r, s2, q.y, s1);                         // r = S2 - S1
    if (fe_is_zero(h)) {
        if (fe_is_zero(r)) { jac_dbl_cold(p); } else { p.inf = 1; }
        return;
    }
    fe_sqr(hh, h); fe_mul(hhh, hh, h); fe_mul(v, u1, hh);
    fe_mul(p.z, p.z, q.z); fe_mul(p.z, p.z, h);
    fe_dbl_lazy(hh, v); fe_add_lazy(hh, hh, hhh);          // 2V + H^3
    fe_sqr_sub<4>(p.x, r, hh);                             // X' = r^2 - H^3 - 2V
    fe_sub_lazy<2>(t, v, p.x);
    fe_neg_lazy(v, s1);
    fe_muladd(p.y, r, t, v, hhh);                          // r(V - X') - S1*H^3 with one fold
}

// Jacobian p == affine (ax, ay)?  (ProjectivePoint == AffinePoint, rust-k256/src/lib.rs:117,122); a_inf = the affine side is the identity
PLUME_HD bool jac_eq_affine(const jac& p, const fe& ax, const fe& ay, bool a_inf) {
    if (p.inf || a_inf) return p.inf && a_inf;
    fe z2, z3, t;
    fe_sqr(z2, p.z); fe_mul(z3, z2, p.z);
    fe_mul(t, ax, z2);
    if (!fe_eq(t, p.x)) return false;
    fe_mul(t, ay, z3);
    return fe_eq(t, p.y);
}

// ------------------------------------------------------------------------------------------------ GLV split
// k = k1 + k2*lambda (mod n) with |k1|, |k2| < 2^128; returns magnitudes (4 limbs each) and signs.
// Constants: the standard secp256k1 endomorphism lattice basis (lambda^3 = 1 mod n, beta^3 = 1 mod p,
// lambda*(x, y) = (beta*x, y)); checked against the oracle's plain double-and-add in tests/test_devsim.py.
struct glv_half {
    uint32_t m[4];
    uint32_t neg;
};
// Round 4: exact integer arithmetic on 29-bit limbs, no reduction mod n anywhere.  With the lattice basis (a1, b1), (a2, b2) (a_i + b_i lambda = 0 mod n):
//     c1 = round(k g1 / 2^384), c2 = round(k g2 / 2^384)     (g1 = round(2^384 b2 / n), g2 = round(2^384 (-b1) / n))
//     k1 = k - c1 a1 - c2 a2,   k2 = c1 (-b1) - c2 b2        as INTEGERS: both are below 2^128 in magnitude, so 174 bits of two's complement hold them
// -- the same k1, k2 the earlier form reached through a 256 x 256-bit product mod n and k1 = k - k2 lambda mod n (1700 instructions per split, a third of them register
// moves of the generic 32-bit schoolbook products; this form: product columns accumulate in 64 bits without carries, like the field arithmetic).
PLUME_HD void glv_cmul(uint32_t c[5], const uint32_t kl[9], const uint32_t g[9]) {      // c = round(k g / 2^384) as five 29-bit limbs (c < 2^129)
    uint64_t acc = 0;
    uint32_t top[5];
    PLUME_UNROLL for (int m = 0; m < 17; m++) {
        PLUME_UNROLL for (int i = 0; i < 9; i++) { const int j = m - i; if (j >= 0 && j < 9) acc += (uint64_t)kl[i] * g[j]; }
        if (m >= 13) top[m - 13] = (uint32_t)acc & PLUME_FE_MASK;
        acc >>= 29;
    }
    top[4] = (uint32_t)acc;                                 // limb 17: what is left (k g < 2^512 = 2^(29 * 17 + 19))
    // bit 383 of the product is bit 6 of limb 13: add the rounding half, then drop 7 bits
    uint32_t cy = 64u;
    PLUME_UNROLL for (int i = 0; i < 5; i++) { const uint32_t t = top[i] + cy; cy = i < 4 ? t >> 29 : 0u; top[i] = i < 4 ? (t & PLUME_FE_MASK) : t; }
    PLUME_UNROLL for (int i = 0; i < 5; i++) c[i] = ((top[i] >> 7) | (i < 4 ? top[i + 1] << 22 : 0u)) & PLUME_FE_MASK;
}
// low six limbs (174 bits) of x * y for five-limb x, y
PLUME_HD void glv_mul_lo(uint32_t r[6], const uint32_t x[5], const uint32_t y[5]) {
    uint64_t acc = 0;
    PLUME_UNROLL for (int m = 0; m < 6; m++) {
        PLUME_UNROLL for (int i = 0; i < 5; i++) { const int j = m - i; if (j >= 0 && j < 5) acc += (uint64_t)x[i] * y[j]; }
        r[m] = (uint32_t)acc & PLUME_FE_MASK;
        acc >>= 29;
    }
}
// d = six signed limb differences (|d_i| < 2^31) that sum, MODULO 2^174, to a value below 2^128 in magnitude (the operands were truncated to 174 bits, so what falls off
// the top of the carry chain means nothing) -> magnitude (four 32-bit words) and sign
PLUME_HD void glv_finish(glv_half& h, const int32_t d[6]) {
    int32_t cy = 0;
    uint32_t l[6];
    PLUME_UNROLL for (int i = 0; i < 6; i++) { const int32_t t = d[i] + cy; l[i] = (uint32_t)t & PLUME_FE_MASK; cy = t >> 29; }      // arithmetic shift: the borrow travels up
    const bool neg = (l[5] >> 28) != 0;                     // |value| < 2^128: modulo 2^174 it is either below 2^128 or within 2^128 of the top -- bit 173 tells which
    const uint32_t m = sel_mask(neg);
    uint32_t bw = neg ? 1u : 0u;                            // two's complement of the six limbs when negative: ~l + 1
    PLUME_UNROLL for (int i = 0; i < 6; i++) { const uint32_t t = ((l[i] ^ m) & PLUME_FE_MASK) + bw; l[i] = t & PLUME_FE_MASK; bw = t >> 29; }
    h.m[0] = l[0] | (l[1] << 29);
    h.m[1] = (l[1] >> 3) | (l[2] << 26);
    h.m[2] = (l[2] >> 6) | (l[3] << 23);
    h.m[3] = (l[3] >> 9) | (l[4] << 20);
    h.neg = neg ? 1u : 0u;
}
PLUME_HD void glv_split(glv_half& h1, glv_half& h2, const sc& k) {
    const uint32_t G1[9] = {0x05DBB031u, 0x049904D2u, 0x1A329FFAu, 0x151428E3u, 0x0EB153DAu, 0x08724942u, 0x0F37A1B2u, 0x0434FA8Du, 0x003086D2u};
    const uint32_t G2[9] = {0x0AC47F71u, 0x0B8DA574u, 0x1D41B185u, 0x0411593Bu, 0x1E4C4221u, 0x1FD4855Fu, 0x00A1BD51u, 0x1AC021D1u, 0x00E4437Eu};
    const uint32_t A1[5] = {0x1284EB15u, 0x03648724u, 0x151AF37Au, 0x0DA4434Fu, 0x00000308u};      // a1 = b2
    const uint32_t MB1[5] = {0x0ABFE4C3u, 0x1AA3FD48u, 0x03A20A1Bu, 0x06FDAC02u, 0x00000E44u};     // -b1
    const uint32_t A2[5] = {0x1D44CFD8u, 0x1E08846Cu, 0x18BCFD95u, 0x14A1EF51u, 0x0000114Cu};
    fe kf;
    fe_from_words(kf, k.v);                                 // k as nine 29-bit limbs
    uint32_t c1[5], c2[5], p[6], q[6];
    glv_cmul(c1, kf.v, G1);
    glv_cmul(c2, kf.v, G2);
    int32_t d[6];
    glv_mul_lo(p, c1, MB1); glv_mul_lo(q, c2, A1);          // k2 = c1 (-b1) - c2 b2
    PLUME_UNROLL for (int i = 0; i < 6; i++) d[i] = (int32_t)p[i] - (int32_t)q[i];
    glv_finish(h2, d);
    glv_mul_lo(p, c1, A1); glv_mul_lo(q, c2, A2);           // k1 = k - c1 a1 - c2 a2
    PLUME_UNROLL for (int i = 0; i < 6; i++) d[i] = (int32_t)kf.v[i] - (int32_t)p[i] - (int32_t)q[i];
    glv_finish(h1, d);
}

// The grid the generator's WIDE digits sit on: windows of PLUME_WBITS = 4 bits, PLUME_NDIG = 33 of them covering a 129-bit half (rounds 1-4 walked this grid for every
// base -- 4-bit Booth digits, tables 1P..8P; 5-bit windows were built and measured in rounds 2 and 3 and lost to their table stage, LABNOTES.md.  Since round 5 the per-item
// bases use the Eisenstein digits further down, whose positions are half a window apart: window w = position 2w.)
#define PLUME_WBITS 4
#define PLUME_NDIG ((128 + PLUME_WBITS) / PLUME_WBITS)       // 33
// Booth recoding with a WIDE window for the generator's slots of the verifier: m = sum d_k 2^(W k), d_k in [-2^(W-1), 2^(W-1)].
// W = PLUME_GW must be a multiple of 4 so that digit k lines up with the 4-bit window i = k * W/4 of the shared doubling chain.
// W = 12: 11 digits per 128-bit half (22 generator additions per verify instead of 34 with W = 8) from a 2048-entry table (256 KiB, L2-resident);
// W = 16: 9 digits per half (18 additions) from 32768 entries (4 MiB: L2 / MALL), built once per context in ~0.2 s.
#ifndef PLUME_GW   // (the one width knob that stays: host builds of these headers take -DPLUME_GW=16 / 20, a CPU cannot build 8 M rows per test run)
#define PLUME_GW 24   // round 3: 2^23 entries (1 GiB of the 288), 6 digits per half: 12 generator additions per verify -- multi-scalar kernel -1.1 %, a verify -1.3 % against W = 16 (32768 entries, 4 MiB, 18
                      // additions; W = 20: 64 MiB, 14 additions, -0.4 %; W = 12: 2048 entries, 22 additions).  The rows of a 1 GiB table come from HBM, not from L2 / MALL; the kernel's other wavefronts cover it.
                      // Host builds of these headers (tests/devsim) take -DPLUME_GW=16: a CPU cannot build 8 M rows per test run
#endif
static_assert(PLUME_GW % PLUME_WBITS == 0, "a wide digit must line up with the windows of the shared doubling chain");
static_assert(PLUME_GW <= 24, "the digit rows hold a wide digit as a magnitude of up to three bytes plus its sign");
static_assert(PLUME_GW >= 16, "wide digits are stored as magnitude bytes + a sign byte: needs W / 4 >= 4 positions per digit");
#define PLUME_GW_BYTES (PLUME_GW > 16 ? 3 : 2)          // magnitude bytes of a wide digit in its full form
#define PLUME_GWS (PLUME_GW / PLUME_WBITS)                // windows per wide digit
#define PLUME_NDIGW ((128 + PLUME_GW) / PLUME_GW)         // digits covering 129 bits: 11 for W = 12, 17 for W = 8
#define PLUME_GTAB_ENTRIES (1 << (PLUME_GW - 1))
PLUME_HD int booth_digit_w(const uint32_t m[4], int k) {
    const int W = PLUME_GW, lo = W * k - 1;
    const uint32_t mask = (1u << (W + 1)) - 1u;
    uint32_t u;
    if (lo < 0) {
        u = (m[0] << 1) & mask;
    } else {
        int wi = lo >> 5, sh = lo & 31;
        uint32_t a = wi < 4 ? m[wi] : 0u, b = (wi + 1) < 4 ? m[wi + 1] : 0u;
        u = ((a >> sh) | ((sh > 31 - W) ? (b << (32 - sh)) : 0u)) & mask;
    }
    return (int)(u & 1) + (int)((u >> 1) & ((1u << (W - 1)) - 1u)) - (int)((u >> W) << (W - 1));
}
// wide digits share the 33-slot digit row of a half-scalar: digit k occupies positions S*k .. S*k+S-1 (S = W/4):
// byte 0 = magnitude bits 0..7, byte 1 = magnitude bits 8..11 | sign << 7.  If byte 1 would fall off the row (W = 8, top digit,
// which is 0 or 1) the sign goes to bit 6 of byte 0.
PLUME_HD void booth_store_wide(int8_t* dig, uint32_t stride, const glv_half& h, bool flip) {
    bool neg = (h.neg != 0) != flip;
    PLUME_UNROLL for (int k = 0; k < PLUME_NDIGW; k++) {
        int d = booth_digit_w(h.m, k);
        bool dn = (d < 0) != neg;
        int mag = d < 0 ? -d : d;
        const int pos = PLUME_GWS * k;
        const int8_t sgn = (int8_t)((mag != 0 && dn) ? 1 : 0);
        static_assert(PLUME_GWS >= PLUME_GW_BYTES + 1, "a wide digit's bytes fit its positions");
        if (pos + PLUME_GW_BYTES < PLUME_NDIG) {
            PLUME_UNROLL for (int b = 0; b < PLUME_GW_BYTES; b++) dig[(uint32_t)(pos + b) * stride] = (int8_t)(uint8_t)((mag >> (8 * b)) & 0xFF);
            dig[(uint32_t)(pos + PLUME_GW_BYTES) * stride] = sgn;
        } else if (pos + 1 < PLUME_NDIG) {          // the top digit of a 129-bit half when the row ends early (W = 20, 24: bits 120.., at most 2^9): two bytes, sign in bit 7 of the second
            dig[(uint32_t)pos * stride] = (int8_t)(uint8_t)(mag & 0xFF);
            dig[(uint32_t)(pos + 1) * stride] = (int8_t)(uint8_t)(((mag >> 8) & 0x7F) | (sgn ? 0x80 : 0));
        } else {                                    // W = 16: the top digit (0 or 1) sits on the row's last position
            dig[(uint32_t)pos * stride] = (int8_t)(uint8_t)((mag & 0x3F) | (sgn ? 0x40 : 0));
        }
    }
}
// ---------------------------------------------------------------------------------- digits in the Eisenstein integers (round 5)
// A GLV pair (k1, k2) IS the Eisenstein integer kappa = k1 + k2 w (w^2 + w + 1 = 0; w acts as lambda: (x, y) -> (beta x, y)).  Rounds 1-4 recoded the two halves apart --
// 4-bit Booth windows, a P-slot and a lambda-P-slot per window, tables of 1P..8P -- i.e. 17 x 17 joint digit values of which only the 8 multiples of P were shared.  The
// ring structure gives more: base 4 with ONE digit per position taken from the sixteen residues of Z[w] / 4, represented by their smallest elements
//     0;   the six units  +-1, +-w, +-w^2;   the six associates of theta = 1 - w (norm 3);   2, 2w, 2w^2 and their negatives (norm 4; -2 = 2 mod 4)
// and every associate u d of a digit d costs nothing: u (x, y) = (x | beta x | beta^2 x, +-y).  So a table is THREE rows -- P, theta P = P - lambda P, 2P -- instead of
// eight, built with ONE round of inversions instead of three (the denominators 2y and (beta - 1) x come straight from P), while the chain keeps its counts: a position
// (two doublings) adds one point with probability 15/16, where a 4-bit window (four doublings) added two.  Table stage of a 2^20 verify: see DESIGN.md.
//   digit code (one byte per position):  0 = nothing to add;  otherwise 1 + 6 row + 2 j + neg  =  (-1)^neg w^j (row point),  row 0: P, 1: theta P, 2: 2P
// Recoding walks the pair two bits at a time with carries in {-1, 0, 1}: t = (sign chunk + carry) for both coordinates, the digit is the representative of t mod 4 that
// keeps the next carry small (64-entry table: residues and signs of t), carry = (t - d) / 4.  After the last chunk the carries themselves are a digit (a unit or +-theta), so
// a magnitude below 4^(NP - 1) needs NP positions.
#define PLUME_NPOS 65        // pairs of 128-bit halves (the verifier)
#define PLUME_NPOS64 33      // pairs of 64-bit quarters (the signer's chains of 64 doublings)
#define PLUME_NPOS66 34      // pairs below 2^66 (the verifier's short first equation, plume_eis.h)
// The signer's multiplications by H: each 128-bit GLV half is cut into PLUME_SIGN_K pieces of 128 / K bits, piece j on the table of 2^(128 j / K) H -- K joint slots along ONE
// chain of 128 / K doublings, after 128 (K - 1) / K doublings spent once per item on the shifted bases, which serve both of its multiplications (sk H and r H).  Per item:
// K = 1: 256 doublings; K = 2 (round 4, the default): 64 + 2 x 64 = 192; K = 4: 96 + 2 x 32 = 160 for two more three-row tables and 2 x 4 x 17 instead of 2 x 2 x 33 positions
// -- built and measured in round 6 (profiles/r06_sign_cut32.txt, same box, uniform level 1): k_sign_hmul 9.33 -> 8.31 ms, but k_sign_hdbl 1.57 -> 2.43 and the table stage
// 0.45 -> 0.83: a 2^20 sign 16.11 -> 16.49 ms (+2.4 %).  The doublings saved in the chains come back in the shifted bases and the tables: K stays 2.
#ifndef PLUME_SIGN_K
#define PLUME_SIGN_K 2
#endif
#define PLUME_SIGN_BITS (128 / PLUME_SIGN_K)                  // bits per piece = doublings per hop between the shifted bases = doublings of the chain
#define PLUME_NPOSK (PLUME_SIGN_BITS / 2 + 1)                 // positions of a pair of pieces: 17 for K = 4, 33 for K = 2
static_assert(PLUME_SIGN_K == 2 || PLUME_SIGN_K == 4, "pieces of 64 or 32 bits (whole words of the GLV halves)");
PLUME_HD uint32_t eisd_entry_table(int ta, int tb) {
    // entry = code | (d0 + 2) << 5 | (d1 + 2) << 8 for t = (ta, tb), |ta|, |tb| <= 4; generated (and its invariants checked) by tests/test_devsim.py::test_eisenstein_digit_table
    static const uint16_t T[64] = {0x240, 0x343, 0x44F, 0x144, 0x261, 0x366, 0x469, 0x167, 0x28D, 0x38C, 0x492, 0x10B, 0x222, 0x328, 0x02A, 0x125,
                                   0x240, 0x343, 0x44F, 0x144, 0x261, 0x366, 0x469, 0x167, 0x28D, 0x38C, 0x011, 0x10B, 0x222, 0x328, 0x02A, 0x125,
                                   0x240, 0x343, 0x44F, 0x144, 0x261, 0x366, 0x469, 0x167, 0x28D, 0x38C, 0x011, 0x10B, 0x222, 0x328, 0x02A, 0x125,
                                   0x240, 0x343, 0x44F, 0x144, 0x261, 0x366, 0x469, 0x167, 0x28D, 0x38C, 0x011, 0x10B, 0x222, 0x328, 0x02A, 0x125};
    const uint32_t idx = (((uint32_t)ta & 3u) << 2) | ((uint32_t)tb & 3u) | (ta < 0 ? 16u : 0u) | (tb < 0 ? 32u : 0u);
    return T[idx];
}
// The same table held in registers (what the kernels run: no load per position).  The four sign quadrants differ in ONE entry -- the residue (2, 2): 2w^2 or its negative
// -- so sixteen entries packed into three constants (5-bit codes, 3-bit d0 + 2, d1 + 2) and one special case; tests hold it to the table above for every t.
PLUME_HD uint32_t eisd_entry(int ta, int tb) {
    const uint32_t i = (((uint32_t)ta & 3u) << 2) | ((uint32_t)tb & 3u);
    const uint64_t CLO = 0x5C98D3A4C123C60ull, CHI = 0x2A902ull, D0 = 0x2491246DB492ull, D1 = 0x21A31A31A31Aull;     // codes 0..11 | codes 12..15 | d0 + 2 | d1 + 2
    uint32_t code = (uint32_t)((i < 12u ? CLO >> (5u * i) : CHI >> (5u * (i - 12u))) & 31u);
    uint32_t d0 = (uint32_t)(D0 >> (3u * i)) & 7u, d1 = (uint32_t)(D1 >> (3u * i)) & 7u;
    if (i == 10u && (ta < 0 || tb < 0)) { code = 17u; d0 = 0u; d1 = 0u; }
    return code | (d0 << 5) | (d1 << 8);
}
// NP digit codes of  +-(a + b w),  a = (aneg ? -1 : 1) am,  b likewise, magnitudes below 4^(NP - 1) given as NW little-endian words; dig[i * stride] = position i.
// Returns false if a carry is left over (the magnitude bound did not hold).
template <int NP, int NW>
PLUME_HD bool eisd_store(int8_t* dig, uint32_t stride, const uint32_t (&am)[NW], bool aneg, const uint32_t (&bm)[NW], bool bneg, bool flip) {
    const int sa = (aneg != flip) ? -1 : 1, sb = (bneg != flip) ? -1 : 1;
    int ca = 0, cb = 0;
    PLUME_UNROLL for (int i = 0; i < NP; i++) {
        const int w = (2 * i) >> 5, sh = (2 * i) & 31;
        const int cha = w < NW ? (int)((am[w < NW ? w : 0] >> sh) & 3u) : 0, chb = w < NW ? (int)((bm[w < NW ? w : 0] >> sh) & 3u) : 0;
        const int ta = sa * cha + ca, tb = sb * chb + cb;
        const uint32_t e = eisd_entry(ta, tb);
        dig[(uint32_t)i * stride] = (int8_t)(e & 31u);
        ca = (ta - ((int)((e >> 5) & 7u) - 2)) >> 2;                  // exact: t = d (mod 4)
        cb = (tb - ((int)((e >> 8) & 7u) - 2)) >> 2;
    }
    return ca == 0 && cb == 0;
}
PLUME_HD void eisd_store_glv(int8_t* dig, uint32_t stride, const glv_half& h1, const glv_half& h2, bool flip) {
    (void)eisd_store<PLUME_NPOS, 4>(dig, stride, h1.m, h1.neg != 0, h2.m, h2.neg != 0, flip);
}

// ------------------------------------------------------------------------------------------ window tables
// One table = 3 rows x 32 words (128 B = one cache line): row 0 = P, row 1 = theta P = P - lambda P, row 2 = 2P, affine, as 29-bit limbs of tight field elements:
//     [ x0..x7 | y0..y7 | b0..b7 | x8 y8 b8 0 | 0 0 0 0 ]        b = beta * x (the x of lambda * (row point))
// i.e. eight 16-byte quads; one table addition gathers seven of them (x, b, y, the top limbs) with aligned 16-byte loads from ONE line, and the table kernel writes whole
// lines (112-byte rows, without the padding quad, measured 6 % slower there).  The generator's fixed tables (wide window table, comb, scanned table) keep rows of PLAIN
// multiples in the same row format.  (Rounds 1-4: eight rows 1P..8P per table, 1 KiB.)
#define PLUME_TAB_ENTRIES 3
#define PLUME_FE_W PLUME_FE_WORDS
#define PLUME_JAC_WORDS (3 * PLUME_FE_WORDS)      // Jacobian point in HBM scratch: x | y | z
#define PLUME_TAB_ENTRY_WORDS 32                 // 128-byte rows (112-byte rows without the padding quad: 6 % slower in the table passes, round 2)
#define PLUME_TAB_WORDS (PLUME_TAB_ENTRIES * PLUME_TAB_ENTRY_WORDS)

// table entry access (layout above)
PLUME_HD void ld_tab_xy(fe& x, fe& y, const uint32_t* e, bool lambda_half) {
    const uint32_t* xs = lambda_half ? e + 16 : e;
    PLUME_UNROLL for (int i = 0; i < 8; i++) { x.v[i] = xs[i]; y.v[i] = e[8 + i]; }
    x.v[8] = lambda_half ? e[26] : e[24];
    y.v[8] = e[25];
}
// A row of a fixed table of plain multiples WITHOUT a digit-dependent address (the signer's uniform schedule, level 2: the scanned table of G): all ROWS rows of the window
// are read, row `ad` (1..ROWS) is kept by masked selects: 18 selects and 128 B of reads per row beside an addition of ~1400 instructions.
template <int ROWS>
PLUME_HD void ld_tab_xy_scan(fe& x, fe& y, const uint32_t* tab, int ad, bool lambda_half) {
    ld_tab_xy(x, y, tab, lambda_half);
    PLUME_NOUNROLL for (int e = 1; e < ROWS; e++) {
        fe rx, ry;
        ld_tab_xy(rx, ry, tab + (size_t)e * PLUME_TAB_ENTRY_WORDS, lambda_half);
        const bool take = ad == e + 1;
        fe_cmov(x, rx, take); fe_cmov(y, ry, take);
    }
}
PLUME_HD void st_tab_entry(uint32_t* e, const fe& x, const fe& y, const fe& bx) {   // (non-temporal stores here: 3.6x slower, they defeat write combining)
    PLUME_UNROLL for (int i = 0; i < 8; i++) { e[i] = x.v[i]; e[8 + i] = y.v[i]; e[16 + i] = bx.v[i]; }
    e[24] = x.v[8]; e[25] = y.v[8]; e[26] = bx.v[8]; e[27] = 0;
    e[28] = 0; e[29] = 0; e[30] = 0; e[31] = 0;     // padding quad: the row is written as a whole cache line
}
PLUME_HD void ld_fe(fe& r, const uint32_t* p) { PLUME_UNROLL for (int i = 0; i < PLUME_FE_W; i++) r.v[i] = p[i]; }
PLUME_HD void st_fe(uint32_t* p, const fe& a) { PLUME_UNROLL for (int i = 0; i < PLUME_FE_W; i++) p[i] = a.v[i]; }
// strided (SoA) field element: word w of element j lives at base[w*stride + j]
PLUME_HD void ld_fe_soa(fe& r, const uint32_t* base, size_t stride, size_t j) { PLUME_UNROLL for (int i = 0; i < PLUME_FE_W; i++) r.v[i] = base[(size_t)i * stride + j]; }
PLUME_HD void st_fe_soa(uint32_t* base, size_t stride, size_t j, const fe& a) { PLUME_UNROLL for (int i = 0; i < PLUME_FE_W; i++) base[(size_t)i * stride + j] = a.v[i]; }
// Jacobian point SoA: x words 0..8, y 9..17, z 18..26
PLUME_HD void ld_jac_soa(jac& p, const uint32_t* base, size_t stride, size_t j) {
    ld_fe_soa(p.x, base, stride, j); ld_fe_soa(p.y, base + PLUME_FE_W * stride, stride, j); ld_fe_soa(p.z, base + 2 * PLUME_FE_W * stride, stride, j);
}
PLUME_HD void st_jac_soa(uint32_t* base, size_t stride, size_t j, const jac& p) {
    st_fe_soa(base, stride, j, p.x); st_fe_soa(base + PLUME_FE_W * stride, stride, j, p.y); st_fe_soa(base + 2 * PLUME_FE_W * stride, stride, j, p.z);
}
// The base of a table job in HBM: ONE 112-byte record per job, job-major -- x0..x8 | y0..y8 | z0..z8 | 0 -- seven 16-byte quads.  (Rounds 1-2 kept the bases word-major,
// word w of job j at bases[w * njobs + j].  The table kernel's lanes walk L CONSECUTIVE jobs each, so a wavefront's load of one word touched 64 words L * 4 bytes apart: every
// 128-byte line was fetched L times over -- FETCH_SIZE of the two passes that read the bases: 2.3 GB per 2^20 verifies for 0.38 GB of records -- and the ingest kernel's
// stores, 3 words apart per lane, wrote every line in pieces: WRITE_SIZE 1.5 GB for 0.34 GB.  With one record per job a lane reads its job with five or seven aligned
// 16-byte loads from at most two lines nobody else needs, and a wavefront of the ingest kernel writes one contiguous 21 KB stretch.)
#define PLUME_BASE_WORDS 28
struct alignas(16) quad32 { uint32_t a, b, c, d; };
PLUME_HD void st_base(uint32_t* bases, size_t job, const jac& p) {
    quad32* q = reinterpret_cast<quad32*>(bases + job * PLUME_BASE_WORDS);
    q[0] = quad32{p.x.v[0], p.x.v[1], p.x.v[2], p.x.v[3]}; q[1] = quad32{p.x.v[4], p.x.v[5], p.x.v[6], p.x.v[7]};
    q[2] = quad32{p.x.v[8], p.y.v[0], p.y.v[1], p.y.v[2]}; q[3] = quad32{p.y.v[3], p.y.v[4], p.y.v[5], p.y.v[6]};
    q[4] = quad32{p.y.v[7], p.y.v[8], p.z.v[0], p.z.v[1]}; q[5] = quad32{p.z.v[2], p.z.v[3], p.z.v[4], p.z.v[5]};
    q[6] = quad32{p.z.v[6], p.z.v[7], p.z.v[8], 0u};
}
// x, y (five quads) and, for a Jacobian base, z (two more)
PLUME_HD void ld_base(jac& p, const uint32_t* bases, size_t job, bool with_z) {
    const quad32* q = reinterpret_cast<const quad32*>(bases + job * PLUME_BASE_WORDS);
    const quad32 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    p.x.v[0] = q0.a; p.x.v[1] = q0.b; p.x.v[2] = q0.c; p.x.v[3] = q0.d; p.x.v[4] = q1.a; p.x.v[5] = q1.b; p.x.v[6] = q1.c; p.x.v[7] = q1.d; p.x.v[8] = q2.a;
    p.y.v[0] = q2.b; p.y.v[1] = q2.c; p.y.v[2] = q2.d; p.y.v[3] = q3.a; p.y.v[4] = q3.b; p.y.v[5] = q3.c; p.y.v[6] = q3.d; p.y.v[7] = q4.a; p.y.v[8] = q4.b;
    if (with_z) {
        const quad32 q5 = q[5], q6 = q[6];
        p.z.v[0] = q4.c; p.z.v[1] = q4.d; p.z.v[2] = q5.a; p.z.v[3] = q5.b; p.z.v[4] = q5.c; p.z.v[5] = q5.d; p.z.v[6] = q6.a; p.z.v[7] = q6.b; p.z.v[8] = q6.c;
    } else {
        p.z = fe_small(1);
    }
}

#define PLUME_JOB_OK 0u
#define PLUME_JOB_INF 1u      // base is the identity: its slots are skipped
#define PLUME_JOB_INVALID 2u  // base failed validation: a dummy (G) table is built, the item is rejected elsewhere
#define PLUME_JOB_AFFINE 0x80u  // OR-ed in: the base has Z = 1, its chain uses mixed additions
PLUME_HD uint32_t job_state(uint8_t f) { return f & 3u; }
// the generator's wide table for the verifier's fixed-base slots: (1..2^(W-1))*G, same row format
#define PLUME_GTAB_WORDS (PLUME_GTAB_ENTRIES * PLUME_TAB_ENTRY_WORDS)
// the signer's doubling-free comb (below): 2^(W-1) entries per W-bit window
#ifndef PLUME_COMB_W
#define PLUME_COMB_W 18   // round 3: 15 windows x 131072 entries (252 MiB), 15 additions per multiplication: signer's comb kernel 1.59 -> 1.26 ms (W = 20: 13 x 524288 entries, 872 MiB, 1.10 ms -- but the
                          // kernels after it lose what it gains, the table sweeps the MALL); round 2: W = 14, 19 x 8192 entries (19.9 MiB), 1.77 ms; W = 11: 24 x 1024, 3 MiB, 2.31 ms.  Host builds: -DPLUME_COMB_W=14
#endif
#define PLUME_COMB_ENTRIES (1 << (PLUME_COMB_W - 1))
#define PLUME_COMB_WINDOW_WORDS (PLUME_COMB_ENTRIES * PLUME_TAB_ENTRY_WORDS)

// ------------------------------------------------------------------------------ window tables: P, theta P, 2P by ONE round of inversions (round 5)
// Rounds 2-4 built 1P..8P by affine chains in three levels (three inversion rounds, every level reading the rows of the one before: 2 KB of HBM traffic per table, the
// stage HBM-bound at 0.55 ms per job and 2^20 items) and, for small batches, by a Jacobian chain with one inversion.  The Eisenstein digits (above) need three rows, and
// both computed rows come straight from P:
//     2P:       slope 3 x^2 / (2 y)                          theta P = P - lambda P = (x, y) + (beta x, -y):   slope -2 y / ((beta - 1) x)
// so a job has two denominators, 2y and (beta - 1) x -- and a Jacobian base (H from hash_to_curve, 2^64 H of the signer) a third, Z -- whose PRODUCT D joins the lane's
// running product; one inversion per 8 lanes (k_tab_invert, Montgomery's trick twice over) returns 1/D, from which the job peels its own inverses with the denominators it
// recomputes from the base.  Pass A reads the base and parks one prefix product per job; pass B reads the base again, writes three rows.  ~0.7 KB of traffic and ~20-30
// multiplications per table instead of 2 KB and ~55.
// Denominators cannot vanish for a point of the group: y = 0 would be a point of order 2, x = 0 a point fixed by lambda, i.e. of order 3, and the order n is prime; whatever
// is not such a point (identity, failed validation) has been replaced by G before it gets here (tab_base).  Should a lane's product come out zero all the same (a test
// feeding unvalidated garbage), the pass is redone with zero denominators replaced by 1: the rows of that job are garbage, the other jobs of the lane are not.
PLUME_HD void pre_st(uint32_t* scr, size_t sstride, size_t slane, size_t q, const fe& a) {
    PLUME_UNROLL for (int i = 0; i < PLUME_FE_W; i++) scr[((q * PLUME_FE_W + (size_t)i) * sstride) + slane] = a.v[i];
}
PLUME_HD void pre_ld(fe& r, const uint32_t* scr, size_t sstride, size_t slane, size_t q) {
    PLUME_UNROLL for (int i = 0; i < PLUME_FE_W; i++) r.v[i] = scr[((q * PLUME_FE_W + (size_t)i) * sstride) + slane];
}
#define PLUME_TAB_SCR_WORDS PLUME_FE_WORDS                  // one parked prefix product per job
// 2P from affine P = (x, y) (tight) and l = 1 / (2y)
PLUME_HD void aff_dbl(fe& x3, fe& y3, const fe& x, const fe& y, const fe& l) {
    fe lam, t;
    fe_sqr3(t, x);                                              // 3x^2
    fe_mul(lam, t, l);
    fe_sqr_sub2<2>(x3, lam, x);                                 // lambda^2 - 2x   (round 3: the subtractions ride in the products' folds, no carry passes)
    fe_sub_lazy<2>(t, x, x3);
    fe_mul_sub<2>(y3, lam, t, y);                               // lambda (x - x3) - y
}
// theta P = (x, y) + (bx, -y) from affine P, bx = beta x and dinv = 1 / (bx - x)
PLUME_HD void aff_theta(fe& x3, fe& y3, const fe& x, const fe& y, const fe& bx, const fe& dinv) {
    fe lam, t, s, m2y;
    const fe z = fe_zero();
    fe_dbl_lazy(t, y); fe_sub_lazy<3>(m2y, z, t);               // (-y) - y = 3p - 2y, unreduced: a product operand next to a tight one
    fe_mul(lam, m2y, dinv);
    fe_add_lazy(s, x, bx);
    fe_sqr_sub<3>(x3, lam, s);                                  // lambda^2 - x - beta x
    fe_sub_lazy<2>(t, x, x3);
    fe_mul_sub<2>(y3, lam, t, y);                               // lambda (x - x3) - y
}
struct DirectRowSinkSync {                                       // host / single-lane builds: rows are stored by the lane that built them
    PLUME_HD void operator()(uint32_t* e, const fe& x, const fe& y, const fe& bx) const { st_tab_entry(e, x, y, bx); }
    PLUME_HD void sync() const {}
    PLUME_HD void inv(fe& r, const fe& a, int) const { fe_inv(r, a); }
};
PLUME_HD void guard_one(fe& d, bool guard) { if (guard) { if (fe_is_zero(d)) d = fe_small(1); } }
// park the running product, then acc *= D
PLUME_HD void tab_park(fe& acc, uint32_t* scr, size_t sstride, size_t slane, size_t q, const fe& D) {
    pre_st(scr, sstride, slane, q, acc);
    fe_mul(acc, acc, D);
}
// 1/D of the job whose product was parked at q; inv moves on to the jobs parked before it
PLUME_HD void tab_unpark(fe& Dinv, fe& inv, const uint32_t* scr, size_t sstride, size_t slane, size_t q, const fe& D) {
    fe pre;
    pre_ld(pre, scr, sstride, slane, q);
    fe_mul(Dinv, inv, pre);
    fe_mul(inv, inv, D);
}
// the base of a job as the builder sees it: affine (x, y) for Z = 1 bases, Jacobian otherwise; anything that is not a usable point becomes G.  The Z words of an
// affine base are not even loaded.
PLUME_HD bool tab_base(jac& b, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, size_t job) {
    bool zone = (jobflags[job] & PLUME_JOB_AFFINE) != 0;
    (void)njobs;
    ld_base(b, bases, job, !zone);
    b.inf = 0;
    if (job_state(jobflags[job]) != PLUME_JOB_OK) { b.x = fe_gx(); b.y = fe_gy(); b.z = fe_small(1); zone = true; }
    return zone;
}
PLUME_HD fe fe_beta_m1() { return fe_set(0x7AE96A2Bu, 0x657C0710u, 0x6E64479Eu, 0xAC3434E9u, 0x9CF04975u, 0x12F58995u, 0xC1396C28u, 0x719501EDu); }   // beta - 1
// the denominators of one job: dy = 2Y, dx = (beta - 1) X, dz = Z (Jacobian base) and their product D
PLUME_HD void tab_dens(fe& dy, fe& dx, fe& dz, fe& pyx, fe& D, const jac& b, bool zone, bool guard) {
    fe_dbl_lazy(dy, b.y); guard_one(dy, guard);
    fe_mul_k(dx, fe_beta_m1(), b.x); guard_one(dx, guard);
    fe_mul(pyx, dy, dx);
    if (zone) { dz = fe_small(1); D = pyx; } else { dz = b.z; guard_one(dz, guard); fe_mul(D, pyx, dz); }
}
// Pass A (jobs ascending): every job's D joins the lane's product -> carry.
PLUME_HD void tab_pass_a(const uint32_t* bases, const uint8_t* jobflags, size_t njobs, size_t j0, int cnt, uint32_t* scr, size_t sstride, size_t slane, fe& carry, bool& guard) {
    fe acc;
    guard = false;
    PLUME_NOUNROLL for (int pass = 0; pass < 2; pass++) {
        acc = fe_small(1);
        PLUME_NOUNROLL for (int jj = 0; jj < cnt; jj++) {
            jac b;
            const bool zone = tab_base(b, bases, jobflags, njobs, j0 + (size_t)jj);
            fe dy, dx, dz, pyx, D;
            tab_dens(dy, dx, dz, pyx, D, b, zone, guard);
            tab_park(acc, scr, sstride, slane, (size_t)jj, D);
        }
        if (guard || !fe_is_zero(acc)) break;
        guard = true;                                            // unreachable for points of prime order: redo with zero denominators replaced by 1
    }
    carry = acc;
}
// Pass B (jobs descending): carry = 1 / (the lane's product) in; rows P, theta P, 2P of every job out.
template <class RowSink>
PLUME_HD void tab_pass_b(uint32_t* tab, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, size_t j0, int cnt, const uint32_t* scr, size_t sstride, size_t slane,
                         const fe& carry, bool guard, const RowSink& sink) {
    const fe beta = fe_beta();
    constexpr size_t TW = PLUME_TAB_ENTRIES * PLUME_TAB_ENTRY_WORDS, EW = PLUME_TAB_ENTRY_WORDS;
    fe inv = carry;
    PLUME_NOUNROLL for (int jj = cnt - 1; jj >= 0; jj--) {
        const size_t job = j0 + (size_t)jj;
        jac b;
        const bool zone = tab_base(b, bases, jobflags, njobs, job);
        fe dy, dx, dz, pyx, D, Dinv, x, y, iy, ix;
        tab_dens(dy, dx, dz, pyx, D, b, zone, guard);
        tab_unpark(Dinv, inv, scr, sstride, slane, (size_t)jj, D);
        if (zone) {
            x = b.x; y = b.y;
            fe_mul(iy, Dinv, dx);                               // 1 / (2y)
            fe_mul(ix, Dinv, dy);                               // 1 / ((beta - 1) x)
        } else {
            // 1 / Z = Dinv (dy dx);  1 / (2Y) = Dinv dx Z;  1 / ((beta - 1) X) = Dinv dy Z;  affine x = X / Z^2, y = Y / Z^3, so
            // 1 / (2y) = Z^3 / (2Y) and 1 / ((beta - 1) x) = Z^2 / ((beta - 1) X)
            fe zi, zi2, z2, t;
            fe_mul(zi, Dinv, pyx);
            fe_sqr(zi2, zi);
            fe_mul(x, b.x, zi2);
            fe_mul(t, zi2, zi); fe_mul(y, b.y, t);
            fe_sqr(z2, dz);
            fe_mul(t, Dinv, dz);                                // 1 / (dy dx)
            fe_mul(iy, t, dx); fe_mul(iy, iy, z2); fe_mul(iy, iy, dz);      // Z^3 / (2Y)
            fe_mul(ix, t, dy); fe_mul(ix, ix, z2);                          // Z^2 / ((beta - 1) X)
        }
        uint32_t* rows = tab + job * TW;
        fe bx, x2, y2, b2;
        fe_mul_k(bx, beta, x);
        sink(rows, x, y, bx);                                   // row 0: P
        aff_theta(x2, y2, x, y, bx, ix);
        fe_mul_k(b2, beta, x2);
        sink(rows + EW, x2, y2, b2);                            // row 1: theta P = P - lambda P
        aff_dbl(x2, y2, x, y, iy);
        fe_mul_k(b2, beta, x2);
        sink(rows + 2 * EW, x2, y2, b2);                        // row 2: 2P
    }
}
// carry[.] <- 1 / carry[.] for nl lane products: thread t of T takes lanes t, t + T, ..., t + (K-1) T and spends ONE inversion on their product
template <int K>
PLUME_HD void tab_invert_group(uint32_t* carry, size_t nl, size_t T, size_t t) {
    fe v[K], pre[K], acc = fe_small(1), inv;
    PLUME_UNROLL for (int j = 0; j < K; j++) {
        const size_t l = t + (size_t)j * T;
        if (l < nl) ld_fe_soa(v[j], carry, nl, l); else v[j] = fe_small(1);
        pre[j] = acc;
        fe_mul(acc, acc, v[j]);
    }
    fe_inv(inv, acc);
    PLUME_UNROLL for (int j = K - 1; j >= 0; j--) {
        const size_t l = t + (size_t)j * T;
        fe o;
        fe_mul(o, inv, pre[j]);
        fe_mul(inv, inv, v[j]);
        if (l < nl) st_fe_soa(carry, nl, l, o);
    }
}
// Both passes in one function with the inversion in place: single-lane builds (the host harness holds the pass sequence to it).
template <class RowSink = DirectRowSinkSync>
PLUME_HD void table_build(uint32_t* tab, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, size_t j0, int cnt, uint32_t* scr, size_t sstride, size_t slane,
                          const RowSink& sink = RowSink()) {
    fe carry, inv;
    bool guard;
    tab_pass_a(bases, jobflags, njobs, j0, cnt, scr, sstride, slane, carry, guard);
    sink.inv(inv, carry, 1);
    tab_pass_b(tab, bases, jobflags, njobs, j0, cnt, scr, sstride, slane, inv, guard, sink);
}

// ------------------------------------------------------------------------------ the generator's fixed tables, one entry per lane (round 3)
// Rounds 1-2 built the verifier's window table of G and the signer's comb with the chained builder above, ONE lane per table: 175.6 + 56.7 ms per context once the
// tables had grown to 32768 and 19 x 8192 entries.  Every entry is independent: lane (w, e) computes (e + 1) * B_w by MSB-first double-and-add
// (<= 15 doublings + 15 mixed additions for a multiplier below 2^16) from the affine window base B_w = 2^(W w) * G and pays its own inversion -- ~45 k instructions per
// lane, all lanes in parallel.  The additions are the checked ones: a multiplier k < 2^16 never meets k' P = +-P on the way (the order is ~2^256), the checks are there
// because nothing here is hot.
PLUME_HD void fixed_table_entry(uint32_t* row, const fe& px, const fe& py, uint32_t k /* 1 <= k < 2^31 */) {
    jac acc; acc.x = px; acc.y = py; acc.z = fe_small(1); acc.inf = 0;
    int top = 30;
    while (top > 0 && !((k >> top) & 1u)) top--;
    PLUME_NOUNROLL for (int b = top - 1; b >= 0; b--) {
        jac_dbl(acc);
        if ((k >> b) & 1u) jac_madd<true>(acc, px, py);
    }
    fe zi, zi2, x, y, bx;
    fe_inv(zi, acc.z); fe_sqr(zi2, zi);
    fe_mul(x, acc.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(y, acc.y, zi2);
    fe_mul_k(bx, fe_beta(), x);
    st_tab_entry(row, x, y, bx);
}
// B = 2^doublings * G in affine form: out18 = x | y (limbs)
PLUME_HD void fixed_window_base(uint32_t* out18, uint32_t doublings) {
    jac g; g.x = fe_gx(); g.y = fe_gy(); g.z = fe_small(1); g.inf = 0;
    PLUME_NOUNROLL for (uint32_t d = 0; d < doublings; d++) jac_dbl(g);
    fe zi, zi2, x, y;
    fe_inv(zi, g.z); fe_sqr(zi2, zi);
    fe_mul(x, g.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(y, g.y, zi2);
    st_fe(out18, x); st_fe(out18 + PLUME_FE_W, y);
}
// entry e of window w of a fixed table with `entries` rows per window: (e + 1) * 2^(W w) * G
PLUME_HD void fixed_table_lane(uint32_t* rows, const uint32_t* base18, uint32_t entries, size_t lane) {
    const size_t w = lane / entries, e = lane % entries;
    fe px, py;
    ld_fe(px, base18 + w * 2 * PLUME_FE_W); ld_fe(py, base18 + w * 2 * PLUME_FE_W + PLUME_FE_W);
    fixed_table_entry(rows + lane * PLUME_TAB_ENTRY_WORDS, px, py, (uint32_t)e + 1u);
}

// ------------------------------------------------------------------------------------ batched affine conversion
// Jacobian -> affine for the points of a SoA array, PLUME_NORM_K points per lane sharing ONE field inversion
// (Montgomery's trick, prefix products kept in registers).  Lane `lane` of `nlanes` handles points lane + j*nlanes, so
// every load/store is coalesced across the wavefront.  X and Y are overwritten with the affine coordinates (Z is left
// as it was and must no longer be used); points flagged infinite are skipped.
#define PLUME_NORM_K 8
PLUME_HD void normalize_points(uint32_t* pts, const uint8_t* inf, size_t npts, size_t lane, size_t nlanes) {
    fe z[PLUME_NORM_K], pre[PLUME_NORM_K];
    fe acc = fe_small(1);
    PLUME_UNROLL for (int j = 0; j < PLUME_NORM_K; j++) {
        const size_t idx = lane + (size_t)j * nlanes;
        const bool live = idx < npts && !inf[idx < npts ? idx : 0];
        if (live) ld_fe_soa(z[j], pts + 2 * PLUME_FE_W * npts, npts, idx); else z[j] = fe_small(1);
        pre[j] = acc;
        fe_mul(acc, acc, z[j]);
    }
    fe inv;
    fe_inv(inv, acc);
    PLUME_UNROLL for (int j = PLUME_NORM_K - 1; j >= 0; j--) {
        const size_t idx = lane + (size_t)j * nlanes;
        const bool live = idx < npts && !inf[idx < npts ? idx : 0];
        fe zi, zi2, x, y;
        fe_mul(zi, inv, pre[j]);
        fe_mul(inv, inv, z[j]);
        if (live) {
            ld_fe_soa(x, pts, npts, idx); ld_fe_soa(y, pts + PLUME_FE_W * npts, npts, idx);
            fe_sqr(zi2, zi);
            fe_mul(x, x, zi2);
            fe_mul(zi2, zi2, zi); fe_mul(y, y, zi2);
            st_fe_soa(pts, npts, idx, x); st_fe_soa(pts + PLUME_FE_W * npts, npts, idx, y);
        }
    }
}

// ------------------------------------------------------------------------------- fixed-base comb (generator only)
// k*G with NO doublings: k = sum d_i 2^(W i) (Booth, d_i in [-2^(W-1), 2^(W-1)], i = 0..NW-1) and a precomputed table
// comb[i][e] = (e+1) * 2^(W i) * G.  W = 14: 19 windows x 8192 entries (19.9 MiB, read through L2 / MALL), 19 mixed additions per multiplication; W = 11: 24 windows x 1024 entries (3 MiB), 24 mixed additions per
// multiplication (W = 8: 33 windows x 128 entries, 33 additions); used by the signer's pk = sk*G and R = r*G
// (rust-k256/src/randomizedsigner.rs:51,53).
#define PLUME_COMB_WINDOWS ((256 + PLUME_COMB_W) / PLUME_COMB_W)     // windows covering 257 bits
#define PLUME_COMB_WORDS (PLUME_COMB_WINDOWS * PLUME_COMB_WINDOW_WORDS)
template <int W>
PLUME_HD int booth_digit_sc(const uint32_t m[8], int k) {   // k is a runtime loop index here (no unrolling)
    const int lo = W * k - 1;
    const uint32_t mask = (1u << (W + 1)) - 1u;
    uint32_t u;
    if (lo < 0) {
        u = (m[0] << 1) & mask;
    } else {
        const uint32_t wi = (uint32_t)lo >> 5, sh = (uint32_t)lo & 31;
        uint32_t a = 0, b = 0;
        PLUME_UNROLL for (int i = 0; i < 8; i++) { a = (wi == (uint32_t)i) ? m[i] : a; b = (wi + 1 == (uint32_t)i) ? m[i] : b; }
        u = ((a >> sh) | (sh > (uint32_t)(31 - W) ? (b << (32 - sh)) : 0u)) & mask;
    }
    return (int)(u & 1) + (int)((u >> 1) & ((1u << (W - 1)) - 1u)) - (int)((u >> W) << (W - 1));
}
PLUME_HD int booth_digit_comb(const uint32_t m[8], int k) { return booth_digit_sc<PLUME_COMB_W>(m, k); }
template <bool CHECKED>
PLUME_HD void comb_mul_g_impl(jac& acc, const sc& k, const uint32_t* comb) {
    acc.x = fe_small(1); acc.y = fe_small(1); acc.z = fe_small(0); acc.inf = 1;
    PLUME_NOUNROLL for (int i = 0; i < PLUME_COMB_WINDOWS; i++) {
        const int d = booth_digit_comb(k.v, i);
        if (d != 0) {
            const int ad = d < 0 ? -d : d;
            const uint32_t* e = comb + ((size_t)i * PLUME_COMB_ENTRIES + (size_t)(ad - 1)) * PLUME_TAB_ENTRY_WORDS;
            fe qx, qy;
            ld_tab_xy(qx, qy, e, false);
            if (d < 0) fe_neg_lazy(qy, qy);
            jac_madd<CHECKED>(acc, qx, qy);
        }
    }
}
// acc += k G by the comb (the verifier's short first equation: the generator's term joins an accumulator that already holds the chain over pk and R)
template <bool CHECKED>
PLUME_HD void comb_add_g(jac& acc, const sc& k, const uint32_t* comb) {
    PLUME_NOUNROLL for (int i = 0; i < PLUME_COMB_WINDOWS; i++) {
        const int d = booth_digit_comb(k.v, i);
        if (d != 0) {
            const int ad = d < 0 ? -d : d;
            const uint32_t* e = comb + ((size_t)i * PLUME_COMB_ENTRIES + (size_t)(ad - 1)) * PLUME_TAB_ENTRY_WORDS;
            fe qx, qy;
            ld_tab_xy(qx, qy, e, false);
            if (d < 0) fe_neg_lazy(qy, qy);
            jac_madd<CHECKED>(acc, qx, qy);
        }
    }
}
// the comb with the uniform schedule: every window adds (a zero digit adds row 1 to a copy that is dropped), the accumulator starts at the offset point B and loses it at the end
template <bool CHECKED>
PLUME_HD void comb_mul_g_uniform_impl(jac& acc, const sc& k, const uint32_t* comb) {
    acc.x = fe_off_x(); acc.y = fe_off_y(); acc.z = fe_small(1); acc.inf = 0;
    PLUME_NOUNROLL for (int i = 0; i < PLUME_COMB_WINDOWS; i++) {
        const int d = booth_digit_comb(k.v, i);
        const int ad = (d < 0 ? -d : d) + (d == 0 ? 1 : 0);
        const uint32_t* e = comb + ((size_t)i * PLUME_COMB_ENTRIES + (size_t)(ad - 1)) * PLUME_TAB_ENTRY_WORDS;
        fe qx, qy;
        ld_tab_xy(qx, qy, e, false);
        jac_madd_uniform<CHECKED>(acc, qx, qy, d);
    }
}
PLUME_HD void comb_mul_g_uniform(jac& acc, const sc& k, const uint32_t* comb) {
    comb_mul_g_uniform_impl<false>(acc, k, comb);
    if (fe_is_zero(acc.z)) {                                    // met p == +-q (probability ~2^-250 for honest keys; crafted tiny keys cannot reach the offset point either)
        PLUME_COUNT_FALLBACK();
        comb_mul_g_uniform_impl<true>(acc, k, comb);
    }
    jac_madd<true>(acc, fe_off_x(), fe_off_neg_y());            // - B
}
// Level 2 of the uniform schedule: no digit-dependent ADDRESS either.  The big comb cannot be scanned (2^17 rows per window), so this form has its own small table:
// gscan[i][e] = (e + 1) * 2^(W i) * G, W = 5: 52 windows x 16 rows (104 KiB; every lane reads the same addresses, so the rows come through the scalar cache), 52 uniform
// additions, each after a scan of its window's 16 rows.  2^20 signatures, both multiplications: W = 4 (65 x 8) 7.34 ms, W = 5 6.98 ms, W = 6 (43 x 32) 7.42 ms
// (tests/gpu_debug/gscan_width.py; the comb's 15 additions at level 1: 1.49 ms).
#define PLUME_GSCAN_W 5
#define PLUME_GSCAN_ENTRIES (1 << (PLUME_GSCAN_W - 1))
#define PLUME_GSCAN_WINDOWS ((256 + PLUME_GSCAN_W) / PLUME_GSCAN_W)
#define PLUME_GSCAN_WORDS (PLUME_GSCAN_WINDOWS * PLUME_GSCAN_ENTRIES * PLUME_TAB_ENTRY_WORDS)
template <bool CHECKED>
PLUME_HD void comb_mul_g_scan_impl(jac& acc, const sc& k, const uint32_t* gscan) {
    acc.x = fe_off_x(); acc.y = fe_off_y(); acc.z = fe_small(1); acc.inf = 0;
    PLUME_NOUNROLL for (int i = 0; i < PLUME_GSCAN_WINDOWS; i++) {
        const int d = booth_digit_sc<PLUME_GSCAN_W>(k.v, i);
        const int ad = (d < 0 ? -d : d) + (d == 0 ? 1 : 0);
        fe qx, qy;
        ld_tab_xy_scan<PLUME_GSCAN_ENTRIES>(qx, qy, gscan + (size_t)i * PLUME_GSCAN_ENTRIES * PLUME_TAB_ENTRY_WORDS, ad, false);
        jac_madd_uniform<CHECKED>(acc, qx, qy, d);
    }
}
PLUME_HD void comb_mul_g_scan(jac& acc, const sc& k, const uint32_t* gscan) {
    comb_mul_g_scan_impl<false>(acc, k, gscan);
    if (fe_is_zero(acc.z)) {
        PLUME_COUNT_FALLBACK();
        comb_mul_g_scan_impl<true>(acc, k, gscan);
    }
    jac_madd<true>(acc, fe_off_x(), fe_off_neg_y());            // - B
}
// unchecked additions first; a lane that met p == +-q (Z = 0 mod p, see jac_madd) is recomputed with the checked form
PLUME_HD void comb_mul_g(jac& acc, const sc& k, const uint32_t* comb) {
    comb_mul_g_impl<false>(acc, k, comb);
    if (!acc.inf && fe_is_zero(acc.z)) {
        PLUME_COUNT_FALLBACK();
        comb_mul_g_impl<true>(acc, k, comb);
    }
}

// --------------------------------------------------------------------------------------- multi-scalar loop
// acc = sum over JOINT slots of (digit of the slot at position p) 4^p.  A joint slot is a window table (P, theta P, 2P) with the Eisenstein digits of one GLV pair
// (eisd_store): one byte per position, at most one addition per position.  Two joint slots: tab0 with rows [0, NP) of the lane's digit area, tab1 with rows [NP, 2 NP).
// wide0 (the verifier's first equation in its long form): tab0 is the generator's WIDE table and rows [0, 2 PLUME_NDIG) hold s in its 24-bit Booth digits, one for each
// half, on the grid of 4-bit windows the rounds before used (booth_store_wide) -- digit k sits at window 6k, i.e. position 12k; tab1's rows follow at 2 PLUME_NDIG.
// A NULL table (a job flagged INF) contributes nothing.
// the wide digit of generator slot s (0: G, 1: lambda G) at 4-bit window w; 0 = nothing to add
PLUME_HD int msm_wide_digit(const int8_t* dig, uint32_t stride, int w, int s) {
    const int i = w;
    int mag = dig[(uint32_t)(s * PLUME_NDIG + i) * stride] & 0xFF;
    bool dn;
    if (i + PLUME_GW_BYTES < PLUME_NDIG) {
        PLUME_UNROLL for (int b = 1; b < PLUME_GW_BYTES; b++) mag |= (dig[(uint32_t)(s * PLUME_NDIG + i + b) * stride] & 0xFF) << (8 * b);
        dn = dig[(uint32_t)(s * PLUME_NDIG + i + PLUME_GW_BYTES) * stride] != 0;
    } else if (i + 1 < PLUME_NDIG) {
        const int hi = dig[(uint32_t)(s * PLUME_NDIG + i + 1) * stride];
        mag |= (hi & 0x7F) << 8; dn = (hi & 0x80) != 0;
    } else { dn = (mag & 0x40) != 0; mag &= 0x3F; }
    return dn ? -mag : mag;
}
// the row of a digit code (1..18): (-1)^neg w^j (row point) = (x | beta x | beta^2 x, +-y).  beta^2 x = -(x + beta x) is formed here, unreduced (3p - x - beta x: limbs
// below 2^31, limb 8 below 2^26 -- a product operand next to a tight one, which is all jac_madd does with it); the three candidates are selected without a branch (every
// wavefront holds all three kinds).  Seven of a row's eight 16-byte quads are read, from one cache line.
PLUME_HD void ld_tab_unit(fe& qx, fe& qy, const uint32_t* tab, int code) {
    const uint32_t c = (uint32_t)(code - 1), row = c / 6u, u = c - 6u * row, j = u >> 1;
    const uint32_t* e = tab + row * PLUME_TAB_ENTRY_WORDS;
    fe x, bx, s, t;
    PLUME_UNROLL for (int i = 0; i < 8; i++) { x.v[i] = e[i]; qy.v[i] = e[8 + i]; bx.v[i] = e[16 + i]; }
    x.v[8] = e[24]; qy.v[8] = e[25]; bx.v[8] = e[26];
    fe_add_lazy(s, x, bx);
    const fe z = fe_zero();
    fe_sub_lazy<3>(t, z, s);
    qx = x;
    fe_cmov(qx, bx, j == 1u);
    fe_cmov(qx, t, j == 2u);
    if (u & 1u) fe_neg_lazy(qy, qy);
}
// wave-uniform "every accumulator is still the identity": the doublings of leading all-zero positions are skipped
PLUME_HD bool msm_all_inf(const jac& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __all(acc.inf != 0) != 0;
#else
    return acc.inf != 0;
#endif
}
template <bool CHECKED, int NP = PLUME_NPOS>
PLUME_HD void msm_run_impl(jac& acc, const uint32_t* tab0, const uint32_t* tab1, const int8_t* dig, uint32_t stride, bool wide0) {
    acc.x = fe_small(1); acc.y = fe_small(1); acc.z = fe_small(0); acc.inf = 1;
    const uint32_t r1 = wide0 ? 2u * PLUME_NDIG : (uint32_t)NP;        // first digit row of tab1's joint slot
    PLUME_NOUNROLL for (int p = NP - 1; p >= 0; p--) {
        if (p != NP - 1 && !msm_all_inf(acc)) { jac_dbl_neg(acc); jac_dbl_neg(acc); }      // (an even number of sign-flipping doublings)
        if (wide0) {
            if (p % (2 * PLUME_GWS) == 0 && p / 2 < PLUME_NDIG) {    // wave-uniform: a wide digit sits at every 6th 4-bit window only
                PLUME_NOUNROLL for (int s = 0; s < 2; s++) {
                    const int d = tab0 ? msm_wide_digit(dig, stride, p / 2, s) : 0;
                    if (d != 0) {
                        fe qx, qy;
                        ld_tab_xy(qx, qy, tab0 + ((d < 0 ? -d : d) - 1) * PLUME_TAB_ENTRY_WORDS, s != 0);
                        if (d < 0) fe_neg_lazy(qy, qy);
                        jac_madd<CHECKED>(acc, qx, qy);
                    }
                }
            }
        } else {
            const int c = tab0 ? dig[(uint32_t)p * stride] : 0;
            if (c != 0) { fe qx, qy; ld_tab_unit(qx, qy, tab0, c); jac_madd<CHECKED>(acc, qx, qy); }
        }
        const int c = tab1 ? dig[(r1 + (uint32_t)p) * stride] : 0;
        if (c != 0) { fe qx, qy; ld_tab_unit(qx, qy, tab1, c); jac_madd<CHECKED>(acc, qx, qy); }
    }
}
// One joint-slot addition of the UNIFORM schedule (the signer): the same instructions whatever the digit code is.  A zero code adds code 1's point to a copy that a masked
// select drops; unit and sign are masked selects (ld_tab_unit's own, plus the sign here).  Level 1: the row's ADDRESS still depends on the code; SCAN (level 2): all three
// rows are read and one is kept by masked selects.
template <bool CHECKED, bool SCAN>
PLUME_HD void msm_add_uniform(jac& acc, const uint32_t* tab, int code) {
    const bool zero = code == 0;
    const uint32_t c = zero ? 0u : (uint32_t)(code - 1), row = c / 6u, u = c - 6u * row, j = u >> 1;
    fe x, y, bx;
    if (SCAN) {
        PLUME_UNROLL for (int i = 0; i < 8; i++) { x.v[i] = tab[i]; y.v[i] = tab[8 + i]; bx.v[i] = tab[16 + i]; }
        x.v[8] = tab[24]; y.v[8] = tab[25]; bx.v[8] = tab[26];
        PLUME_NOUNROLL for (uint32_t r = 1; r < PLUME_TAB_ENTRIES; r++) {       // (rolled: one row's 27 words in flight at a time -- unrolled, the three rows' loads were hoisted together and spilled)
            const uint32_t* e = tab + r * PLUME_TAB_ENTRY_WORDS;
            fe rx, ry, rb;
            PLUME_UNROLL for (int i = 0; i < 8; i++) { rx.v[i] = e[i]; ry.v[i] = e[8 + i]; rb.v[i] = e[16 + i]; }
            rx.v[8] = e[24]; ry.v[8] = e[25]; rb.v[8] = e[26];
            fe_cmov(x, rx, row == r); fe_cmov(y, ry, row == r); fe_cmov(bx, rb, row == r);
        }
    } else {
        const uint32_t* e = tab + row * PLUME_TAB_ENTRY_WORDS;
        PLUME_UNROLL for (int i = 0; i < 8; i++) { x.v[i] = e[i]; y.v[i] = e[8 + i]; bx.v[i] = e[16 + i]; }
        x.v[8] = e[24]; y.v[8] = e[25]; bx.v[8] = e[26];
    }
    fe s, t, qx;
    fe_add_lazy(s, x, bx);
    const fe z = fe_zero();
    fe_sub_lazy<3>(t, z, s);
    qx = x;
    fe_cmov(qx, bx, j == 1u);
    fe_cmov(qx, t, j == 2u);
    jac_madd_uniform<CHECKED>(acc, qx, y, zero ? 0 : ((u & 1u) ? -1 : 1));
}
// The signer's chains: K joint slots -- slot j's table at tab + j * tstride words, its digit rows [j NP, (j + 1) NP) -- along NP positions of two doublings.  One addition body
// in the loop (the slots are walked by a rolled inner loop).
template <bool CHECKED, int NP, int K>
PLUME_HD void msm_runk_impl(jac& acc, const uint32_t* tab, size_t tstride, const int8_t* dig, uint32_t stride) {
    acc.x = fe_small(1); acc.y = fe_small(1); acc.z = fe_small(0); acc.inf = 1;
    PLUME_NOUNROLL for (int p = NP - 1; p >= 0; p--) {
        if (p != NP - 1 && !msm_all_inf(acc)) { jac_dbl_neg(acc); jac_dbl_neg(acc); }
        PLUME_NOUNROLL for (uint32_t j = 0; j < (uint32_t)K; j++) {
            const int c = tab ? dig[(j * (uint32_t)NP + (uint32_t)p) * stride] : 0;
            if (c != 0) { fe qx, qy; ld_tab_unit(qx, qy, tab + j * tstride, c); jac_madd<CHECKED>(acc, qx, qy); }
        }
    }
}
template <int NP, int K>
PLUME_HD void msm_runk(jac& acc, const uint32_t* tab, size_t tstride, const int8_t* dig, uint32_t stride) {
    msm_runk_impl<false, NP, K>(acc, tab, tstride, dig, stride);
    if (!acc.inf && fe_is_zero(acc.z)) {                        // met p == +-q inside an unchecked addition (Z = 0 mod p forever after, see jac_madd): redo that lane
        PLUME_COUNT_FALLBACK();
        msm_runk_impl<true, NP, K>(acc, tab, tstride, dig, stride);
    }
}
// The same chain with the UNIFORM schedule (the signer, levels 1 and 2): no position is skipped and no branch depends on a digit.  `live` false (the table's base was the
// identity / invalid: a public fact) turns every digit into 0.  Starts at the offset point B; 4^(NP - 1) B comes off again at the end.
template <bool CHECKED, int NP, bool SCAN, int K>
PLUME_HD void msm_run_uniform_impl(jac& acc, const uint32_t* tab, size_t tstride, bool live, const int8_t* dig, uint32_t stride) {
    acc.x = fe_off_x(); acc.y = fe_off_y(); acc.z = fe_small(1); acc.inf = 0;
    PLUME_NOUNROLL for (int p = NP - 1; p >= 0; p--) {
        if (p != NP - 1) { jac_dbl_neg(acc); jac_dbl_neg(acc); }
        PLUME_NOUNROLL for (uint32_t j = 0; j < (uint32_t)K; j++) {
            const int c = live ? dig[(j * (uint32_t)NP + (uint32_t)p) * stride] : 0;
            msm_add_uniform<CHECKED, SCAN>(acc, tab + j * tstride, c);
        }
    }
}
static_assert(2 * (PLUME_NPOS - 1) == 128 && 2 * (PLUME_NPOS64 - 1) == 64 && (PLUME_NPOSK == PLUME_NPOS64 || 2 * (PLUME_NPOSK - 1) == 32), "the uniform chains' offset constants are -(2^128 B), -(2^64 B), -(2^32 B)");
template <int NP, bool SCAN, int K>
PLUME_HD void msm_run_uniform(jac& acc, const uint32_t* tab, size_t tstride, bool live, const int8_t* dig, uint32_t stride) {
    msm_run_uniform_impl<false, NP, SCAN, K>(acc, tab, tstride, live, dig, stride);
    if (fe_is_zero(acc.z)) {
        PLUME_COUNT_FALLBACK();
        msm_run_uniform_impl<true, NP, SCAN, K>(acc, tab, tstride, live, dig, stride);
    }
    const fe cx = NP == 17 ? fe_off_c32_x() : NP == PLUME_NPOS64 ? fe_off_c64_x() : fe_off_c128_x(), cy = NP == 17 ? fe_off_c32_y() : NP == PLUME_NPOS64 ? fe_off_c64_y() : fe_off_c128_y();     // - 4^(NP - 1) B
    if (!acc.inf) jac_madd<true>(acc, cx, cy);
    else { acc.x = cx; acc.y = cy; acc.z = fe_small(1); acc.inf = 0; }              // (acc.inf: only after a checked redo that hit the identity)
}
// the same chain with the checked additions only (the redo kernel of the verifier: tasks whose unchecked chain met p == +-q)
PLUME_HD void msm_run_checked(jac& acc, const uint32_t* tab0, const uint32_t* tab1, const int8_t* dig, uint32_t stride, bool wide0 = false) {
    PLUME_COUNT_FALLBACK();
    msm_run_impl<true>(acc, tab0, tab1, dig, stride, wide0);
}
// the unchecked chain; returns false when the accumulator met p == +-q on the way (Z = 0 mod p: the result is garbage and the task has to be redone with checked additions)
PLUME_HD bool msm_run_unchecked(jac& acc, const uint32_t* tab0, const uint32_t* tab1, const int8_t* dig, uint32_t stride, bool wide0 = false) {
    msm_run_impl<false>(acc, tab0, tab1, dig, stride, wide0);
    return acc.inf || !fe_is_zero(acc.z);
}
}  // namespace plume
