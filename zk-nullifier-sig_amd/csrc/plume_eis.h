// Short scalars for the verifier's FIRST equation when R is given (V1 verify, verify_non_zk): a half-GCD in the Eisenstein integers.
//
// rust-k256/src/lib.rs:101,115-121 computes R' = s G - c pk and compares it with the given r_point.  That is an identity check, s G - c pk - R = O, and an identity check
// may be multiplied by any tau != 0 (the group has prime order n):   (tau s) G - (tau c) pk - tau R = O.
// Z[w] (w^2 + w + 1 = 0) maps onto Z/n by w -> lambda, with kernel (pi), pi = a1 + b1 w, N(pi) = a1^2 - a1 b1 + b1^2 = n (the GLV lattice's first basis vector).  With
// gamma = c1 + c2 w the GLV split of c, the Euclidean algorithm on (pi, gamma) in Z[w], stopped at the first remainder of norm < 2^128, leaves
//     tau gamma = upsilon (mod pi),   tau = t0 + t1 w,  upsilon = u0 + u1 w,   all four coefficients of about 64 bits
// (Antipa et al.'s accelerated verification, in the endomorphism ring: the four-dimensional form).  So
//     k G - upsilon pk - tau R = O,   k = tau s mod n,
// where pk and R carry 64-bit coefficients on (P, lambda P): a chain of 64 doublings instead of 128, and k G costs fifteen additions from the doubling-free comb the signer
// already uses.  The multi-scalar kernel evaluates  k G - upsilon pk - (tau - 1) R  and the finalize stage compares it with R exactly as before -- a valid signature's
// accumulator ends at R, not at the identity, so the hot loop's unchecked additions meet no exceptional case on honest inputs.  (One valid input does end at the identity: the
// signature with nonce r = 0, R = Hr = identity, which the reference accepts.  R's table is then absent, k G - upsilon pk closes at the identity in the comb's last addition, the
// task is filed and the redo launch's checked chain returns the identity: tests/golden edge cases "r=0", tests/test_gpu_round6.py.)
//
// Quotients are ESTIMATED in double precision (any quotient gives a unimodular step, so the relation above holds whatever the estimates are; good estimates make the
// coefficients short).  Measured over 10^5 random and 10^3 crafted c: coefficients <= 65 bits, 39 steps on average, 51 at most (tests/test_devsim.py).  The caller checks
// the bound it needs (< 2^66: thirty-four positions of base-4 Eisenstein digits) and falls back to the long form of the equation when it fails -- no input is known that does.
#pragma once
#include "plume_ec.h"

namespace plume {

// Numbers here are SIGNED and kept as 29-bit limbs in int32 words (the layout the field code uses, for the same reason: a column of 32 x 32-bit products accumulates in
// 64 bits with no carry steps): value = sum l_i 2^(29 i), lower limbs in [0, 2^29), the top limb carries the sign.  Remainders: five limbs (145 bits; they start below 2^129),
// cofactors: three (87 bits; they end below 2^67).
constexpr int kEisR = 5, kEisT = 3;
constexpr int32_t kEisMask = (1 << 29) - 1;

PLUME_HD double eis_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <int N>
PLUME_HD double eis_to_double(const int32_t (&l)[N]) {
    double d = (double)l[N - 1];
    PLUME_UNROLL for (int i = N - 2; i >= 0; i--) d = eis_fma(d, 536870912.0, (double)l[i]);
    return d;
}
// x -= q y in Z[w] for q = q0 + q1 w with |q0|, |q1| < 2^31:   q y = (q0 y0 - q1 y1) + (q0 y1 + q1 (y0 - y1)) w      (w^2 = -1 - w)
// One pass per coordinate: two products per limb into a 64-bit column, the carry travels up by an arithmetic shift.
template <int N>
PLUME_HD void eis_reduce(int32_t (&x0)[N], int32_t (&x1)[N], const int32_t (&y0)[N], const int32_t (&y1)[N], int32_t q0, int32_t q1) {
    int64_t c0 = 0, c1 = 0;
    PLUME_UNROLL for (int i = 0; i < N; i++) {
        const int32_t z = y0[i] - y1[i];
        c0 += (int64_t)x0[i] - (int64_t)q0 * y0[i] + (int64_t)q1 * y1[i];
        c1 += (int64_t)x1[i] - (int64_t)q0 * y1[i] - (int64_t)q1 * z;
        if (i < N - 1) { x0[i] = (int32_t)c0 & kEisMask; c0 >>= 29; x1[i] = (int32_t)c1 & kEisMask; c1 >>= 29; }
        else { x0[i] = (int32_t)c0; x1[i] = (int32_t)c1; }
    }
}
// the same with the quotient shifted up by k limbs (crafted challenges only: an honest quotient is a few bits long): x -= (q 2^(29 k)) y.  Out of line, operands by value:
// the hot loop's arrays never have their address taken (they stay in registers).
template <int N> struct eis_pair { int32_t c0[N], c1[N]; };
template <int N>
PLUME_HD_NOINLINE eis_pair<N> eis_reduce_shifted(eis_pair<N> x, const eis_pair<N> y, int32_t q0, int32_t q1, int k) {
    // y 2^(29 k) has to fit the N limbs, so y itself fits N - k of them and the limbs above are its sign extension: limb N - 1 - k is read as the (signed) top limb
    const int32_t s0 = y.c0[N - 1] < 0 ? (1 << 29) : 0, s1 = y.c1[N - 1] < 0 ? (1 << 29) : 0;
    int64_t c0 = 0, c1 = 0;
    for (int i = 0; i < N; i++) {
        const int j = i - k;
        int32_t a = j >= 0 ? y.c0[j] : 0, b = j >= 0 ? y.c1[j] : 0;
        if (j == N - 1 - k && k > 0) { a -= s0; b -= s1; }
        c0 += (int64_t)x.c0[i] - (int64_t)q0 * a + (int64_t)q1 * b;
        c1 += (int64_t)x.c1[i] - (int64_t)q0 * b - (int64_t)q1 * (a - b);
        if (i < N - 1) { x.c0[i] = (int32_t)c0 & kEisMask; c0 >>= 29; x.c1[i] = (int32_t)c1 & kEisMask; c1 >>= 29; }
        else { x.c0[i] = (int32_t)c0; x.c1[i] = (int32_t)c1; }
    }
    return x;
}
// the quotient estimate round(x / y) for x = (x0, x1), y = (y0, y1) given in double precision; k = how many 29-bit limbs the (31-bit) estimate is shifted by
PLUME_HD void eis_quotient(int32_t& q0, int32_t& q1, int& k, double x0, double x1, double y0, double y1) {
    const double nb = eis_fma(y0, y0, eis_fma(-y0, y1, y1 * y1));            // N(y) > 0
    const double inv = 1.0 / nb;
    const double cb0 = y0 - y1, cb1 = -y1;                                   // conj(y) = (y0 - y1) - y1 w
    const double n0 = eis_fma(x0, cb0, -(x1 * cb1));
    const double n1 = eis_fma(x0, cb1, eis_fma(x1, cb0, -(x1 * cb1)));
    double r0 = n0 * inv, r1 = n1 * inv;
    const double a0 = r0 < 0 ? -r0 : r0, a1 = r1 < 0 ? -r1 : r1, m = a0 > a1 ? a0 : a1;
    k = 0;
    const double lim = 1073741824.0, dn = 1.0 / 536870912.0;                 // 2^30, 2^-29
    if (m >= lim) {                                                          // (a quotient of 2^30 and more: never on honest input)
        double mm = m;
        PLUME_NOUNROLL while (mm >= lim && k < 4) { mm *= dn; r0 *= dn; r1 *= dn; k++; }
        r0 = r0 > lim ? lim : (r0 < -lim ? -lim : r0);
        r1 = r1 > lim ? lim : (r1 < -lim ? -lim : r1);
    }
    q0 = (int32_t)__builtin_rint(r0);
    q1 = (int32_t)__builtin_rint(r1);
}
PLUME_HD double eis_norm(double x0, double x1) { return eis_fma(x0, x0, eis_fma(-x0, x1, x1 * x1)); }
template <int N>
PLUME_HD void eis_step(int32_t (&x0)[N], int32_t (&x1)[N], const int32_t (&y0)[N], const int32_t (&y1)[N], int32_t q0, int32_t q1, int k) {
    if (k == 0) { eis_reduce(x0, x1, y0, y1, q0, q1); return; }
    eis_pair<N> x, y;
    PLUME_UNROLL for (int i = 0; i < N; i++) { x.c0[i] = x0[i]; x.c1[i] = x1[i]; y.c0[i] = y0[i]; y.c1[i] = y1[i]; }
    x = eis_reduce_shifted<N>(x, y, q0, q1, k);
    PLUME_UNROLL for (int i = 0; i < N; i++) { x0[i] = x.c0[i]; x1[i] = x.c1[i]; }
}

struct eis_short {
    uint32_t t[2][3], u[2][3];      // magnitudes of tau - 1 = (t0 - 1) + t1 w and of upsilon = u0 + u1 w (below 2^66 when ok)
    uint32_t tneg[2], uneg[2];      // their signs
    sc tau;                         // tau = t0 + t1 lambda mod n
    bool ok;                        // every coefficient is below 2^66 in magnitude
};
// a GLV half (magnitude of at most 129 bits in four words... the splits stay below 2^128) as signed limbs
template <int N>
PLUME_HD void eis_from_half(int32_t (&l)[N], const glv_half& h) {
    const uint32_t* m = h.m;
    int32_t v[5];
    v[0] = (int32_t)(m[0] & (uint32_t)kEisMask);
    v[1] = (int32_t)(((m[0] >> 29) | (m[1] << 3)) & (uint32_t)kEisMask);
    v[2] = (int32_t)(((m[1] >> 26) | (m[2] << 6)) & (uint32_t)kEisMask);
    v[3] = (int32_t)(((m[2] >> 23) | (m[3] << 9)) & (uint32_t)kEisMask);
    v[4] = (int32_t)(m[3] >> 20);
    int32_t cy = 0;
    PLUME_UNROLL for (int i = 0; i < N; i++) {
        const int32_t t = (h.neg ? -(i < 5 ? v[i] : 0) : (i < 5 ? v[i] : 0)) + cy;
        if (i < N - 1) { l[i] = t & kEisMask; cy = t >> 29; } else l[i] = t;
    }
}
// |value| as three 32-bit words + sign, and whether it is below 2^66 (thirty-four positions of Eisenstein digits, plume_ec.h); the limbs are normalised (lower limbs in [0, 2^29), sign in the top limb)
template <int N>
PLUME_HD bool eis_abs66(uint32_t (&mag)[3], uint32_t& neg, const int32_t (&l)[N]) {
    neg = l[N - 1] < 0 ? 1u : 0u;
    int32_t v[N];
    int32_t cy = 0;
    PLUME_UNROLL for (int i = 0; i < N; i++) {
        const int32_t t = (neg ? -l[i] : l[i]) + cy;
        if (i < N - 1) { v[i] = t & kEisMask; cy = t >> 29; } else v[i] = t;
    }
    // v: non-negative value in canonical limbs.  Words: bits 0..95 from limbs 0..3
    const uint32_t a = (uint32_t)v[0], b = (uint32_t)v[1], c = (uint32_t)v[2], d = N > 3 ? (uint32_t)v[N > 3 ? 3 : 0] : 0u;
    mag[0] = a | (b << 29);
    mag[1] = (b >> 3) | (c << 26);
    mag[2] = (c >> 6) | (d << 23);
    bool small = (mag[2] >> 2) == 0 && (N <= 3 || (d >> 9) == 0);       // below 2^66: nothing above bit 65
    PLUME_UNROLL for (int i = 4; i < N; i++) small = small && v[i] == 0;
    return small;
}

// (tau, upsilon) for the challenge c (canonical, any value including 0) given by its GLV split: the half-GCD of (pi, gamma).  At most kEisMaxSteps reductions; every honest input
// finishes in about forty.
constexpr int kEisMaxSteps = 400;
PLUME_HD void eis_half_gcd(eis_short& out, const glv_half& g0, const glv_half& g1) {      // gamma = g0 + g1 w: the GLV split of c (glv_split)
    // pi = a1 + b1 w:  a1 = 0x3086d221a7d46bcde86c90e49284eb15, b1 = -0xe4437ed6010e88286f547fa90abfe4c3 (the lattice vector glv_split calls (a1, b1))
    const glv_half pa = {{0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u}, 0u}, pb = {{0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u}, 1u};
    int32_t x0[kEisR], x1[kEisR], y0[kEisR], y1[kEisR];
    eis_from_half(x0, pa); eis_from_half(x1, pb);
    eis_from_half(y0, g0); eis_from_half(y1, g1);
    int32_t tx0[kEisT] = {0, 0, 0}, tx1[kEisT] = {0, 0, 0}, ty0[kEisT] = {1, 0, 0}, ty1[kEisT] = {0, 0, 0};
    double dx0 = eis_to_double(x0), dx1 = eis_to_double(x1), dy0 = eis_to_double(y0), dy1 = eis_to_double(y1);
    const double T = 340282366920938463463374607431768211456.0;      // 2^128
    int which = -1;                                                  // 0: (tx, x) is the answer, 1: (ty, y)
    PLUME_NOUNROLL for (int it = 0; it < kEisMaxSteps; it++) {
        if (eis_norm(dy0, dy1) < T) { which = 1; break; }
        int32_t q0, q1; int k;
        eis_quotient(q0, q1, k, dx0, dx1, dy0, dy1);
        eis_step(x0, x1, y0, y1, q0, q1, k);
        eis_step(tx0, tx1, ty0, ty1, q0, q1, k);
        dx0 = eis_to_double(x0); dx1 = eis_to_double(x1);
        if (eis_norm(dx0, dx1) < T) { which = 0; break; }
        eis_quotient(q0, q1, k, dy0, dy1, dx0, dx1);
        eis_step(y0, y1, x0, x1, q0, q1, k);
        eis_step(ty0, ty1, tx0, tx1, q0, q1, k);
        dy0 = eis_to_double(y0); dy1 = eis_to_double(y1);
    }
    const bool sel = which == 1;
    int32_t r0[kEisR], r1[kEisR], t0[kEisT], t1[kEisT];
    PLUME_UNROLL for (int i = 0; i < kEisR; i++) { r0[i] = sel ? y0[i] : x0[i]; r1[i] = sel ? y1[i] : x1[i]; }
    PLUME_UNROLL for (int i = 0; i < kEisT; i++) { t0[i] = sel ? ty0[i] : tx0[i]; t1[i] = sel ? ty1[i] : tx1[i]; }
    bool ok = which >= 0;
    uint32_t m0[3], m1[3], n0, n1;
    ok = eis_abs66(m0, n0, t0) && ok;
    ok = eis_abs66(m1, n1, t1) && ok;
    {   // tau = t0 + t1 lambda mod n
        sc a, b, lam, bl;
        PLUME_UNROLL for (int i = 0; i < 8; i++) { a.v[i] = i < 3 ? m0[i] : 0u; b.v[i] = i < 3 ? m1[i] : 0u; }
        const uint32_t L[8] = {0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu};      // lambda, little-endian words
        PLUME_UNROLL for (int i = 0; i < 8; i++) lam.v[i] = L[i];
        sc_mul(bl, b, lam);
        if (n1) sc_neg(bl, bl);
        if (n0) sc_neg(a, a);
        sc_add(out.tau, a, bl);
        ok = ok && !sc_is_zero(out.tau);
    }
    {   // tau - 1: one off t0's bottom limb (the normalisation inside eis_abs66 carries it)
        t0[0] -= 1;
        int32_t cy = 0;
        PLUME_UNROLL for (int i = 0; i < kEisT; i++) { const int32_t t = t0[i] + cy; if (i < kEisT - 1) { t0[i] = t & kEisMask; cy = t >> 29; } else t0[i] = t; }
    }
    ok = eis_abs66(out.t[0], out.tneg[0], t0) && ok;
    PLUME_UNROLL for (int i = 0; i < 3; i++) out.t[1][i] = m1[i];
    out.tneg[1] = n1;
    ok = eis_abs66(out.u[0], out.uneg[0], r0) && ok;
    ok = eis_abs66(out.u[1], out.uneg[1], r1) && ok;
    out.ok = ok;
}

// tau c == upsilon (mod n)?  The relation holds by construction -- every step applies one exact integer transformation to (x, y) and (tx, ty), the double-precision estimates only
// choose the quotient -- so this never fails; it is CHECKED all the same before the short form is used, because a pair that broke it would make equation 1 accept or reject
// the wrong signatures without any other symptom.  Two scalar multiplications per item (3 % of the scalar stage); a failure sends the item to the long form (verify_scalars).
PLUME_HD bool eis_consistent(const eis_short& e, const sc& c) {
    const uint32_t L[8] = {0x1B23BD72u, 0xDF02967Cu, 0x20816678u, 0x122E22EAu, 0x8812645Au, 0xA5261C02u, 0xC05C30E0u, 0x5363AD4Cu};      // lambda, little-endian words
    sc a, b, lam, bl, ups, lhs;
    PLUME_UNROLL for (int i = 0; i < 8; i++) { a.v[i] = i < 3 ? e.u[0][i] : 0u; b.v[i] = i < 3 ? e.u[1][i] : 0u; lam.v[i] = L[i]; }
    sc_mul(bl, b, lam);
    if (e.uneg[1]) sc_neg(bl, bl);
    if (e.uneg[0]) sc_neg(a, a);
    sc_add(ups, a, bl);                                               // upsilon = u0 + u1 lambda mod n
    sc_mul(lhs, e.tau, c);
    uint32_t d = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) d |= lhs.v[i] ^ ups.v[i];
    return d == 0;
}

}  // namespace plume
