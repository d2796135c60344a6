// RFC 9380 hash_to_curve, suite secp256k1_XMD:SHA-256_SSWU_RO_, for PLUME's H = h2c(m || SEC1c(pk))
// (rust-k256/src/utils.rs:11-20; signer path rust-k256/src/randomizedsigner.rs:57-61).
//   expand_message_xmd / hash_to_field : semantics rust-arkworks/src/fixed_hasher/expander.rs:89-134, mod.rs:32-62
//   simplified SWU on E' (A', B' = 1771, Z = -11) : constants rust-arkworks/src/secp256k1/curves/mod.rs:70-81
//   3-isogeny E' -> secp256k1 : coefficient tables curves/mod.rs:87-112
// Inversion-free: SSWU keeps x = xn/xd, the isogeny is evaluated on the fraction and lands directly in
// Jacobian coordinates; Q0 + Q1 is a Jacobian addition.  The affine H (needed for the V1 c-hash) falls out of
// H's window table for free (table_build shares one inversion across all tables of a lane).
#pragma once
#include "plume_ec.h"
#include "plume_sha256.h"

namespace plume {

PLUME_HD fe fe_iso_a() { return fe_set(0x3F8731ABu, 0xDD661ADCu, 0xA08A5558u, 0xF0F5D272u, 0xE953D363u, 0xCB6F0E5Du, 0x405447C0u, 0x1A444533u); }
PLUME_HD fe fe_sqrt_neg_z() { return fe_set(0x31FDF302u, 0x724013E5u, 0x7AD13FB3u, 0x8F842AFEu, 0xEC184F00u, 0xA74789DDu, 0x286729C8u, 0x303C4A59u); }  // sqrt(11)

// DST' = DST || I2OSP(49, 1)  (rust-k256/src/lib.rs:61; expander.rs:53-57), 50 bytes
PLUME_HD uint32_t dst_prime_byte(uint32_t k) {
    // "QUUX-V01-CS02-with-secp256k1_XMD:SHA-256_SSWU_RO_" 0x31, packed big-endian into words
    const uint32_t w[13] = {0x51555558u, 0x2D563031u, 0x2D435330u, 0x322D7769u, 0x74682D73u, 0x65637032u, 0x35366B31u,
                            0x5F584D44u, 0x3A534841u, 0x2D323536u, 0x5F535357u, 0x555F524Fu, 0x5F310000u};
    uint32_t idx = k >> 2, word = w[0];
    PLUME_UNROLL for (int i = 1; i < 13; i++) word = (idx == (uint32_t)i) ? w[i] : word;
    return (word >> (8 * (3 - (k & 3)))) & 0xFF;
}

// how the public key is appended to the message before hashing
#define PLUME_ENC_NONE 0u      // nothing appended (raw hash_from_bytes(&[msg]) — KAT pinning)
#define PLUME_ENC_IDENTITY 1u  // the single byte 00
#define PLUME_ENC_POINT 33u    // tag || x
// b0 = SHA256( 0^64 || msg || enc(pk) || 00 60 00 || DST' ); elen = length of enc(pk): 0, 1 or 33; pkx must be CANONICAL
PLUME_HD void xmd_b0(uint32_t b0[8], const uint8_t* msg, uint32_t mlen, const fe& pkx, uint32_t tag, uint32_t elen) {
    const uint32_t len = mlen + elen + 53u;
    uint32_t xw[8];
    fe_to_words(xw, pkx);
    sha256_init_after_zero_block(b0);
    sha256_absorb_pad(b0, 64u, len, [&](uint32_t pos) -> uint32_t {
        if (pos < mlen) return msg[pos];
        uint32_t k = pos - mlen;
        if (k < elen) return k == 0 ? (elen == PLUME_ENC_IDENTITY ? 0u : tag) : be_byte_of_limbs(xw, k - 1);
        k -= elen;
        if (k < 3) return k == 1 ? 0x60u : 0u;   // I2OSP(96, 2) || I2OSP(0, 1)
        return dst_prime_byte(k - 3);
    });
}
// b_i = SHA256( x[32] || i || DST' ): 83 bytes = block A (x, i, DST'[0..31)) + constant block B (DST'[31..50), pad, len)
PLUME_HD void xmd_bi(uint32_t out[8], const uint32_t x[8], uint32_t idx) {
    uint32_t w[16];
    PLUME_UNROLL for (int i = 0; i < 8; i++) w[i] = x[i];
    w[8] = (idx << 24) | 0x00515555u;  // idx 'Q' 'U' 'U'
    w[9] = 0x582D5630u; w[10] = 0x312D4353u; w[11] = 0x30322D77u; w[12] = 0x6974682Du; w[13] = 0x73656370u; w[14] = 0x3235366Bu; w[15] = 0x315F584Du;
    sha256_init(out);
    sha256_compress(out, w);
    // tail: "D:SHA-256_SSWU_RO_" 0x31 | 0x80 | zeros | bitlen 664
    w[0] = 0x443A5348u; w[1] = 0x412D3235u; w[2] = 0x365F5353u; w[3] = 0x57555F52u; w[4] = 0x4F5F3180u;
    PLUME_UNROLL for (int i = 5; i < 15; i++) w[i] = 0;
    w[15] = 664u;
    sha256_compress(out, w);
}
// hash_to_field: two field elements from 96 uniform bytes, each OS2IP(48 B) mod p (mod.rs:32-50)
PLUME_HD void fe_from_be48_words(fe& r, const uint32_t* w /* 12 big-endian words, most significant first */) {
    uint32_t t[16];
    PLUME_UNROLL for (int i = 0; i < 12; i++) t[i] = w[11 - i];
    PLUME_UNROLL for (int i = 12; i < 16; i++) t[i] = 0;
    fe_from_words16(r, t);
}
PLUME_HD void hash_to_field2(fe& u0, fe& u1, const uint8_t* msg, uint32_t mlen, const fe& pkx, uint32_t tag, uint32_t elen) {
    uint32_t b0[8], uni[24], x[8];
    xmd_b0(b0, msg, mlen, pkx, tag, elen);
    xmd_bi(uni, b0, 1);
    PLUME_UNROLL for (int i = 0; i < 8; i++) x[i] = b0[i] ^ uni[i];
    xmd_bi(uni + 8, x, 2);
    PLUME_UNROLL for (int i = 0; i < 8; i++) x[i] = b0[i] ^ uni[8 + i];
    xmd_bi(uni + 16, x, 3);
    fe_from_be48_words(u0, uni);
    fe_from_be48_words(u1, uni + 12);
}

// What the straight-line map knows beyond its result, for the circuit witness hints (plume_stages.h h2c_intermediates): tv1 = Z u^2, the square-root candidate of
// sqrt_ratio and which branch was taken
struct sswu_extra {
    fe tv1;        // Z * u^2
    fe root;       // is_sq ? sqrt(gx1) : sqrt(Z * gx1)      (RFC 9380 F.2.1.2 sqrt_ratio's second return value; sign as the exponentiation gives it)
    bool is_sq;    // gx1 is a square
};
// RFC 9380 F.2 straight-line SSWU without the final division: x = xn/xd, y affine on E'
PLUME_HD void sswu_frac(fe& xn, fe& xd, fe& y, const fe& u, sswu_extra* extra = nullptr) {
    const fe A = fe_iso_a();
    fe tv1, tv2, tv3, tv4, tv5, tv6, y1, t;
    fe_sqr(tv1, u);
    fe_mul_small(tv1, tv1, 11); fe_neg(tv1, tv1);          // tv1 = Z*u^2, Z = -11
    fe_sqr(tv2, tv1);
    fe_add(tv2, tv2, tv1);
    fe one = fe_small(1);
    fe_add(tv3, tv2, one);
    fe_mul_small(tv3, tv3, 1771);                          // B'
    if (fe_is_zero(tv2)) { tv4 = fe_small(11); fe_neg(tv4, tv4); } else { fe_neg(tv4, tv2); }
    fe_mul_k(tv4, A, tv4);
    fe_sqr(tv2, tv3);
    fe_sqr(tv6, tv4);
    fe_mul_k(tv5, A, tv6);
    fe_add(tv2, tv2, tv5);
    fe_mul(tv2, tv2, tv3);
    fe_mul(tv6, tv6, tv4);
    fe_mul_small(tv5, tv6, 1771);
    fe_add(tv2, tv2, tv5);                                  // gx1 numerator
    fe_mul(xn, tv1, tv3);
    // sqrt_ratio_3mod4(tv2, tv6)  (RFC 9380 F.2.1.2)
    fe s1, s2, s3;
    fe_sqr(s1, tv6);
    fe_mul(s2, tv2, tv6);
    fe_mul(s1, s1, s2);
    fe_pow_c1(y1, s1);
    fe_mul(y1, y1, s2);
    fe_sqr(s3, y1); fe_mul(s3, s3, tv6);
    bool is_sq = fe_eq(s3, tv2);
    fe y2; fe_mul_k(y2, fe_sqrt_neg_z(), y1);
    fe_cmov(y1, y2, !is_sq);
    if (extra) { extra->tv1 = tv1; extra->root = y1; extra->is_sq = is_sq; }
    fe_mul(y, tv1, u);
    fe_mul(y, y, y1);
    fe_cmov(xn, tv3, is_sq);
    fe_cmov(y, y1, is_sq);
    bool e1 = fe_is_odd(u) == fe_is_odd(y);
    fe_neg(t, y);
    fe_cmov(y, t, !e1);
    xd = tv4;
}

// 3-isogeny on the fraction (RFC 9380 E.1), output Jacobian with Z = Dx*Dy [* yd].  y is affine, or the fraction y / *yd when yd is given.
PLUME_HD void iso3_frac_to_jac(jac& q, const fe& xn, const fe& xd, const fe& y, const fe* yd = nullptr) {
    const fe k10 = fe_set(0x8E38E38Eu, 0x38E38E38u, 0xE38E38E3u, 0x8E38E38Eu, 0x38E38E38u, 0xE38E38E3u, 0x8E38E38Du, 0xAAAAA8C7u);
    const fe k11 = fe_set(0x07D3D4C8u, 0x0BC321D5u, 0xB9F315CEu, 0xA7FD44C5u, 0xD595D2FCu, 0x0BF63B92u, 0xDFFF1044u, 0xF17C6581u);
    const fe k12 = fe_set(0x534C328Du, 0x23F234E6u, 0xE2A413DEu, 0xCA25CAECu, 0xE4506144u, 0x037C4031u, 0x4ECBD0B5u, 0x3D9DD262u);
    const fe k13 = fe_set(0x8E38E38Eu, 0x38E38E38u, 0xE38E38E3u, 0x8E38E38Eu, 0x38E38E38u, 0xE38E38E3u, 0x8E38E38Du, 0xAAAAA88Cu);
    const fe k20 = fe_set(0xD3577119u, 0x3D94918Au, 0x9CA34CCBu, 0xB7B640DDu, 0x86CD4095u, 0x42F8487Du, 0x9FE6B745u, 0x781EB49Bu);
    const fe k21 = fe_set(0xEDADC6F6u, 0x4383DC1Du, 0xF7C4B2D5u, 0x1B542254u, 0x06D36B64u, 0x1F5E41BBu, 0xC52A5661u, 0x2A8C6D14u);
    const fe k30 = fe_set(0x4BDA12F6u, 0x84BDA12Fu, 0x684BDA12u, 0xF684BDA1u, 0x2F684BDAu, 0x12F684BDu, 0xA12F684Bu, 0x8E38E23Cu);
    const fe k31 = fe_set(0xC75E0C32u, 0xD5CB7C0Fu, 0xA9D0A54Bu, 0x12A0A6D5u, 0x647AB046u, 0xD686DA6Fu, 0xDFFC90FCu, 0x201D71A3u);
    const fe k32 = fe_set(0x29A61946u, 0x91F91A73u, 0x715209EFu, 0x6512E576u, 0x722830A2u, 0x01BE2018u, 0xA765E85Au, 0x9ECEE931u);
    const fe k33 = fe_set(0x2F684BDAu, 0x12F684BDu, 0xA12F684Bu, 0xDA12F684u, 0xBDA12F68u, 0x4BDA12F6u, 0x84BDA12Fu, 0x38E38D84u);
    const fe k40 = fe_set(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu, 0xFFFFF93Bu);
    const fe k41 = fe_set(0x7A06534Bu, 0xB8BDB49Fu, 0xD5E9E663u, 0x2722C298u, 0x9467C1BFu, 0xC8E8D978u, 0xDFB425D2u, 0x685C2573u);
    const fe k42 = fe_set(0x6484AA71u, 0x6545CA2Cu, 0xF3A70C3Fu, 0xA8FE337Eu, 0x0A3D2116u, 0x2F0D6299u, 0xA7BF8192u, 0xBFD2A76Fu);
    fe xd2, xd3, xn2, xn3, n2d, nd2, t, nx, dx, ny, dy;
    fe_sqr(xd2, xd); fe_mul(xd3, xd2, xd);
    fe_sqr(xn2, xn); fe_mul(xn3, xn2, xn);
    fe_mul(n2d, xn2, xd); fe_mul(nd2, xn, xd2);
    // Nx = k13 xn^3 + k12 xn^2 xd + k11 xn xd^2 + k10 xd^3
    fe_mul_k(nx, k13, xn3); fe_mul_k(t, k12, n2d); fe_add(nx, nx, t); fe_mul_k(t, k11, nd2); fe_add(nx, nx, t); fe_mul_k(t, k10, xd3); fe_add(nx, nx, t);
    // Dx = xd * (xn^2 + k21 xn xd + k20 xd^2)
    fe_mul_k(dx, k21, xn); fe_mul(dx, dx, xd); fe_add(dx, dx, xn2); fe_mul_k(t, k20, xd2); fe_add(dx, dx, t); fe_mul(dx, dx, xd);
    // Ny = k33 xn^3 + k32 xn^2 xd + k31 xn xd^2 + k30 xd^3
    fe_mul_k(ny, k33, xn3); fe_mul_k(t, k32, n2d); fe_add(ny, ny, t); fe_mul_k(t, k31, nd2); fe_add(ny, ny, t); fe_mul_k(t, k30, xd3); fe_add(ny, ny, t);
    // Dy = xn^3 + k42 xn^2 xd + k41 xn xd^2 + k40 xd^3
    fe_mul_k(dy, k42, n2d); fe_add(dy, dy, xn3); fe_mul_k(t, k41, nd2); fe_add(dy, dy, t); fe_mul_k(t, k40, xd3); fe_add(dy, dy, t);
    // x' = Nx/Dx, y' = y Ny/Dy;  Z = Dx Dy, X = Nx Dx Dy^2, Y = y Ny Dx^3 Dy^2
    fe dy2, w;
    fe_mul(q.z, dx, dy);
    fe_sqr(dy2, dy);
    fe_mul(w, dx, dy2);            // Dx Dy^2
    if (yd) {                      // y = y / yd:  Z = Dx Dy yd,  X = Nx Dx Dy^2 yd^2,  Y = y Ny Dx^3 Dy^2 yd^2
        fe yd2;
        fe_mul(q.z, q.z, *yd);
        fe_sqr(yd2, *yd);
        fe_mul(w, w, yd2);
    }
    fe_mul(q.x, nx, w);
    fe_sqr(t, dx); fe_mul(w, w, t);  // Dx^3 Dy^2 [yd^2]
    fe_mul(t, y, ny);
    fe_mul(q.y, t, w);
    q.inf = fe_is_zero(q.z) ? 1u : 0u;
}
// (a/b, y0) + (c/d, y1) on E', x as fractions, y affine; the sum as fractions x3 = xn/xd, y3 = yn/yd.  Returns false when the sum is the identity (opposite points).
//   distinct x (the case that occurs): the chord, which does not involve the curve's coefficients --
//       lambda = (y1 - y0) bd / (cb - ad) = Ln / Ld;   x3 = lambda^2 - x0 - x1 = (Ln^2 bd - (cb + ad) Ld^2) / (Ld^2 bd);   y3 = lambda (x0 - x3) - y0
//   equal x, equal y (cryptographically unreachable; a rare divergent branch): the tangent, lambda = (3 a^2 + A' b^2) / (2 y0 b^2), then the same two lines with
//       x1 = x0; y0 != 0 because E'(Fp) has odd (prime) order
PLUME_HD bool eprime_add_frac(fe& xn, fe& xd, fe& yn, fe& yd, const fe& a, const fe& b, const fe& y0, const fe& c, const fe& d, const fe& y1) {
    fe bd, cb, ad, Ld, Sm, Ln, Ln2, Ld2, t, W;
    fe_mul(bd, b, d); fe_mul(cb, c, b); fe_mul(ad, a, d);
    fe_sub(Ld, cb, ad);
    fe_add(Sm, cb, ad);
    fe_sub(t, y1, y0); fe_mul(Ln, t, bd);
    if (fe_is_zero(Ld)) {
        if (!fe_is_zero(t)) return false;                  // y1 = -y0: opposite points
        fe a2, b2;
        fe_sqr(a2, a); fe_sqr(b2, b);
        fe_mul_small(Ln, a2, 3); fe_mul_k(t, fe_iso_a(), b2); fe_add(Ln, Ln, t);     // 3 a^2 + A' b^2
        fe_mul(Ld, y0, b2); fe_dbl(Ld, Ld);                                              // 2 y0 b^2
        bd = b;                                                                          // x0 + x1 = 2a / b, x1 - x0 folded into lambda
        fe_dbl(Sm, a); ad = a;
    }
    fe_sqr(Ln2, Ln); fe_sqr(Ld2, Ld);
    fe_mul(xd, Ld2, bd);                                   // D
    fe_mul(xn, Ln2, bd); fe_mul(t, Sm, Ld2); fe_sub(xn, xn, t);
    fe_mul(W, ad, Ld2); fe_sub(W, W, xn);                  // (x0 - x3) D
    fe_mul(yd, Ld, xd);
    fe_mul(yn, Ln, W); fe_mul(t, y0, yd); fe_sub(yn, yn, t);
    return true;
}
// H = h2c(msg || enc(pk)) as a Jacobian point; enc: PLUME_ENC_POINT / PLUME_ENC_IDENTITY / PLUME_ENC_NONE.
// Round 3: ONE isogeny evaluation instead of two.  iso_map is a group homomorphism, so iso(Q0') + iso(Q1') = iso(Q0' + Q1') (RFC 9380 section 6.6.3 names exactly this
// optimisation): the two simplified-SWU outputs are added on E' -- as fractions, no inversion -- and the sum goes through the isogeny once: 13 + 33 field multiplications
// instead of 2 x 29 + 16.  E'(Fp) has the prime order of secp256k1 (isogenous curves), so the isogeny's kernel holds no rational point but the identity and the result is
// the same point in every case, Q0' = +-Q1' included (eprime_add_frac).
PLUME_HD void map2_to_curve_jac(jac& h, const fe& u0, const fe& u1) {
    fe a = fe_zero(), b = fe_zero(), y0 = fe_zero(), c, d, y1;
    // one body for both maps (code size), but no arrays indexed by the loop counter (those would live in scratch memory: round 1, 288 B per lane) and no branch on it
    // either: every pass shifts the previous map's result along, the sum does not care about the order
    PLUME_NOUNROLL for (int i = 0; i < 2; i++) {
        fe u, xn, xd, y;
        PLUME_UNROLL for (int k = 0; k < 9; k++) u.v[k] = i ? u1.v[k] : u0.v[k];
        sswu_frac(xn, xd, y, u);
        c = a; d = b; y1 = y0;
        a = xn; b = xd; y0 = y;
    }
    fe xn, xd, yn, yd;
    if (eprime_add_frac(xn, xd, yn, yd, a, b, y0, c, d, y1)) {
        iso3_frac_to_jac(h, xn, xd, yn, &yd);
    } else {
        h.x = fe_small(1); h.y = fe_small(1); h.z = fe_small(0); h.inf = 1;
    }
}
// the same sum from the two maps' results, for callers that ran the maps elsewhere (the two-role ingest kernel of small batches: one map per wavefront role).
// (a, b, y0) = map(u1), (c, d, y1) = map(u0): the operand order of map2_to_curve_jac, so that the Jacobian H is the same representative.
PLUME_HD void maps_to_curve_jac(jac& h, const fe& a, const fe& b, const fe& y0, const fe& c, const fe& d, const fe& y1) {
    fe xn, xd, yn, yd;
    if (eprime_add_frac(xn, xd, yn, yd, a, b, y0, c, d, y1)) {
        iso3_frac_to_jac(h, xn, xd, yn, &yd);
    } else {
        h.x = fe_small(1); h.y = fe_small(1); h.z = fe_small(0); h.inf = 1;
    }
}
PLUME_HD void hash_to_curve_jac(jac& h, const uint8_t* msg, uint32_t mlen, const fe& pkx, uint32_t tag, uint32_t enc) {
    fe u0, u1;
    hash_to_field2(u0, u1, msg, mlen, pkx, tag, enc);
    map2_to_curve_jac(h, u0, u1);
}

}  // namespace plume
