/* PLUME (ERC-7524) on secp256k1 — plain-C ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the reported CPU baseline.  The product (zk-nullifier-sig_amd/) never links or calls it.
 *
 * CPU restatement of the reference's hot path, one item at a time, single-threaded per call (an optional
 * pthread fan-out only splits the batch; it does not change any per-item computation):
 *   verify           rust-k256/src/lib.rs:93-145
 *   c-hash           rust-k256/src/lib.rs:159-168, rust-arkworks/src/lib.rs:120-163
 *   hash_to_curve    rust-k256/src/utils.rs:11-20  (k256 GroupDigest::hash_from_bytes, ExpandMsgXmd<Sha256>)
 *   encode_pt        rust-k256/src/utils.rs:23-25, rust-arkworks/src/lib.rs:76-88,112-118
 *   sign             rust-k256/src/randomizedsigner.rs:43-112, rust-arkworks/src/lib.rs:229-278
 *   XMD / h2f        rust-arkworks/src/fixed_hasher/expander.rs:89-134, mod.rs:32-62
 *   constants        rust-arkworks/src/secp256k1/fields/fq.rs:12, fr.rs:19, curves/mod.rs:36-112
 * The arithmetic is in the un-vendored crate k256 ~0.13.3 (rust-k256/Cargo.toml:18), absent from
 * /root/reference and unbuildable here (no rustc); its algorithm is restated from RFC 9380 (SSWU F.2,
 * sqrt_ratio F.2.1.2, 3-isogeny E.1), SEC1 and FIPS 180-4.  Parity is PINNED: tests/test_oracle_c.py checks
 * this library against the reference's own KATs (tests/golden/reference_kats.json) and against the Python
 * oracle's seeded golden batches (tests/golden/*.bin).
 *
 * Representation: Fp / Fn elements are 4x64-bit little-endian limbs, always fully reduced; products use
 * unsigned __int128.  Points are Jacobian (X,Y,Z) with an explicit infinity flag; scalar multiplication is a
 * 4-bit fixed window over all 256 bits (same algorithm class as k256's generic `ProjectivePoint * Scalar`,
 * minus its endomorphism split).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;     /* mod p */
typedef struct { uint64_t l[4]; } sc;     /* mod n */
typedef struct { fe x, y, z; int inf; } jac;
typedef struct { fe x, y; int inf; } aff;

/* ------------------------------------------------------------------ constants (curves/mod.rs, fq.rs, fr.rs) */
static const uint64_t P_[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};
static const uint64_t N_[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
#define PC 0x1000003D1ULL /* 2^256 mod p */
static const fe FE_GX = {{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}};
static const fe FE_GY = {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}};
static const fe FE_ISO_A = {{0x405447C01A444533ULL, 0xE953D363CB6F0E5DULL, 0xA08A5558F0F5D272ULL, 0x3F8731ABDD661ADCULL}};
static const fe FE_ISO_B = {{1771, 0, 0, 0}};
static const fe FE_Z = {{0xFFFFFFFEFFFFFC24ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}}; /* -11 */
static const fe FE_ONE = {{1, 0, 0, 0}};
/* 3-isogeny coefficient tables, ascending degree (curves/mod.rs:88-111): filled from big-endian hex in init_consts() */
static fe YNUM[4], YDEN[4], XNUM[4], XDEN[3];
static fe FE_C2;      /* sqrt(-Z) */
static int consts_ready = 0;

/* ------------------------------------------------------------------------------------------- SHA-256 */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; } sha256_ctx;
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const uint8_t *p) {
    uint32_t w[64], a, b, c, d, e, f, g, hh;
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = h[0]; b = h[1]; c = h[2]; d = h[3]; e = h[4]; f = h[5]; g = h[6]; hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25), ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
static void sha256_init(sha256_ctx *c) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(c->h, iv, sizeof iv);
    c->len = 0;
}
static void sha256_update(sha256_ctx *c, const void *data, size_t n) {
    const uint8_t *p = (const uint8_t *)data;
    while (n) {
        size_t off = c->len % 64, take = 64 - off;
        if (take > n) take = n;
        memcpy(c->buf + off, p, take);
        c->len += take; p += take; n -= take;
        if (c->len % 64 == 0) sha256_block(c->h, c->buf);
    }
}
static void sha256_final(sha256_ctx *c, uint8_t out[32]) {
    uint64_t bits = c->len * 8;
    uint8_t pad = 0x80, zero = 0, lenb[8];
    sha256_update(c, &pad, 1);
    while (c->len % 64 != 56) sha256_update(c, &zero, 1);
    for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha256_update(c, lenb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = c->h[i] >> 24; out[4 * i + 1] = c->h[i] >> 16; out[4 * i + 2] = c->h[i] >> 8; out[4 * i + 3] = c->h[i]; }
}

/* ------------------------------------------------------------------------------- 256-bit helpers (mod m) */
static int ge256(const uint64_t a[4], const uint64_t m[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] > m[i]) return 1; if (a[i] < m[i]) return 0; }
    return 1;
}
static int is_zero256(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static uint64_t add256(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t sub256(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static void from_be32(uint64_t r[4], const uint8_t b[32]) {
    for (int i = 0; i < 4; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | b[8 * (3 - i) + j]; r[i] = w; }
}
static void to_be32(uint8_t b[32], const uint64_t a[4]) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) b[8 * (3 - i) + j] = (uint8_t)(a[i] >> (56 - 8 * j));
}
static void mul256(uint64_t r[8], const uint64_t a[4], const uint64_t b[4]) {
    memset(r, 0, 64);
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a[i] * b[j] + r[i + j]; r[i + j] = (uint64_t)c; c >>= 64; }
        r[i + 4] = (uint64_t)c;
    }
}

/* ---------------------------------------------------------------------------------------------- Fp */
static void fe_add(fe *r, const fe *a, const fe *b) {
    uint64_t c = add256(r->l, a->l, b->l);
    if (c || ge256(r->l, P_)) sub256(r->l, r->l, P_);
}
static void fe_sub(fe *r, const fe *a, const fe *b) {
    if (sub256(r->l, a->l, b->l)) add256(r->l, r->l, P_);
}
static void fe_neg(fe *r, const fe *a) { fe z = {{0, 0, 0, 0}}; fe_sub(r, &z, a); }
static int fe_is_zero(const fe *a) { return is_zero256(a->l); }
static int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }
static void fe_reduce512(fe *r, const uint64_t t[8]) { /* t mod p, 2^256 = PC (mod p) */
    uint64_t lo[5];
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)t[i] + (u128)t[i + 4] * PC; lo[i] = (uint64_t)c; c >>= 64; }
    lo[4] = (uint64_t)c;                       /* < 2^34 */
    c = (u128)lo[4] * PC;
    uint64_t out[4];
    for (int i = 0; i < 4; i++) { c += lo[i]; out[i] = (uint64_t)c; c >>= 64; }
    if (c) { /* wrapped once more: add PC (cannot carry again) */
        u128 d = PC;
        for (int i = 0; i < 4; i++) { d += out[i]; out[i] = (uint64_t)d; d >>= 64; }
    }
    if (ge256(out, P_)) sub256(out, out, P_);
    memcpy(r->l, out, 32);
}
static void fe_mul(fe *r, const fe *a, const fe *b) { uint64_t t[8]; mul256(t, a->l, b->l); fe_reduce512(r, t); }
static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }
static void fe_pow(fe *r, const fe *a, const uint64_t e[4]) {
    fe acc = FE_ONE, base = *a;
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) fe_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) { if (started) fe_mul(&acc, &acc, &base); else { acc = base; started = 1; } }
    }
    *r = acc;
}
static void fe_inv(fe *r, const fe *a) { uint64_t e[4]; static const uint64_t two[4] = {2, 0, 0, 0}; sub256(e, P_, two); fe_pow(r, a, e); }
static int fe_from_be_checked(fe *r, const uint8_t b[32]) { from_be32(r->l, b); return !ge256(r->l, P_); }
static void fe_from_hex(fe *r, const char *hex) { /* 64 hex digits */
    uint8_t b[32];
    for (int i = 0; i < 32; i++) {
        unsigned v = 0;
        for (int k = 0; k < 2; k++) { char ch = hex[2 * i + k]; v = v * 16 + (ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10); }
        b[i] = (uint8_t)v;
    }
    from_be32(r->l, b);
}

static void init_consts(void) {
    if (consts_ready) return;
    fe_from_hex(&XNUM[0], "8e38e38e38e38e38e38e38e38e38e38e38e38e38e38e38e38e38e38daaaaa8c7");
    fe_from_hex(&XNUM[1], "07d3d4c80bc321d5b9f315cea7fd44c5d595d2fc0bf63b92dfff1044f17c6581");
    fe_from_hex(&XNUM[2], "534c328d23f234e6e2a413deca25caece4506144037c40314ecbd0b53d9dd262");
    fe_from_hex(&XNUM[3], "8e38e38e38e38e38e38e38e38e38e38e38e38e38e38e38e38e38e38daaaaa88c");
    fe_from_hex(&XDEN[0], "d35771193d94918a9ca34ccbb7b640dd86cd409542f8487d9fe6b745781eb49b");
    fe_from_hex(&XDEN[1], "edadc6f64383dc1df7c4b2d51b54225406d36b641f5e41bbc52a56612a8c6d14");
    XDEN[2] = FE_ONE;
    fe_from_hex(&YNUM[0], "4bda12f684bda12f684bda12f684bda12f684bda12f684bda12f684b8e38e23c");
    fe_from_hex(&YNUM[1], "c75e0c32d5cb7c0fa9d0a54b12a0a6d5647ab046d686da6fdffc90fc201d71a3");
    fe_from_hex(&YNUM[2], "29a6194691f91a73715209ef6512e576722830a201be2018a765e85a9ecee931");
    fe_from_hex(&YNUM[3], "2f684bda12f684bda12f684bda12f684bda12f684bda12f684bda12f38e38d84");
    fe_from_hex(&YDEN[0], "fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffff93b");
    fe_from_hex(&YDEN[1], "7a06534bb8bdb49fd5e9e6632722c2989467c1bfc8e8d978dfb425d2685c2573");
    fe_from_hex(&YDEN[2], "6484aa716545ca2cf3a70c3fa8fe337e0a3d21162f0d6299a7bf8192bfd2a76f");
    YDEN[3] = FE_ONE;
    /* c2 = sqrt(-Z) = 11^((p+1)/4) */
    fe eleven = {{11, 0, 0, 0}};
    uint64_t e[4]; static const uint64_t one[4] = {1, 0, 0, 0};
    add256(e, P_, one);                        /* p+1 wraps to 2^256: compute (p+1)/4 as (p>>2)+1 since p = 3 mod 4 */
    for (int i = 0; i < 4; i++) e[i] = (P_[i] >> 2) | (i < 3 ? P_[i + 1] << 62 : 0);
    add256(e, e, one);
    fe_pow(&FE_C2, &eleven, e);
    consts_ready = 1;
}

/* ---------------------------------------------------------------------------------------------- Fn */
static void sc_reduce512(sc *r, const uint64_t t[8]) {
    /* generic: schoolbook long reduction by repeated folding with NC = 2^256 - n (129 bits) */
    static const uint64_t NC[3] = {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1};
    uint64_t cur[8];
    memcpy(cur, t, 64);
    for (int round = 0; round < 4; round++) { /* hi*NC + lo; shrinks by ~127 bits each round */
        uint64_t hi[4] = {cur[4], cur[5], cur[6], cur[7]}, acc[8] = {cur[0], cur[1], cur[2], cur[3], 0, 0, 0, 0};
        if (is_zero256(hi)) break;
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 3; j++) { c += (u128)hi[i] * NC[j] + acc[i + j]; acc[i + j] = (uint64_t)c; c >>= 64; }
            for (int k = i + 3; k < 8 && c; k++) { c += acc[k]; acc[k] = (uint64_t)c; c >>= 64; }
        }
        memcpy(cur, acc, 64);
    }
    uint64_t out[4] = {cur[0], cur[1], cur[2], cur[3]};
    while (ge256(out, N_)) sub256(out, out, N_);
    memcpy(r->l, out, 32);
}
static void sc_mul(sc *r, const sc *a, const sc *b) { uint64_t t[8]; mul256(t, a->l, b->l); sc_reduce512(r, t); }
static void sc_add(sc *r, const sc *a, const sc *b) {
    uint64_t c = add256(r->l, a->l, b->l);
    if (c || ge256(r->l, N_)) sub256(r->l, r->l, N_);
}
static void sc_neg(sc *r, const sc *a) { if (is_zero256(a->l)) *r = *a; else sub256(r->l, N_, a->l); }

/* ------------------------------------------------------------------------------------------ group law */
static void jac_set_inf(jac *r) { memset(r, 0, sizeof *r); r->inf = 1; }
static void jac_from_aff(jac *r, const aff *a) {
    if (a->inf) { jac_set_inf(r); return; }
    r->x = a->x; r->y = a->y; r->z = FE_ONE; r->inf = 0;
}
static void jac_dbl(jac *r, const jac *p) {
    if (p->inf || fe_is_zero(&p->y)) { jac_set_inf(r); return; }
    fe a, b, c, d, e, f, t;
    fe_sqr(&a, &p->x); fe_sqr(&b, &p->y); fe_sqr(&c, &b);
    fe_add(&t, &p->x, &b); fe_sqr(&t, &t); fe_sub(&t, &t, &a); fe_sub(&t, &t, &c); fe_add(&d, &t, &t);
    fe_add(&e, &a, &a); fe_add(&e, &e, &a);
    fe_sqr(&f, &e);
    fe z3; fe_mul(&z3, &p->y, &p->z); fe_add(&z3, &z3, &z3);
    fe x3; fe_sub(&x3, &f, &d); fe_sub(&x3, &x3, &d);
    fe y3; fe_sub(&y3, &d, &x3); fe_mul(&y3, &e, &y3);
    fe c8; fe_add(&c8, &c, &c); fe_add(&c8, &c8, &c8); fe_add(&c8, &c8, &c8);
    fe_sub(&y3, &y3, &c8);
    r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}
static void jac_add(jac *r, const jac *p, const jac *q) {
    if (p->inf) { *r = *q; return; }
    if (q->inf) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
    fe_sqr(&z1z1, &p->z); fe_sqr(&z2z2, &q->z);
    fe_mul(&u1, &p->x, &z2z2); fe_mul(&u2, &q->x, &z1z1);
    fe_mul(&s1, &p->y, &q->z); fe_mul(&s1, &s1, &z2z2);
    fe_mul(&s2, &q->y, &p->z); fe_mul(&s2, &s2, &z1z1);
    fe_sub(&h, &u2, &u1); fe_sub(&rr, &s2, &s1);
    if (fe_is_zero(&h)) { if (fe_is_zero(&rr)) jac_dbl(r, p); else jac_set_inf(r); return; }
    fe h2, h3, v;
    fe_sqr(&h2, &h); fe_mul(&h3, &h2, &h); fe_mul(&v, &u1, &h2);
    fe x3; fe_sqr(&x3, &rr); fe_sub(&x3, &x3, &h3); fe_sub(&x3, &x3, &v); fe_sub(&x3, &x3, &v);
    fe y3; fe_sub(&t, &v, &x3); fe_mul(&y3, &rr, &t); fe_mul(&t, &s1, &h3); fe_sub(&y3, &y3, &t);
    fe z3; fe_mul(&z3, &p->z, &q->z); fe_mul(&z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}
static void jac_neg(jac *r, const jac *p) { *r = *p; if (!p->inf) fe_neg(&r->y, &p->y); }
static void jac_to_aff(aff *r, const jac *p) {
    if (p->inf) { memset(r, 0, sizeof *r); r->inf = 1; return; }
    fe zi, zi2, zi3;
    fe_inv(&zi, &p->z); fe_sqr(&zi2, &zi); fe_mul(&zi3, &zi2, &zi);
    fe_mul(&r->x, &p->x, &zi2); fe_mul(&r->y, &p->y, &zi3); r->inf = 0;
}
static int jac_eq_aff(const jac *p, const aff *a) { /* ProjectivePoint == AffinePoint (lib.rs:117,122) */
    if (p->inf || a->inf) return p->inf && a->inf;
    fe z2, z3, t;
    fe_sqr(&z2, &p->z); fe_mul(&z3, &z2, &p->z);
    fe_mul(&t, &a->x, &z2); if (!fe_eq(&t, &p->x)) return 0;
    fe_mul(&t, &a->y, &z3); return fe_eq(&t, &p->y);
}
/* k*P, 4-bit fixed window, MSB first (`ProjectivePoint * Scalar`, lib.rs:101,109; randomizedsigner.rs:51,53,67,70) */
static void jac_mul(jac *r, const sc *k, const jac *p) {
    jac tab[16];
    jac_set_inf(&tab[0]); tab[1] = *p;
    for (int i = 2; i < 16; i++) { if (i & 1) jac_add(&tab[i], &tab[i - 1], p); else jac_dbl(&tab[i], &tab[i / 2]); }
    jac acc; jac_set_inf(&acc);
    for (int w = 63; w >= 0; w--) {
        for (int d = 0; d < 4; d++) jac_dbl(&acc, &acc);
        unsigned dig = (unsigned)(k->l[w / 16] >> (4 * (w % 16))) & 15;
        if (dig) jac_add(&acc, &acc, &tab[dig]);
    }
    *r = acc;
}
static int aff_on_curve(const aff *a) {
    if (a->inf) return 1;
    fe l, r3, seven = {{7, 0, 0, 0}};
    fe_sqr(&l, &a->y); fe_sqr(&r3, &a->x); fe_mul(&r3, &r3, &a->x); fe_add(&r3, &r3, &seven);
    return fe_eq(&l, &r3);
}
/* C-ABI point format: 64 bytes x||y big-endian, all-zero = identity. returns 0 if not canonical / not on curve */
static int aff_from_bytes(aff *r, const uint8_t b[64]) {
    int allz = 1;
    for (int i = 0; i < 64; i++) if (b[i]) { allz = 0; break; }
    if (allz) { memset(r, 0, sizeof *r); r->inf = 1; return 1; }
    r->inf = 0;
    if (!fe_from_be_checked(&r->x, b) || !fe_from_be_checked(&r->y, b + 32)) return 0;
    return aff_on_curve(r);
}
static void aff_to_bytes(uint8_t b[64], const aff *a) {
    if (a->inf) { memset(b, 0, 64); return; }
    to_be32(b, a->x.l); to_be32(b + 32, a->y.l);
}
/* encode_pt: 33-byte SEC1 compressed; identity = single 00 (utils.rs:23-25; arkworks lib.rs:112-118) */
static size_t sec1c(uint8_t out[33], const aff *a) {
    if (a->inf) { out[0] = 0; return 1; }
    out[0] = 2 + (uint8_t)(a->y.l[0] & 1);
    to_be32(out + 1, a->x.l);
    return 33;
}

/* ------------------------------------------------------------------------------------- hash_to_curve */
static const char DST_PRIME[51] = "QUUX-V01-CS02-with-secp256k1_XMD:SHA-256_SSWU_RO_\x31"; /* DST || len(DST)=49 */
static void expand_message_xmd96(uint8_t out[96], const uint8_t *m1, size_t n1, const uint8_t *m2, size_t n2) {
    /* expander.rs:89-134 with n = 96, ell = 3; msg = m1 || m2 */
    static const uint8_t zpad[64] = {0};
    uint8_t b0[32], bi[32], x[32], tag;
    const uint8_t lib[3] = {0, 96, 0};
    sha256_ctx c;
    sha256_init(&c); sha256_update(&c, zpad, 64); sha256_update(&c, m1, n1); sha256_update(&c, m2, n2);
    sha256_update(&c, lib, 3); sha256_update(&c, DST_PRIME, 50); sha256_final(&c, b0);
    tag = 1;
    sha256_init(&c); sha256_update(&c, b0, 32); sha256_update(&c, &tag, 1); sha256_update(&c, DST_PRIME, 50); sha256_final(&c, bi);
    memcpy(out, bi, 32);
    for (int i = 2; i <= 3; i++) {
        for (int j = 0; j < 32; j++) x[j] = b0[j] ^ bi[j];
        tag = (uint8_t)i;
        sha256_init(&c); sha256_update(&c, x, 32); sha256_update(&c, &tag, 1); sha256_update(&c, DST_PRIME, 50); sha256_final(&c, bi);
        memcpy(out + 32 * (i - 1), bi, 32);
    }
}
static void fe_from_be48(fe *r, const uint8_t b[48]) { /* OS2IP(48 B) mod p (fixed_hasher/mod.rs:43-45) */
    uint64_t t[8] = {0};
    for (int i = 0; i < 6; i++) { uint64_t w = 0; for (int j = 0; j < 8; j++) w = (w << 8) | b[8 * (5 - i) + j]; t[i] = w; }
    fe_reduce512(r, t);
}
static int fe_sgn0(const fe *a) { return (int)(a->l[0] & 1); }
static int sqrt_ratio(fe *y, const fe *u, const fe *v) { /* RFC 9380 F.2.1.2 */
    fe tv1, tv2, tv3, y1, y2;
    uint64_t c1[4];
    for (int i = 0; i < 4; i++) c1[i] = (P_[i] >> 2) | (i < 3 ? P_[i + 1] << 62 : 0);   /* (p-3)/4 = p >> 2 */
    fe_sqr(&tv1, v); fe_mul(&tv2, u, v); fe_mul(&tv1, &tv1, &tv2);
    fe_pow(&y1, &tv1, c1); fe_mul(&y1, &y1, &tv2); fe_mul(&y2, &y1, &FE_C2);
    fe_sqr(&tv3, &y1); fe_mul(&tv3, &tv3, v);
    int qr = fe_eq(&tv3, u);
    *y = qr ? y1 : y2;
    return qr;
}
static void sswu(fe *xo, fe *yo, const fe *u) { /* RFC 9380 F.2, on E' */
    fe tv1, tv2, tv3, tv4, tv5, tv6, x, y, y1;
    fe_sqr(&tv1, u); fe_mul(&tv1, &FE_Z, &tv1); fe_sqr(&tv2, &tv1); fe_add(&tv2, &tv2, &tv1);
    fe_add(&tv3, &tv2, &FE_ONE); fe_mul(&tv3, &FE_ISO_B, &tv3);
    if (fe_is_zero(&tv2)) tv4 = FE_Z; else fe_neg(&tv4, &tv2);
    fe_mul(&tv4, &FE_ISO_A, &tv4);
    fe_sqr(&tv2, &tv3); fe_sqr(&tv6, &tv4); fe_mul(&tv5, &FE_ISO_A, &tv6); fe_add(&tv2, &tv2, &tv5);
    fe_mul(&tv2, &tv2, &tv3); fe_mul(&tv6, &tv6, &tv4); fe_mul(&tv5, &FE_ISO_B, &tv6); fe_add(&tv2, &tv2, &tv5);
    fe_mul(&x, &tv1, &tv3);
    int sq = sqrt_ratio(&y1, &tv2, &tv6);
    fe_mul(&y, &tv1, u); fe_mul(&y, &y, &y1);
    if (sq) { x = tv3; y = y1; }
    if (fe_sgn0(u) != fe_sgn0(&y)) fe_neg(&y, &y);
    fe inv; fe_inv(&inv, &tv4); fe_mul(xo, &x, &inv); *yo = y;
}
static void poly(fe *r, const fe *c, int deg, const fe *x) {
    fe acc = c[deg];
    for (int i = deg - 1; i >= 0; i--) { fe_mul(&acc, &acc, x); fe_add(&acc, &acc, &c[i]); }
    *r = acc;
}
static void iso3(aff *r, const fe *x, const fe *y) { /* RFC 9380 E.1 */
    fe xn, xd, yn, yd, t;
    poly(&xn, XNUM, 3, x); poly(&xd, XDEN, 2, x); poly(&yn, YNUM, 3, x); poly(&yd, YDEN, 3, x);
    if (fe_is_zero(&xd) || fe_is_zero(&yd)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
    fe_inv(&t, &xd); fe_mul(&r->x, &xn, &t);
    fe_inv(&t, &yd); fe_mul(&r->y, &yn, &t); fe_mul(&r->y, &r->y, y);
    r->inf = 0;
}
/* h2c over m1||m2; optional intermediates u0,u1,q0,q1 */
static void hash_to_curve_raw(aff *out, const uint8_t *m1, size_t n1, const uint8_t *m2, size_t n2, fe u_out[2], aff q_out[2]) {
    uint8_t uni[96];
    fe u[2], x, y;
    aff q[2];
    jac j0, j1, s;
    expand_message_xmd96(uni, m1, n1, m2, n2);
    fe_from_be48(&u[0], uni); fe_from_be48(&u[1], uni + 48);
    for (int i = 0; i < 2; i++) { sswu(&x, &y, &u[i]); iso3(&q[i], &x, &y); }
    jac_from_aff(&j0, &q[0]); jac_from_aff(&j1, &q[1]); jac_add(&s, &j0, &j1);
    jac_to_aff(out, &s);
    if (u_out) { u_out[0] = u[0]; u_out[1] = u[1]; }
    if (q_out) { q_out[0] = q[0]; q_out[1] = q[1]; }
}
static void plume_h2c(aff *out, const uint8_t *msg, size_t mlen, const aff *pk) { /* utils.rs:11-20 */
    uint8_t enc[33];
    size_t n = sec1c(enc, pk);
    hash_to_curve_raw(out, msg, mlen, enc, n, 0, 0);
}

/* ------------------------------------------------------------------------------------------- c-hash */
static void c_hash(uint8_t out[32], int version, const aff *pk, const aff *h, const aff *nul, const aff *rp, const aff *hr) {
    /* lib.rs:159-168; order lib.rs:128-135 (V1) / :139-143 (V2) */
    sha256_ctx c;
    uint8_t e[33];
    aff g; g.x = FE_GX; g.y = FE_GY; g.inf = 0;
    sha256_init(&c);
    if (version == 1) {
        sha256_update(&c, e, sec1c(e, &g)); sha256_update(&c, e, sec1c(e, pk)); sha256_update(&c, e, sec1c(e, h));
    }
    sha256_update(&c, e, sec1c(e, nul)); sha256_update(&c, e, sec1c(e, rp)); sha256_update(&c, e, sec1c(e, hr));
    sha256_final(&c, out);
}
static int sc_from_be_nonzero(sc *r, const uint8_t b[32]) { from_be32(r->l, b); return !is_zero256(r->l) && !ge256(r->l, N_); }
static void sc_from_digest(sc *r, const uint8_t d[32], int *canonical) { /* Scalar::reduce (lib.rs:128) */
    from_be32(r->l, d);
    *canonical = !is_zero256(r->l) && !ge256(r->l, N_);
    if (ge256(r->l, N_)) sub256(r->l, r->l, N_);
}

/* ---------------------------------------------------------------------------------------- per item */
static int verify_one(int version, const uint8_t *msg, size_t mlen, const uint8_t *pk_b, const uint8_t *nul_b,
                      const uint8_t *c_b, const uint8_t *s_b, const uint8_t *r_b, const uint8_t *hr_b) {
    aff pk, nul, rp, hrp, h, ra, hra;
    sc c, s, negc, cc;
    if (!sc_from_be_nonzero(&c, c_b) || !sc_from_be_nonzero(&s, s_b)) return 0;       /* NonZeroScalar (lib.rs:75-77) */
    if (!aff_from_bytes(&pk, pk_b) || !aff_from_bytes(&nul, nul_b)) return 0;         /* AffinePoint invariant */
    if (version == 1 && (!aff_from_bytes(&rp, r_b) || !aff_from_bytes(&hrp, hr_b))) return 0;
    jac g, pkj, nulj, hj, t1, t2, rcalc, hrcalc;
    aff ga; ga.x = FE_GX; ga.y = FE_GY; ga.inf = 0;
    jac_from_aff(&g, &ga); jac_from_aff(&pkj, &pk); jac_from_aff(&nulj, &nul);
    sc_neg(&negc, &c);
    jac_mul(&t1, &s, &g); jac_mul(&t2, &c, &pkj); jac_neg(&t2, &t2); jac_add(&rcalc, &t1, &t2);      /* lib.rs:101 */
    plume_h2c(&h, msg, mlen, &pk);                                                                  /* lib.rs:103 */
    jac_from_aff(&hj, &h);
    jac_mul(&t1, &s, &hj); jac_mul(&t2, &c, &nulj); jac_neg(&t2, &t2); jac_add(&hrcalc, &t1, &t2);   /* lib.rs:109 */
    if (version == 1) {
        if (!jac_eq_aff(&rcalc, &rp)) return 0;                                                     /* lib.rs:117 */
        if (!jac_eq_aff(&hrcalc, &hrp)) return 0;                                                   /* lib.rs:122 */
    }
    jac_to_aff(&ra, &rcalc); jac_to_aff(&hra, &hrcalc);
    uint8_t d[32]; int canon;
    c_hash(d, version, &pk, &h, &nul, &ra, &hra);
    sc_from_digest(&cc, d, &canon);
    return memcmp(cc.l, c.l, 32) == 0;                                                              /* lib.rs:127-143 */
}

/* plume_arkworks' verify_non_zk (rust-arkworks/src/tests.rs:28-78): 1 = Ok(true), 0 = Ok(false), 2 = Err(HashToCurveError) (pk is the identity:
 * rust-arkworks/src/lib.rs:99-101).  s and digest_private are Fr elements (zero allowed); values the types cannot hold (>= n, coordinates >= p,
 * points off the curve) give 0. */
static int verify_non_zk_one(int version, const uint8_t *msg, size_t mlen, const uint8_t *pk_b, const uint8_t *nul_b, const uint8_t *s_b,
                             const uint8_t *r_b, const uint8_t *hr_b, const uint8_t *dp_b) {
    aff pk, nul, rp, hrp, h;
    sc c, s, cc;
    from_be32(s.l, s_b); from_be32(c.l, dp_b);
    if (ge256(s.l, N_) || ge256(c.l, N_)) return 0;
    if (!aff_from_bytes(&pk, pk_b) || !aff_from_bytes(&nul, nul_b) || !aff_from_bytes(&rp, r_b) || !aff_from_bytes(&hrp, hr_b)) return 0;
    if (pk.inf) return 2;                                                                           /* tests.rs:36 -> lib.rs:99-101 */
    plume_h2c(&h, msg, mlen, &pk);                                                                  /* tests.rs:36 */
    uint8_t d[32]; int canon;
    c_hash(d, version, &pk, &h, &nul, &rp, &hrp);                                                   /* tests.rs:40-51: the GIVEN r_point, hashed_to_curve_r */
    sc_from_digest(&cc, d, &canon);                                                                 /* tests.rs:52 from_be_bytes_mod_order */
    jac g, pkj, nulj, hj, t1, t2, rcalc, hrcalc;
    aff ga; ga.x = FE_GX; ga.y = FE_GY; ga.inf = 0;
    jac_from_aff(&g, &ga); jac_from_aff(&pkj, &pk); jac_from_aff(&nulj, &nul); jac_from_aff(&hj, &h);
    jac_mul(&t1, &s, &g); jac_mul(&t2, &c, &pkj); jac_neg(&t2, &t2); jac_add(&rcalc, &t1, &t2);      /* tests.rs:55-57 */
    if (!jac_eq_aff(&rcalc, &rp)) return 0;                                                         /* tests.rs:59-61 */
    jac_mul(&t1, &s, &hj); jac_mul(&t2, &c, &nulj); jac_neg(&t2, &t2); jac_add(&hrcalc, &t1, &t2);   /* tests.rs:64-66 */
    if (!jac_eq_aff(&hrcalc, &hrp)) return 0;                                                       /* tests.rs:68-70 */
    return memcmp(cc.l, c.l, 32) == 0;                                                              /* tests.rs:73-75 */
}

static void sign_one(int version, const uint8_t *msg, size_t mlen, const uint8_t *sk_b, const uint8_t *r_b, const uint8_t *pk_in,
                     uint8_t *pk_o, uint8_t *nul_o, uint8_t *c_o, uint8_t *s_o, uint8_t *r_o, uint8_t *hr_o, uint8_t *h_o, uint8_t *status) {
    sc sk, r, c, s, t;
    uint8_t st = 0;
    from_be32(sk.l, sk_b); from_be32(r.l, r_b);
    if (is_zero256(sk.l) || ge256(sk.l, N_) || is_zero256(r.l) || ge256(r.l, N_)) st |= 2;
    if (ge256(sk.l, N_)) sub256(sk.l, sk.l, N_);
    if (ge256(r.l, N_)) sub256(r.l, r.l, N_);
    aff ga, pk, rp, h, nul, hr; ga.x = FE_GX; ga.y = FE_GY; ga.inf = 0;
    jac g, hj, tj;
    jac_from_aff(&g, &ga);
    jac_mul(&tj, &r, &g); jac_to_aff(&rp, &tj);                                 /* randomizedsigner.rs:51 */
    if (pk_in) { if (!aff_from_bytes(&pk, pk_in)) { st |= 2; memset(&pk, 0, sizeof pk); pk.inf = 1; } }  /* arkworks: pk supplied (lib.rs:230,238) */
    else { jac_mul(&tj, &sk, &g); jac_to_aff(&pk, &tj); }                       /* :53 */
    plume_h2c(&h, msg, mlen, &pk);                                              /* :57-61 */
    if (h.inf) st |= 4;
    jac_from_aff(&hj, &h);
    jac_mul(&tj, &r, &hj); jac_to_aff(&hr, &tj);                                /* :67 */
    jac_mul(&tj, &sk, &hj); jac_to_aff(&nul, &tj);                              /* :70 */
    uint8_t d[32]; int canon;
    c_hash(d, version, &pk, &h, &nul, &rp, &hr);                                /* :73-89 */
    sc_from_digest(&c, d, &canon);
    if (!canon) st |= 1;                                                        /* :90-91 */
    sc_mul(&t, &c, &sk); sc_add(&s, &r, &t);                                    /* :94 */
    if (is_zero256(s.l)) st |= 4;                                               /* :95 */
    if (pk_o) aff_to_bytes(pk_o, &pk);
    aff_to_bytes(nul_o, &nul); to_be32(c_o, c.l); to_be32(s_o, s.l);
    aff_to_bytes(r_o, &rp); aff_to_bytes(hr_o, &hr);
    if (h_o) aff_to_bytes(h_o, &h);
    *status = st;
}

/* ------------------------------------------------------------------------------------ batch drivers */
typedef struct {
    int kind, version; size_t lo, hi;
    const uint8_t *msgs; const uint64_t *off;
    const uint8_t *a0, *a1, *a2, *a3, *a4, *a5;
    uint8_t *o0, *o1, *o2, *o3, *o4, *o5, *o6, *o7;
} job;
static void *worker(void *arg) {
    job *j = (job *)arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        const uint8_t *m = j->msgs + j->off[i];
        size_t ml = (size_t)(j->off[i + 1] - j->off[i]);
        if (j->kind == 0)
            j->o0[i] = (uint8_t)verify_one(j->version, m, ml, j->a0 + 64 * i, j->a1 + 64 * i, j->a2 + 32 * i, j->a3 + 32 * i,
                                           j->a4 ? j->a4 + 64 * i : 0, j->a5 ? j->a5 + 64 * i : 0);
        else if (j->kind == 1)
            sign_one(j->version, m, ml, j->a0 + 32 * i, j->a1 + 32 * i, j->a2 ? j->a2 + 64 * i : 0, j->o0 ? j->o0 + 64 * i : 0, j->o1 + 64 * i,
                     j->o2 + 32 * i, j->o3 + 32 * i, j->o4 + 64 * i, j->o5 + 64 * i, j->o6 ? j->o6 + 64 * i : 0, j->o7 + i);
        else if (j->kind == 3)
            j->o0[i] = (uint8_t)verify_non_zk_one(j->version, m, ml, j->a0 + 64 * i, j->a1 + 64 * i, j->a3 + 32 * i, j->a4 + 64 * i, j->a5 + 64 * i, j->a2 + 32 * i);
        else {
            aff pk, h;
            if (!aff_from_bytes(&pk, j->a0 + 64 * i)) { memset(j->o0 + 64 * i, 0, 64); continue; }
            plume_h2c(&h, m, ml, &pk);
            aff_to_bytes(j->o0 + 64 * i, &h);
        }
    }
    return 0;
}
static void run(job *proto, size_t n, int nthreads) {
    init_consts();
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    job *jobs = (job *)malloc(sizeof(job) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = *proto; jobs[t].lo = n * t / nthreads; jobs[t].hi = n * (t + 1) / nthreads;
        if (t > 0) pthread_create(&th[t], 0, worker, &jobs[t]);
    }
    worker(&jobs[0]);
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], 0);
    free(th); free(jobs);
}

/* same argument meaning as include/plume_hip.h's plume_verify_batch / plume_sign_batch (host buffers) */
int oracle_verify_batch(int version, size_t n, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *pk, const uint8_t *nullifier,
                        const uint8_t *c, const uint8_t *s, const uint8_t *r_point, const uint8_t *hashed_to_curve_r, uint8_t *ok, int nthreads) {
    if ((version != 1 && version != 2) || (version == 1 && (!r_point || !hashed_to_curve_r))) return -1;
    job j; memset(&j, 0, sizeof j);
    j.kind = 0; j.version = version; j.msgs = msgs; j.off = msg_off; j.a0 = pk; j.a1 = nullifier; j.a2 = c; j.a3 = s;
    j.a4 = version == 1 ? r_point : 0; j.a5 = version == 1 ? hashed_to_curve_r : 0; j.o0 = ok;
    run(&j, n, nthreads);
    return 0;
}
/* same argument meaning as plume_verify_non_zk_batch; ok: 1 Ok(true), 0 Ok(false), 2 Err */
int oracle_verify_non_zk_batch(int version, size_t n, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *pk, const uint8_t *nullifier,
                               const uint8_t *s, const uint8_t *r_point, const uint8_t *hashed_to_curve_r, const uint8_t *digest_private, uint8_t *ok, int nthreads) {
    if ((version != 1 && version != 2) || !r_point || !hashed_to_curve_r) return -1;
    job j; memset(&j, 0, sizeof j);
    j.kind = 3; j.version = version; j.msgs = msgs; j.off = msg_off; j.a0 = pk; j.a1 = nullifier; j.a2 = digest_private; j.a3 = s;
    j.a4 = r_point; j.a5 = hashed_to_curve_r; j.o0 = ok;
    run(&j, n, nthreads);
    return 0;
}
/* pk_in != NULL selects the arkworks-shaped sign_with_r (pk supplied, not derived); h_out optional */
int oracle_sign_batch(int version, size_t n, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sk, const uint8_t *r, const uint8_t *pk_in,
                      uint8_t *pk, uint8_t *nullifier, uint8_t *c, uint8_t *s, uint8_t *r_point, uint8_t *hashed_to_curve_r, uint8_t *h_out,
                      uint8_t *status, int nthreads) {
    if (version != 1 && version != 2) return -1;
    job j; memset(&j, 0, sizeof j);
    j.kind = 1; j.version = version; j.msgs = msgs; j.off = msg_off; j.a0 = sk; j.a1 = r; j.a2 = pk_in;
    j.o0 = pk; j.o1 = nullifier; j.o2 = c; j.o3 = s; j.o4 = r_point; j.o5 = hashed_to_curve_r; j.o6 = h_out; j.o7 = status;
    run(&j, n, nthreads);
    return 0;
}
/* ---------------------------------------------------------------------------- aggregate check (NOT a reference function)
 * SURVEY.md §8f rank 4's optional "aggregate-only random-linear-combination batch check".  The reference has no such function; what the oracle
 * restates here is its DEFINITION (include/plume_hip.h, plume_aggregate_check), item by item on the reference-shaped arithmetic above:
 *   an item whose inputs are not values of the reference's types (or, mode 1, whose pk is the identity) is "bad" and takes no part in the sum;
 *   hash_ok = the challenge over the GIVEN encodings equals c (lib.rs:128-135 / tests.rs:40-52);
 *   E1 = s*G - c*pk - r_point, E2 = s*H - c*nullifier - hashed_to_curve_r  (lib.rs:101,109,117,122 / tests.rs:55-70);
 *   a | b = SHA256(seed || be64(index_base + i)), each half a big-endian 128-bit integer with its top bit cleared;
 *   A = sum a*E1 + b*E2;   record = all_ok | A is identity | 0 | 0 | n_bad u32 LE | A affine (zeros = identity).
 * mode 0: PlumeSignature::verify types (c, s in [1, n-1]); mode 1: verify_non_zk types (Fr elements, zero allowed). */
typedef struct {
    int version, mode; size_t lo, hi;
    const uint8_t *msgs; const uint64_t *msg_off; const uint8_t *pk_b, *nul_b, *c_b, *s_b, *r_b, *hr_b, *seed; uint64_t index_base;
    uint8_t *hash_ok; jac tot; uint32_t nbad;
} agg_job;
static void *agg_worker(void *arg) {
    agg_job *J = (agg_job *)arg;
    const int version = J->version, mode = J->mode;
    jac tot; jac_set_inf(&tot);
    uint32_t nbad = 0;
    aff ga; ga.x = FE_GX; ga.y = FE_GY; ga.inf = 0;
    jac g; jac_from_aff(&g, &ga);
    for (size_t i = J->lo; i < J->hi; i++) {
        aff pk, nul, rp, hrp, h;
        sc c, s, cc;
        int ok;
        from_be32(c.l, J->c_b + 32 * i); from_be32(s.l, J->s_b + 32 * i);
        if (mode == 0) ok = sc_from_be_nonzero(&c, J->c_b + 32 * i) && sc_from_be_nonzero(&s, J->s_b + 32 * i);
        else ok = !ge256(c.l, N_) && !ge256(s.l, N_);
        ok = ok && J->msg_off[i + 1] >= J->msg_off[i];
        ok = ok && aff_from_bytes(&pk, J->pk_b + 64 * i) && aff_from_bytes(&nul, J->nul_b + 64 * i) && aff_from_bytes(&rp, J->r_b + 64 * i) && aff_from_bytes(&hrp, J->hr_b + 64 * i);
        if (ok && mode == 1 && pk.inf) ok = 0;
        if (!ok) { if (J->hash_ok) J->hash_ok[i] = 0; nbad++; continue; }
        plume_h2c(&h, J->msgs + J->msg_off[i], (size_t)(J->msg_off[i + 1] - J->msg_off[i]), &pk);
        uint8_t d[32]; int canon;
        c_hash(d, version, &pk, &h, &nul, &rp, &hrp);
        sc_from_digest(&cc, d, &canon);
        const int hok = memcmp(cc.l, c.l, 32) == 0;
        if (J->hash_ok) J->hash_ok[i] = (uint8_t)hok;
        if (!hok) nbad++;
        jac pkj, nulj, hj, rj, hrj, t1, t2, e1, e2;
        jac_from_aff(&pkj, &pk); jac_from_aff(&nulj, &nul); jac_from_aff(&hj, &h); jac_from_aff(&rj, &rp); jac_from_aff(&hrj, &hrp);
        jac_mul(&t1, &s, &g); jac_mul(&t2, &c, &pkj); jac_neg(&t2, &t2); jac_add(&e1, &t1, &t2); jac_neg(&rj, &rj); jac_add(&e1, &e1, &rj);
        jac_mul(&t1, &s, &hj); jac_mul(&t2, &c, &nulj); jac_neg(&t2, &t2); jac_add(&e2, &t1, &t2); jac_neg(&hrj, &hrj); jac_add(&e2, &e2, &hrj);
        uint8_t pre[40], dg[32];
        memcpy(pre, J->seed, 32);
        const uint64_t idx = J->index_base + (uint64_t)i;
        for (int k = 0; k < 8; k++) pre[32 + k] = (uint8_t)(idx >> (8 * (7 - k)));
        sha256_ctx sh; sha256_init(&sh); sha256_update(&sh, pre, 40); sha256_final(&sh, dg);
        uint8_t ab[32]; sc a, b;
        memset(ab, 0, 32); memcpy(ab + 16, dg, 16); ab[16] &= 0x7F; from_be32(a.l, ab);
        memset(ab, 0, 32); memcpy(ab + 16, dg + 16, 16); ab[16] &= 0x7F; from_be32(b.l, ab);
        jac_mul(&t1, &a, &e1); jac_add(&tot, &tot, &t1);
        jac_mul(&t2, &b, &e2); jac_add(&tot, &tot, &t2);
    }
    J->tot = tot; J->nbad = nbad;
    return 0;
}
/* nthreads only splits the item range; the partial sums are added in range order */
int oracle_aggregate_check(int version, int mode, size_t n, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *pk_b, const uint8_t *nul_b,
                           const uint8_t *c_b, const uint8_t *s_b, const uint8_t *r_b, const uint8_t *hr_b, const uint8_t seed[32], uint64_t index_base,
                           uint8_t *hash_ok, uint8_t result[72], int nthreads) {
    if ((version != 1 && version != 2) || (mode != 0 && mode != 1) || (mode == 0 && version != 1)) return -1;
    init_consts();
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    agg_job *jobs = (agg_job *)malloc(sizeof(agg_job) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        agg_job *J = &jobs[t];
        J->version = version; J->mode = mode; J->lo = n * t / nthreads; J->hi = n * (t + 1) / nthreads;
        J->msgs = msgs; J->msg_off = msg_off; J->pk_b = pk_b; J->nul_b = nul_b; J->c_b = c_b; J->s_b = s_b; J->r_b = r_b; J->hr_b = hr_b; J->seed = seed;
        J->index_base = index_base; J->hash_ok = hash_ok;
        if (t > 0) pthread_create(&th[t], 0, agg_worker, J);
    }
    agg_worker(&jobs[0]);
    jac tot = jobs[0].tot;
    uint32_t nbad = jobs[0].nbad;
    for (int t = 1; t < nthreads; t++) { pthread_join(th[t], 0); jac_add(&tot, &tot, &jobs[t].tot); nbad += jobs[t].nbad; }
    free(th); free(jobs);
    aff ta; jac_to_aff(&ta, &tot);
    memset(result, 0, 72);
    result[0] = (uint8_t)(ta.inf && nbad == 0);
    result[1] = (uint8_t)(ta.inf != 0);
    result[4] = (uint8_t)nbad; result[5] = (uint8_t)(nbad >> 8); result[6] = (uint8_t)(nbad >> 16); result[7] = (uint8_t)(nbad >> 24);
    aff_to_bytes(result + 8, &ta);
    return 0;
}
int oracle_hash_to_curve_batch(size_t n, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *pk, uint8_t *h_out, int nthreads) {
    job j; memset(&j, 0, sizeof j);
    j.kind = 2; j.msgs = msgs; j.off = msg_off; j.a0 = pk; j.o0 = h_out;
    run(&j, n, nthreads);
    return 0;
}
/* raw h2c over arbitrary bytes with intermediates (RFC vector): out = u0|u1 (32 B each) | q0 | q1 | P (64 B each) = 256 B */
void oracle_h2c_raw(const uint8_t *data, size_t len, uint8_t out[256]) {
    init_consts();
    fe u[2]; aff q[2], p;
    hash_to_curve_raw(&p, data, len, 0, 0, u, q);
    to_be32(out, u[0].l); to_be32(out + 32, u[1].l);
    aff_to_bytes(out + 64, &q[0]); aff_to_bytes(out + 128, &q[1]); aff_to_bytes(out + 192, &p);
}
/* k*P on 64-byte points (test hook); returns 0 on invalid point */
int oracle_point_mul(const uint8_t k[32], const uint8_t p[64], uint8_t out[64]) {
    init_consts();
    aff a, r; sc s; jac j, t;
    if (!aff_from_bytes(&a, p)) return 0;
    from_be32(s.l, k);
    while (ge256(s.l, N_)) sub256(s.l, s.l, N_);
    jac_from_aff(&j, &a); jac_mul(&t, &s, &j); jac_to_aff(&r, &t); aff_to_bytes(out, &r);
    return 1;
}
size_t oracle_sec1_compress(const uint8_t p[64], uint8_t out[33]) {
    aff a;
    init_consts();
    if (!aff_from_bytes(&a, p)) return 0;
    return sec1c(out, &a);
}
void oracle_sha256(const uint8_t *data, size_t len, uint8_t out[32]) { sha256_ctx c; sha256_init(&c); sha256_update(&c, data, len); sha256_final(&c, out); }
