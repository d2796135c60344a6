/* PLUME verify on the CPU with a THIRD PARTY's group arithmetic: OpenSSL libcrypto (NID_secp256k1, EC_POINT_mul).  TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * SURVEY.md §8d names this as the optional independent CPU leg beside the oracle: the reference's own CPU path (rust-k256) cannot be built here (no rustc / cargo), and the two
 * CPU legs in oracle/ were written by the same hands as the GPU code.  Here every scalar multiplication, point addition and point comparison of PlumeSignature::verify
 * (rust-k256/src/lib.rs:93-145) is OpenSSL's; only what OpenSSL does not have comes from the plain oracle it is compiled with (plume_oracle.c): hash_to_curve (RFC 9380),
 * SHA-256 over the SEC1 encodings, the byte formats.  Inputs with an identity point or anything the reference's types cannot hold take the plain oracle's path.
 * tests/test_openssl_leg.py holds its verdicts to the plain oracle's; bench.py's cpu_baseline times it ("openssl").  Nothing under zk-nullifier-sig_amd/ loads it.
 */
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/obj_mac.h>

#include "plume_oracle.c"

typedef struct { EC_GROUP *g; BN_CTX *ctx; BIGNUM *order; } ossl;

static int ossl_point(const ossl *o, EC_POINT *p, const aff *a) {
    uint8_t xb[32], yb[32];
    to_be32(xb, a->x.l); to_be32(yb, a->y.l);
    BIGNUM *x = BN_bin2bn(xb, 32, NULL), *y = BN_bin2bn(yb, 32, NULL);
    const int ok = EC_POINT_set_affine_coordinates(o->g, p, x, y, o->ctx);      /* also checks the curve equation */
    BN_free(x); BN_free(y);
    return ok == 1;
}
static int ossl_to_aff(const ossl *o, aff *a, const EC_POINT *p) {
    if (EC_POINT_is_at_infinity(o->g, p)) { memset(a, 0, sizeof *a); a->inf = 1; return 1; }
    BIGNUM *x = BN_new(), *y = BN_new();
    uint8_t xb[32], yb[32];
    const int ok = EC_POINT_get_affine_coordinates(o->g, p, x, y, o->ctx) == 1 && BN_bn2binpad(x, xb, 32) == 32 && BN_bn2binpad(y, yb, 32) == 32;
    BN_free(x); BN_free(y);
    if (!ok) return 0;
    from_be32(a->x.l, xb); from_be32(a->y.l, yb); a->inf = 0;
    return 1;
}
static int ossl_verify_one(const ossl *o, int version, const uint8_t *msg, size_t mlen, const uint8_t *pk_b, const uint8_t *nul_b, const uint8_t *c_b, const uint8_t *s_b,
                           const uint8_t *r_b, const uint8_t *hr_b) {
    static const uint8_t zero64[64] = {0};
    if (!memcmp(pk_b, zero64, 64) || !memcmp(nul_b, zero64, 64) || (version == 1 && (!memcmp(r_b, zero64, 64) || !memcmp(hr_b, zero64, 64))))
        return verify_one(version, msg, mlen, pk_b, nul_b, c_b, s_b, r_b, hr_b);
    aff pk, nul, rp, hrp, h, ra, hra;
    sc c, s, cc;
    if (!sc_from_be_nonzero(&c, c_b) || !sc_from_be_nonzero(&s, s_b)) return 0;
    if (!aff_from_bytes(&pk, pk_b) || !aff_from_bytes(&nul, nul_b)) return 0;
    if (version == 1 && (!aff_from_bytes(&rp, r_b) || !aff_from_bytes(&hrp, hr_b))) return 0;
    plume_h2c(&h, msg, mlen, &pk);                                                                  /* lib.rs:103 */
    if (h.inf) return verify_one(version, msg, mlen, pk_b, nul_b, c_b, s_b, r_b, hr_b);
    int res = 0;
    BIGNUM *bs = BN_bin2bn(s_b, 32, NULL), *bc = BN_bin2bn(c_b, 32, NULL), *negc = BN_new();
    EC_POINT *P = EC_POINT_new(o->g), *N = EC_POINT_new(o->g), *H = EC_POINT_new(o->g), *R = EC_POINT_new(o->g), *T1 = EC_POINT_new(o->g), *T2 = EC_POINT_new(o->g), *G2 = EC_POINT_new(o->g);
    if (!BN_sub(negc, o->order, bc)) goto done;
    if (!ossl_point(o, P, &pk) || !ossl_point(o, N, &nul) || !ossl_point(o, H, &h)) goto done;
    if (EC_POINT_mul(o->g, R, bs, P, negc, o->ctx) != 1) goto done;                                  /* s*G - c*pk           lib.rs:101 */
    if (EC_POINT_mul(o->g, T1, NULL, H, bs, o->ctx) != 1 || EC_POINT_mul(o->g, T2, NULL, N, negc, o->ctx) != 1 || EC_POINT_add(o->g, T1, T1, T2, o->ctx) != 1) goto done;   /* lib.rs:109 */
    if (version == 1) {
        if (!ossl_point(o, G2, &rp) || EC_POINT_cmp(o->g, R, G2, o->ctx) != 0) goto done;            /* lib.rs:117 */
        if (!ossl_point(o, G2, &hrp) || EC_POINT_cmp(o->g, T1, G2, o->ctx) != 0) goto done;          /* lib.rs:122 */
        ra = rp; hra = hrp;
    } else if (!ossl_to_aff(o, &ra, R) || !ossl_to_aff(o, &hra, T1)) goto done;
    {
        uint8_t d[32]; int canon;
        c_hash(d, version, &pk, &h, &nul, &ra, &hra);                                               /* lib.rs:127-143 */
        sc_from_digest(&cc, d, &canon);
        res = memcmp(cc.l, c.l, 32) == 0;
    }
done:
    EC_POINT_free(P); EC_POINT_free(N); EC_POINT_free(H); EC_POINT_free(R); EC_POINT_free(T1); EC_POINT_free(T2); EC_POINT_free(G2);
    BN_free(bs); BN_free(bc); BN_free(negc);
    return res;
}

typedef struct { int version; size_t lo, hi; const uint8_t *msgs; const uint64_t *off; const uint8_t *pk, *nul, *c, *s, *r, *hr; uint8_t *ok; } ojob;
static void *ossl_worker(void *arg) {
    ojob *j = (ojob *)arg;
    ossl o;
    o.g = EC_GROUP_new_by_curve_name(NID_secp256k1); o.ctx = BN_CTX_new(); o.order = BN_new();      /* one group / context per thread */
    EC_GROUP_get_order(o.g, o.order, o.ctx);
    for (size_t i = j->lo; i < j->hi; i++)
        j->ok[i] = (uint8_t)ossl_verify_one(&o, j->version, j->msgs + j->off[i], (size_t)(j->off[i + 1] - j->off[i]), j->pk + 64 * i, j->nul + 64 * i, j->c + 32 * i, j->s + 32 * i,
                                            j->r ? j->r + 64 * i : 0, j->hr ? j->hr + 64 * i : 0);
    BN_free(o.order); BN_CTX_free(o.ctx); EC_GROUP_free(o.g);
    return 0;
}
/* same arguments as oracle_verify_batch / plume_verify_batch (host buffers) */
int ossl_verify_batch(int version, size_t n, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *pk, const uint8_t *nullifier, const uint8_t *c, const uint8_t *s,
                      const uint8_t *r_point, const uint8_t *hashed_to_curve_r, uint8_t *ok, int nthreads) {
    if ((version != 1 && version != 2) || (version == 1 && (!r_point || !hashed_to_curve_r))) return -1;
    init_consts();
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    ojob *jobs = (ojob *)malloc(sizeof(ojob) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        ojob j = {version, n * t / nthreads, n * (t + 1) / nthreads, msgs, msg_off, pk, nullifier, c, s, version == 1 ? r_point : 0, version == 1 ? hashed_to_curve_r : 0, ok};
        jobs[t] = j;
        if (t > 0) pthread_create(&th[t], 0, ossl_worker, &jobs[t]);
    }
    ossl_worker(&jobs[0]);
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], 0);
    free(th); free(jobs);
    return 0;
}
