// Per-lane bodies of the PLUME batch pipeline.  The __global__ kernels in plume_kernels.hip are thin wrappers
// that map (blockIdx, threadIdx) to an item / job / task index and call these; tests/devsim calls the same
// bodies from plain host loops so the exact device logic is checked on the CPU against the oracle.
//
// verify (rust-k256/src/lib.rs:93-145), n items:
//   S1 ingest_h2c   lane = item     validate inputs; H = h2c(m || enc(pk)) (Jacobian); publish the table jobs of the item: 3i = pk, 3i+1 = H,
//                                   3i+2 = nullifier (and 3n+i = R when equation 1 runs in its short form)
//   S1' scalars     lane = item     every digit row the multi-scalar stage reads: Eisenstein digits of s and -c; equation 1's own rows (plume_eis.h)
//   S2 tables       lane = L jobs   window tables P, theta P, 2P (+ beta*x), one inversion per 8 lanes
//   S3 msm          lane = task     task 2i: R' = s*G - c*pk (or k*G - upsilon*pk - (tau-1)*R) ;  task 2i+1: Hr' = s*H - c*nul   (Jacobian results)
//   S4 finalize     lane = item     V1: R' == r_point, Hr' == hashed_to_curve_r, c == SHA256(G,pk,H,nul,R,Hr) mod n
//                                   V2: c == SHA256(nul, R', Hr') mod n
// sign (rust-k256/src/randomizedsigner.rs:43-112; arkworks flavour rust-arkworks/src/lib.rs:229-278), n items:
//   G1 sign_gmul    lane = task     task 2i: pk = sk*G ; task 2i+1: R = r*G
//   G2 sign_h2c     lane = item     pk -> affine (or caller-supplied), H = h2c(m || enc(pk)); table job i = H
//   S2 tables       (as above, 2 jobs per item: H and 2^64 H)
//   G3 sign_hmul    lane = task     task 2i: nullifier = sk*H ; task 2i+1: Hr = r*H
//   G4 sign_final   lane = item     affine outputs, c = SHA256(..) mod n, s = r + sk*c, status bits
#pragma once
#include "plume_h2c.h"
#include "plume_eis.h"

namespace plume {

// ------------------------------------------------------------------------------------------- point ingest
// 64-byte affine x||y big-endian, all-zero = identity (include/plume_hip.h).  flag: 0 ok, 1 identity, 2 invalid
// x, y come out CANONICAL (their limbs are exactly the 256-bit integers read, which are checked to be < p)
PLUME_HD uint32_t load_affine_be(fe& x, fe& y, const uint8_t* p) {
    uint32_t wx[8], wy[8];
    words_from_be_aligned(wx, p);
    words_from_be_aligned(wy, p + 32);
    fe_from_words(x, wx);
    fe_from_words(y, wy);
    uint32_t nz = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) nz |= wx[i] | wy[i];
    if (nz == 0) return PLUME_JOB_INF;
    if (!words_lt_p(wx) || !words_lt_p(wy)) return PLUME_JOB_INVALID;
    return affine_on_curve(x, y) ? PLUME_JOB_OK : PLUME_JOB_INVALID;
}
// the same record when an earlier stage of the same call has already validated it (the item would carry a reject flag otherwise): no curve equation
PLUME_HD uint32_t reload_affine_be(fe& x, fe& y, const uint8_t* p) {
    uint32_t wx[8], wy[8];
    words_from_be_aligned(wx, p);
    words_from_be_aligned(wy, p + 32);
    fe_from_words(x, wx);
    fe_from_words(y, wy);
    uint32_t nz = 0;
    PLUME_UNROLL for (int i = 0; i < 8; i++) nz |= wx[i] | wy[i];
    return nz == 0 ? PLUME_JOB_INF : PLUME_JOB_OK;
}
PLUME_HD void store_affine_be(uint8_t* p, fe x, fe y, bool inf) {
    fe_normalize(x); fe_normalize(y);
    if (inf) { x = fe_zero(); y = fe_zero(); }
    fe_to_be_aligned(p, x);
    fe_to_be_aligned(p + 32, y);
}
// SEC1-compressed record: 02|03 || x, or 00 followed by 32 zero bytes for the identity (33-byte stride: byte stores)
PLUME_HD void store_sec1_be(uint8_t* p, fe x, fe y, bool inf) {
    fe_normalize(x);
    uint32_t w[8];
    fe_to_words(w, x);
    p[0] = (uint8_t)(inf ? 0u : 2u + (fe_is_odd(y) ? 1u : 0u));
    PLUME_UNROLL for (int i = 0; i < 8; i++) {
        const uint32_t v = inf ? 0u : w[7 - i];
        p[1 + 4 * i] = (uint8_t)(v >> 24); p[2 + 4 * i] = (uint8_t)(v >> 16); p[3 + 4 * i] = (uint8_t)(v >> 8); p[4 + 4 * i] = (uint8_t)v;
    }
}
PLUME_HD void store_point_be(uint8_t* base, size_t i, int out33, const fe& x, const fe& y, bool inf) {
    if (out33) store_sec1_be(base + 33 * i, x, y, inf); else store_affine_be(base + 64 * i, x, y, inf);
}
// scalar in [1, n-1]?
PLUME_HD bool load_scalar_be(sc& k, const uint8_t* p) {
    sc_from_be_aligned(k, p);
    return !sc_is_zero(k) && sc_lt_n(k);
}

// --------------------------------------------------------------------------------------------- c-hash
// One SEC1-compressed operand of the c-hash: canonical x, parity of y, or the identity (encoded as the single byte 00;
// rust-k256/src/utils.rs:23-25, rust-arkworks/src/lib.rs:112-118)
struct enc_pt {
    uint32_t xw[8];  // canonical x as little-endian 32-bit words
    uint32_t tag;    // 2 | 3, or 0 for the identity
};
PLUME_HD enc_pt enc_of(fe x, const fe& y, bool inf) {
    enc_pt e;
    fe_normalize(x);
    fe_to_words(e.xw, x);
    e.tag = inf ? 0u : (2u + (fe_is_odd(y) ? 1u : 0u));
    return e;
}
// SHA256 over npts encodings (order given by the caller).  Fast path: no identity among them -> static layout of
// 33-byte records; slow path: generic byte stream.
template <int NPTS>
PLUME_HD void c_hash(uint32_t out[8], const enc_pt* pts) {
    bool any_inf = false;
    PLUME_UNROLL for (int i = 0; i < NPTS; i++) any_inf |= (pts[i].tag == 0);
    sha256_init(out);
    if (!any_inf) {
        constexpr int LEN = 33 * NPTS, NW = ((LEN + 9 + 63) / 64) * 16;
        uint32_t w[NW];
        PLUME_UNROLL for (int i = 0; i < NW; i++) w[i] = 0;
        PLUME_UNROLL for (int i = 0; i < NPTS; i++) {
            // record i occupies bytes [33i, 33i+33): tag then x big-endian; static shifts after unrolling
            PLUME_UNROLL for (int k = 0; k < 33; k++) {
                const int pos = 33 * i + k;
                uint32_t byte = k == 0 ? pts[i].tag : ((pts[i].xw[7 - ((k - 1) >> 2)] >> (8 * (3 - ((k - 1) & 3)))) & 0xFF);
                w[pos >> 2] |= byte << (8 * (3 - (pos & 3)));
            }
        }
        w[LEN >> 2] |= 0x80u << (8 * (3 - (LEN & 3)));
        w[NW - 1] = LEN * 8;
        PLUME_UNROLL for (int b = 0; b < NW / 16; b++) sha256_compress(out, w + 16 * b);
    } else {
        uint32_t len = 0;
        PLUME_UNROLL for (int i = 0; i < NPTS; i++) len += pts[i].tag ? 33u : 1u;
        sha256_absorb_pad(out, 0u, len, [&](uint32_t pos) -> uint32_t {
            uint32_t v = 0, base = 0;
            PLUME_UNROLL for (int i = 0; i < NPTS; i++) {
                uint32_t l = pts[i].tag ? 33u : 1u;
                if (pos >= base && pos < base + l) v = (pos == base) ? pts[i].tag : be_byte_of_limbs(pts[i].xw, pos - base - 1);
                base += l;
            }
            return v;
        });
    }
}

// message i = msgs[msg_off[i] .. msg_off[i+1]): false (and an empty span) when the offsets decrease, reach past the buffer or span more than the
// 32-bit length the SHA-256 front end takes -- the lane then never touches msgs
PLUME_HD bool msg_span(uint64_t& o0, uint32_t& len, const uint64_t* msg_off, size_t i, uint64_t msgs_bytes) {
    o0 = msg_off[i];
    const uint64_t o1 = msg_off[i + 1];
    const bool ok = o1 >= o0 && o1 <= msgs_bytes && o1 - o0 <= 0xFFFFFF00ull;
    len = ok ? (uint32_t)(o1 - o0) : 0u;
    if (!ok) o0 = 0;
    return ok;
}

// ===================================================================================================== verify
#define PLUME_MODE_VERIFY 0   // PlumeSignature::verify (rust-k256/src/lib.rs:93-145)
#define PLUME_MODE_NON_ZK 1   // plume_arkworks' verify_non_zk (rust-arkworks/src/tests.rs:28-78): c' hashed from the GIVEN r_point / hashed_to_curve_r,
                              // both EC equations checked for V1 AND V2, scalars are Fr elements (zero allowed), pk = identity is Err(HashToCurveError)
#define PLUME_ITEM_REJECT 1u  // itemflags: the item's inputs cannot be represented by the reference's types -> ok = 0
#define PLUME_ITEM_ERR 2u     // itemflags (non-zk mode): the reference returns Err (pk is the identity, rust-arkworks/src/lib.rs:99-101) -> ok = 2
struct VerifyArgs {
    int version;       // 1 | 2
    int mode;          // PLUME_MODE_*
    uint32_t n;
    // caller arrays (device memory): SoA of big-endian records
    const uint8_t* msgs; const uint64_t* msg_off;
    uint64_t msgs_bytes;   // size of msgs: an item whose offsets are decreasing or reach past it is rejected, never read
    const uint8_t *pk, *nul, *c, *s, *rpt, *hr;
    uint8_t* ok;
    const uint8_t* preflags;  // optional, n bytes: non-zero = reject (set by the SEC1 decompression stage)
    const uint8_t *rpt33, *hr33;   // optional (V1, SEC1 ingest): r_point / hashed_to_curve_r as 33-byte records, used INSTEAD of rpt / hr:
                                   // they are only compared with computed points and hashed, which needs x and the parity -- no square root
    // scratch (device memory)
    uint32_t* bases;      // 3n job-major records of PLUME_BASE_WORDS words (plume_ec.h st_base): jobs 3i, 3i+1, 3i+2 = pk, H, nullifier of item i
    uint8_t* jobflags;    // 3n
    uint8_t* itemflags;   // n : 1 = rejected at ingest (bad scalar / invalid point)
    uint32_t* tab;        // 3n tables of PLUME_TAB_WORDS
    uint32_t* res;        // PLUME_JAC_WORDS x (2n) words, Jacobian SoA of R' (task 2i) and Hr' (task 2i+1)
    uint8_t* resinf;      // 2n
    const uint32_t* gtab; // wide table of G (PLUME_GTAB_WORDS): (1..2^(W-1))*G
    uint32_t* redo;       // redo[0] = number of tasks whose unchecked chain met p == +-q, redo[1 + k] = the k-th such task (2i + eq); capacity 2n; zeroed before the multi-scalar kernel
    int8_t* digs;         // PLUME_VDIG_ROWS x n signed window digits, row-major (row r of item i at digs[r * n + i]: a wavefront reads / writes 64 consecutive bytes per row).
                          // Written once per item by the scalar stage (verify_scalars; rounds 4: by the ingest kernel; rounds 1-3: each multi-scalar lane split s and c again)
    // Calls that GIVE r_point as a 64-byte record (V1 verify, verify_non_zk) may run equation 1 in its SHORT form (plume_eis.h).  They do iff eq1fall is set
    // (verify_eq1_short below); every item of such a call then has a flag here:
    uint8_t* eq1fall;     // n : 0 = the item's equation 1 runs in the short form (digit set B holds the 64-bit coefficients, eq1k the generator's scalar), 1 = the item FALLS
                          //     BACK to the long form (set B holds s in the generator's wide digits): the half-GCD's coefficients did not fit (no such input is known) or the
                          //     caller forces it (tests).  NULL = the whole call runs the long form
    uint32_t* eq1k;       // 8 x n words, word-major: k = tau s mod n, multiplied by G through the comb
    const uint32_t* gcomb;  // the doubling-free comb of G (PLUME_COMB_WORDS), shared with the signer
    int msm_pair;            // calls of a few thousand items: the long-form chains run as two halves on two lanes (k_verify_msm_pair; verify_msm_half below)
    int scalars_in_ingest;   // the two-role ingest kernel (small calls) runs verify_scalars in its role B: no k_verify_scalars launch for this call (plume_kernels.hip)
    unsigned long long* clk;   // optional (stage timing on): clk[0] += shader-clock cycles, clk[1] += constant-rate wall-clock ticks that sampled workgroups of the multi-scalar kernel lived
                               // for -- their ratio is the clock the kernel ran at (plume_last_msm_clock): what turns a kernel time into a box-independent cycle count
    int eq1force;         // test knob: 1 = file every item's equation 1 as "long form" (everything then runs through the redo launch's checked chain)
};
// jobs: 3 per item (pk, H, nullifier) at 3i, 3i+1, 3i+2; in the short form a fourth, R, at 3n + i -- behind the others so that the job kinds of the first 3n stay aligned
// across the lanes of the table passes and the last n are all of one kind (affine)
PLUME_HD bool verify_eq1_short(const VerifyArgs& a) { return a.eq1fall != nullptr; }      // does this call run equation 1 in the short form?
PLUME_HD size_t verify_njobs(const VerifyArgs& a) { return (verify_eq1_short(a) ? 4 : 3) * (size_t)a.n; }
// the digit rows of an item: three sets of 2 x PLUME_NDIG rows (the two halves of a GLV split): s in 4-bit windows (equation 2), s with the generator's wide digits
// (equation 1), -c in 4-bit windows (both equations)
#define PLUME_VDIG_SET PLUME_NPOS                    // sets A and C: the Eisenstein digits of one GLV pair, one row per position
#define PLUME_VDIG_SETB (2 * PLUME_NPOS66)          // set B: s in the generator's wide digits (2 x 33 rows: long form) OR the two coefficient pairs of the short form (2 x 34)
#define PLUME_VDIG_ROWS (2 * PLUME_VDIG_SET + PLUME_VDIG_SETB)
static_assert(PLUME_VDIG_SETB >= 2 * PLUME_NDIG, "set B holds either form");
// One crafted item (pk = +-k G with small k, s = +-c, ...) steers its accumulator into p == +-q inside an UNCHECKED addition.  Rounds 1-2 redid such a lane on the spot with
// the checked additions -- and its 63 neighbours waited: one crafted item per wavefront doubled the kernel.  Now the lane only files its task; a second,
// dense launch (k_verify_msm_redo: one filed task per lane, grid-stride) redoes the filed tasks.  Honest batches file nothing and the second launch costs its launch; a batch
// salted with one crafted item per wavefront pays 1/64 of the kernel again instead of all of it; a batch made ENTIRELY of crafted items pays the checked chain once more
// (the worst case is bounded by ~2.2x, and that batch is all rejects or all self-inflicted).
PLUME_HD uint32_t redo_file(uint32_t* redo, uint32_t task) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t k = atomicAdd(redo, 1u);
#else
    const uint32_t k = redo[0]++;
#endif
    redo[1 + k] = task;
    return k;
}

// the digit rows of one item (d = the item's column of the row-major digit array, n = its row pitch): set A = the Eisenstein digits of s = k1 + k2 lambda (equation 2's H),
// set B = equation 1's own rows (below), set C = the digits of -c (equation 2's nullifier, and pk in equation 1's long form)
PLUME_HD void verify_item_digits(int8_t* d, uint32_t n, const sc& s, const glv_half& c1, const glv_half& c2, bool long_b) {
    glv_half h1, h2;
    glv_split(h1, h2, s);
    eisd_store_glv(d, n, h1, h2, false);
    if (long_b) { booth_store_wide(d + (size_t)PLUME_VDIG_SET * n, n, h1, false); booth_store_wide(d + (size_t)(PLUME_VDIG_SET + PLUME_NDIG) * n, n, h2, false); }
    eisd_store_glv(d + (size_t)(PLUME_VDIG_SET + PLUME_VDIG_SETB) * n, n, c1, c2, true);
}
// The scalar stage: every digit row the multi-scalar kernel reads, once per item.  Short form of equation 1 (verify_eq1_short(a); plume_eis.h): (tau, upsilon) from the
// half-GCD of c, k = tau s mod n for the comb, and set B = the Eisenstein digits of -upsilon (pk's joint slot) and of -(tau - 1) (R's).
PLUME_HD void verify_scalars(const VerifyArgs& a, uint32_t i) {
    sc c, s;
    bool okc = load_scalar_be(c, a.c + 32 * (size_t)i), oks = load_scalar_be(s, a.s + 32 * (size_t)i);
    if (a.mode == PLUME_MODE_NON_ZK) { okc = sc_lt_n(c); oks = sc_lt_n(s); }
    if (!okc || !oks) { if (verify_eq1_short(a)) a.eq1fall[i] = 1; return; }            // (the ingest stage rejects the item; its digit rows are never read)
    bool lng = true;
    glv_half c1, c2;
    glv_split(c1, c2, c);                                                     // once: the half-GCD starts from it, the digits of -c are its
    if (verify_eq1_short(a)) {
        eis_short e;
        eis_half_gcd(e, c1, c2);
        lng = !e.ok || a.eq1force != 0 || !eis_consistent(e, c);
        if (!lng) {
            sc k;
            sc_mul(k, e.tau, s);
            PLUME_UNROLL for (int w = 0; w < 8; w++) a.eq1k[(size_t)w * a.n + i] = k.v[w];
            int8_t* b = a.digs + (size_t)PLUME_VDIG_SET * a.n + i;
            (void)eisd_store<PLUME_NPOS66, 3>(b, a.n, e.u[0], e.uneg[0] != 0, e.u[1], e.uneg[1] != 0, true);                                   // - upsilon
            (void)eisd_store<PLUME_NPOS66, 3>(b + (size_t)PLUME_NPOS66 * a.n, a.n, e.t[0], e.tneg[0] != 0, e.t[1], e.tneg[1] != 0, true);      // - (tau - 1)
        }
        a.eq1fall[i] = lng ? 1 : 0;
    }
    verify_item_digits(a.digs + i, a.n, s, c1, c2, lng);
}

PLUME_HD void verify_ingest_h2c(const VerifyArgs& a, uint32_t i) {
    fe pkx, pky, nx, ny;
    uint32_t fpk = load_affine_be(pkx, pky, a.pk + 64 * (size_t)i);
    uint32_t fnul = load_affine_be(nx, ny, a.nul + 64 * (size_t)i);
    sc c, s;
    bool okc = load_scalar_be(c, a.c + 32 * (size_t)i), oks = load_scalar_be(s, a.s + 32 * (size_t)i);
    if (a.mode == PLUME_MODE_NON_ZK) { okc = sc_lt_n(c); oks = sc_lt_n(s); }          // Fr elements: zero is a value
    uint64_t o0; uint32_t mlen;
    const bool okm = msg_span(o0, mlen, a.msg_off, i, a.msgs_bytes);
    bool bad = !okc || !oks || !okm || fpk == PLUME_JOB_INVALID || fnul == PLUME_JOB_INVALID || (a.preflags && a.preflags[i]);
    const bool err = a.mode == PLUME_MODE_NON_ZK && !bad && fpk == PLUME_JOB_INF;      // hash_to_curve(message, pk)? -> Err
    a.itemflags[i] = (uint8_t)(bad ? PLUME_ITEM_REJECT : err ? PLUME_ITEM_ERR : 0u);
    bad = bad || err;
    // (The window digits of s and c are the scalar stage's work since round 5: verify_scalars.)
    // Short form of equation 1: R is a base of the multi-scalar chain, so it is validated HERE (an r_point that is no curve point rejects the item, as it does in the
    // finalize stage of the long form) and becomes job 3n + i of the table stage.
    if (verify_eq1_short(a)) {
        fe rx, ry;
        const uint32_t fr = load_affine_be(rx, ry, a.rpt + 64 * (size_t)i);
        if (fr == PLUME_JOB_INVALID && !err) { bad = true; a.itemflags[i] = (uint8_t)PLUME_ITEM_REJECT; }
        jac p; p.inf = 0; p.z = fe_small(1); p.x = rx; p.y = ry;
        st_base(a.bases, 3 * (size_t)a.n + i, p); a.jobflags[3 * (size_t)a.n + i] = (uint8_t)(fr | PLUME_JOB_AFFINE);
    }
    // the records of pk and the nullifier go out BEFORE hash_to_curve: nothing below needs y(nullifier) or more of pk than its x and parity, and 27 registers less are
    // live through the two exponentiations (the kernel spills at its 128-register budget)
    {
        jac p; p.inf = 0; p.z = fe_small(1);
        p.x = nx; p.y = ny;
        st_base(a.bases, 3 * (size_t)i + 2, p); a.jobflags[3 * (size_t)i + 2] = (uint8_t)(fnul | PLUME_JOB_AFFINE);
        p.x = pkx; p.y = pky;
        st_base(a.bases, 3 * (size_t)i + 0, p); a.jobflags[3 * (size_t)i + 0] = (uint8_t)(fpk | PLUME_JOB_AFFINE);
    }
    const uint32_t pktag = 2u + (fe_is_odd(pky) ? 1u : 0u);
    jac h;
    if (!bad) {
        hash_to_curve_jac(h, a.msgs + o0, mlen, pkx, pktag, fpk == PLUME_JOB_INF ? PLUME_ENC_IDENTITY : PLUME_ENC_POINT);
    } else {
        h.x = fe_gx(); h.y = fe_gy(); h.z = fe_small(1); h.inf = 0;
    }
    st_base(a.bases, 3 * (size_t)i + 1, h); a.jobflags[3 * (size_t)i + 1] = (uint8_t)(h.inf ? PLUME_JOB_INF : PLUME_JOB_OK);
}

// ---- the ingest stage in TWO ROLES per item (round 4, small batches) ----------------------------------------------------------------------------------------------
// Up to 2^16 items the ingest kernel is latency-bound (measured: 2^14 items 0.239 -> 0.150 ms, 2^16 0.273 -> 0.235, 2^17 no change): one wavefront per SIMD walks 96 k dependent instructions, 61 k of them the two square-root exponentiations of the
// two simplified-SWU maps, which do not depend on each other.  Here an item is served by two lanes in DIFFERENT wavefronts of a workgroup (roles are wave-uniform: no
// divergence): role A takes pk, the message and expand_message_xmd and maps u0; role B takes the nullifier, c, s, the window digits and maps u1; A adds the two results on
// E', runs the isogeny and stores H.  They meet twice (ingest_xch, through LDS on the device, with a workgroup barrier each time).  Same outputs, byte for byte.
struct ingest_xch {
    fe u1;            // A -> B at the first meeting
    uint32_t fb;      // B -> A at the first meeting: nullifier flag | bad scalar << 8
    fe xn, xd, y;     // B -> A at the second meeting: map(u1) on E', x as a fraction
};
struct ingest_a_state { fe u0; fe c, d, y1; uint32_t fpk; bool bad_a, bad; };
// role B, before the first meeting: nullifier record, scalar range checks (the window digits are the scalar stage's: verify_scalars)
PLUME_HD void verify_ingest_b1(const VerifyArgs& a, uint32_t i, ingest_xch& x) {
    fe nx, ny;
    const uint32_t fnul = load_affine_be(nx, ny, a.nul + 64 * (size_t)i);
    sc c, s;
    bool okc = load_scalar_be(c, a.c + 32 * (size_t)i), oks = load_scalar_be(s, a.s + 32 * (size_t)i);
    if (a.mode == PLUME_MODE_NON_ZK) { okc = sc_lt_n(c); oks = sc_lt_n(s); }
    jac p; p.inf = 0; p.z = fe_small(1); p.x = nx; p.y = ny;
    st_base(a.bases, 3 * (size_t)i + 2, p); a.jobflags[3 * (size_t)i + 2] = (uint8_t)(fnul | PLUME_JOB_AFFINE);
    bool okr = true;
    if (verify_eq1_short(a)) {                                                          // short form of equation 1: R is validated here and becomes job 3n + i (verify_ingest_h2c)
        fe rx, ry;
        const uint32_t fr = load_affine_be(rx, ry, a.rpt + 64 * (size_t)i);
        okr = fr != PLUME_JOB_INVALID;
        p.x = rx; p.y = ry;
        st_base(a.bases, 3 * (size_t)a.n + i, p); a.jobflags[3 * (size_t)a.n + i] = (uint8_t)(fr | PLUME_JOB_AFFINE);
    }
    x.fb = fnul | ((okc && oks && okr) ? 0u : 0x100u);
}
// role A, before the first meeting: pk record, message span, hash_to_field
PLUME_HD void verify_ingest_a1(const VerifyArgs& a, uint32_t i, ingest_xch& x, ingest_a_state& st) {
    fe pkx, pky;
    st.fpk = load_affine_be(pkx, pky, a.pk + 64 * (size_t)i);
    uint64_t o0; uint32_t mlen;
    const bool okm = msg_span(o0, mlen, a.msg_off, i, a.msgs_bytes);
    st.bad_a = !okm || st.fpk == PLUME_JOB_INVALID || (a.preflags && a.preflags[i]);
    jac p; p.inf = 0; p.z = fe_small(1); p.x = pkx; p.y = pky;
    st_base(a.bases, 3 * (size_t)i + 0, p); a.jobflags[3 * (size_t)i + 0] = (uint8_t)(st.fpk | PLUME_JOB_AFFINE);
    const uint32_t pktag = 2u + (fe_is_odd(pky) ? 1u : 0u);
    if (!st.bad_a) {
        hash_to_field2(st.u0, x.u1, a.msgs + o0, mlen, pkx, pktag, st.fpk == PLUME_JOB_INF ? PLUME_ENC_IDENTITY : PLUME_ENC_POINT);
    } else {                                                                   // (a malformed span is never read)
        st.u0 = fe_small(1); x.u1 = fe_small(1);
    }
}
// role A after the first meeting: the verdict of the checks, then map(u0)
PLUME_HD void verify_ingest_a2(const VerifyArgs& a, uint32_t i, const ingest_xch& x, ingest_a_state& st) {
    bool bad = st.bad_a || (x.fb & 0x100u) != 0 || (x.fb & 0xFFu) == PLUME_JOB_INVALID;
    const bool err = a.mode == PLUME_MODE_NON_ZK && !bad && st.fpk == PLUME_JOB_INF;
    a.itemflags[i] = (uint8_t)(bad ? PLUME_ITEM_REJECT : err ? PLUME_ITEM_ERR : 0u);
    st.bad = bad || err;
    sswu_frac(st.c, st.d, st.y1, st.u0);
}
// role B after the first meeting: map(u1)
PLUME_HD void verify_ingest_b2(ingest_xch& x) {
    const fe u = x.u1;
    sswu_frac(x.xn, x.xd, x.y, u);
}
// role A after the second meeting: Q0' + Q1' on E', the isogeny, H
PLUME_HD void verify_ingest_a3(const VerifyArgs& a, uint32_t i, const ingest_xch& x, const ingest_a_state& st) {
    jac h;
    if (!st.bad) {
        maps_to_curve_jac(h, x.xn, x.xd, x.y, st.c, st.d, st.y1);
    } else {
        h.x = fe_gx(); h.y = fe_gy(); h.z = fe_small(1); h.inf = 0;
    }
    st_base(a.bases, 3 * (size_t)i + 1, h); a.jobflags[3 * (size_t)i + 1] = (uint8_t)(h.inf ? PLUME_JOB_INF : PLUME_JOB_OK);
}

// task t = 2*item + eq;  eq 0: s*G - c*pk, eq 1: s*H - c*nul.   dig: this lane's digit area (LDS), element stride.
// CHECKED = false: the hot form; a task whose chain met p == +-q is filed in a.redo and stores nothing.  CHECKED = true: the redo launch's form.
// FORM: which forms of equation 1 the instantiation carries -- 0: the long form only (calls whose equation 1 runs in the long form), 1: the short form only (the hot kernel of a short-form call:
// an item the scalar stage left in the long form is filed for the redo launch), 2: both (the redo launch, host harness)
template <bool CHECKED, int FORM = 2>
PLUME_HD void verify_msm(const VerifyArgs& a, uint32_t item, uint32_t eq, const uint32_t* gtab, int8_t* dig, uint32_t stride) {
    const size_t nt = 2 * (size_t)a.n, t = 2 * (size_t)item + eq;
    jac acc;
    const bool short_call = FORM == 1 || (FORM == 2 && verify_eq1_short(a));
    if (a.itemflags[item]) {
        acc.x = fe_small(1); acc.y = fe_small(1); acc.z = fe_small(0); acc.inf = 1;
    } else if (FORM != 0 && eq == 0 && short_call && !a.eq1fall[item]) {
        // Equation 1, short form (plume_eis.h): k G - upsilon pk - (tau - 1) R, to be compared with R by the finalize stage.  Two joint slots, pk and R, thirty-four positions
        // (66 doublings at most; leading all-zero positions are skipped); then the generator's term from the doubling-free comb, fifteen additions.
        const int8_t* db = a.digs + (size_t)PLUME_VDIG_SET * a.n + item;
        PLUME_UNROLL for (int r = 0; r < PLUME_VDIG_SETB; r++) dig[(uint32_t)r * stride] = db[(size_t)r * a.n];
        const size_t jp = 3 * (size_t)item, jr = 3 * (size_t)a.n + item;
        const uint32_t* tabp = job_state(a.jobflags[jp]) == PLUME_JOB_OK ? a.tab + jp * PLUME_TAB_WORDS : nullptr;
        const uint32_t* tabr = job_state(a.jobflags[jr]) == PLUME_JOB_OK ? a.tab + jr * PLUME_TAB_WORDS : nullptr;
        msm_run_impl<CHECKED, PLUME_NPOS66>(acc, tabp, tabr, dig, stride, false);
        sc k;                                                                 // (loaded after the chain: eight registers the chain does not have to carry)
        PLUME_UNROLL for (int w = 0; w < 8; w++) k.v[w] = a.eq1k[(size_t)w * a.n + item];
        comb_add_g<CHECKED>(acc, k, a.gcomb);
        if (!CHECKED && !acc.inf && fe_is_zero(acc.z)) { redo_file(a.redo, (uint32_t)t); return; }
    } else if (FORM == 1 && eq == 0) {
        redo_file(a.redo, (uint32_t)t);                                       // a long-form item of a short-form call (never on known inputs): the redo launch takes it
        return;
    } else {
        if (FORM == 2 && !CHECKED && eq == 0 && short_call) { redo_file(a.redo, (uint32_t)t); return; }
        // the rows the scalar stage left for this item, into the lane's digit area: s first (equation 1: its wide digits for the generator's table, 2 x 33 rows; equation 2: its
        // Eisenstein digits for H, 65 rows), then the Eisenstein digits of -c (pk / the nullifier)
        const int8_t* dc = a.digs + (size_t)(PLUME_VDIG_SET + PLUME_VDIG_SETB) * a.n + item;
        if (eq == 0) {
            const int8_t* ds = a.digs + (size_t)PLUME_VDIG_SET * a.n + item;
            PLUME_UNROLL for (int r = 0; r < 2 * PLUME_NDIG; r++) dig[(uint32_t)r * stride] = ds[(size_t)r * a.n];
            PLUME_UNROLL for (int r = 0; r < PLUME_NPOS; r++) dig[(uint32_t)(2 * PLUME_NDIG + r) * stride] = dc[(size_t)r * a.n];
        } else {
            const int8_t* ds = a.digs + item;
            PLUME_UNROLL for (int r = 0; r < PLUME_NPOS; r++) { dig[(uint32_t)r * stride] = ds[(size_t)r * a.n]; dig[(uint32_t)(PLUME_NPOS + r) * stride] = dc[(size_t)r * a.n]; }
        }
        const size_t ja = 3 * (size_t)item + 1, jb = 3 * (size_t)item + (eq ? 2 : 0);
        const uint32_t* tab0 = eq ? (job_state(a.jobflags[ja]) == PLUME_JOB_OK ? a.tab + ja * PLUME_TAB_WORDS : nullptr) : gtab;
        const uint32_t* tab1 = job_state(a.jobflags[jb]) == PLUME_JOB_OK ? a.tab + jb * PLUME_TAB_WORDS : nullptr;
        if (CHECKED) {
            msm_run_checked(acc, tab0, tab1, dig, stride, eq == 0);
        } else if (!msm_run_unchecked(acc, tab0, tab1, dig, stride, eq == 0)) {
            redo_file(a.redo, (uint32_t)t);
            return;
        }
    }
    st_jac_soa(a.res, nt, t, acc);
    a.resinf[t] = (uint8_t)acc.inf;
}

// ---- calls of a few thousand items (k_verify_msm_pair, round 5) ----------------------------------------------------------------------------------------------------
// Such a call leaves most SIMDs one wavefront or none: the multi-scalar kernel's time is the LATENCY of one chain, 130 doublings + 122 additions for equation 2.  There the
// long-form chain of a task is cut in two -- half 0 walks the first joint slot alone (equation 1: the generator's wide digits; equation 2: H), half 1 the second (pk / the
// nullifier): 130 doublings + 61 additions each, on two lanes of different wavefronts -- and the halves are joined by one checked addition.  More work in total (the doublings
// run twice), so only where the machine is empty: VerifyArgs::msm_pair, set by the host for calls of at most 2^14 items.
// Returns false when the unchecked half met p == +-q (the join then files the task for the redo launch, which runs the whole chain with checked additions as ever).
PLUME_HD bool verify_msm_half(const VerifyArgs& a, uint32_t item, uint32_t eq, uint32_t half, const uint32_t* gtab, int8_t* dig, uint32_t stride, jac& acc) {
    acc.x = fe_small(1); acc.y = fe_small(1); acc.z = fe_small(0); acc.inf = 1;
    if (a.itemflags[item]) return true;
    const int8_t* dc = a.digs + (size_t)(PLUME_VDIG_SET + PLUME_VDIG_SETB) * a.n + item;
    if (eq == 0) {       // rows as verify_msm lays them out: s's wide digits first, then the digits of -c
        const int8_t* ds = a.digs + (size_t)PLUME_VDIG_SET * a.n + item;
        if (half == 0) { PLUME_UNROLL for (int r = 0; r < 2 * PLUME_NDIG; r++) dig[(uint32_t)r * stride] = ds[(size_t)r * a.n]; }
        else { PLUME_UNROLL for (int r = 0; r < PLUME_NPOS; r++) dig[(uint32_t)(2 * PLUME_NDIG + r) * stride] = dc[(size_t)r * a.n]; }
    } else {
        const int8_t* ds = a.digs + item;
        if (half == 0) { PLUME_UNROLL for (int r = 0; r < PLUME_NPOS; r++) dig[(uint32_t)r * stride] = ds[(size_t)r * a.n]; }
        else { PLUME_UNROLL for (int r = 0; r < PLUME_NPOS; r++) dig[(uint32_t)(PLUME_NPOS + r) * stride] = dc[(size_t)r * a.n]; }
    }
    const size_t ja = 3 * (size_t)item + 1, jb = 3 * (size_t)item + (eq ? 2 : 0);
    const uint32_t* tab0 = eq ? (job_state(a.jobflags[ja]) == PLUME_JOB_OK ? a.tab + ja * PLUME_TAB_WORDS : nullptr) : gtab;
    const uint32_t* tab1 = job_state(a.jobflags[jb]) == PLUME_JOB_OK ? a.tab + jb * PLUME_TAB_WORDS : nullptr;
    return msm_run_unchecked(acc, half == 0 ? tab0 : nullptr, half == 1 ? tab1 : nullptr, dig, stride, eq == 0);
}
// the join, by half 0's lane: acc0 += acc1 with the checked Jacobian addition (the two halves may well meet in p == +-q: a valid equation 2 whose halves are H-multiples)
PLUME_HD void verify_msm_join(const VerifyArgs& a, uint32_t item, uint32_t eq, jac& acc0, bool ok0, const jac& acc1, bool ok1) {
    const size_t nt = 2 * (size_t)a.n, t = 2 * (size_t)item + eq;
    if (!ok0 || !ok1) { redo_file(a.redo, (uint32_t)t); return; }
    jac_add(acc0, acc1);
    st_jac_soa(a.res, nt, t, acc0);
    a.resinf[t] = (uint8_t)acc0.inf;
}

PLUME_HD void verify_finalize(const VerifyArgs& a, uint32_t i) {
    const size_t nt = 2 * (size_t)a.n;
    uint8_t ok = 0;
    if (a.itemflags[i] == PLUME_ITEM_ERR) {
        // Err(HashToCurveError) needs every argument to be a value of its type first: an r_point / hashed_to_curve_r that is no curve point rejects
        fe x, y;
        ok = (load_affine_be(x, y, a.rpt + 64 * (size_t)i) != PLUME_JOB_INVALID && load_affine_be(x, y, a.hr + 64 * (size_t)i) != PLUME_JOB_INVALID) ? 2 : 0;
    }
    if (!a.itemflags[i]) {
        jac rc, hc;
        ld_jac_soa(rc, a.res, nt, 2 * (size_t)i); rc.inf = a.resinf[2 * (size_t)i];
        ld_jac_soa(hc, a.res, nt, 2 * (size_t)i + 1); hc.inf = a.resinf[2 * (size_t)i + 1];
        fe pkx, pky, nx, ny;
        uint32_t fpk = reload_affine_be(pkx, pky, a.pk + 64 * (size_t)i);     // validated by verify_ingest_h2c (itemflags == 0 here)
        uint32_t fnul = reload_affine_be(nx, ny, a.nul + 64 * (size_t)i);
        sc c;
        sc_from_be_aligned(c, a.c + 32 * (size_t)i);
        uint32_t dg[8];
        bool hashed = false;
        if (a.version == 1 && a.rpt33) {
            // SEC1 records for R and Hr.  The reference decodes them (failing on a bad tag, x >= p, or an x with no curve point) and then
            // compares with the computed points: a computed point IS on the curve, so "x matches and the parity matches" already implies that
            // the record decodes, and the comparison needs the computed points in affine form: one shared inversion instead of two square roots.
            const uint8_t* r33 = a.rpt33 + 33 * (size_t)i;
            const uint8_t* h33 = a.hr33 + 33 * (size_t)i;
            uint32_t rw[8], hw[8];
            words_from_be(rw, r33 + 1); words_from_be(hw, h33 + 1);
            const uint32_t rtag = r33[0], htag = h33[0];
            const bool rfmt = rtag == 0u || ((rtag == 2u || rtag == 3u) && words_lt_p(rw));
            const bool hfmt = htag == 0u || ((htag == 2u || htag == 3u) && words_lt_p(hw));
            bool match = rfmt && hfmt && (rc.inf != 0) == (rtag == 0u) && (hc.inf != 0) == (htag == 0u);
            fe rxa = fe_zero(), rya = fe_zero(), hxa = fe_zero(), hya = fe_zero();
            if (match) {
                // 1 / (Zr * Zh) -> both inverses (an identity contributes Z = 1)
                fe zr = rc.inf ? fe_small(1) : rc.z, zh = hc.inf ? fe_small(1) : hc.z, zz, zi, t, t2;
                fe_mul(zz, zr, zh); fe_inv(zz, zz);
                fe_mul(zi, zz, zh);                               // 1 / Zr
                fe_sqr(t, zi); fe_mul(rxa, rc.x, t); fe_mul(t2, t, zi); fe_mul(rya, rc.y, t2);
                fe_mul(zi, zz, zr);                               // 1 / Zh
                fe_sqr(t, zi); fe_mul(hxa, hc.x, t); fe_mul(t2, t, zi); fe_mul(hya, hc.y, t2);
                fe gx, gh;
                fe_from_words(gx, rw); fe_from_words(gh, hw);
                if (!rc.inf) match = match && fe_eq(rxa, gx) && (uint32_t)fe_is_odd(rya) == (rtag & 1u);
                if (!hc.inf) match = match && fe_eq(hxa, gh) && (uint32_t)fe_is_odd(hya) == (htag & 1u);
            }
            if (match) {                                                                                           // lib.rs:117,122
                const size_t jh = 3 * (size_t)i + 1;
                bool hinf = job_state(a.jobflags[jh]) == PLUME_JOB_INF;
                fe Hx, Hy;
                ld_tab_xy(Hx, Hy, a.tab + jh * PLUME_TAB_WORDS, false);
                enc_pt pts[6];
                pts[0] = enc_of(fe_gx(), fe_gy(), false);
                pts[1] = enc_of(pkx, pky, fpk == PLUME_JOB_INF);
                pts[2] = enc_of(Hx, Hy, hinf);
                pts[3] = enc_of(nx, ny, fnul == PLUME_JOB_INF);
                pts[4] = enc_of(rxa, rya, rc.inf != 0);
                pts[5] = enc_of(hxa, hya, hc.inf != 0);
                c_hash<6>(dg, pts);                                                                                  // lib.rs:128-135
                hashed = true;
            }
        } else if (a.version == 1 || a.mode == PLUME_MODE_NON_ZK) {
            // the given R, Hr are compared with the computed points (lib.rs:117,122; verify_non_zk tests.rs:55-70) and the challenge is hashed
            // from the GIVEN encodings (non-zk: tests.rs:40-51, for V2 too)
            fe rx, ry, hx, hy;
            uint32_t fr = load_affine_be(rx, ry, a.rpt + 64 * (size_t)i);
            uint32_t fh = load_affine_be(hx, hy, a.hr + 64 * (size_t)i);
            if (fr != PLUME_JOB_INVALID && fh != PLUME_JOB_INVALID &&
                jac_eq_affine(rc, rx, ry, fr == PLUME_JOB_INF) && jac_eq_affine(hc, hx, hy, fh == PLUME_JOB_INF)) {   // lib.rs:117,122
                // affine H = entry 0 of H's table; identity H has no table
                const size_t jh = 3 * (size_t)i + 1;
                bool hinf = job_state(a.jobflags[jh]) == PLUME_JOB_INF;
                fe Hx, Hy;
                ld_tab_xy(Hx, Hy, a.tab + jh * PLUME_TAB_WORDS, false);
                enc_pt pts[6];
                pts[0] = enc_of(fe_gx(), fe_gy(), false);
                pts[1] = enc_of(pkx, pky, fpk == PLUME_JOB_INF);
                pts[2] = enc_of(Hx, Hy, hinf);
                pts[3] = enc_of(nx, ny, fnul == PLUME_JOB_INF);
                pts[4] = enc_of(rx, ry, fr == PLUME_JOB_INF);
                pts[5] = enc_of(hx, hy, fh == PLUME_JOB_INF);
                if (a.version == 1) c_hash<6>(dg, pts);                                                              // lib.rs:128-135
                else c_hash<3>(dg, pts + 3);                                                                         // compute_c_v2(nul, r_point, hashed_to_curve_r)
                hashed = true;
            }
        } else {
            // R', Hr' were made affine by the batched conversion stage (normalize_points): X, Y are the coordinates
            enc_pt pts[3];
            pts[0] = enc_of(nx, ny, fnul == PLUME_JOB_INF);
            pts[1] = enc_of(rc.x, rc.y, rc.inf != 0);
            pts[2] = enc_of(hc.x, hc.y, hc.inf != 0);
            c_hash<3>(dg, pts);                                                                                      // lib.rs:139-143
            hashed = true;
        }
        if (hashed) {
            sc cc; bool canon;
            sc_from_digest_words(cc, dg, canon);                                                                     // Scalar::reduce
            uint32_t diff = 0;
            PLUME_UNROLL for (int k = 0; k < 8; k++) diff |= cc.v[k] ^ c.v[k];
            ok = diff == 0 ? 1 : 0;
        }
    }
    a.ok[i] = ok;
}

// ======================================================================================================= sign
#define PLUME_ST_C_NOT_CANONICAL 1u  // digest == 0 or >= n (k256 sign panics there, randomizedsigner.rs:90-91; arkworks reduces, lib.rs:257)
#define PLUME_ST_BAD_SCALAR 2u       // sk or r outside [1, n-1], or a supplied pk that is not a curve point
#define PLUME_ST_IDENTITY 4u         // H == identity (randomizedsigner.rs:61) or s == 0 (:95)

struct SignArgs {
    int version;
    uint32_t n;
    const uint8_t* msgs; const uint64_t* msg_off;
    uint64_t msgs_bytes;    // size of msgs (see VerifyArgs)
    const uint8_t *sk, *r;
    const uint8_t* pk_in;   // optional (arkworks-shaped sign_with_r: pk supplied, not derived)
    uint8_t *pk, *nul, *c, *s, *rpt, *hr, *status;
    uint8_t* h_out;         // optional 64 B/item
    int out33;              // 0: pk, nul, rpt, hr are 64-byte affine records; 1: 33-byte SEC1-compressed records (stride 33)
    // scratch
    uint32_t* gres;  uint8_t* gresinf;   // 2n tasks: sk*G, r*G (Jacobian SoA)
    uint32_t* bases; uint8_t* jobflags;  // PLUME_SIGN_K n jobs: H of item i at i, 2^(j PLUME_SIGN_BITS) H at j n + i (the signer's chains are PLUME_SIGN_BITS doublings long, sign_hdbl)
    uint8_t* itemflags;                  // n: status bits accumulated across stages
    uint32_t* pkaff;                     // 2 * PLUME_FE_WORDS x n words SoA: affine pk (x, y), canonical, for the final stage
    uint32_t* tab;                       // PLUME_SIGN_K n tables, same order
    uint32_t* hres;  uint8_t* hresinf;   // 2n tasks: sk*H, r*H
    const uint32_t* gcomb;               // fixed-base comb of G (PLUME_COMB_WORDS)
    const uint32_t* gscan;               // level 2 only: the small scanned table of G (PLUME_GSCAN_WORDS)
    int uniform;                         // plume_set_sign_uniform: 1 = the uniform-schedule kernels (no branch on a secret digit), 2 = and no table address from a secret digit
};

// scalars reduced mod n for the arithmetic, status bit if out of range (the Rust types cannot hold such values)
PLUME_HD uint32_t load_scalar_reduced(sc& k, const uint8_t* p) {
    sc_from_be_aligned(k, p);
    bool ok = !sc_is_zero(k) && sc_lt_n(k);
    sc_cond_sub_n(k);
    return ok ? 0u : PLUME_ST_BAD_SCALAR;
}
// task t = 2*item + which: which 0 -> sk, 1 -> r;  result = k * H.
// k = k1 + k2 lambda (GLV), each 128-bit half cut into PLUME_SIGN_K pieces of PLUME_SIGN_BITS bits: piece j of the pair on the table of 2^(j PLUME_SIGN_BITS) H (job j n + item)
// -- K joint slots along one chain of PLUME_SIGN_BITS doublings (plume_ec.h; K = 2 in rounds 4-5, K = 4 since round 6.  The doublings that make the shifted bases are spent
// ONCE per item and serve both of its multiplications, sk * H and r * H: sign_hdbl).
template <int UNIFORM = 0>
PLUME_HD void sign_mul(const SignArgs& a, uint32_t item, uint32_t which, bool live, uint32_t* res, uint8_t* resinf, int8_t* dig, uint32_t stride) {
    const size_t nt = 2 * (size_t)a.n, t = 2 * (size_t)item + which;
    sc k;
    (void)load_scalar_reduced(k, (which ? a.r : a.sk) + 32 * (size_t)item);
    glv_half h1, h2;
    glv_split(h1, h2, k);
    constexpr int WPP = PLUME_SIGN_BITS / 32;                                           // words per piece
    PLUME_UNROLL for (int j = 0; j < PLUME_SIGN_K; j++) {                              // the Eisenstein digits of piece j's pair, for the table of 2^(j PLUME_SIGN_BITS) H
        uint32_t p1[WPP], p2[WPP];
        PLUME_UNROLL for (int w = 0; w < WPP; w++) { p1[w] = h1.m[j * WPP + w]; p2[w] = h2.m[j * WPP + w]; }
        (void)eisd_store<PLUME_NPOSK, WPP>(dig + (uint32_t)(j * PLUME_NPOSK) * stride, stride, p1, h1.neg != 0, p2, h2.neg != 0, false);
    }
    const uint32_t* t0 = a.tab + (size_t)item * PLUME_TAB_WORDS;                      // (every job has a table: a dummy one when its base was no usable point)
    const size_t ts = (size_t)a.n * PLUME_TAB_WORDS;                                  // job j n + item: the next shifted base's table
    jac acc;
    if (UNIFORM == 2) msm_run_uniform<PLUME_NPOSK, true, PLUME_SIGN_K>(acc, t0, ts, live, dig, stride);
    else if (UNIFORM == 1) msm_run_uniform<PLUME_NPOSK, false, PLUME_SIGN_K>(acc, t0, ts, live, dig, stride);
    else msm_runk<PLUME_NPOSK, PLUME_SIGN_K>(acc, live ? t0 : nullptr, ts, dig, stride);
    st_jac_soa(res, nt, t, acc);
    resinf[t] = (uint8_t)acc.inf;
}
// The shifted bases of item i next to H (job i): 2^(j PLUME_SIGN_BITS) H at job j n + i, j = 1 .. K - 1 -- PLUME_SIGN_BITS doublings of the Jacobian point per hop, once per
// item.  A job whose H is no usable point (identity: a status bit, randomizedsigner.rs:61) gets G under the same flag, like every other placeholder base.
PLUME_HD void sign_hdbl(const SignArgs& a, uint32_t i) {
    const uint8_t f = a.jobflags[i];
    jac h;
    const bool ok = job_state(f) == PLUME_JOB_OK;
    if (ok) { ld_base(h, a.bases, i, true); h.inf = 0; }
    else { h.x = fe_gx(); h.y = fe_gy(); h.z = fe_small(1); h.inf = 0; }
    PLUME_NOUNROLL for (uint32_t j = 1; j < (uint32_t)PLUME_SIGN_K; j++) {
        if (ok) { PLUME_NOUNROLL for (int d = 0; d < PLUME_SIGN_BITS; d++) jac_dbl_neg(h); }      // an even number of sign-flipping doublings
        st_base(a.bases, (size_t)j * a.n + i, h);
        a.jobflags[(size_t)j * a.n + i] = f;
    }
}
// task t = 2*item + which: sk*G (which 0) or r*G (which 1) by the doubling-free comb
template <int UNIFORM = 0>
PLUME_HD void sign_gmul(const SignArgs& a, uint32_t item, uint32_t which) {
    const size_t nt = 2 * (size_t)a.n, t = 2 * (size_t)item + which;
    if (which == 0 && a.pk_in) { a.gresinf[t] = 1; return; }   // pk supplied: sk*G not needed (flagged so the affine conversion skips it)
    sc k;
    (void)load_scalar_reduced(k, (which ? a.r : a.sk) + 32 * (size_t)item);
    jac acc;
    if (UNIFORM == 2) comb_mul_g_scan(acc, k, a.gscan);
    else if (UNIFORM == 1) comb_mul_g_uniform(acc, k, a.gcomb);
    else comb_mul_g(acc, k, a.gcomb);
    st_jac_soa(a.gres, nt, t, acc);
    a.gresinf[t] = (uint8_t)acc.inf;
}
PLUME_HD void sign_h2c(const SignArgs& a, uint32_t i) {
    const size_t nt = 2 * (size_t)a.n;
    uint32_t st = 0;
    sc tmp;
    st |= load_scalar_reduced(tmp, a.sk + 32 * (size_t)i);
    st |= load_scalar_reduced(tmp, a.r + 32 * (size_t)i);
    fe px, py;
    bool pinf;
    if (a.pk_in) {
        uint32_t f = load_affine_be(px, py, a.pk_in + 64 * (size_t)i);
        if (f == PLUME_JOB_INVALID) { st |= PLUME_ST_BAD_SCALAR; f = PLUME_JOB_INF; }
        pinf = f == PLUME_JOB_INF;
    } else {
        jac p;
        ld_jac_soa(p, a.gres, nt, 2 * (size_t)i); p.inf = a.gresinf[2 * (size_t)i];
        pinf = p.inf != 0;
        px = p.x; py = p.y;   // affine already (normalize_points ran on gres)
        fe_normalize(px); fe_normalize(py);
    }
    if (pinf) { px = fe_zero(); py = fe_zero(); }
    st_fe_soa(a.pkaff, a.n, i, px); st_fe_soa(a.pkaff + PLUME_FE_W * (size_t)a.n, a.n, i, py);
    jac h;
    uint64_t o0; uint32_t mlen;
    if (!msg_span(o0, mlen, a.msg_off, i, a.msgs_bytes)) st |= PLUME_ST_BAD_SCALAR;   // malformed offsets: flagged, the (empty) span is hashed
    hash_to_curve_jac(h, a.msgs + o0, mlen, px, 2u + (fe_is_odd(py) ? 1u : 0u), pinf ? PLUME_ENC_IDENTITY : PLUME_ENC_POINT);
    if (h.inf) st |= PLUME_ST_IDENTITY;
    st_base(a.bases, i, h);
    a.jobflags[i] = (uint8_t)(h.inf ? PLUME_JOB_INF : PLUME_JOB_OK);
    a.itemflags[i] = (uint8_t)(st | (pinf ? 0x80u : 0u));
}
template <int UNIFORM = 0>
PLUME_HD void sign_hmul(const SignArgs& a, uint32_t item, uint32_t which, int8_t* dig, uint32_t stride) {
    sign_mul<UNIFORM>(a, item, which, job_state(a.jobflags[item]) == PLUME_JOB_OK, a.hres, a.hresinf, dig, stride);
}
PLUME_HD void sign_final(const SignArgs& a, uint32_t i) {
    const size_t nt = 2 * (size_t)a.n;
    uint32_t st = a.itemflags[i] & 0x7Fu;
    bool pinf = (a.itemflags[i] & 0x80u) != 0;
    jac R, nul, hr;
    ld_jac_soa(R, a.gres, nt, 2 * (size_t)i + 1); R.inf = a.gresinf[2 * (size_t)i + 1];
    ld_jac_soa(nul, a.hres, nt, 2 * (size_t)i); nul.inf = a.hresinf[2 * (size_t)i];
    ld_jac_soa(hr, a.hres, nt, 2 * (size_t)i + 1); hr.inf = a.hresinf[2 * (size_t)i + 1];
    // R, nullifier, Hr are affine already (normalize_points ran on gres and hres)
    fe px, py, Hx, Hy;
    ld_fe_soa(px, a.pkaff, a.n, i); ld_fe_soa(py, a.pkaff + PLUME_FE_W * (size_t)a.n, a.n, i);
    bool hinf = job_state(a.jobflags[i]) == PLUME_JOB_INF;
    ld_tab_xy(Hx, Hy, a.tab + (size_t)i * PLUME_TAB_WORDS, false);
    uint32_t dg[8];
    enc_pt e_nul = enc_of(nul.x, nul.y, nul.inf != 0), e_r = enc_of(R.x, R.y, R.inf != 0), e_hr = enc_of(hr.x, hr.y, hr.inf != 0);
    if (a.version == 1) {
        enc_pt pts[6] = {enc_of(fe_gx(), fe_gy(), false), enc_of(px, py, pinf), enc_of(Hx, Hy, hinf), e_nul, e_r, e_hr};
        c_hash<6>(dg, pts);                                                          // randomizedsigner.rs:80-87
    } else {
        enc_pt pts[3] = {e_nul, e_r, e_hr};
        c_hash<3>(dg, pts);
    }
    sc c, sk, r, s, tt;
    bool canon;
    sc_from_digest_words(c, dg, canon);
    if (!canon) st |= PLUME_ST_C_NOT_CANONICAL;                                      // :90-91
    (void)load_scalar_reduced(sk, a.sk + 32 * (size_t)i);
    (void)load_scalar_reduced(r, a.r + 32 * (size_t)i);
    sc_mul(tt, c, sk); sc_add(s, r, tt);                                             // :94
    if (sc_is_zero(s)) st |= PLUME_ST_IDENTITY;                                      // :95
    if (a.pk) store_point_be(a.pk, i, a.out33, px, py, pinf);
    store_point_be(a.nul, i, a.out33, nul.x, nul.y, nul.inf != 0);
    sc_to_be_aligned(a.c + 32 * (size_t)i, c);
    sc_to_be_aligned(a.s + 32 * (size_t)i, s);
    store_point_be(a.rpt, i, a.out33, R.x, R.y, R.inf != 0);
    store_point_be(a.hr, i, a.out33, hr.x, hr.y, hr.inf != 0);
    if (a.h_out) store_affine_be(a.h_out + 64 * (size_t)i, Hx, Hy, hinf);
    a.status[i] = (uint8_t)st;
}

// ================================================================================= SEC1-compressed ingest ("next" row f-1)
// 33-byte records: 02|03 || x (big-endian), or 00 (+ 32 ignored bytes) for the identity — the wire format of the
// reference's serde / wasm layer (javascript/src/lib.rs:95-118,147-184; rust-arkworks/src/lib.rs:76-88).  Any other tag,
// x >= p, or an x with no point on the curve is a decoding error there; here it rejects the item.
struct DecompressArgs {
    uint32_t n;
    int npts;                      // 2 (pk, nullifier) or 4 (+ r_point, hashed_to_curve_r)
    const uint8_t* in[4];          // 33 B / item each
    uint8_t* out[4];               // 64 B / item each (the engine's affine format)
    uint8_t* preflags;             // n
};
// returns true when the record decodes; out64 always gets a well-formed 64-byte record
PLUME_HD bool decompress_point(uint8_t* out64, const uint8_t* in33) {
    const uint32_t tag = in33[0];
    fe x, y, rhs, t;
    uint32_t wx[8];
    words_from_be(wx, in33 + 1);
    fe_from_words(x, wx);
    bool ok = (tag == 2u || tag == 3u) && words_lt_p(wx);
    fe_sqr(rhs, x); fe_mul(rhs, rhs, x);
    fe seven = fe_small(7);
    fe_add(rhs, rhs, seven);
    fe_sqrt_candidate(y, rhs);
    fe_sqr(t, y);
    ok = ok && fe_eq(t, rhs);
    if (fe_is_odd(y) != ((tag & 1u) != 0)) fe_neg(y, y);
    const bool inf = tag == 0u;
    if (!ok) { x = fe_gx(); y = fe_gy(); }      // placeholder; the item is rejected through preflags
    store_affine_be(out64, x, y, inf);
    return ok || inf;
}
PLUME_HD void decompress_item(const DecompressArgs& a, uint32_t i) {
    bool ok = true;
    PLUME_NOUNROLL for (int k = 0; k < a.npts; k++) ok = decompress_point(a.out[k] + 64 * (size_t)i, a.in[k] + 33 * (size_t)i) && ok;
    a.preflags[i] = ok ? 0 : 1;
}

// ============================================================================================ h2c only (KAT pinning)
struct H2cArgs {
    uint32_t n;
    const uint8_t* msgs; const uint64_t* msg_off;
    uint64_t msgs_bytes; // size of msgs (see VerifyArgs)
    const uint8_t* pk;   // may be NULL: hash the raw message bytes (no encoding appended)
    uint8_t* h_out;      // 64 B/item, all-zero for identity or invalid pk
};
PLUME_HD void h2c_only(const H2cArgs& a, uint32_t i) {
    uint64_t o0; uint32_t mlen;
    jac h;
    fe x = fe_zero(), y = fe_zero();
    bool bad = !msg_span(o0, mlen, a.msg_off, i, a.msgs_bytes);
    if (bad) {
        h.inf = 1;
    } else if (a.pk) {
        fe px, py;
        uint32_t f = load_affine_be(px, py, a.pk + 64 * (size_t)i);
        bad = f == PLUME_JOB_INVALID;
        if (!bad) hash_to_curve_jac(h, a.msgs + o0, mlen, px, 2u + (fe_is_odd(py) ? 1u : 0u), f == PLUME_JOB_INF ? PLUME_ENC_IDENTITY : PLUME_ENC_POINT);
    } else {
        hash_to_curve_jac(h, a.msgs + o0, mlen, x, 0u, PLUME_ENC_NONE);   // raw message bytes only
    }
    if (!bad && !h.inf) {
        fe zi, zi2;
        fe_inv(zi, h.z); fe_sqr(zi2, zi);
        fe_mul(x, h.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(y, h.y, zi2);
    }
    store_affine_be(a.h_out + 64 * (size_t)i, x, y, bad || h.inf);
}

// ================================================================== circuit witness hints ("next" row f-3, the pinnable part)
// What a prover farm needs to fill the circom verifier's inputs (circuits/circom/verify_nullifier.circom:14-31,152-162) from the same h2c the
// verifier runs: u0, u1 (hash_to_field), the simplified-SWU outputs on the isogenous curve E' (q{0,1}_x_mapped, q{0,1}_y_mapped), Q0, Q1 (after
// the 3-isogeny) and H = Q0 + Q1; each value either as 32 big-endian bytes or as the circuit's 4 x 64-bit little-endian registers
// (circuits/circom/utils.ts:11-17 scalarToCircuitValue; verify_nullifier.circom:380-385).  q*_gx1_sqrt, q*_gx2_sqrt and q*_y_pos are NOT produced:
// their definitions live in the un-vendored secp256k1_hash_to_curve_circom/ts/generate_inputs (circuits/circom/test/v1.test.ts:5,38-40), unpinnable here.
struct H2cInterArgs {
    uint32_t n;
    const uint8_t* msgs; const uint64_t* msg_off;
    uint64_t msgs_bytes;
    const uint8_t* pk;     // may be NULL: hash the raw message bytes (RFC 9380 J.8.1 vectors)
    int registers;         // 0: 32 big-endian bytes per value; 1: 4 little-endian 64-bit registers per value
    uint8_t* u;            // optional, n x 64 : u0 | u1
    uint8_t* mapped;       // optional, n x 128: q0_x_mapped | q0_y_mapped | q1_x_mapped | q1_y_mapped   (affine, on E')
    uint8_t* q;            // optional, n x 128: Q0.x | Q0.y | Q1.x | Q1.y   (affine, on secp256k1; identity = zeros)
    uint8_t* h;            // optional, n x 64 : H.x | H.y
    uint8_t* hints;        // optional, n x 192: q0_gx1_sqrt | q0_gx2_sqrt | q0_y_pos | q1_gx1_sqrt | q1_gx2_sqrt | q1_y_pos   (UNPINNED definitions: include/plume_hip.h)
};
// The three square-root hints of one map (circuits/circom/verify_nullifier.circom:21-23,27-29).  UNPINNED: their generator (secp256k1_hash_to_curve_circom/ts/
// generate_inputs) is not in the reference tree, so these follow RFC 9380 F.2 / F.2.1.2 with every free choice fixed here and stated in include/plume_hip.h:
//   gx1 = x1^3 + A' x1 + B',  gx2 = x2^3 + A' x2 + B' with x2 = Z u^2 x1  (so gx2 = (Z u^2)^3 gx1: exactly one of gx1, gx2 is a square, Z being a non-residue)
//   gxk_sqrt = the EVEN square root (sgn0 = 0) of gxk when gxk is a square, of Z * gxk when it is not (sqrt_ratio's second return value: the witness that gxk is a non-residue)
//   y_pos    = the square root of the chosen gx (gx1 if it is a square, else gx2) whose sgn0 equals sgn0(u): the y the map returns on E'
// All four roots come out of the map's ONE exponentiation: with r = sqrt_ratio's candidate, sqrt(gx1) = r and sqrt(Z gx2) = Z u tv1 r on the square branch,
// sqrt(Z gx1) = r and sqrt(gx2) = tv1 u r on the other (tv1 = Z u^2).
PLUME_HD void fe_even_root(fe& r) { if (fe_is_odd(r)) fe_neg(r, r); }
PLUME_HD void sswu_hints(fe& gx1_sqrt, fe& gx2_sqrt, const fe& u, const sswu_extra& e) {
    fe t;
    gx1_sqrt = e.root;
    fe_mul(t, e.tv1, u); fe_mul(t, t, e.root);                   // tv1 u r
    if (e.is_sq) { fe_mul_small(t, t, 11); fe_neg(gx2_sqrt, t); }   // Z = -11:  Z u tv1 r
    else gx2_sqrt = t;
    fe_even_root(gx1_sqrt); fe_even_root(gx2_sqrt);
}
// one 256-bit value: canonical, big-endian bytes or little-endian registers (= little-endian bytes); dst is 4-byte aligned
PLUME_HD void store_value(uint8_t* dst, fe v, int registers) {
    fe_normalize(v);
    uint32_t w[8];
    fe_to_words(w, v);
    if (registers) { uint32_t* q = (uint32_t*)dst; PLUME_UNROLL for (int k = 0; k < 8; k++) q[k] = w[k]; }
    else words_to_be_aligned(dst, w);
}
PLUME_HD void h2c_intermediates(const H2cInterArgs& a, uint32_t i) {
    uint64_t o0; uint32_t mlen;
    bool bad = !msg_span(o0, mlen, a.msg_off, i, a.msgs_bytes);
    fe px = fe_zero(), py = fe_zero();
    uint32_t enc = PLUME_ENC_NONE;
    if (a.pk) {
        const uint32_t f = load_affine_be(px, py, a.pk + 64 * (size_t)i);
        bad = bad || f == PLUME_JOB_INVALID;
        enc = f == PLUME_JOB_INF ? PLUME_ENC_IDENTITY : PLUME_ENC_POINT;
    }
    fe u[2], xn[2], xd[2], y[2];
    jac q[2], h;
    if (bad) {   // all outputs zero
        const fe z = fe_zero();
        PLUME_NOUNROLL for (int k = 0; k < 6; k++) {
            if (a.u && k < 2) store_value(a.u + 64 * (size_t)i + 32 * k, z, a.registers);
            if (a.mapped && k < 4) store_value(a.mapped + 128 * (size_t)i + 32 * k, z, a.registers);
            if (a.q && k < 4) store_value(a.q + 128 * (size_t)i + 32 * k, z, a.registers);
            if (a.h && k < 2) store_value(a.h + 64 * (size_t)i + 32 * k, z, a.registers);
            if (a.hints) store_value(a.hints + 192 * (size_t)i + 32 * k, z, a.registers);
        }
        return;
    }
    hash_to_field2(u[0], u[1], a.msgs + o0, mlen, px, 2u + (fe_is_odd(py) ? 1u : 0u), enc);
    PLUME_NOUNROLL for (int k = 0; k < 2; k++) {
        sswu_extra ex;
        sswu_frac(xn[k], xd[k], y[k], u[k], &ex);
        if (a.hints) {
            fe g1, g2;
            sswu_hints(g1, g2, u[k], ex);
            uint8_t* dst = a.hints + 192 * (size_t)i + 96 * k;
            store_value(dst, g1, a.registers); store_value(dst + 32, g2, a.registers); store_value(dst + 64, y[k], a.registers);   // y_pos: the map's y (sgn0 = sgn0(u))
        }
        iso3_frac_to_jac(q[k], xn[k], xd[k], y[k]);
    }
    h = q[0];
    jac_add(h, q[1]);
    // one inversion for the five denominators xd0, xd1, Z(Q0), Z(Q1), Z(H) (an identity contributes 1; xd is never zero)
    fe d[5], pre[5], acc = fe_small(1), inv;
    d[0] = xd[0]; d[1] = xd[1];
    d[2] = q[0].inf ? fe_small(1) : q[0].z; d[3] = q[1].inf ? fe_small(1) : q[1].z; d[4] = h.inf ? fe_small(1) : h.z;
    PLUME_UNROLL for (int k = 0; k < 5; k++) { pre[k] = acc; fe_mul(acc, acc, d[k]); }
    fe_inv(inv, acc);
    fe di[5];
    PLUME_UNROLL for (int k = 4; k >= 0; k--) { fe_mul(di[k], inv, pre[k]); fe_mul(inv, inv, d[k]); }
    if (a.u) { store_value(a.u + 64 * (size_t)i, u[0], a.registers); store_value(a.u + 64 * (size_t)i + 32, u[1], a.registers); }
    if (a.mapped) {
        PLUME_NOUNROLL for (int k = 0; k < 2; k++) {
            fe x; fe_mul(x, xn[k], di[k]);
            store_value(a.mapped + 128 * (size_t)i + 64 * k, x, a.registers);
            store_value(a.mapped + 128 * (size_t)i + 64 * k + 32, y[k], a.registers);
        }
    }
    PLUME_NOUNROLL for (int k = 0; k < 3; k++) {
        const jac& p = k == 2 ? h : q[k];
        uint8_t* dst = k == 2 ? (a.h ? a.h + 64 * (size_t)i : nullptr) : (a.q ? a.q + 128 * (size_t)i + 64 * k : nullptr);
        if (!dst) continue;
        fe x = fe_zero(), yy = fe_zero(), zi2;
        if (!p.inf) { fe_sqr(zi2, di[2 + k]); fe_mul(x, p.x, zi2); fe_mul(zi2, zi2, di[2 + k]); fe_mul(yy, p.y, zi2); }
        store_value(dst, x, a.registers); store_value(dst + 32, yy, a.registers);
    }
}
// ================================================================== SEC1-DER scalar marshalling ("next" row f-2, the wasm wire format)
// SecretKey::from(scalar).to_sec1_der() (javascript/src/lib.rs:98-110 uses it for `s` and `digest_private`): RFC 5915 ECPrivateKey with the public key
//   30 6b | 02 01 01 | 04 20 <scalar, 32 B> | a1 44 03 42 00 | 04 <x, 32 B> <y, 32 B>        = 109 bytes, public key = scalar * G
// The generator multiplication is what costs: one doubling-free comb per scalar here.  A scalar outside [1, n-1] (no SecretKey exists for it) gets
// status PLUME_ST_BAD_SCALAR and an all-zero record.
#define PLUME_DER_LEN 109
struct DerArgs {
    uint32_t n;
    const uint8_t* scalars;   // n x 32 big-endian
    uint8_t* der;             // n x 109
    uint8_t* status;          // n
    const uint32_t* gcomb;
    const uint32_t* gscan;    // level 2 only (PLUME_GSCAN_WORDS)
    int uniform;              // the comb's schedule (plume_set_sign_uniform: 0, 1, 2): the scalars are secret keys
};
PLUME_HD void scalar_to_sec1_der(const DerArgs& a, uint32_t i) {
    sc k;
    uint8_t* out = a.der + (size_t)PLUME_DER_LEN * i;
    if (!load_scalar_be(k, a.scalars + 32 * (size_t)i)) {
        PLUME_NOUNROLL for (int j = 0; j < PLUME_DER_LEN; j++) out[j] = 0;
        a.status[i] = (uint8_t)PLUME_ST_BAD_SCALAR;
        return;
    }
    jac p;
    if (a.uniform == 2) comb_mul_g_scan(p, k, a.gscan);
    else if (a.uniform) comb_mul_g_uniform(p, k, a.gcomb);
    else comb_mul_g(p, k, a.gcomb);     // never the identity for k in [1, n-1]
    fe zi, zi2, x, y;
    fe_inv(zi, p.z); fe_sqr(zi2, zi); fe_mul(x, p.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(y, p.y, zi2);
    fe_normalize(x); fe_normalize(y);
    uint32_t xw[8], yw[8];
    fe_to_words(xw, x); fe_to_words(yw, y);
    const uint8_t head[7] = {0x30, 0x6b, 0x02, 0x01, 0x01, 0x04, 0x20}, mid[6] = {0xa1, 0x44, 0x03, 0x42, 0x00, 0x04};
    PLUME_UNROLL for (int j = 0; j < 7; j++) out[j] = head[j];
    PLUME_UNROLL for (int j = 0; j < 32; j++) out[7 + j] = (uint8_t)(k.v[7 - (j >> 2)] >> (8 * (3 - (j & 3))));
    PLUME_UNROLL for (int j = 0; j < 6; j++) out[39 + j] = mid[j];
    PLUME_UNROLL for (int j = 0; j < 32; j++) { out[45 + j] = (uint8_t)(xw[7 - (j >> 2)] >> (8 * (3 - (j & 3)))); out[77 + j] = (uint8_t)(yw[7 - (j >> 2)] >> (8 * (3 - (j & 3)))); }
    a.status[i] = 0;
}

// 32-byte big-endian values -> the circuit's 4 x 64-bit little-endian registers (c, s, pk, nullifier ... of a signature): a byte reversal
PLUME_HD void registers_from_be(uint8_t* out, const uint8_t* in, size_t k) {
    const uint32_t* src = (const uint32_t*)(in + 32 * k);
    uint32_t* dst = (uint32_t*)(out + 32 * k);
    uint32_t w[8];
    PLUME_UNROLL for (int j = 0; j < 8; j++) w[j] = bswap32(src[7 - j]);
    PLUME_UNROLL for (int j = 0; j < 8; j++) dst[j] = w[j];
}

}  // namespace plume
