// Aggregate random-linear-combination check of a batch of PLUME signatures whose r_point / hashed_to_curve_r are GIVEN
// (V1 verify, rust-k256/src/lib.rs:93-135; verify_non_zk for V1 and V2, rust-arkworks/src/tests.rs:28-78) -- SURVEY.md §8f rank 4,
// "an aggregate-only random-linear-combination batch check (one big MSM) as an optional fast pre-filter".
//
// DIFFERENT SEMANTICS from verify: all-or-nothing and probabilistic, never a replacement for the per-item `ok`.  Per item the hash
// check (c == SHA256(...) mod n over the given encodings) is exact; the two group equations
//     E1_i = s_i*G - c_i*pk_i - R_i = O        E2_i = s_i*H_i - c_i*nul_i - Hr_i = O
// are checked only through  A = sum_i a_i*E1_i + b_i*E2_i == O  with 127-bit coefficients a_i | b_i = SHA256(seed || be64(index)) that the
// producer of the batch must not be able to predict (the caller draws `seed` afresh per call): a batch with a false equation passes with
// probability <= 2^-126.  A is
//     (sum a_i s_i)*G  -  sum (a_i c_i)*pk_i  -  sum a_i*R_i  +  sum (b_i s_i)*H_i  -  sum (b_i c_i)*nul_i  -  sum b_i*Hr_i
// i.e. one multi-scalar multiplication over 5n points, done by the bucket method: signed W-bit windows, one bucket per
// (window, |digit|), terms counting-sorted by bucket (histogram, scan, scatter), one lane per bucket summing its run with mixed
// additions, then running sums over chunks of buckets, a tree over the chunks, and a Horner step over the windows.
//
// Per-lane bodies only (the __global__ wrappers are in plume_agg_kernels.hip, tests/devsim drives the same bodies on the host).
#pragma once
#include "plume_stages.h"

namespace plume {

#define PLUME_AGG_TERMS 5          // per item: 0 pk, 1 r_point, 2 H, 3 nullifier, 4 hashed_to_curve_r
#define PLUME_AGG_CHUNK 8          // buckets per running-sum lane
#define PLUME_AGG_GROUP 8          // partial sums per tree lane
#define PLUME_AGG_NORM_K 8         // H points sharing one inversion
#define PLUME_AGG_SUM_K 64         // scalars per lane of the (sum a_i s_i) reduction
#ifndef PLUME_AGG_RESULT_BYTES
#define PLUME_AGG_RESULT_BYTES 72  // all_ok | aggregate_is_identity | 0 | 0 | n_bad (u32 LE) | aggregate point (64 B affine big-endian, zeros = identity)
#endif

struct AggArgs {
    int version, mode;             // as VerifyArgs
    uint32_t n;
    int W;                         // window width: 4, 8 or 16 bits (a divisor of 256, see agg_window_bits in plume_capi.hip)
    int nw_long, nw_short;         // windows of a scalar < 2^255 / < 2^127: ceil(256 / W), ceil(128 / W)
    uint32_t nbuckets;             // 2^(W-1) per window
    uint32_t nkeys;                // nw_long * nbuckets
    uint64_t index_base;           // coefficient index of item 0 (so that pieces of one batch draw distinct coefficients)
    uint8_t seed[32];
    // caller arrays (device)
    const uint8_t *pk, *nul, *c, *s, *rpt, *hr;
    // left behind by verify_ingest_h2c
    const uint32_t* bases; const uint8_t* jobflags; const uint8_t* itemflags;
    // workspace
    uint8_t* haff;                 // n x 64: affine H, big-endian (zeros: identity or rejected item)
    uint32_t* scal;                // (5 x 8) x n words, SoA: |scalar| of term t, word w of item i at scal[(8t + w) n + i]
    uint8_t* tlive;                // n: bit t = term t takes part
    uint8_t* tneg;                 // n: bit t = term t enters negated
    uint32_t* gs;                  // 8 x n words, SoA: a_i * s_i mod n
    uint8_t* hash_ok;              // n: 1 = inputs representable and c matches the hash
    uint32_t* nbad;                // items with hash_ok == 0
    uint32_t* count;               // nkeys + 1: histogram, then exclusive offsets
    uint32_t* sorted;              // one word per (term, window) pair with a non-zero digit: (5 item + t) << 1 | negative
    uint32_t* bsum; uint8_t* bsuminf;   // bucket sums, Jacobian SoA over nkeys
    const uint32_t* gcomb;
    uint8_t* result;               // PLUME_AGG_RESULT_BYTES
};

PLUME_HD uint32_t agg_atomic_inc(uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, 1u);
#else
    return (*p)++;
#endif
}

PLUME_HD const uint8_t* agg_term_point(const AggArgs& a, uint32_t item, uint32_t t) {
    const uint8_t* base = t == 0 ? a.pk : t == 1 ? a.rpt : t == 2 ? a.haff : t == 3 ? a.nul : a.hr;
    return base + 64 * (size_t)item;
}

// ---------------------------------------------------------------------------------------- stage 1: H -> affine
// lane handles items lane + j*nlanes (coalesced), one inversion for PLUME_AGG_NORM_K values of Z (Montgomery's trick)
PLUME_HD void agg_normalize_h(const AggArgs& a, size_t lane, size_t nlanes) {
    fe z[PLUME_AGG_NORM_K], pre[PLUME_AGG_NORM_K];
    fe acc = fe_small(1);
    bool live[PLUME_AGG_NORM_K];
    PLUME_UNROLL for (int j = 0; j < PLUME_AGG_NORM_K; j++) {
        const size_t idx = lane + (size_t)j * nlanes;
        const size_t safe = idx < a.n ? idx : 0;
        live[j] = idx < a.n && !a.itemflags[safe] && job_state(a.jobflags[3 * safe + 1]) == PLUME_JOB_OK;
        if (live[j]) { jac hz; ld_base(hz, a.bases, 3 * idx + 1, true); z[j] = hz.z; } else z[j] = fe_small(1);
        pre[j] = acc;
        fe_mul(acc, acc, z[j]);
    }
    fe inv;
    fe_inv(inv, acc);
    PLUME_UNROLL for (int j = PLUME_AGG_NORM_K - 1; j >= 0; j--) {
        const size_t idx = lane + (size_t)j * nlanes;
        fe zi, zi2, x = fe_zero(), y = fe_zero();
        fe_mul(zi, inv, pre[j]);
        fe_mul(inv, inv, z[j]);
        if (live[j]) {
            jac hxy; ld_base(hxy, a.bases, 3 * idx + 1, false); x = hxy.x; y = hxy.y;
            fe_sqr(zi2, zi);
            fe_mul(x, x, zi2);
            fe_mul(zi2, zi2, zi); fe_mul(y, y, zi2);
        }
        if (idx < a.n) store_affine_be(a.haff + 64 * idx, x, y, !live[j]);
    }
}

// ------------------------------------------------------------------------- stage 2: hash check, coefficients, term scalars
PLUME_HD void agg_store_scalar(const AggArgs& a, uint32_t i, int t, const sc& k) {
    PLUME_UNROLL for (int w = 0; w < 8; w++) a.scal[(size_t)(8 * t + w) * a.n + i] = k.v[w];
}
// k -> min(k, n - k) (< 2^255); returns true when the negation was taken
PLUME_HD bool agg_fold_half(sc& k) {
    sc m;
    sc_neg(m, k);
    bool lt = false;                          // m < k ?
    PLUME_UNROLL for (int w = 0; w < 8; w++) { if (m.v[w] != k.v[w]) lt = m.v[w] < k.v[w]; }
    if (lt) k = m;
    return lt;
}
PLUME_HD void agg_item_terms(const AggArgs& a, uint32_t i) {
    sc zero; PLUME_UNROLL for (int w = 0; w < 8; w++) zero.v[w] = 0;
    fe pkx, pky, nx, ny, rx, ry, hx, hy, Hx, Hy;
    uint32_t fr = PLUME_JOB_INVALID, fh = PLUME_JOB_INVALID;
    bool bad = a.itemflags[i] != 0;
    if (!bad) {
        fr = load_affine_be(rx, ry, a.rpt + 64 * (size_t)i);
        fh = load_affine_be(hx, hy, a.hr + 64 * (size_t)i);
        bad = fr == PLUME_JOB_INVALID || fh == PLUME_JOB_INVALID;
    }
    if (bad) {
        a.hash_ok[i] = 0; a.tlive[i] = 0; a.tneg[i] = 0;
        PLUME_UNROLL for (int w = 0; w < 8; w++) a.gs[(size_t)w * a.n + i] = 0;
        (void)agg_atomic_inc(a.nbad);
        return;
    }
    const uint32_t fpk = reload_affine_be(pkx, pky, a.pk + 64 * (size_t)i);      // validated by verify_ingest_h2c
    const uint32_t fnul = reload_affine_be(nx, ny, a.nul + 64 * (size_t)i);
    const uint32_t fH = reload_affine_be(Hx, Hy, a.haff + 64 * (size_t)i);
    sc c, s;
    sc_from_be_aligned(c, a.c + 32 * (size_t)i);
    sc_from_be_aligned(s, a.s + 32 * (size_t)i);
    // the challenge over the GIVEN encodings (lib.rs:128-135; tests.rs:40-52)
    uint32_t dg[8];
    enc_pt pts[6];
    pts[0] = enc_of(fe_gx(), fe_gy(), false);
    pts[1] = enc_of(pkx, pky, fpk == PLUME_JOB_INF);
    pts[2] = enc_of(Hx, Hy, fH == PLUME_JOB_INF);
    pts[3] = enc_of(nx, ny, fnul == PLUME_JOB_INF);
    pts[4] = enc_of(rx, ry, fr == PLUME_JOB_INF);
    pts[5] = enc_of(hx, hy, fh == PLUME_JOB_INF);
    if (a.version == 1) c_hash<6>(dg, pts); else c_hash<3>(dg, pts + 3);
    sc cc; bool canon;
    sc_from_digest_words(cc, dg, canon);
    uint32_t diff = 0;
    PLUME_UNROLL for (int k = 0; k < 8; k++) diff |= cc.v[k] ^ c.v[k];
    a.hash_ok[i] = diff == 0 ? 1 : 0;
    if (diff != 0) (void)agg_atomic_inc(a.nbad);
    // coefficients: SHA256(seed || be64(index_base + i)), one block
    uint32_t w[16], st[8];
    PLUME_UNROLL for (int k = 0; k < 8; k++)
        w[k] = ((uint32_t)a.seed[4 * k] << 24) | ((uint32_t)a.seed[4 * k + 1] << 16) | ((uint32_t)a.seed[4 * k + 2] << 8) | (uint32_t)a.seed[4 * k + 3];
    const uint64_t idx = a.index_base + i;
    w[8] = (uint32_t)(idx >> 32); w[9] = (uint32_t)idx;
    w[10] = 0x80000000u;
    PLUME_UNROLL for (int k = 11; k < 15; k++) w[k] = 0;
    w[15] = 40 * 8;
    sha256_init(st);
    sha256_compress(st, w);
    sc ca = zero, cb = zero;
    ca.v[3] = st[0] & 0x7FFFFFFFu; ca.v[2] = st[1]; ca.v[1] = st[2]; ca.v[0] = st[3];
    cb.v[3] = st[4] & 0x7FFFFFFFu; cb.v[2] = st[5]; cb.v[1] = st[6]; cb.v[0] = st[7];
    sc k0, k2, k3, g;
    sc_mul(k0, ca, c); sc_mul(k2, cb, s); sc_mul(k3, cb, c); sc_mul(g, ca, s);
    uint32_t neg = 0x1Bu;                      // pk, R, nullifier, Hr enter negated; H does not
    if (agg_fold_half(k0)) neg ^= 1u;
    if (agg_fold_half(k2)) neg ^= 4u;
    if (agg_fold_half(k3)) neg ^= 8u;
    uint32_t live = 0;
    if (fpk != PLUME_JOB_INF && !sc_is_zero(k0)) live |= 1u;
    if (fr != PLUME_JOB_INF && !sc_is_zero(ca)) live |= 2u;
    if (fH != PLUME_JOB_INF && !sc_is_zero(k2)) live |= 4u;
    if (fnul != PLUME_JOB_INF && !sc_is_zero(k3)) live |= 8u;
    if (fh != PLUME_JOB_INF && !sc_is_zero(cb)) live |= 16u;
    agg_store_scalar(a, i, 0, k0); agg_store_scalar(a, i, 1, ca); agg_store_scalar(a, i, 2, k2); agg_store_scalar(a, i, 3, k3); agg_store_scalar(a, i, 4, cb);
    PLUME_UNROLL for (int k = 0; k < 8; k++) a.gs[(size_t)k * a.n + i] = g.v[k];
    a.tlive[i] = (uint8_t)live; a.tneg[i] = (uint8_t)neg;
}

// ------------------------------------------------------------------------------------ stage 3: counting sort by bucket
// signed Booth digit of window j (bits [W j, W j + W) plus the bit below) of term t's scalar, read from the SoA words
PLUME_HD int agg_digit(const AggArgs& a, uint32_t i, int t, int j) {
    const int W = a.W, lo = W * j - 1;
    const uint32_t mask = (1u << (W + 1)) - 1u;
    const uint32_t* k = a.scal + (size_t)(8 * t) * a.n + i;
    uint32_t u;
    if (lo < 0) {
        u = (k[0] << 1) & mask;
    } else {
        const uint32_t wi = (uint32_t)lo >> 5, sh = (uint32_t)lo & 31;
        const uint64_t v = (wi < 8 ? (uint64_t)k[(size_t)wi * a.n] : 0ull) | (wi + 1 < 8 ? (uint64_t)k[(size_t)(wi + 1) * a.n] << 32 : 0ull);
        u = (uint32_t)(v >> sh) & mask;
    }
    return (int)(u & 1) + (int)((u >> 1) & ((1u << (W - 1)) - 1u)) - (int)((u >> W) << (W - 1));
}
// The sort is tiled: workgroup (window j, tile b) owns `tile` consecutive items and one window, and keeps that window's nbuckets bins in LDS
// (128 KiB at W = 16).  Pass 1 counts into the bins and leaves them in tiles[(j ntiles + b) nbuckets + e]; agg_tile_totals / the scan / agg_tile_offsets
// turn the counts into each (window, tile, bucket)'s first position in `sorted`; pass 2 reloads them as cursors and places the pairs.  All atomics
// are LDS atomics; the order inside a bucket is arbitrary (the sum does not depend on it).
template <bool SCATTER>
PLUME_HD void agg_tile_pairs(const AggArgs& a, uint32_t j, uint32_t b, uint32_t tile, uint32_t tid, uint32_t nthreads, uint32_t* bins) {
    const uint32_t lo = b * tile, hi = lo + tile < a.n ? lo + tile : a.n;
    PLUME_NOUNROLL for (uint32_t i = lo + tid; i < hi; i += nthreads) {
        const uint32_t live = a.tlive[i], neg = a.tneg[i];
        PLUME_NOUNROLL for (int t = 0; t < PLUME_AGG_TERMS; t++) {
            if (!((live >> t) & 1u)) continue;
            if ((int)j >= ((t == 1 || t == 4) ? a.nw_short : a.nw_long)) continue;
            const int d = agg_digit(a, i, t, (int)j);
            if (d == 0) continue;
            uint32_t* bin = bins + (uint32_t)((d < 0 ? -d : d) - 1);
            if (SCATTER) {
                const uint32_t pos = agg_atomic_inc(bin);
                a.sorted[pos] = ((5u * i + (uint32_t)t) << 1) | (((neg >> t) & 1u) ^ (d < 0 ? 1u : 0u));
            } else {
                (void)agg_atomic_inc(bin);
            }
        }
    }
}
// lane = key (window j, bucket e): count[key] = sum over the tiles
PLUME_HD void agg_tile_totals(const AggArgs& a, const uint32_t* tiles, uint32_t ntiles, uint32_t key) {
    const uint32_t j = key / a.nbuckets, e = key - j * a.nbuckets;
    uint32_t tot = 0;
    PLUME_NOUNROLL for (uint32_t b = 0; b < ntiles; b++) tot += tiles[((size_t)j * ntiles + b) * a.nbuckets + e];
    a.count[key] = tot;
}
// lane = key, after the scan: tiles[..] <- first position of (window, tile, bucket)
PLUME_HD void agg_tile_offsets(const AggArgs& a, uint32_t* tiles, uint32_t ntiles, uint32_t key) {
    const uint32_t j = key / a.nbuckets, e = key - j * a.nbuckets;
    uint32_t run = a.count[key];
    PLUME_NOUNROLL for (uint32_t b = 0; b < ntiles; b++) {
        uint32_t* p = tiles + ((size_t)j * ntiles + b) * a.nbuckets + e;
        const uint32_t v = *p;
        *p = run;
        run += v;
    }
}
// exclusive scan of v[0 .. total) by `nthreads` cooperating lanes, lane t owning a contiguous range: phase 0 leaves the range sums in part[];
// part[] is scanned (by one lane: agg_scan_mid, or by the same three phases one level up); phase 1 rewrites v[] with the offsets
PLUME_HD void agg_scan_phase0(const uint32_t* v, uint32_t total, uint32_t tid, uint32_t nthreads, uint32_t* part) {
    const uint32_t per = (total + nthreads - 1) / nthreads;
    const uint32_t lo = tid * per < total ? tid * per : total, hi = lo + per < total ? lo + per : total;
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; k++) s += v[k];
    part[tid] = s;
}
PLUME_HD void agg_scan_mid(uint32_t nthreads, uint32_t* part) {
    uint32_t run = 0;
    for (uint32_t t = 0; t < nthreads; t++) { const uint32_t x = part[t]; part[t] = run; run += x; }
}
PLUME_HD void agg_scan_phase1(uint32_t* v, uint32_t total, uint32_t tid, uint32_t nthreads, const uint32_t* part) {
    const uint32_t per = (total + nthreads - 1) / nthreads;
    const uint32_t lo = tid * per < total ? tid * per : total, hi = lo + per < total ? lo + per : total;
    uint32_t run = part[tid];
    for (uint32_t k = lo; k < hi; k++) {
        const uint32_t x = v[k];
        v[k] = run;
        run += x;
    }
}

// ------------------------------------------------------------------------- stage 3b: buckets ordered by run length
// A wavefront of the bucket-sum kernel takes as long as its longest run (lengths are Poisson around 5n / 2^(W-1): the longest of 64 is ~20 % above
// the mean).  perm[] lists the keys by DECREASING run length (a counting sort over min(length, 255), tiled like the pair sort: workgroup `blk` owns
// a contiguous range of keys, bins in LDS, hist[blk][bin] in HBM), so that a wavefront's 64 runs have almost the same length and the longest runs
// start first.  The windows are handled in two groups (upper half first, see aggregate_device): each group's keys are ordered on their own.
#define PLUME_AGG_LEN_BINS 256
PLUME_HD uint32_t agg_len_bin(const AggArgs& a, uint32_t key) {
    const uint32_t len = a.count[key + 1] - a.count[key];
    return len < PLUME_AGG_LEN_BINS ? len : PLUME_AGG_LEN_BINS - 1;
}
template <bool SCATTER>
PLUME_HD void agg_perm_pairs(const AggArgs& a, uint32_t k0, uint32_t k1, uint32_t blk, uint32_t nblk, uint32_t tid, uint32_t nthreads, uint32_t* bins, uint32_t* perm) {
    // the keys [k0, k1) (a group of windows); their order goes to perm[k0 .. k1)
    const uint32_t per = (k1 - k0 + nblk - 1) / nblk;
    const uint32_t lo = k0 + blk * per < k1 ? k0 + blk * per : k1, hi = lo + per < k1 ? lo + per : k1;
    PLUME_NOUNROLL for (uint32_t key = lo + tid; key < hi; key += nthreads) {
        uint32_t* bin = bins + agg_len_bin(a, key);
        if (SCATTER) perm[k0 + agg_atomic_inc(bin)] = key; else (void)agg_atomic_inc(bin);
    }
}
// one workgroup, lane = bin: phase 0 totals per bin; (one lane: exclusive scan from the LONGEST bin down;) phase 1: hist[blk][bin] <- first position
PLUME_HD void agg_perm_phase0(const uint32_t* hist, uint32_t nblk, uint32_t bin, uint32_t* part) {
    uint32_t tot = 0;
    PLUME_NOUNROLL for (uint32_t b = 0; b < nblk; b++) tot += hist[(size_t)b * PLUME_AGG_LEN_BINS + bin];
    part[bin] = tot;
}
PLUME_HD void agg_perm_mid(uint32_t* part) {
    uint32_t run = 0;
    for (int bin = PLUME_AGG_LEN_BINS - 1; bin >= 0; bin--) { const uint32_t x = part[bin]; part[bin] = run; run += x; }
}
PLUME_HD void agg_perm_phase1(uint32_t* hist, uint32_t nblk, uint32_t bin, const uint32_t* part) {
    uint32_t run = part[bin];
    PLUME_NOUNROLL for (uint32_t b = 0; b < nblk; b++) {
        uint32_t* p = hist + (size_t)b * PLUME_AGG_LEN_BINS + bin;
        const uint32_t x = *p;
        *p = run;
        run += x;
    }
}

// ----------------------------------------------------------------------------------------- stage 4: bucket sums
PLUME_HD void agg_load_term(const AggArgs& a, uint32_t v, uint32_t wx[8], uint32_t wy[8]) {
    const uint32_t tid = v >> 1, item = tid / 5u, t = tid - 5u * item;
    const uint8_t* src = agg_term_point(a, item, t);
    words_from_be_aligned(wx, src); words_from_be_aligned(wy, src + 32);
}
PLUME_HD void agg_bucket_sum(const AggArgs& a, uint32_t key) {   // key = perm[lane]
    jac acc;
    acc.x = fe_small(1); acc.y = fe_small(1); acc.z = fe_small(0); acc.inf = 1;
    const uint32_t p0 = a.count[key], p1 = a.count[key + 1];
    uint32_t v = 0, wx[8], wy[8];
    if (p0 < p1) { v = a.sorted[p0]; agg_load_term(a, v, wx, wy); }
    PLUME_NOUNROLL for (uint32_t p = p0; p < p1; p++) {
        fe qx, qy;
        fe_from_words(qx, wx); fe_from_words(qy, wy);
        const bool negate = (v & 1u) != 0;
        if (p + 1 < p1) { v = a.sorted[p + 1]; agg_load_term(a, v, wx, wy); }   // the next term's gather flies while this addition runs
        if (negate) fe_neg_lazy(qy, qy);
        jac_madd<true>(acc, qx, qy);          // checked: one signer's pk (or any repeated point) may meet itself in a bucket
    }
    st_jac_soa(a.bsum, a.nkeys, key, acc);
    a.bsuminf[key] = (uint8_t)acc.inf;
}

// ------------------------------------------------------------------------------------ stage 5: sum_e (e + 1) * S_e per window
PLUME_HD void agg_ld_pt(jac& p, const uint32_t* pts, const uint8_t* inf, size_t stride, size_t idx) {
    p.inf = inf[idx];
    if (p.inf) { p.x = fe_small(1); p.y = fe_small(1); p.z = fe_small(0); } else ld_jac_soa(p, pts, stride, idx);
}
PLUME_HD void agg_st_pt(uint32_t* pts, uint8_t* inf, size_t stride, size_t idx, const jac& p) {
    st_jac_soa(pts, stride, idx, p);
    inf[idx] = (uint8_t)p.inf;
}
// lane (window j, chunk k) -> out[j * nchunks + k] = sum over the chunk's buckets e of (e + 1) * S_e:
// running sums give T = sum (e - lo + 1) S_e and R = sum S_e; the chunk's offset adds lo * R (double-and-add, lo < 2^15)
// out holds the windows [j0, j0 + nwin) of one group
PLUME_HD void agg_chunk_reduce(const AggArgs& a, uint32_t j, uint32_t k, uint32_t chunk, uint32_t nchunks, uint32_t* out, uint8_t* outinf, uint32_t j0, uint32_t nwin) {
    const uint32_t lo = k * chunk, hi = lo + chunk < a.nbuckets ? lo + chunk : a.nbuckets;
    jac R, T, S;
    R.x = fe_small(1); R.y = fe_small(1); R.z = fe_small(0); R.inf = 1;
    T = R;
    PLUME_NOUNROLL for (uint32_t e = hi; e > lo; e--) {
        agg_ld_pt(S, a.bsum, a.bsuminf, a.nkeys, (size_t)j * a.nbuckets + (e - 1));
        jac_add(R, S);
        jac_add(T, R);
    }
    if (lo != 0 && !R.inf) {
        jac P = R;                              // lo * R, most significant bit first
        int top = 31;
        while (!((lo >> top) & 1u)) top--;
        PLUME_NOUNROLL for (int b = top - 1; b >= 0; b--) {
            jac_dbl(P);
            if ((lo >> b) & 1u) jac_add(P, R);
        }
        jac_add(T, P);
    }
    agg_st_pt(out, outinf, (size_t)nwin * nchunks, (size_t)(j - j0) * nchunks + k, T);
}
// lane (window j, group q): out[j * m_out + q] = sum of in[j * m_in + q g .. q g + g)
PLUME_HD void agg_group_sum(const uint32_t* in, const uint8_t* ininf, uint32_t m_in, uint32_t g, uint32_t* out, uint8_t* outinf, uint32_t m_out, uint32_t nw, uint32_t j, uint32_t q) {
    jac acc, S;
    acc.x = fe_small(1); acc.y = fe_small(1); acc.z = fe_small(0); acc.inf = 1;
    const uint32_t lo = q * g, hi = lo + g < m_in ? lo + g : m_in;
    PLUME_NOUNROLL for (uint32_t e = lo; e < hi; e++) {
        agg_ld_pt(S, in, ininf, (size_t)nw * m_in, (size_t)j * m_in + e);
        jac_add(acc, S);
    }
    agg_st_pt(out, outinf, (size_t)nw * m_out, (size_t)j * m_out + q, acc);
}
// lane j in [j0, j0 + nwin): pts[j - j0] <- 2^(W j) * pts[j - j0]   (the Horner weights, all windows of the group in parallel)
PLUME_HD void agg_window_shift(const AggArgs& a, uint32_t* pts, uint8_t* inf, uint32_t j, uint32_t j0, uint32_t nwin) {
    jac P;
    agg_ld_pt(P, pts, inf, (size_t)nwin, j - j0);
    if (!P.inf) {
        PLUME_NOUNROLL for (int d = 0; d < a.W * (int)j; d++) jac_dbl(P);
    }
    agg_st_pt(pts, inf, (size_t)nwin, j - j0, P);
}

// ------------------------------------------------------------------------------------ sum of a_i s_i mod n (SoA words)
PLUME_HD void agg_scalar_sum(const uint32_t* in, size_t nin, uint32_t* out, size_t nout, size_t lane) {
    sc acc; PLUME_UNROLL for (int w = 0; w < 8; w++) acc.v[w] = 0;
    PLUME_NOUNROLL for (int q = 0; q < PLUME_AGG_SUM_K; q++) {
        const size_t idx = lane + (size_t)q * nout;
        if (idx >= nin) break;
        sc v; PLUME_UNROLL for (int w = 0; w < 8; w++) v.v[w] = in[(size_t)w * nin + idx];
        sc_add(acc, acc, v);
    }
    PLUME_UNROLL for (int w = 0; w < 8; w++) out[(size_t)w * nout + lane] = acc.v[w];
}

// ------------------------------------------------------------------------------------------------- last lane
PLUME_HD uint32_t agg_record_nbad(const uint8_t* rec) { return (uint32_t)rec[4] | ((uint32_t)rec[5] << 8) | ((uint32_t)rec[6] << 16) | ((uint32_t)rec[7] << 24); }
PLUME_HD void agg_write_record(uint8_t* rec, const jac& tot, uint32_t nbad) {
    fe x = fe_zero(), y = fe_zero();
    if (!tot.inf) {
        fe zi, zi2;
        fe_inv(zi, tot.z); fe_sqr(zi2, zi); fe_mul(x, tot.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(y, tot.y, zi2);
    }
    rec[0] = (uint8_t)((tot.inf && nbad == 0) ? 1 : 0);
    rec[1] = (uint8_t)(tot.inf ? 1 : 0);
    rec[2] = 0; rec[3] = 0;
    rec[4] = (uint8_t)nbad; rec[5] = (uint8_t)(nbad >> 8); rec[6] = (uint8_t)(nbad >> 16); rec[7] = (uint8_t)(nbad >> 24);
    store_affine_be(rec + 8, x, y, tot.inf != 0);
}
// tot += the aggregate point of another record (written by this library: a curve point or zeros)
PLUME_HD void agg_add_record(jac& tot, uint32_t& nbad, const uint8_t* rec) {
    fe x, y;
    nbad += agg_record_nbad(rec);
    if (reload_affine_be(x, y, rec + 8) == PLUME_JOB_OK) jac_madd<true>(tot, x, y);
}
// gout <- gsum * G (doubling-free comb), Jacobian
PLUME_HD void agg_gterm(const AggArgs& a, const uint32_t* gsum, uint32_t* gout, uint8_t* goutinf) {
    sc g; PLUME_UNROLL for (int w = 0; w < 8; w++) g.v[w] = gsum[w];
    jac P;
    comb_mul_g(P, g, a.gcomb);
    agg_st_pt(gout, goutinf, 1, 0, P);
}
// total = gsum*G + the shifted window sums of both groups (+ the record of the pieces before this one); result record
PLUME_HD void agg_final(const AggArgs& a, const uint32_t* lo, const uint8_t* loinf, uint32_t nlo, const uint32_t* hi, const uint8_t* hiinf, uint32_t nhi, const uint32_t* gpt,
                        const uint8_t* gptinf, const uint8_t* carry) {
    jac tot, S;
    agg_ld_pt(tot, gpt, gptinf, 1, 0);
    PLUME_NOUNROLL for (uint32_t j = 0; j < nlo; j++) { agg_ld_pt(S, lo, loinf, (size_t)nlo, (size_t)j); jac_add(tot, S); }
    PLUME_NOUNROLL for (uint32_t j = 0; j < nhi; j++) { agg_ld_pt(S, hi, hiinf, (size_t)nhi, (size_t)j); jac_add(tot, S); }
    uint32_t nbad = *a.nbad;
    if (carry) agg_add_record(tot, nbad, carry);
    agg_write_record(a.result, tot, nbad);
}
// several records (shards of one batch) -> one
PLUME_HD void agg_combine(const uint8_t* records, uint32_t m, uint8_t* result) {
    jac tot;
    tot.x = fe_small(1); tot.y = fe_small(1); tot.z = fe_small(0); tot.inf = 1;
    uint32_t nbad = 0;
    PLUME_NOUNROLL for (uint32_t k = 0; k < m; k++) agg_add_record(tot, nbad, records + (size_t)PLUME_AGG_RESULT_BYTES * k);
    agg_write_record(result, tot, nbad);
}

}  // namespace plume
