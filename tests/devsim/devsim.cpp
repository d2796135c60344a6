// Device-code simulation harness — TEST INFRASTRUCTURE ONLY (never linked into the product library).
// Compiles the product's device headers (zk-nullifier-sig_amd/csrc/plume_*.h) as plain host C++ and drives the
// per-lane stage bodies with the same index mapping the HIP kernels use, so the exact device arithmetic and
// pipeline logic can be checked against the oracle in a container without a GPU.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "plume_stages.h"
#include "plume_aggregate.h"
#include "plume_dedup.h"

using namespace plume;

static void fe_from_le_words(fe& r, const uint32_t* w) { fe_from_words(r, w); }   // any 256-bit integer, also >= p
static int g_sign_uniform = 0;        // the signer's (and the DER export's) uniform-schedule bodies (plume_set_sign_uniform) instead of the default ones

extern "C" {
void ds_set_sign_uniform(int on) { g_sign_uniform = on; }

// how many multi-scalar chains were redone with checked additions since the library was loaded (p == +-q inside a chain)
unsigned long ds_fallback_count(void) { return fallback_counter(); }

// op: 0 mul, 1 sqr, 2 add, 3 sub, 4 neg, 5 inv, 6 pow_c1, 7 normalize, 8 mul_small(b[0]), 9 is_zero->out[0], 10 eq->out[0], 11 is_odd->out[0],
//     12 (a+b)*(a+2p-b) on unreduced operands, 13 (a+4p-2b)^2 through fe_carry, 14 words round trip, 15 inversion by divsteps,
//     16..20 the fused multiply-subtract forms of the group law (round 3): xy - x, x^2 - (2y + x), x^2 - 2y, x(y - x) - 2y^2, (x - y)(x + y) - y
// operands/outputs: 256-bit integers as 8 little-endian 32-bit words (outputs canonical); count elements
void ds_fe_op(int op, size_t count, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    for (size_t i = 0; i < count; i++) {
        fe x, y, r = fe_zero();
        bool flag = false;
        fe_from_le_words(x, a + 8 * i);
        fe_from_le_words(y, b + 8 * i);
        switch (op) {
            case 0: fe_mul(r, x, y); break;
            case 1: fe_sqr(r, x); break;
            case 2: fe_add(r, x, y); break;
            case 3: fe_sub(r, x, y); break;
            case 4: fe_neg(r, x); break;
            case 5: fe_inv_fermat(r, x); break;
            case 6: fe_pow_c1(r, x); break;
            case 7: r = x; fe_normalize(r); break;
            case 8: fe_mul_small(r, x, b[8 * i]); break;
            case 9: flag = fe_is_zero(x); break;
            case 10: flag = fe_eq(x, y); break;
            case 11: flag = fe_is_odd(x); break;
            case 12: { fe s, d; fe_add_lazy(s, x, y); fe_sub_lazy<2>(d, x, y); fe_mul(r, d, s); break; }          // (a+b)(a-b), both operands unreduced
            case 13: { fe t; fe_add_lazy(t, y, y); fe_sub_lazy<4>(t, x, t); fe_carry(t); fe_sqr(r, t); break; }  // (a-2b)^2
            case 14: r = x; break;
            case 15: fe_inv_gcd(r, x); break;
            case 16: fe_mul_sub<2>(r, x, y, x); break;                                                               // xy - x
            case 17: { fe hh; fe_dbl_lazy(hh, y); fe_add_lazy(hh, hh, x); fe_sqr_sub<4>(r, x, hh); break; }           // x^2 - (2y + x): the mixed addition's X3
            case 18: fe_sqr_sub2<2>(r, x, y); break;                                                                 // x^2 - 2y: the doubling's X'
            case 19: { fe t, nb, db; fe_sub_lazy<2>(t, y, x); fe_neg_lazy(nb, y); fe_dbl_lazy(db, y); fe_muladd(r, x, t, nb, db); break; }   // x(y - x) - 2y^2: the doubling's Y' at its operand bounds
            case 20: { fe d, sm; fe_add_lazy(sm, x, y); fe_sub_lazy<2>(d, x, y); fe_mul_sub<2>(r, d, sm, y); break; }  // (x - y)(x + y) - y, unreduced factors
        }
        if (op >= 9 && op <= 11) { for (int k = 0; k < 8; k++) out[8 * i + k] = k == 0 ? (uint32_t)flag : 0u; continue; }
        fe_normalize(r);
        fe_to_words(out + 8 * i, r);
    }
}
// The products on RAW limbs (9 words per operand, NOT canonical: whatever magnitudes the caller's contract allows) -> raw result limbs, so that a test can drive the
// column sums to their bounds and look at the tightness of what comes back.  op: 0 fe_mul(a, b), 1 fe_sqr(a), 2 fe_muladd(a, b, c, e), 3 fe_mul_sub<2>(a, b, c), 4 fe_sqr_sub<4>(a, c),
// 5 fe_sqr_sub2<2>(a, c), 6 fe_sqr3(a), 7 fe_sqr2(a)
void ds_fe_raw(int op, size_t count, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* e, uint32_t* out) {
    for (size_t i = 0; i < count; i++) {
        fe x, y, z, w, r = fe_zero();
        for (int k = 0; k < 9; k++) { x.v[k] = a[9 * i + k]; y.v[k] = b[9 * i + k]; z.v[k] = c[9 * i + k]; w.v[k] = e[9 * i + k]; }
        switch (op) {
            case 0: fe_mul(r, x, y); break;
            case 1: fe_sqr(r, x); break;
            case 2: fe_muladd(r, x, y, z, w); break;
            case 3: fe_mul_sub<2>(r, x, y, z); break;
            case 4: fe_sqr_sub<4>(r, x, z); break;
            case 5: fe_sqr_sub2<2>(r, x, z); break;
            case 6: fe_sqr3(r, x); break;
            case 7: fe_sqr2(r, x); break;
        }
        for (int k = 0; k < 9; k++) out[9 * i + k] = r.v[k];
    }
}
// The group law on RAW limbs: every lazy sum / difference that feeds a product inside jac_dbl / jac_dbl_neg / jac_madd / jac_add, driven by operands whose limbs
// sit at the ends of the "tight" range -- the host build's PLUME_FE_CHECK assertions (fe_mul_inputs_ok: limbs 0..7 by the column bound, limb 8 <= 2^26) abort on a violation.
// op: 0 jac_dbl, 1 jac_dbl_neg, 2 jac_madd<false>, 3 jac_madd<true>, 4 the same with the row's y negated lazily (what a negative digit does), 5 jac_add (q = (qx, qy, p.z)).
// p = (x, y, z) and the affine (qx, qy): 9 words each; out: the resulting x, y, z (27 words).
void ds_group_raw(int op, size_t count, const uint32_t* px, const uint32_t* py, const uint32_t* pz, const uint32_t* qx, const uint32_t* qy, uint32_t* out) {
    for (size_t i = 0; i < count; i++) {
        jac p; p.inf = 0;
        fe ax, ay;
        for (int k = 0; k < 9; k++) { p.x.v[k] = px[9 * i + k]; p.y.v[k] = py[9 * i + k]; p.z.v[k] = pz[9 * i + k]; ax.v[k] = qx[9 * i + k]; ay.v[k] = qy[9 * i + k]; }
        switch (op) {
            case 0: jac_dbl(p); break;
            case 1: jac_dbl_neg(p); break;
            case 2: jac_madd<false>(p, ax, ay); break;
            case 3: jac_madd<true>(p, ax, ay); break;
            case 4: { fe ny; fe_neg_lazy(ny, ay); jac_madd<false>(p, ax, ny); break; }
            case 5: { jac q; q.inf = 0; q.x = ax; q.y = ay; q.z = p.z; jac_add(p, q); break; }
        }
        for (int k = 0; k < 9; k++) { out[27 * i + k] = p.x.v[k]; out[27 * i + 9 + k] = p.y.v[k]; out[27 * i + 18 + k] = p.z.v[k]; }
    }
}
// op: 0 mul, 1 add, 2 neg, 3 reduce of 512-bit a|b (a low)
void ds_sc_op(int op, size_t count, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    for (size_t i = 0; i < count; i++) {
        sc x, y, r;
        for (int k = 0; k < 8; k++) { x.v[k] = a[8 * i + k]; y.v[k] = b[8 * i + k]; r.v[k] = 0; }
        if (op == 0) sc_mul(r, x, y);
        else if (op == 1) sc_add(r, x, y);
        else if (op == 2) sc_neg(r, x);
        else { uint32_t t[16]; for (int k = 0; k < 8; k++) { t[k] = x.v[k]; t[8 + k] = y.v[k]; } sc_reduce_wide(r, t); }
        for (int k = 0; k < 8; k++) out[8 * i + k] = r.v[k];
    }
}
uint32_t ds_wbits() { return PLUME_WBITS; }                  // width of the grid the generator's wide digits sit on (4)
// k (8 limbs) -> m1[4], neg1, m2[4], neg2 (10 words) and the PLUME_NPOS Eisenstein digit codes of the pair (int8, rows of 65)
void ds_glv(size_t count, const uint32_t* k, uint32_t* out, int8_t* digits) {
    for (size_t i = 0; i < count; i++) {
        sc x; for (int j = 0; j < 8; j++) x.v[j] = k[8 * i + j];
        glv_half h1, h2;
        glv_split(h1, h2, x);
        for (int j = 0; j < 4; j++) { out[10 * i + j] = h1.m[j]; out[10 * i + 5 + j] = h2.m[j]; }
        out[10 * i + 4] = h1.neg; out[10 * i + 9] = h2.neg;
        eisd_store_glv(digits + PLUME_NPOS * i, 1, h1, h2, false);
    }
}
void ds_sha256(const uint8_t* data, uint32_t len, uint8_t out[32]) {
    uint32_t st[8];
    sha256_init(st);
    sha256_absorb_pad(st, 0u, len, [&](uint32_t pos) -> uint32_t { return data[pos]; });
    for (int i = 0; i < 8; i++) { out[4 * i] = st[i] >> 24; out[4 * i + 1] = st[i] >> 16; out[4 * i + 2] = st[i] >> 8; out[4 * i + 3] = st[i]; }
}

// the generator's fixed tables exactly as the library builds them at plume_init (plume_ec.h fixed_window_base / fixed_table_lane: one entry per "lane")
static void build_gtab(std::vector<uint32_t>& gtab) {
    static std::vector<uint32_t> cached;           // the table only depends on G: build it once per process
    if (!cached.empty()) { gtab = cached; return; }
    gtab.assign(PLUME_GTAB_WORDS, 0);
    uint32_t base18[2 * PLUME_FE_WORDS];
    fixed_window_base(base18, 0);
    for (size_t lane = 0; lane < (size_t)PLUME_GTAB_ENTRIES; lane++) fixed_table_lane(gtab.data(), base18, PLUME_GTAB_ENTRIES, lane);
    cached = gtab;
}

static void build_gcomb(std::vector<uint32_t>& comb) {
    comb.assign(PLUME_COMB_WORDS, 0);
    std::vector<uint32_t> base18((size_t)PLUME_COMB_WINDOWS * 2 * PLUME_FE_WORDS);
    for (uint32_t w = 0; w < PLUME_COMB_WINDOWS; w++) fixed_window_base(base18.data() + (size_t)w * 2 * PLUME_FE_WORDS, PLUME_COMB_W * w);      // mirrors k_fixed_bases
    for (size_t lane = 0; lane < (size_t)PLUME_COMB_ENTRIES * PLUME_COMB_WINDOWS; lane++) fixed_table_lane(comb.data(), base18.data(), PLUME_COMB_ENTRIES, lane);
}
// the scanned table of G of the uniform schedule's level 2 (mirrors launch_fixed_tables' gscan leg)
static const std::vector<uint32_t>& shared_gscan() {
    static std::vector<uint32_t> t;
    if (t.empty()) {
        t.assign(PLUME_GSCAN_WORDS, 0);
        std::vector<uint32_t> base18((size_t)PLUME_GSCAN_WINDOWS * 2 * PLUME_FE_WORDS);
        for (uint32_t w = 0; w < PLUME_GSCAN_WINDOWS; w++) fixed_window_base(base18.data() + (size_t)w * 2 * PLUME_FE_WORDS, PLUME_GSCAN_W * w);
        for (size_t lane = 0; lane < (size_t)PLUME_GSCAN_ENTRIES * PLUME_GSCAN_WINDOWS; lane++) fixed_table_lane(t.data(), base18.data(), PLUME_GSCAN_ENTRIES, lane);
    }
    return t;
}
static const std::vector<uint32_t>& shared_gcomb() {
    static std::vector<uint32_t> gcomb;
    if (gcomb.empty()) build_gcomb(gcomb);
    return gcomb;
}

// host stand-in for launch_tables: the same lane -> jobs mapping, the same lane-interleaved scratch indexing and the same PASS sequence as the table stage (one loop over the
// lanes per pass, the lanes' running products in a word-major array, tab_invert_group between the two passes); a batch of one lane also runs the one-function form
// (table_build) and the two must agree
void ds_set_tables_small(int) {}                                  // (rounds 4's small-batch table path is gone: one path for every size)
static void run_tables(uint32_t* tab, const uint32_t* bases, const uint8_t* jobflags, size_t njobs, int L) {
    const size_t lanes = (njobs + L - 1) / L, stride = ((lanes + 7) / 8) * 8;     // "grid" rounded up like the kernel's
    std::vector<uint32_t> scr(stride * (size_t)L * PLUME_TAB_SCR_WORDS), carry(stride * PLUME_FE_WORDS);
    std::vector<uint8_t> guardf(stride, 0);
    constexpr int K = 8;
    const size_t T = (stride + K - 1) / K;
    const DirectRowSinkSync sink;
    auto span = [&](size_t lane, size_t& j0, int& cnt) { j0 = lane * (size_t)L; cnt = j0 < njobs ? (int)(njobs - j0 < (size_t)L ? njobs - j0 : (size_t)L) : 0; };
    for (size_t lane = 0; lane < stride; lane++) {
        size_t j0; int cnt; span(lane, j0, cnt);
        fe c; bool g = false;
        tab_pass_a(bases, jobflags, njobs, j0, cnt, scr.data(), stride, lane, c, g);
        st_fe_soa(carry.data(), stride, lane, c); guardf[lane] = g ? 1 : 0;
    }
    for (size_t t = 0; t < T; t++) tab_invert_group<K>(carry.data(), stride, T, t);
    for (size_t lane = 0; lane < stride; lane++) {
        size_t j0; int cnt; span(lane, j0, cnt);
        fe c; ld_fe_soa(c, carry.data(), stride, lane);
        tab_pass_b(tab, bases, jobflags, njobs, j0, cnt, scr.data(), stride, lane, c, guardf[lane] != 0, sink);
    }
    if (lanes == 1) {                                                              // the one-function form on the same input
        std::vector<uint32_t> tab2((size_t)njobs * PLUME_TAB_WORDS), scr2((size_t)L * PLUME_TAB_SCR_WORDS);
        table_build(tab2.data(), bases, jobflags, njobs, 0, (int)njobs, scr2.data(), 1, 0);
        if (memcmp(tab2.data(), tab, tab2.size() * 4) != 0) { fprintf(stderr, "devsim: table_build and the pass sequence disagree\n"); abort(); }
    }
}

// window tables of `nb` affine bases given as raw 64-byte records, all flagged usable WITHOUT validation, built by ONE lane (test hook for the
// zero-denominator guard of the table passes); out: nb x PLUME_TAB_ENTRIES x 64 bytes (x || y of the rows P, theta P, 2P)
void ds_tables_raw(uint32_t nb, const uint8_t* pts, uint8_t* out) {
    std::vector<uint32_t> bases(PLUME_BASE_WORDS * (size_t)nb), tab((size_t)nb * PLUME_TAB_WORDS);
    std::vector<uint8_t> flags(nb, (uint8_t)(PLUME_JOB_OK | PLUME_JOB_AFFINE));
    for (uint32_t j = 0; j < nb; j++) {
        alignas(16) uint8_t rec[64]; memcpy(rec, pts + 64 * j, 64);
        jac p; p.inf = 0; p.z = fe_small(1);
        fe_from_be_aligned(p.x, rec); fe_from_be_aligned(p.y, rec + 32);
        st_base(bases.data(), j, p);
    }
    run_tables(tab.data(), bases.data(), flags.data(), nb, (int)nb);
    for (uint32_t j = 0; j < nb; j++)
        for (int k = 0; k < PLUME_TAB_ENTRIES; k++) {
            fe x, y; alignas(16) uint8_t rec[64];
            ld_tab_xy(x, y, tab.data() + (size_t)j * PLUME_TAB_WORDS + k * PLUME_TAB_ENTRY_WORDS, false);
            store_affine_be(rec, x, y, false);
            memcpy(out + 64 * (PLUME_TAB_ENTRIES * (size_t)j + k), rec, 64);
        }
}
// the digit table of eisd_store in its two forms (the 64-entry table, the register-resident packing the kernels run) for every t = (ta, tb) in [-4, 4]^2: 81 x 2 words
void ds_eisd_entries(uint32_t* out) {
    int k = 0;
    for (int ta = -4; ta <= 4; ta++) for (int tb = -4; tb <= 4; tb++) { out[k++] = eisd_entry_table(ta, tb); out[k++] = eisd_entry(ta, tb); }
}
uint32_t ds_tab_entries() { return PLUME_TAB_ENTRIES; }     // rows per window table (3: P, theta P = P - lambda P, 2P)

// simulated launch geometry: blocks of B lanes sharing a digit buffer with element stride B (as LDS does on the GPU)
static const uint32_t B = 8;

static int g_eq1_short = 1;      // mirrors ctx->eq1_short (plume_capi.hip): 0 = long form always, 1 = short form where R is given, 2 = every item takes the fallback (long form through the checked chain)
void ds_set_eq1_short(int v) { g_eq1_short = v; }
static int g_msm_pair = 0;
void ds_set_msm_pair(int on) { g_msm_pair = on; }                     // the verify harness then runs every long-form chain as two halves + a join (k_verify_msm_pair)
static int g_ingest_two_roles = 0;
void ds_set_ingest_two_roles(int on) { g_ingest_two_roles = on; }      // the verify harness then runs the ingest stage in its two-role form
static int verify_impl(int version, uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nul, const uint8_t* c,
                       const uint8_t* s, const uint8_t* rpt, const uint8_t* hr, uint8_t* ok, int L, const uint8_t* preflags, const uint8_t* rpt33 = nullptr,
                       const uint8_t* hr33 = nullptr, int mode = PLUME_MODE_VERIFY, uint64_t msgs_bytes = ~0ull);
// plume_arkworks' verify_non_zk through the same per-lane bodies (PLUME_MODE_NON_ZK); c = digest_private
int ds_verify_non_zk_batch(int version, uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nul, const uint8_t* s,
                           const uint8_t* rpt, const uint8_t* hr, const uint8_t* digest_private, uint8_t* ok, int L) {
    return verify_impl(version, n, msgs, msg_off, pk, nul, digest_private, s, rpt, hr, ok, L, nullptr, nullptr, nullptr, PLUME_MODE_NON_ZK);
}
// verify with an explicit msgs buffer size (malformed offsets must reject the item without reading out of bounds)
int ds_verify_batch_bounded(int version, uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, uint64_t msgs_bytes, const uint8_t* pk, const uint8_t* nul, const uint8_t* c,
                            const uint8_t* s, const uint8_t* rpt, const uint8_t* hr, uint8_t* ok) {
    return verify_impl(version, n, msgs, msg_off, pk, nul, c, s, rpt, hr, ok, 3, nullptr, nullptr, nullptr, PLUME_MODE_VERIFY, msgs_bytes);
}
int ds_verify_batch(int version, uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nul, const uint8_t* c,
                    const uint8_t* s, const uint8_t* rpt, const uint8_t* hr, uint8_t* ok, int L) {
    return verify_impl(version, n, msgs, msg_off, pk, nul, c, s, rpt, hr, ok, L, nullptr);
}
// SEC1-compressed ingest: decompress_item per lane, then the same pipeline
int ds_verify_batch_sec1(int version, uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk33, const uint8_t* nul33, const uint8_t* c,
                         const uint8_t* s, const uint8_t* r33, const uint8_t* hr33, uint8_t* ok) {
    if (version != 1 && version != 2) return -1;
    std::vector<uint8_t> dec[4], pre(n);
    DecompressArgs d; d.n = n; d.npts = 2;       // mirrors verify_sec1_device: r_point / hashed_to_curve_r stay compressed
    d.in[0] = pk33; d.in[1] = nul33; d.in[2] = r33; d.in[3] = hr33;
    for (int k = 0; k < 4; k++) { dec[k].resize(64 * (size_t)n + 16); d.out[k] = dec[k].data(); }
    d.preflags = pre.data();
    for (uint32_t i = 0; i < n; i++) decompress_item(d, i);
    return verify_impl(version, n, msgs, msg_off, d.out[0], d.out[1], c, s, nullptr, nullptr, ok, 3, pre.data(), version == 1 ? r33 : nullptr, version == 1 ? hr33 : nullptr);
}
static int verify_impl(int version, uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nul, const uint8_t* c,
                       const uint8_t* s, const uint8_t* rpt, const uint8_t* hr, uint8_t* ok, int L, const uint8_t* preflags, const uint8_t* rpt33, const uint8_t* hr33, int mode,
                       uint64_t msgs_bytes) {
    if (version != 1 && version != 2) return -1;
    std::vector<uint32_t> gtab; build_gtab(gtab);
    const bool eq1short = g_eq1_short != 0 && rpt != nullptr && rpt33 == nullptr && (version == 1 || mode == PLUME_MODE_NON_ZK);          // mirrors verify_device
    const size_t J = eq1short ? 4 : 3;
    std::vector<uint32_t> bases(PLUME_BASE_WORDS * J * (size_t)n), tab((size_t)J * n * PLUME_TAB_WORDS), res(PLUME_JAC_WORDS * 2 * (size_t)n), eq1k(8 * (size_t)n + 1);
    std::vector<uint8_t> jobflags(J * (size_t)n), itemflags(n), resinf(2 * (size_t)n), eq1fall(n + 1);
    VerifyArgs a; memset(&a, 0, sizeof a);
    if (eq1short) { a.eq1fall = eq1fall.data(); a.eq1k = eq1k.data(); a.gcomb = shared_gcomb().data(); a.eq1force = g_eq1_short == 2 ? 1 : 0; }
    a.mode = mode; a.msgs_bytes = msgs_bytes == ~0ull ? msg_off[n] : msgs_bytes;
    a.version = version; a.n = n; a.msgs = msgs; a.msg_off = msg_off; a.pk = pk; a.nul = nul; a.c = c; a.s = s; a.rpt = rpt; a.hr = hr; a.ok = ok; a.preflags = preflags; a.rpt33 = rpt33; a.hr33 = hr33;
    a.bases = bases.data(); a.jobflags = jobflags.data(); a.itemflags = itemflags.data(); a.tab = tab.data(); a.res = res.data(); a.resinf = resinf.data();
    a.gtab = gtab.data();
    std::vector<int8_t> digs((size_t)PLUME_VDIG_ROWS * n + 1, 0);
    a.digs = digs.data();
    if (g_ingest_two_roles) {            // the small-batch form of the stage (k_verify_ingest_split): role B and role A of every item, meeting twice
        std::vector<ingest_xch> x(n);
        std::vector<ingest_a_state> stt(n);
        for (uint32_t i = 0; i < n; i++) verify_ingest_b1(a, i, x[i]);
        for (uint32_t i = 0; i < n; i++) verify_ingest_a1(a, i, x[i], stt[i]);
        for (uint32_t i = 0; i < n; i++) verify_ingest_b2(x[i]);
        for (uint32_t i = 0; i < n; i++) verify_ingest_a2(a, i, x[i], stt[i]);
        for (uint32_t i = 0; i < n; i++) verify_ingest_a3(a, i, x[i], stt[i]);
        a.scalars_in_ingest = 1;
        for (uint32_t i = 0; i < n; i++) verify_scalars(a, i);            // role B's last duty in that kernel: no scalar launch of its own
    } else {
        for (uint32_t i = 0; i < n; i++) verify_ingest_h2c(a, i);
        for (uint32_t i = 0; i < n; i++) verify_scalars(a, i);
    }
    const size_t nj = J * (size_t)n;
    run_tables(a.tab, a.bases, a.jobflags, nj, L);
    std::vector<int8_t> dig((2 * PLUME_NDIG + PLUME_NPOS) * B);
    std::vector<uint32_t> redo(2 * (size_t)n + 1, 0);
    a.redo = redo.data();
    if (g_msm_pair && !eq1short) {          // the small-call form (k_verify_msm_pair): the two halves of every chain, joined by a checked addition
        a.msm_pair = 1;
        for (uint32_t eq = 0; eq < 2; eq++)
            for (uint32_t i = 0; i < n; i++) {
                jac h0, h1;
                const bool ok1 = verify_msm_half(a, i, eq, 1, a.gtab, dig.data() + (i % B), B, h1);
                const bool ok0 = verify_msm_half(a, i, eq, 0, a.gtab, dig.data() + (i % B), B, h0);
                verify_msm_join(a, i, eq, h0, ok0, h1, ok1);
            }
    } else
    for (uint32_t eq = 0; eq < 2; eq++)
        for (uint32_t i = 0; i < n; i++) verify_msm<false>(a, i, eq, a.gtab, dig.data() + (i % B), B);
    for (uint32_t k = 0; k < redo[0]; k++) verify_msm<true>(a, redo[1 + k] >> 1, redo[1 + k] & 1u, a.gtab, dig.data() + (k % B), B);      // mirrors k_verify_msm_redo
    if (version == 2 && mode == PLUME_MODE_VERIFY) {      // mirrors launch_normalize
        const size_t npts = 2 * (size_t)n, nlanes = (npts + PLUME_NORM_K - 1) / PLUME_NORM_K;
        for (size_t lane = 0; lane < nlanes; lane++) normalize_points(a.res, a.resinf, npts, lane, nlanes);
    }
    for (uint32_t i = 0; i < n; i++) verify_finalize(a, i);
    return 0;
}

// The aggregate random-linear-combination check (plume_aggregate.h) through the same per-lane bodies and the same launch sequence as
// aggregate_device in plume_capi.hip.  W = 0 picks the window width the library would; carry = the record of earlier pieces (or NULL).
int ds_aggregate_check(int version, int mode, uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nul, const uint8_t* c, const uint8_t* s,
                       const uint8_t* rpt, const uint8_t* hr, const uint8_t* seed, uint64_t index_base, int W, const uint8_t* carry, uint8_t* hash_ok, uint8_t* result) {
    if (version != 1 && version != 2) return -1;
    AggArgs a; memset(&a, 0, sizeof a);
    a.version = version; a.mode = mode; a.n = n;
    if (W == 0) W = n >= (1u << 15) ? 16 : n >= 64 ? 8 : 4;
    a.W = W; a.nw_long = (256 + W - 1) / W; a.nw_short = (128 + W - 1) / W; a.nbuckets = 1u << (W - 1); a.nkeys = (uint32_t)a.nw_long * a.nbuckets;
    a.index_base = index_base; memcpy(a.seed, seed, 32);
    const size_t nn = n ? n : 1, npairs = (size_t)n * (3 * a.nw_long + 2 * a.nw_short);
    std::vector<uint32_t> bases(PLUME_BASE_WORDS * 3 * nn), scal((size_t)PLUME_AGG_TERMS * 8 * nn), gs(8 * nn, 0), count((size_t)a.nkeys + 1, 0), sorted(npairs + 1),
        bsum((size_t)PLUME_JAC_WORDS * a.nkeys), nbad(4, 0);
    std::vector<uint8_t> jobflags(3 * nn), itemflags(nn), haff(64 * nn + 16), flags(2 * nn), bsuminf(a.nkeys), hok(nn);
    a.pk = pk; a.nul = nul; a.c = c; a.s = s; a.rpt = rpt; a.hr = hr;
    a.bases = bases.data(); a.jobflags = jobflags.data(); a.itemflags = itemflags.data();
    a.haff = haff.data(); a.scal = scal.data(); a.tlive = flags.data(); a.tneg = flags.data() + nn; a.gs = gs.data(); a.hash_ok = hash_ok ? hash_ok : hok.data(); a.nbad = nbad.data();
    a.count = count.data(); a.sorted = sorted.data(); a.bsum = bsum.data(); a.bsuminf = bsuminf.data();
    a.gcomb = shared_gcomb().data(); a.result = result;
    if (n) {
        VerifyArgs v; memset(&v, 0, sizeof v);
        v.version = version; v.mode = mode; v.n = n; v.msgs = msgs; v.msg_off = msg_off; v.msgs_bytes = msg_off[n]; v.pk = pk; v.nul = nul; v.c = c; v.s = s; v.rpt = rpt; v.hr = hr;
        v.bases = bases.data(); v.jobflags = jobflags.data(); v.itemflags = itemflags.data();
        for (uint32_t i = 0; i < n; i++) verify_ingest_h2c(v, i);
        const size_t nlanes = ((size_t)n + PLUME_AGG_NORM_K - 1) / PLUME_AGG_NORM_K;
        for (size_t lane = 0; lane < nlanes; lane++) agg_normalize_h(a, lane, nlanes);
        for (uint32_t i = 0; i < n; i++) agg_item_terms(a, i);
        // the tiled counting sort; `tile` small so that several tiles and ragged last tiles occur, 3 cooperating lanes per workgroup
        const uint32_t tile = n > 40 ? 17 : 5, ntiles = (n + tile - 1) / tile, T = 7, SL = 3;
        std::vector<uint32_t> tiles((size_t)a.nw_long * ntiles * a.nbuckets, 0), bins(a.nbuckets), part(T);
        for (uint32_t blk = 0; blk < (uint32_t)a.nw_long * ntiles; blk++) {
            std::fill(bins.begin(), bins.end(), 0u);
            for (uint32_t tid = 0; tid < SL; tid++) agg_tile_pairs<false>(a, blk / ntiles, blk % ntiles, tile, tid, SL, bins.data());
            std::copy(bins.begin(), bins.end(), tiles.begin() + (size_t)blk * a.nbuckets);
        }
        for (uint32_t key = 0; key < a.nkeys; key++) agg_tile_totals(a, tiles.data(), ntiles, key);
        {   // two-level scan as the kernels do it (T range lanes, T2 lanes over their sums)
            const uint32_t T2 = 3;
            std::vector<uint32_t> top(T2);
            for (uint32_t t = 0; t < T; t++) agg_scan_phase0(a.count, a.nkeys + 1, t, T, part.data());
            for (uint32_t t = 0; t < T2; t++) agg_scan_phase0(part.data(), T, t, T2, top.data());
            agg_scan_mid(T2, top.data());
            for (uint32_t t = 0; t < T2; t++) agg_scan_phase1(part.data(), T, t, T2, top.data());
            for (uint32_t t = 0; t < T; t++) agg_scan_phase1(a.count, a.nkeys + 1, t, T, part.data());
        }
        for (uint32_t key = 0; key < a.nkeys; key++) agg_tile_offsets(a, tiles.data(), ntiles, key);
        for (uint32_t blk = (uint32_t)a.nw_long * ntiles; blk-- > 0;) {     // any workgroup order
            std::copy(tiles.begin() + (size_t)blk * a.nbuckets, tiles.begin() + (size_t)(blk + 1) * a.nbuckets, bins.begin());
            for (uint32_t tid = SL; tid-- > 0;) agg_tile_pairs<true>(a, blk / ntiles, blk % ntiles, tile, tid, SL, bins.data());
        }
    }
    // the windows in two groups, upper half first, as aggregate_device does (there the upper group's reduction runs on a second stream)
    const uint32_t nlo = (uint32_t)a.nw_long / 2, nhi = (uint32_t)a.nw_long - nlo;
    const uint32_t chunk = a.nbuckets < PLUME_AGG_CHUNK ? a.nbuckets : PLUME_AGG_CHUNK, m0 = (a.nbuckets + chunk - 1) / chunk;
    std::vector<uint32_t> perm(a.nkeys, 0xFFFFFFFFu);
    std::vector<uint32_t> red[4]; std::vector<uint8_t> rinf[4];
    int cur[2] = {0, 0};
    for (int grp = 1; grp >= 0; grp--) {
        const uint32_t j0 = grp ? nlo : 0, nwin = grp ? nhi : nlo, k0 = j0 * a.nbuckets, k1 = (j0 + nwin) * a.nbuckets;
        {   // keys by decreasing run length (3 workgroups of 2 lanes), then one lane per key in that order
            const uint32_t PB = 3, PL = 2;
            std::vector<uint32_t> hist((size_t)PB * PLUME_AGG_LEN_BINS, 0), bins(PLUME_AGG_LEN_BINS), part(PLUME_AGG_LEN_BINS);
            for (uint32_t blk = 0; blk < PB; blk++) {
                std::fill(bins.begin(), bins.end(), 0u);
                for (uint32_t tid = 0; tid < PL; tid++) agg_perm_pairs<false>(a, k0, k1, blk, PB, tid, PL, bins.data(), perm.data());
                std::copy(bins.begin(), bins.end(), hist.begin() + (size_t)blk * PLUME_AGG_LEN_BINS);
            }
            for (uint32_t bin = 0; bin < PLUME_AGG_LEN_BINS; bin++) agg_perm_phase0(hist.data(), PB, bin, part.data());
            agg_perm_mid(part.data());
            for (uint32_t bin = 0; bin < PLUME_AGG_LEN_BINS; bin++) agg_perm_phase1(hist.data(), PB, bin, part.data());
            for (uint32_t blk = 0; blk < PB; blk++) {
                std::copy(hist.begin() + (size_t)blk * PLUME_AGG_LEN_BINS, hist.begin() + (size_t)(blk + 1) * PLUME_AGG_LEN_BINS, bins.begin());
                for (uint32_t tid = 0; tid < PL; tid++) agg_perm_pairs<true>(a, k0, k1, blk, PB, tid, PL, bins.data(), perm.data());
            }
            std::vector<uint8_t> seen(a.nkeys, 0);
            uint32_t prev = 0xFFFFFFFFu;
            for (uint32_t lane = k0; lane < k1; lane++) {
                const uint32_t key = perm[lane];
                if (key < k0 || key >= k1 || seen[key]) return -2;                // a permutation of the group's keys ...
                seen[key] = 1;
                const uint32_t bin = agg_len_bin(a, key);
                if (bin > prev) return -3;                                        // ... by non-increasing length
                prev = bin;
                agg_bucket_sum(a, key);
            }
        }
        uint32_t m = m0;
        const size_t nred = (size_t)nwin * m;
        for (int k = 0; k < 2; k++) { red[2 * grp + k].assign((size_t)PLUME_JAC_WORDS * nred, 0); rinf[2 * grp + k].assign(nred, 0); }
        uint32_t** r = nullptr; (void)r;
        for (uint32_t lane = 0; lane < nwin * m; lane++) agg_chunk_reduce(a, j0 + lane / m, lane % m, chunk, m, red[2 * grp].data(), rinf[2 * grp].data(), j0, nwin);
        int c = 0;
        while (m > 1) {
            const uint32_t g = m < PLUME_AGG_GROUP ? m : PLUME_AGG_GROUP, mo = (m + g - 1) / g;
            for (uint32_t lane = 0; lane < nwin * mo; lane++)
                agg_group_sum(red[2 * grp + c].data(), rinf[2 * grp + c].data(), m, g, red[2 * grp + (c ^ 1)].data(), rinf[2 * grp + (c ^ 1)].data(), mo, nwin, lane / mo, lane % mo);
            c ^= 1; m = mo;
        }
        for (uint32_t k = 0; k < nwin; k++) agg_window_shift(a, red[2 * grp + c].data(), rinf[2 * grp + c].data(), j0 + k, j0, nwin);
        cur[grp] = c;
    }
    std::vector<uint32_t> sbuf[2];
    const uint32_t* in = gs.data();
    size_t nin = nn;
    int sc = 0;
    do {
        const size_t nout = (nin + PLUME_AGG_SUM_K - 1) / PLUME_AGG_SUM_K;
        sbuf[sc].assign(8 * nout, 0);
        for (size_t lane = 0; lane < nout; lane++) agg_scalar_sum(in, nin, sbuf[sc].data(), nout, lane);
        in = sbuf[sc].data(); nin = nout; sc ^= 1;
    } while (nin > 1);
    std::vector<uint32_t> gpt(PLUME_JAC_WORDS); uint8_t gptinf = 0;
    agg_gterm(a, in, gpt.data(), &gptinf);
    agg_final(a, red[cur[0]].data(), rinf[cur[0]].data(), nlo, red[2 + cur[1]].data(), rinf[2 + cur[1]].data(), nhi, gpt.data(), &gptinf, carry);
    return 0;
}
void ds_aggregate_combine(const uint8_t* records, uint32_t m, uint8_t* result) { agg_combine(records, m, result); }

int ds_sign_batch(int version, uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r, const uint8_t* pk_in,
                  uint8_t* pk, uint8_t* nul, uint8_t* c, uint8_t* s, uint8_t* rpt, uint8_t* hr, uint8_t* h_out, uint8_t* status, int L) {
    if (version != 1 && version != 2) return -1;
    // (the signer reads the comb only)
    const std::vector<uint32_t>& gcomb = shared_gcomb();
    std::vector<uint32_t> gres(PLUME_JAC_WORDS * 2 * (size_t)n), hres(PLUME_JAC_WORDS * 2 * (size_t)n), bases(PLUME_BASE_WORDS * PLUME_SIGN_K * (size_t)n), pkaff(2 * PLUME_FE_WORDS * (size_t)n), tab(PLUME_SIGN_K * (size_t)n * PLUME_TAB_WORDS);
    std::vector<uint8_t> gresinf(2 * (size_t)n), hresinf(2 * (size_t)n), jobflags(PLUME_SIGN_K * (size_t)n), itemflags(n);
    SignArgs a; memset(&a, 0, sizeof a);
    a.version = version; a.n = n; a.msgs = msgs; a.msg_off = msg_off; a.msgs_bytes = msg_off[n]; a.sk = sk; a.r = r; a.pk_in = pk_in;
    a.pk = pk; a.nul = nul; a.c = c; a.s = s; a.rpt = rpt; a.hr = hr; a.status = status; a.h_out = h_out;
    a.gres = gres.data(); a.gresinf = gresinf.data(); a.bases = bases.data(); a.jobflags = jobflags.data(); a.itemflags = itemflags.data();
    a.pkaff = pkaff.data(); a.tab = tab.data(); a.hres = hres.data(); a.hresinf = hresinf.data(); a.gcomb = gcomb.data(); a.gscan = shared_gscan().data(); a.uniform = g_sign_uniform;
    std::vector<int8_t> dig((2 * PLUME_NDIG + PLUME_NPOS) * B);
    for (uint32_t w = 0; w < 2; w++)
        for (uint32_t i = 0; i < n; i++) { if (g_sign_uniform == 2) sign_gmul<2>(a, i, w); else if (g_sign_uniform) sign_gmul<1>(a, i, w); else sign_gmul(a, i, w); }
    {
        const size_t npts = 2 * (size_t)n, nlanes = (npts + PLUME_NORM_K - 1) / PLUME_NORM_K;
        for (size_t lane = 0; lane < nlanes; lane++) normalize_points(a.gres, a.gresinf, npts, lane, nlanes);
    }
    for (uint32_t i = 0; i < n; i++) sign_h2c(a, i);
    for (uint32_t i = 0; i < n; i++) sign_hdbl(a, i);
    run_tables(a.tab, a.bases, a.jobflags, PLUME_SIGN_K * (size_t)n, L);
    for (uint32_t w = 0; w < 2; w++)
        for (uint32_t i = 0; i < n; i++) {
            int8_t* dg = dig.data() + (i % B);
            if (g_sign_uniform == 2) sign_hmul<2>(a, i, w, dg, B); else if (g_sign_uniform) sign_hmul<1>(a, i, w, dg, B); else sign_hmul(a, i, w, dg, B);
        }
    {
        const size_t npts = 2 * (size_t)n, nlanes = (npts + PLUME_NORM_K - 1) / PLUME_NORM_K;
        for (size_t lane = 0; lane < nlanes; lane++) normalize_points(a.hres, a.hresinf, npts, lane, nlanes);
    }
    for (uint32_t i = 0; i < n; i++) sign_final(a, i);
    return 0;
}

// R' = s*G - c*pk through the verify multi-scalar body (equation 1: wide generator digits + the pk window table)
int ds_eq1(const uint8_t s_be[32], const uint8_t c_be[32], const uint8_t pk_be[64], uint8_t out[64]) {
    std::vector<uint32_t> gtab; build_gtab(gtab);
    alignas(16) uint8_t sb[32], cb[32], pb[64], ob[64];
    memcpy(sb, s_be, 32); memcpy(cb, c_be, 32); memcpy(pb, pk_be, 64);
    fe x, y;
    uint32_t f = load_affine_be(x, y, pb);
    if (f == PLUME_JOB_INVALID) return 0;
    std::vector<uint32_t> bases(PLUME_BASE_WORDS * 3), tab(3 * PLUME_TAB_WORDS), res(PLUME_JAC_WORDS * 2);
    std::vector<uint8_t> jobflags(3), itemflags(1, 0), resinf(2);
    VerifyArgs a; memset(&a, 0, sizeof a);
    a.version = 2; a.mode = PLUME_MODE_VERIFY; a.n = 1; a.c = cb; a.s = sb; a.bases = bases.data(); a.jobflags = jobflags.data(); a.itemflags = itemflags.data();
    a.tab = tab.data(); a.res = res.data(); a.resinf = resinf.data(); a.gtab = gtab.data();
    jac p; p.x = x; p.y = y; p.z = fe_small(1); p.inf = 0;
    for (int j = 0; j < 3; j++) { st_base(a.bases, j, p); a.jobflags[j] = (uint8_t)(f | PLUME_JOB_AFFINE); }
    run_tables(a.tab, a.bases, a.jobflags, 3, 3);
    std::vector<int8_t> dig(2 * PLUME_NDIG + PLUME_NPOS), digs(PLUME_VDIG_ROWS);
    { sc sv, cv; sc_from_be_aligned(sv, sb); sc_from_be_aligned(cv, cb); glv_half c1, c2; glv_split(c1, c2, cv); verify_item_digits(digs.data(), 1, sv, c1, c2, true); }     // what the scalar stage leaves for the item (long form)
    a.digs = digs.data();
    uint32_t redo[3] = {0, 0, 0};
    a.redo = redo;
    verify_msm<false>(a, 0, 0, a.gtab, dig.data(), 1);
    if (redo[0]) verify_msm<true>(a, 0, 0, a.gtab, dig.data(), 1);
    jac r; ld_jac_soa(r, a.res, 2, 0); r.inf = a.resinf[0];
    fe ox = fe_zero(), oy = fe_zero();
    if (!r.inf) { fe zi, zi2; fe_inv(zi, r.z); fe_sqr(zi2, zi); fe_mul(ox, r.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(oy, r.y, zi2); }
    store_affine_be(ob, ox, oy, r.inf != 0);
    memcpy(out, ob, 64);
    return 1;
}

// the half-GCD in Z[w] (plume_eis.h) for n challenges c (32-byte big-endian, canonical): out[i] = t0 - 1, t1, u0, u1 as signed 128-bit little-endian integers (4 x 16 bytes),
// tau[i] = t0 + t1 lambda mod n (32 bytes BE), okf[i] = the bound flag
int ds_eis_half_gcd(uint32_t n, const uint8_t* c_be, uint8_t* out, uint8_t* tau_be, uint8_t* okf) {
    for (uint32_t i = 0; i < n; i++) {
        alignas(16) uint8_t cb[32], tb[32];
        memcpy(cb, c_be + 32 * (size_t)i, 32);
        sc c; sc_from_be_aligned(c, cb);
        eis_short e;
        glv_half g0, g1;
        glv_split(g0, g1, c);
        eis_half_gcd(e, g0, g1);
        const uint32_t (*mags[4])[3] = {&e.t[0], &e.t[1], &e.u[0], &e.u[1]};
        const uint32_t negs[4] = {e.tneg[0], e.tneg[1], e.uneg[0], e.uneg[1]};
        for (int k = 0; k < 4; k++) {
            __int128 v = 0;
            for (int w = 2; w >= 0; w--) v = (v << 32) | (*mags[k])[w];
            if (negs[k]) v = -v;
            memcpy(out + 64 * (size_t)i + 16 * k, &v, 16);
        }
        sc_to_be_aligned(tb, e.tau);
        memcpy(tau_be + 32 * (size_t)i, tb, 32);
        okf[i] = e.ok ? 1 : 0;
    }
    return 0;
}
// eis_consistent(half-GCD of c, c) for n challenges, and the same after one tamper of the pair: which = 0 none, 1 upsilon's first coefficient + 1, 2 upsilon's second
// coefficient's sign flipped, 3 tau + 1, 4 the pair of ANOTHER challenge (c + 1)
int ds_eis_consistent(uint32_t n, const uint8_t* c_be, int which, uint8_t* out) {
    for (uint32_t i = 0; i < n; i++) {
        alignas(16) uint8_t cb[32];
        memcpy(cb, c_be + 32 * (size_t)i, 32);
        sc c; sc_from_be_aligned(c, cb);
        sc c_for_pair = c;
        if (which == 4) { sc one; for (int k = 0; k < 8; k++) one.v[k] = k == 0 ? 1u : 0u; sc_add(c_for_pair, c, one); }
        eis_short e;
        glv_half g0, g1;
        glv_split(g0, g1, c_for_pair);
        eis_half_gcd(e, g0, g1);
        if (which == 1) e.u[0][0] ^= 1u;
        if (which == 2) e.uneg[1] ^= 1u;
        if (which == 3) { sc one; for (int k = 0; k < 8; k++) one.v[k] = k == 0 ? 1u : 0u; sc_add(e.tau, e.tau, one); }
        out[i] = eis_consistent(e, c) ? 1 : 0;
    }
    return 0;
}
// equation 1 in its SHORT form for one item: returns k G - upsilon pk - (tau - 1) R (which a valid signature makes equal to R) through verify_scalars, the table stage
// and the multi-scalar body; *used_long = the scalar stage fell back to the long form
int ds_eq1_short(const uint8_t s_be[32], const uint8_t c_be[32], const uint8_t pk_be[64], const uint8_t r_be[64], uint8_t out[64], int* used_long) {
    std::vector<uint32_t> gtab; build_gtab(gtab);
    alignas(16) uint8_t sb[32], cb[32], pb[64], rb[64], ob[64];
    memcpy(sb, s_be, 32); memcpy(cb, c_be, 32); memcpy(pb, pk_be, 64); memcpy(rb, r_be, 64);
    fe x, y, rx, ry;
    const uint32_t f = load_affine_be(x, y, pb), fr = load_affine_be(rx, ry, rb);
    if (f == PLUME_JOB_INVALID || fr == PLUME_JOB_INVALID) return 0;
    std::vector<uint32_t> bases(PLUME_BASE_WORDS * 4), tab(4 * PLUME_TAB_WORDS), res(PLUME_JAC_WORDS * 2), eq1k(8);
    std::vector<uint8_t> jobflags(4), itemflags(1, 0), resinf(2), eq1fall(1);
    VerifyArgs a; memset(&a, 0, sizeof a);
    a.version = 1; a.mode = PLUME_MODE_VERIFY; a.n = 1; a.c = cb; a.s = sb; a.bases = bases.data(); a.jobflags = jobflags.data(); a.itemflags = itemflags.data();
    a.tab = tab.data(); a.res = res.data(); a.resinf = resinf.data(); a.gtab = gtab.data(); a.gcomb = shared_gcomb().data(); a.eq1fall = eq1fall.data(); a.eq1k = eq1k.data();
    a.eq1force = g_eq1_short == 2 ? 1 : 0;
    jac p; p.x = x; p.y = y; p.z = fe_small(1); p.inf = 0;
    for (int j = 0; j < 3; j++) { st_base(a.bases, j, p); a.jobflags[j] = (uint8_t)(f | PLUME_JOB_AFFINE); }
    p.x = rx; p.y = ry;
    st_base(a.bases, 3, p); a.jobflags[3] = (uint8_t)(fr | PLUME_JOB_AFFINE);
    run_tables(a.tab, a.bases, a.jobflags, 4, 4);
    std::vector<int8_t> dig(2 * PLUME_NDIG + PLUME_NPOS), digs(PLUME_VDIG_ROWS);
    a.digs = digs.data();
    verify_scalars(a, 0);
    if (used_long) *used_long = eq1fall[0];
    uint32_t redo[3] = {0, 0, 0};
    a.redo = redo;
    verify_msm<false>(a, 0, 0, a.gtab, dig.data(), 1);
    if (redo[0]) verify_msm<true>(a, 0, 0, a.gtab, dig.data(), 1);
    jac r; ld_jac_soa(r, a.res, 2, 0); r.inf = a.resinf[0];
    fe ox = fe_zero(), oy = fe_zero();
    if (!r.inf) { fe zi, zi2; fe_inv(zi, r.z); fe_sqr(zi2, zi); fe_mul(ox, r.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(oy, r.y, zi2); }
    store_affine_be(ob, ox, oy, r.inf != 0);
    memcpy(out, ob, 64);
    return 1;
}

int ds_h2c_batch(uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, uint8_t* h_out) {
    H2cArgs a; a.n = n; a.msgs = msgs; a.msg_off = msg_off; a.msgs_bytes = msg_off[n]; a.pk = pk; a.h_out = h_out;
    for (uint32_t i = 0; i < n; i++) h2c_only(a, i);
    return 0;
}

int ds_h2c_intermediates(uint32_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, int registers, uint8_t* u, uint8_t* mapped, uint8_t* q, uint8_t* h, uint8_t* hints) {
    H2cInterArgs a; a.n = n; a.msgs = msgs; a.msg_off = msg_off; a.msgs_bytes = msg_off[n]; a.pk = pk; a.registers = registers; a.u = u; a.mapped = mapped; a.q = q; a.h = h; a.hints = hints;
    for (uint32_t i = 0; i < n; i++) h2c_intermediates(a, i);
    return 0;
}
// map_to_curve(u0) + map_to_curve(u1) through the device code's one-isogeny path (and its equal-x fallback): u big-endian 32 bytes each, out = affine x || y (zeros = identity)
void ds_map2_to_curve(const uint8_t* u0b, const uint8_t* u1b, uint8_t* out) {
    alignas(16) uint8_t ua[32], ub[32], ob[64];
    memcpy(ua, u0b, 32); memcpy(ub, u1b, 32);
    fe u0, u1;
    fe_from_be_aligned(u0, ua); fe_from_be_aligned(u1, ub);
    jac h;
    map2_to_curve_jac(h, u0, u1);
    fe x = fe_zero(), y = fe_zero();
    if (!h.inf) { fe zi, zi2; fe_inv(zi, h.z); fe_sqr(zi2, zi); fe_mul(x, h.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(y, h.y, zi2); }
    store_affine_be(ob, x, y, h.inf != 0);
    memcpy(out, ob, 64);
}
int ds_scalars_to_der(uint32_t n, const uint8_t* scalars, uint8_t* der, uint8_t* status) {
    const std::vector<uint32_t>& gcomb = shared_gcomb();
    DerArgs a; a.n = n; a.scalars = scalars; a.der = der; a.status = status; a.gcomb = gcomb.data(); a.gscan = shared_gscan().data(); a.uniform = g_sign_uniform;
    for (uint32_t i = 0; i < n; i++) scalar_to_sec1_der(a, i);
    return 0;
}
void ds_registers_from_be(size_t nvalues, const uint8_t* in, uint8_t* out) { for (size_t k = 0; k < nvalues; k++) registers_from_be(out, in, k); }

// k*P through the device table + msm path (single base, affine 64-byte BE in/out); returns 0 for invalid input
int ds_point_mul(const uint8_t k_be[32], const uint8_t p_be[64], uint8_t out[64]) {
    fe x, y;
    alignas(16) uint8_t pb[64], kb[32];
    memcpy(pb, p_be, 64); memcpy(kb, k_be, 32);
    uint32_t f = load_affine_be(x, y, pb);
    if (f == PLUME_JOB_INVALID) return 0;
    alignas(16) uint8_t ob[64];
    if (f == PLUME_JOB_INF) { memset(out, 0, 64); return 1; }
    std::vector<uint32_t> bases(PLUME_BASE_WORDS), tab(PLUME_TAB_WORDS);
    jac p; p.x = x; p.y = y; p.z = fe_small(1); p.inf = 0;
    st_base(bases.data(), 0, p);
    uint8_t flag = 0;
    run_tables(tab.data(), bases.data(), &flag, 1, 1);
    sc k; sc_from_be_aligned(k, kb);
    while (!sc_lt_n(k)) sc_cond_sub_n(k);
    glv_half h1, h2; glv_split(h1, h2, k);
    std::vector<int8_t> dig(2 * PLUME_NPOS);
    eisd_store_glv(dig.data(), 1, h1, h2, false);
    jac acc;
    if (!msm_run_unchecked(acc, tab.data(), nullptr, dig.data(), 1)) msm_run_checked(acc, tab.data(), nullptr, dig.data(), 1);
    fe ox = fe_zero(), oy = fe_zero();
    if (!acc.inf) { fe zi, zi2; fe_inv(zi, acc.z); fe_sqr(zi2, zi); fe_mul(ox, acc.x, zi2); fe_mul(zi2, zi2, zi); fe_mul(oy, acc.y, zi2); }
    store_affine_be(ob, ox, oy, acc.inf != 0);
    memcpy(out, ob, 64);
    return 1;
}

// nullifier-set post-processing through the per-lane bodies (lanes run one after another; order = `order`, to show the result does not depend on it)
int ds_nullifier_first_occurrence(uint32_t n, const uint8_t* nul, const uint8_t* live, const uint64_t* ids, const uint32_t* order, uint8_t* first, uint64_t* n_unique) {
    DedupArgs a;
    a.n = n; a.nul = nul; a.live = live; a.ids = ids; a.first = first;
    const uint32_t m = dedup_table_size(n);
    a.mask = m - 1;
    a.key[0] = 0x1234567u ^ n; a.key[1] = 0x9E3779B9u;
    std::vector<uint32_t> slots(m), myslot(n ? n : 1);
    std::vector<unsigned long long> minid(m);
    unsigned long long cnt = 0;
    a.slots = slots.data(); a.minid = minid.data(); a.myslot = myslot.data(); a.n_unique = &cnt; a.blockcnt = nullptr;
    for (uint32_t s2 = 0; s2 < m; s2++) dedup_clear(a, s2);
    for (uint32_t k = 0; k < n; k++) dedup_insert(a, order ? order[k] : k);
    for (uint32_t i = 0; i < n; i++) cnt += dedup_mark(a, i) ? 1 : 0;
    if (n_unique) *n_unique = cnt;
    return 0;
}
}
