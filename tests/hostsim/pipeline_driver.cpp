// TEST INFRASTRUCTURE ONLY.  Drives the HOST side of libplume_hip.so -- csrc/plume_capi.hip compiled as plain C++ against the mock runtime (mockhip/hip/hip_runtime.h) with
// the kernels as host loops (host_launch.cpp) -- through its C ABI, on several mock devices, and compares every output with the C oracle (oracle/plume_oracle.c).  Built and
// run by tests/test_sanitizers.py under AddressSanitizer + UBSan and under ThreadSanitizer: the contexts, lanes, staging slots, piece pipeline, chunk loop, shard worker
// threads and their teardown then run for real, with every caller array sized exactly on the heap and the mock's lazy / random stream scheduler running whatever the
// library did not order in the worst order.
//   pipeline_driver <group> [seed]        groups: verify sign multi benchseq device misc devapi faults all
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/plume_hip.h"

extern "C" {
int oracle_verify_batch(int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nullifier, const uint8_t* c, const uint8_t* s,
                        const uint8_t* r_point, const uint8_t* hashed_to_curve_r, uint8_t* ok, int nthreads);
int oracle_verify_non_zk_batch(int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, const uint8_t* nullifier, const uint8_t* s,
                               const uint8_t* r_point, const uint8_t* hashed_to_curve_r, const uint8_t* digest_private, uint8_t* ok, int nthreads);
int oracle_sign_batch(int version, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* sk, const uint8_t* r, const uint8_t* pk_in, uint8_t* pk,
                      uint8_t* nullifier, uint8_t* c, uint8_t* s, uint8_t* r_point, uint8_t* hashed_to_curve_r, uint8_t* h_out, uint8_t* status, int nthreads);
int oracle_aggregate_check(int version, int mode, size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk_b, const uint8_t* nul_b, const uint8_t* c_b,
                           const uint8_t* s_b, const uint8_t* r_b, const uint8_t* hr_b, const uint8_t seed[32], uint64_t index_base, uint8_t* hash_ok, uint8_t result[72], int nthreads);
int oracle_hash_to_curve_batch(size_t n, const uint8_t* msgs, const uint64_t* msg_off, const uint8_t* pk, uint8_t* h_out, int nthreads);
int oracle_point_mul(const uint8_t k[32], const uint8_t p[64], uint8_t out[64]);
size_t oracle_sec1_compress(const uint8_t p[64], uint8_t out[33]);
}

#define REQUIRE(c)                                                                                                                   \
    do {                                                                                                                             \
        if (!(c)) { std::fprintf(stderr, "pipeline_driver: %s:%d: %s   [%s] (text of the last library error on this thread, possibly an earlier, expected one: %s)\n", __FILE__, __LINE__, #c, g_what.c_str(), plume_last_error()); std::exit(2); } \
    } while (0)
static std::string g_what;
static std::mt19937_64 rng;
static uint64_t rnd(uint64_t lo, uint64_t hi) { return lo + rng() % (hi - lo + 1); }

// A caller array: exactly `bytes` long.  kind 0 = pageable heap memory, 1 = page-locked by the library's allocator, 2 = heap memory page-locked with plume_host_register
struct Arr {
    uint8_t* p = nullptr;
    size_t bytes = 0;
    int kind = 0;
    Arr() = default;
    Arr(size_t b, int k) { alloc(b, k); }
    Arr(const Arr&) = delete;
    Arr& operator=(const Arr&) = delete;
    void alloc(size_t b, int k) {
        release();
        bytes = b; kind = k;
        if (k == 1) { p = (uint8_t*)plume_host_alloc(b ? b : 1); if (!p) { std::fprintf(stderr, "plume_host_alloc failed\n"); std::exit(2); } }
        else { p = (uint8_t*)std::malloc(b ? b : 1); if (k == 2 && b && plume_host_register(p, b) != 0) { std::fprintf(stderr, "plume_host_register failed\n"); std::exit(2); } }
        std::memset(p, 0xEE, b);
    }
    void release() {
        if (!p) return;
        if (kind == 1) plume_host_free(p);
        else { if (kind == 2 && bytes) plume_host_unregister(p); std::free(p); }
        p = nullptr;
    }
    ~Arr() { release(); }
    void set(const std::vector<uint8_t>& v) { if (v.size() != bytes) { std::fprintf(stderr, "Arr::set size\n"); std::exit(2); } if (bytes) std::memcpy(p, v.data(), bytes); }
};

struct Batch {                     // a signed batch from the oracle, with some items spoiled; everything as plain vectors
    int version = 1;
    size_t n = 0;
    std::vector<uint8_t> msgs, sk, r, pk, nul, c, s, rpt, hr, h, status;
    std::vector<uint64_t> off;
    std::vector<uint8_t> ok_expect, ok_nonzk_expect;
};
static std::vector<uint8_t> random_scalar_bytes(size_t n) {
    std::vector<uint8_t> v(32 * n);
    for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)rng();
    for (size_t i = 0; i < n; i++) { v[32 * i] &= 0x7F; v[32 * i + 31] |= 1; }          // in [1, n-1] for sure
    return v;
}
static Batch make_batch(int version, size_t n, bool spoil) {
    Batch b;
    b.version = version; b.n = n;
    b.off.assign(n + 1, 0);
    for (size_t i = 0; i < n; i++) { const size_t len = rnd(0, 9) == 0 ? 0 : rnd(0, 3) == 0 ? rnd(50, 150) : rnd(1, 40); b.off[i + 1] = b.off[i] + len; }
    b.msgs.resize(b.off[n]);
    for (auto& x : b.msgs) x = (uint8_t)rng();
    b.sk = random_scalar_bytes(n); b.r = random_scalar_bytes(n);
    if (spoil) for (size_t i = 0; i < n; i++) if (rnd(0, 13) == 0) std::memset(&b.r[32 * i], 0, 32);      // nonce zero: R = Hr = identity, which the reference's verify ACCEPTS (both equations end at the identity)
    b.pk.assign(64 * n, 0); b.nul.assign(64 * n, 0); b.c.assign(32 * n, 0); b.s.assign(32 * n, 0); b.rpt.assign(64 * n, 0); b.hr.assign(64 * n, 0); b.h.assign(64 * n, 0); b.status.assign(n, 0);
    if (n) oracle_sign_batch(version, n, b.msgs.data(), b.off.data(), b.sk.data(), b.r.data(), nullptr, b.pk.data(), b.nul.data(), b.c.data(), b.s.data(), b.rpt.data(), b.hr.data(), b.h.data(), b.status.data(), 4);
    if (spoil)
        for (size_t i = 0; i < n; i++) {
            switch (rnd(0, 11)) {
                case 0: b.s[32 * i + 31] ^= 1; break;                                    // wrong s
                case 1: b.c[32 * i + 7] ^= 0x10; break;                                  // wrong c
                case 2: b.nul[64 * i + 63] ^= 1; break;                                  // nullifier off the curve
                case 3: std::memset(&b.s[32 * i], 0, 32); break;                         // s = 0
                case 4: std::memset(&b.pk[64 * i], 0xFF, 32); break;                     // pk.x >= p
                case 5: if (version == 1) b.rpt[64 * i + 5] ^= 4; break;                 // r_point wrong
                case 6: if (i > 0) std::memcpy(&b.nul[64 * i], &b.nul[64 * (i - 1)], 64); break;   // somebody else's nullifier (on the curve, wrong)
                default: break;
            }
        }
    b.ok_expect.assign(n, 0); b.ok_nonzk_expect.assign(n, 0);
    if (n) {
        oracle_verify_batch(version, n, b.msgs.data(), b.off.data(), b.pk.data(), b.nul.data(), b.c.data(), b.s.data(), b.rpt.data(), b.hr.data(), b.ok_expect.data(), 4);
        oracle_verify_non_zk_batch(version, n, b.msgs.data(), b.off.data(), b.pk.data(), b.nul.data(), b.s.data(), b.rpt.data(), b.hr.data(), b.c.data(), b.ok_nonzk_expect.data(), 4);
    }
    return b;
}
static std::vector<uint8_t> compress_all(const std::vector<uint8_t>& pts, size_t n) {
    std::vector<uint8_t> out(33 * n, 0);
    for (size_t i = 0; i < n; i++) oracle_sec1_compress(&pts[64 * i], &out[33 * i]);
    return out;
}

struct Knobs { size_t piece = 0, first = 0, tail = 0, chunk = 0, regmin = 0; int lanes = 0, eq1 = -1, uniform = -1, sub = 0; };
static void apply(plume_ctx* ctx, const Knobs& k) {
    if (k.chunk) REQUIRE(plume_set_chunk(ctx, k.chunk) == 0);
    if (k.piece) REQUIRE(plume_set_host_piece(ctx, k.piece) == 0);
    if (k.first) REQUIRE(plume_set_host_first_piece(ctx, k.first) == 0);
    if (k.tail) REQUIRE(plume_set_host_tail_piece(ctx, k.tail) == 0);
    if (k.lanes) REQUIRE(plume_set_host_lanes(ctx, k.lanes) == 0);
    if (k.regmin) REQUIRE(plume_set_host_register_min(ctx, k.regmin) == 0);
    if (k.eq1 >= 0) REQUIRE(plume_set_eq1_short(ctx, k.eq1) == 0);
    if (k.uniform >= 0) REQUIRE(plume_set_sign_uniform(ctx, k.uniform) == 0);
    if (k.sub) REQUIRE(plume_set_sub_batches(ctx, k.sub) == 0);
}

// every host-pointer verify form on one batch, arrays of memory kind `mk`
static void check_verify(plume_ctx* ctx, const Batch& b, int mk) {
    const size_t n = b.n;
    Arr msgs(b.msgs.size(), mk), off(8 * (n + 1), mk), pk(64 * n, mk), nul(64 * n, mk), c(32 * n, mk), s(32 * n, mk), rpt(64 * n, mk), hr(64 * n, mk), ok(n, mk);
    msgs.set(b.msgs); std::memcpy(off.p, b.off.data(), 8 * (n + 1)); pk.set(b.pk); nul.set(b.nul); c.set(b.c); s.set(b.s); rpt.set(b.rpt); hr.set(b.hr);
    const bool v1 = b.version == 1;
    REQUIRE(plume_verify_batch(ctx, b.version, n, msgs.p, (const uint64_t*)off.p, pk.p, nul.p, c.p, s.p, v1 ? rpt.p : nullptr, v1 ? hr.p : nullptr, ok.p) == 0);
    REQUIRE(n == 0 || std::memcmp(ok.p, b.ok_expect.data(), n) == 0);
    std::memset(ok.p, 0xEE, n);
    REQUIRE(plume_verify_non_zk_batch(ctx, b.version, n, msgs.p, (const uint64_t*)off.p, pk.p, nul.p, s.p, rpt.p, hr.p, c.p, ok.p) == 0);
    REQUIRE(n == 0 || std::memcmp(ok.p, b.ok_nonzk_expect.data(), n) == 0);
    // SEC1 records of the same points (an item whose point is off the curve has no record: its oracle verdict is 0 either way, give it a bad tag)
    Arr pk33(33 * n, mk), nul33(33 * n, mk), r33(33 * n, mk), hr33(33 * n, mk);
    pk33.set(compress_all(b.pk, n)); nul33.set(compress_all(b.nul, n)); r33.set(compress_all(b.rpt, n)); hr33.set(compress_all(b.hr, n));
    std::vector<uint8_t> expect(b.ok_expect);
    for (size_t i = 0; i < n; i++)
        if (!expect[i]) { pk33.p[33 * i] = 0x05; }            // whatever made the 64-byte form fail, the 33-byte form of this item fails on its tag
    std::memset(ok.p, 0xEE, n);
    REQUIRE(plume_verify_batch_sec1(ctx, b.version, n, msgs.p, (const uint64_t*)off.p, pk33.p, nul33.p, c.p, s.p, v1 ? r33.p : nullptr, v1 ? hr33.p : nullptr, ok.p) == 0);
    REQUIRE(n == 0 || std::memcmp(ok.p, expect.data(), n) == 0);
}

static void check_sign(plume_ctx* ctx, const Batch& b, int mk, bool supply_pk) {
    const size_t n = b.n;
    Arr msgs(b.msgs.size(), mk), off(8 * (n + 1), mk), sk(32 * n, mk), r(32 * n, mk), pkin(64 * n, mk);
    msgs.set(b.msgs); std::memcpy(off.p, b.off.data(), 8 * (n + 1)); sk.set(b.sk); r.set(b.r); pkin.set(b.pk);
    // what the oracle signs from these inputs (the batch's own signature fields may have been spoiled)
    std::vector<uint8_t> epk(64 * n), enul(64 * n), ec(32 * n), es(32 * n), erpt(64 * n), ehr(64 * n), eh(64 * n), est(n);
    if (n) oracle_sign_batch(b.version, n, b.msgs.data(), b.off.data(), b.sk.data(), b.r.data(), supply_pk ? b.pk.data() : nullptr, epk.data(), enul.data(), ec.data(), es.data(), erpt.data(), ehr.data(), eh.data(), est.data(), 4);
    {
        Arr pk(64 * n, mk), nul(64 * n, mk), c(32 * n, mk), s(32 * n, mk), rpt(64 * n, mk), hr(64 * n, mk), status(n, mk);
        REQUIRE(plume_sign_batch(ctx, b.version, n, msgs.p, (const uint64_t*)off.p, sk.p, r.p, supply_pk ? pkin.p : nullptr, pk.p, nul.p, c.p, s.p, rpt.p, hr.p, status.p) == 0);
        if (n) {
            REQUIRE(std::memcmp(status.p, est.data(), n) == 0);
            REQUIRE(std::memcmp(nul.p, enul.data(), 64 * n) == 0 && std::memcmp(c.p, ec.data(), 32 * n) == 0 && std::memcmp(s.p, es.data(), 32 * n) == 0);
            REQUIRE(std::memcmp(rpt.p, erpt.data(), 64 * n) == 0 && std::memcmp(hr.p, ehr.data(), 64 * n) == 0);
            if (!supply_pk) REQUIRE(std::memcmp(pk.p, epk.data(), 64 * n) == 0);
        }
    }
    {
        Arr nul33(33 * n, mk), c(32 * n, mk), s(32 * n, mk), r33(33 * n, mk), hr33(33 * n, mk), status(n, mk);
        REQUIRE(plume_sign_batch_sec1(ctx, b.version, n, msgs.p, (const uint64_t*)off.p, sk.p, r.p, supply_pk ? pkin.p : nullptr, nullptr, nul33.p, c.p, s.p, r33.p, hr33.p, status.p) == 0);
        if (n) {
            REQUIRE(std::memcmp(nul33.p, compress_all(enul, n).data(), 33 * n) == 0 && std::memcmp(r33.p, compress_all(erpt, n).data(), 33 * n) == 0);
            REQUIRE(std::memcmp(hr33.p, compress_all(ehr, n).data(), 33 * n) == 0 && std::memcmp(s.p, es.data(), 32 * n) == 0);
        }
    }
}

static void check_aggregate(plume_ctx* ctx, const Batch& b, int mk, int mode) {
    const size_t n = b.n;
    Arr msgs(b.msgs.size(), mk), off(8 * (n + 1), mk), pk(64 * n, mk), nul(64 * n, mk), c(32 * n, mk), s(32 * n, mk), rpt(64 * n, mk), hr(64 * n, mk), hok(n, mk);
    msgs.set(b.msgs); std::memcpy(off.p, b.off.data(), 8 * (n + 1)); pk.set(b.pk); nul.set(b.nul); c.set(b.c); s.set(b.s); rpt.set(b.rpt); hr.set(b.hr);
    uint8_t seed[32], res[72], eres[72];
    for (auto& x : seed) x = (uint8_t)rng();
    std::vector<uint8_t> ehok(n);
    REQUIRE(oracle_aggregate_check(b.version, mode, n, b.msgs.data(), b.off.data(), b.pk.data(), b.nul.data(), b.c.data(), b.s.data(), b.rpt.data(), b.hr.data(), seed, 0, ehok.data(), eres, 4) == 0);
    REQUIRE(plume_aggregate_check(ctx, b.version, mode, n, msgs.p, (const uint64_t*)off.p, pk.p, nul.p, c.p, s.p, rpt.p, hr.p, seed, hok.p, res) == 0);
    REQUIRE(std::memcmp(res, eres, 72) == 0);
    REQUIRE(n == 0 || std::memcmp(hok.p, ehok.data(), n) == 0);
    REQUIRE(plume_aggregate_check(ctx, b.version, mode, n, msgs.p, (const uint64_t*)off.p, pk.p, nul.p, c.p, s.p, rpt.p, hr.p, nullptr, nullptr, res) == 0);     // the library's own seed
    REQUIRE(res[0] == eres[0] || eres[1] == 0);              // an all-true batch stays all-true under any seed
}

static void check_h2c_and_friends(plume_ctx* ctx, const Batch& b, int mk) {
    const size_t n = b.n;
    Arr msgs(b.msgs.size(), mk), off(8 * (n + 1), mk), pk(64 * n, mk), h(64 * n, mk);
    msgs.set(b.msgs); std::memcpy(off.p, b.off.data(), 8 * (n + 1));
    std::vector<uint8_t> good_pk(64 * n), eh(64 * n);
    static const uint8_t G[64] = {0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62, 0x95, 0xCE, 0x87, 0x0B, 0x07, 0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE, 0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98,
                                  0x48, 0x3A, 0xDA, 0x77, 0x26, 0xA3, 0xC4, 0x65, 0x5D, 0xA4, 0xFB, 0xFC, 0x0E, 0x11, 0x08, 0xA8, 0xFD, 0x17, 0xB4, 0x48, 0xA6, 0x85, 0x54, 0x19, 0x9C, 0x47, 0xD0, 0x8F, 0xFB, 0x10, 0xD4, 0xB8};
    for (size_t i = 0; i < n; i++) oracle_point_mul(&b.sk[32 * i], G, &good_pk[64 * i]);
    pk.set(good_pk);
    if (n) oracle_hash_to_curve_batch(n, b.msgs.data(), b.off.data(), good_pk.data(), eh.data(), 4);
    REQUIRE(plume_hash_to_curve_batch(ctx, n, msgs.p, (const uint64_t*)off.p, pk.p, h.p) == 0);
    REQUIRE(n == 0 || std::memcmp(h.p, eh.data(), 64 * n) == 0);
    {   // the intermediates: h must agree with the call above in both encodings; u, mapped, q are sized exactly
        Arr u(64 * n, mk), mapped(128 * n, mk), q(128 * n, mk), h2(64 * n, mk), hints(192 * n, mk);
        REQUIRE(plume_h2c_intermediates_batch(ctx, n, msgs.p, (const uint64_t*)off.p, pk.p, 0, u.p, mapped.p, q.p, h2.p) == 0);
        REQUIRE(n == 0 || std::memcmp(h2.p, eh.data(), 64 * n) == 0);
        REQUIRE(plume_h2c_intermediates_batch(ctx, n, msgs.p, (const uint64_t*)off.p, pk.p, 1, nullptr, nullptr, nullptr, h2.p) == 0);
        std::vector<uint64_t> regs(8 * n);
        REQUIRE(plume_registers_from_be(2 * n, eh.data(), regs.data()) == 0);
        REQUIRE(n == 0 || std::memcmp(h2.p, regs.data(), 64 * n) == 0);
        REQUIRE(plume_h2c_hints_batch(ctx, n, msgs.p, (const uint64_t*)off.p, pk.p, 0, hints.p) == 0);
    }
    {   // SEC1-DER export of the secret keys and the checked import of the records
        Arr sk(32 * n, mk), der(109 * n, mk), st(n, mk), back(32 * n, mk), okf(n, mk);
        sk.set(b.sk);
        if (n > 2) std::memset(sk.p + 32, 0, 32);                                     // a scalar no SecretKey holds
        REQUIRE(plume_scalars_to_sec1_der_batch(ctx, n, sk.p, der.p, st.p) == 0);
        for (size_t i = 0; i < n; i++) {
            const bool bad = n > 2 && i == 1;
            REQUIRE(st.p[i] == (bad ? PLUME_STATUS_BAD_SCALAR : 0));
            if (bad) continue;
            REQUIRE(std::memcmp(der.p + 109 * i + 7, &b.sk[32 * i], 32) == 0 && std::memcmp(der.p + 109 * i + 45, &good_pk[64 * i], 64) == 0);
        }
        if (n > 3) der.p[109 * 3 + 100] ^= 1;                                         // a tampered public-key field
        REQUIRE(plume_sec1_der_to_scalars_checked(ctx, n, der.p, back.p, okf.p) == 0);
        for (size_t i = 0; i < n; i++) {
            const bool bad = (n > 2 && i == 1) || (n > 3 && i == 3);
            REQUIRE(okf.p[i] == (bad ? 0 : 1));
            if (!bad) REQUIRE(std::memcmp(back.p + 32 * i, &b.sk[32 * i], 32) == 0);
        }
    }
    {   // first occurrences among repeated nullifiers
        Arr nul(64 * n, mk), live(n, mk), first(n, mk);
        std::vector<uint8_t> v(b.nul);
        for (size_t i = 0; i + 1 < n; i += 3) std::memcpy(&v[64 * (i + 1)], &v[64 * i], 64);
        nul.set(v);
        for (size_t i = 0; i < n; i++) live.p[i] = (i % 7) != 6;
        uint64_t nu = ~0ull, expect_nu = 0;
        REQUIRE(plume_nullifier_first_occurrence(ctx, n, nul.p, live.p, nullptr, first.p, &nu) == 0);
        for (size_t i = 0; i < n; i++) {
            bool f = live.p[i] != 0;
            for (size_t j = 0; j < i && f; j++) if (live.p[j] && !std::memcmp(&v[64 * j], &v[64 * i], 64)) f = false;
            REQUIRE(first.p[i] == (f ? 1 : 0));
            expect_nu += f;
        }
        REQUIRE(nu == expect_nu);
    }
}

// ------------------------------------------------------------------------------------------------------------------ groups
static void group_verify(uint64_t seed) {
    plume_ctx* ctx = nullptr;
    REQUIRE(plume_init(&ctx, 0) == 0);
    const Batch v1 = make_batch(1, 150 + seed % 7, true), v2 = make_batch(2, 70, true), tiny = make_batch(1, 3, false), empty = make_batch(1, 0, false);
    struct Case { const char* name; Knobs k; int mk; };
    const Case cases[] = {
        {"defaults, pageable arrays", {}, 0},
        {"small pieces, pageable arrays (one lane, staged copies)", {40, 12, 12, 0, 0, 0, 3, -1, 0}, 0},
        {"small pieces, page-locked arrays (two lanes, four slots), short first equation", {40, 12, 12, 0, 0, 2, 3, -1, 0}, 1},
        {"small pieces, registered arrays, one lane, long first equation", {33, 7, 7, 0, 0, 1, 0, -1, 0}, 2},
        {"pieces larger than the chunk, registration for the call, every item through the checked chain", {64, 16, 16, 24, 1, 2, 2, -1, 0}, 0},
        {"sub-batches on the caller's stream", {0, 0, 0, 0, 0, 0, 1, -1, 2}, 1},
    };
    for (const Case& cs : cases) {
        g_what = cs.name;
        apply(ctx, cs.k);
        check_verify(ctx, v1, cs.mk);
        check_verify(ctx, v2, cs.mk);
        check_verify(ctx, tiny, cs.mk);
        check_verify(ctx, empty, cs.mk);
    }
    g_what = "argument errors";
    uint8_t ok1[4] = {9, 9, 9, 9};
    REQUIRE(plume_verify_batch(ctx, 3, 3, tiny.msgs.data(), tiny.off.data(), tiny.pk.data(), tiny.nul.data(), tiny.c.data(), tiny.s.data(), tiny.rpt.data(), tiny.hr.data(), ok1) != 0);
    REQUIRE(plume_verify_batch(ctx, 1, 3, tiny.msgs.data(), tiny.off.data(), tiny.pk.data(), tiny.nul.data(), tiny.c.data(), tiny.s.data(), nullptr, nullptr, ok1) != 0);      // V1 needs R and Hr
    REQUIRE(plume_verify_batch(nullptr, 1, 3, tiny.msgs.data(), tiny.off.data(), tiny.pk.data(), tiny.nul.data(), tiny.c.data(), tiny.s.data(), tiny.rpt.data(), tiny.hr.data(), ok1) != 0);
    {   // message offsets that run backwards: the items are rejected, nothing outside msgs is read
        std::vector<uint64_t> bad(tiny.off);
        bad[1] = bad[3] + 1000;
        REQUIRE(plume_verify_batch(ctx, 1, 3, tiny.msgs.data(), bad.data(), tiny.pk.data(), tiny.nul.data(), tiny.c.data(), tiny.s.data(), tiny.rpt.data(), tiny.hr.data(), ok1) != 0 || (ok1[0] == 0 && ok1[1] == 0));
    }
    plume_destroy(ctx);
}

static void group_sign(uint64_t seed) {
    plume_ctx* ctx = nullptr;
    REQUIRE(plume_init(&ctx, 1) == 0);
    Batch v1 = make_batch(1, 90 + seed % 5, false), v2 = make_batch(2, 40, false), empty = make_batch(2, 0, false);
    std::memset(&v1.sk[32 * 4], 0, 32);                       // sk = 0: BAD_SCALAR
    std::memset(&v1.r[32 * 9], 0xFF, 32);                     // r >= n
    struct Case { const char* name; Knobs k; int mk; bool supply_pk; };
    const Case cases[] = {
        {"defaults (uniform level 1), pageable arrays", {}, 0, false},
        {"tapered small pieces on one lane (pageable arrays), level 0", {24, 0, 6, 0, 0, 0, -1, 0, 0}, 0, false},
        {"uniform small pieces dealt to two lanes (page-locked arrays), level 0", {24, 0, 6, 0, 0, 0, -1, 0, 0}, 1, false},
        {"uniform small pieces on two lanes, registered arrays, level 2 (scanned tables), pk supplied", {24, 0, 6, 0, 0, 0, -1, 2, 0}, 2, true},
        {"chunk smaller than the pieces, registration for the call, level 1", {32, 0, 8, 20, 1, 0, -1, 1, 0}, 0, true},
    };
    for (const Case& cs : cases) {
        g_what = cs.name;
        apply(ctx, cs.k);
        REQUIRE(cs.k.uniform < 0 || plume_get_sign_uniform(ctx) == cs.k.uniform);
        check_sign(ctx, v1, cs.mk, cs.supply_pk);
        check_sign(ctx, v2, cs.mk, cs.supply_pk);
        check_sign(ctx, empty, cs.mk, cs.supply_pk);
    }
    plume_destroy(ctx);
}

static void group_multi(uint64_t seed) {
    int ids[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    plume_ctx* ctx = nullptr;
    g_what = "eight shards, eight devices";
    REQUIRE(plume_init_multi(&ctx, ids, 8) == 0);
    REQUIRE(plume_num_shards(ctx) == 8);
    for (int d = 0; d < 8; d++) REQUIRE(plume_shard_numa_node(ctx, d) == -1);
    Knobs k; k.piece = 16; k.first = 5; k.tail = 5; k.eq1 = 3; k.lanes = 2;
    apply(ctx, k);
    const Batch v1 = make_batch(1, 203 + seed % 11, true), v2 = make_batch(2, 61, true), few = make_batch(1, 5, false), empty = make_batch(1, 0, false);
    for (int mk : {1, 0}) {
        check_verify(ctx, v1, mk); check_verify(ctx, v2, mk); check_verify(ctx, few, mk); check_verify(ctx, empty, mk);
        check_sign(ctx, v2, mk, false); check_sign(ctx, few, mk, true); check_sign(ctx, empty, mk, false);
    }
    const Batch clean = make_batch(1, 97, false), clean2 = make_batch(2, 41, false);
    check_aggregate(ctx, clean, 1, 0); check_aggregate(ctx, clean, 0, 1); check_aggregate(ctx, clean2, 1, 1); check_aggregate(ctx, v1, 1, 0); check_aggregate(ctx, few, 0, 0); check_aggregate(ctx, empty, 0, 0);
    check_h2c_and_friends(ctx, v2, 1);
    check_h2c_and_friends(ctx, few, 0);
    {   // device-resident calls need a single-device context
        uint8_t ok1[8];
        REQUIRE(plume_verify_batch_device(ctx, 1, 5, few.msgs.data(), few.off.data(), few.msgs.size(), few.pk.data(), few.nul.data(), few.c.data(), few.s.data(), few.rpt.data(), few.hr.data(), ok1, nullptr) == PLUME_ERR_ARG);
    }
    plume_destroy(ctx);
    g_what = "three shards on two devices, used from two caller threads in turn";
    int ids2[3] = {2, 2, 5};
    setenv("PLUME_MSM_PAIR_MAX", "0", 1);                    // (these shards: one lane per chain whatever the size)
    REQUIRE(plume_init_multi(&ctx, ids2, 3) == 0);
    unsetenv("PLUME_MSM_PAIR_MAX");
    apply(ctx, k);
    std::thread t([&] { g_what = "second caller thread"; check_verify(ctx, v2, 1); });
    t.join();
    check_verify(ctx, v1, 0);
    plume_destroy(ctx);
    g_what = "bad device lists";
    int bad[2] = {0, 99};
    REQUIRE(plume_init_multi(&ctx, bad, 2) != 0 && ctx == nullptr);
    REQUIRE(plume_init_multi(&ctx, ids, 0) != 0);
    REQUIRE(plume_init(&ctx, -1) != 0);
}

// bench.py --multi-ctx's call sequence, literally (bench.py multi_ctx_main): ONE plume_init_multi context over devices 0..7 with the library's DEFAULT knobs, the set-up sign
// from pageable arrays, then warm-up + timed verify calls of the whole batch from the same page-locked arrays into the same page-locked verdict array, close.  What the first run
// on a real 8-GPU node executes on the host side, with nothing left to discover there.
static void group_benchseq(uint64_t seed) {
    int ids[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    for (int ver : {1, 2}) {
        plume_ctx* ctx = nullptr;
        g_what = "bench.py --multi-ctx sequence";
        REQUIRE(plume_init_multi(&ctx, ids, 8) == 0 && plume_num_shards(ctx) == 8);
        const size_t n = 8 * 23 + 5 + seed % 7;
        Batch b = make_batch(ver, n, false);
        // the set-up pass: eng.sign_batch(ver, msgs, off, sk, r) on pageable arrays
        std::vector<uint8_t> pk(64 * n), nul(64 * n), c(32 * n), s(32 * n), rpt(64 * n), hr(64 * n), st(n, 0xEE);
        REQUIRE(plume_sign_batch(ctx, ver, n, b.msgs.data(), b.off.data(), b.sk.data(), b.r.data(), nullptr, pk.data(), nul.data(), c.data(), s.data(), rpt.data(), hr.data(), st.data()) == 0);
        REQUIRE(pk == b.pk && nul == b.nul && c == b.c && s == b.s && rpt == b.rpt && hr == b.hr);
        for (size_t i = 0; i < n; i++) REQUIRE(st[i] == 0);
        // synth.corrupt_for_verify: items with i mod 16 == 5, kind (i div 16) mod 4
        std::vector<uint8_t> msgs = b.msgs;
        for (size_t i = 5; i < n; i += 16) {
            switch ((i / 16) % 4) {
                case 0: s[32 * i + 31] ^= 1; break;
                case 1: c[32 * i + 31] ^= 1; break;
                case 2: std::memcpy(&nul[64 * i], &b.nul[64 * (i - 1)], 64); break;
                default: if (ver == 1) { std::memcpy(&rpt[64 * i], &b.hr[64 * i], 64); std::memcpy(&hr[64 * i], &b.rpt[64 * i], 64); } else if (b.off[i + 1] > b.off[i]) msgs[b.off[i]] ^= 1; break;
            }
        }
        std::vector<uint8_t> want(n);
        oracle_verify_batch(ver, n, msgs.data(), b.off.data(), pk.data(), nul.data(), c.data(), s.data(), rpt.data(), hr.data(), want.data(), 4);
        // capi.pinned_copy of every array, capi.pinned_empty for the verdicts
        Arr pm(msgs.size(), 1), po(8 * (n + 1), 1), ppk(64 * n, 1), pnul(64 * n, 1), pc(32 * n, 1), ps(32 * n, 1), prp(64 * n, 1), phr(64 * n, 1), pok(n, 1);
        pm.set(msgs); std::memcpy(po.p, b.off.data(), 8 * (n + 1)); ppk.set(pk); pnul.set(nul); pc.set(c); ps.set(s); prp.set(rpt); phr.set(hr);
        for (int step = 0; step < 1 + 3; step++) {                                // --warmup 1 --steps 3
            std::memset(pok.p, 0xEE, n);
            REQUIRE(plume_verify_batch(ctx, ver, n, (const uint8_t*)pm.p, (const uint64_t*)po.p, (const uint8_t*)ppk.p, (const uint8_t*)pnul.p, (const uint8_t*)pc.p, (const uint8_t*)ps.p,
                                       ver == 1 ? (const uint8_t*)prp.p : nullptr, ver == 1 ? (const uint8_t*)phr.p : nullptr, (uint8_t*)pok.p) == 0);
            REQUIRE(std::memcmp(pok.p, want.data(), n) == 0);
        }
        plume_destroy(ctx);
    }
}

// device-resident calls: the caller owns the device arrays and the streams
struct DevArr {
    uint8_t* p = nullptr; size_t bytes;
    DevArr(const void* src, size_t b) : bytes(b) { if (hipMalloc((void**)&p, b ? b : 1) != hipSuccess) std::exit(2); if (src && b) (void)hipMemcpy(p, src, b, hipMemcpyHostToDevice); }
    ~DevArr() { (void)hipFree(p); }
};
static void group_device(uint64_t seed) {
    plume_ctx* ctx = nullptr;
    setenv("PLUME_SPLIT_SCALARS", "0", 1);                   // this context: the two-role ingest kernel WITHOUT the scalar stage in its idle role (the A/B form)
    REQUIRE(plume_init(&ctx, 3) == 0);
    unsetenv("PLUME_SPLIT_SCALARS");
    REQUIRE(hipSetDevice(3) == hipSuccess);
    {   // stage timing is off by default: the hook says so instead of reporting stale times
        float ms0[4]; const char* nm0[4];
        REQUIRE(plume_last_stage_times(ctx, nm0, ms0, 4) == PLUME_ERR_ARG);
        REQUIRE(plume_set_stage_timing(ctx, 1) == 0);
    }
    const Batch a = make_batch(1, 130 + seed % 3, true), b = make_batch(2, 77, true);
    hipStream_t s1, s2;
    REQUIRE(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess);
    for (int in_flight : {1, 2}) {
        for (int sub : {1, 2}) {
            g_what = "device-resident verify: in_flight " + std::to_string(in_flight) + ", sub-batches " + std::to_string(sub);
            REQUIRE(plume_set_in_flight(ctx, in_flight) == 0);
            REQUIRE(plume_set_sub_batches(ctx, sub) == 0);
            REQUIRE(plume_set_eq1_short(ctx, sub == 1 ? 3 : 0) == 0);
            DevArr am(a.msgs.data(), a.msgs.size()), ao(a.off.data(), 8 * (a.n + 1)), apk(a.pk.data(), 64 * a.n), an(a.nul.data(), 64 * a.n), ac(a.c.data(), 32 * a.n), as(a.s.data(), 32 * a.n), ar(a.rpt.data(), 64 * a.n),
                ah(a.hr.data(), 64 * a.n), aok(nullptr, a.n), aok2(nullptr, a.n);
            DevArr bm(b.msgs.data(), b.msgs.size()), bo(b.off.data(), 8 * (b.n + 1)), bpk(b.pk.data(), 64 * b.n), bn(b.nul.data(), 64 * b.n), bc(b.c.data(), 32 * b.n), bs(b.s.data(), 32 * b.n), bok(nullptr, b.n);
            // three calls on two streams and the context's own, nothing waited for in between: they share the context's workspace(s)
            REQUIRE(plume_verify_batch_device(ctx, 1, a.n, am.p, (const uint64_t*)ao.p, a.msgs.size(), apk.p, an.p, ac.p, as.p, ar.p, ah.p, aok.p, s1) == 0);
            REQUIRE(plume_verify_batch_device(ctx, 2, b.n, bm.p, (const uint64_t*)bo.p, b.msgs.size(), bpk.p, bn.p, bc.p, bs.p, nullptr, nullptr, bok.p, s2) == 0);
            REQUIRE(plume_verify_non_zk_batch_device(ctx, 1, a.n, am.p, (const uint64_t*)ao.p, a.msgs.size(), apk.p, an.p, as.p, ar.p, ah.p, ac.p, aok2.p, nullptr) == 0);
            REQUIRE(plume_verify_batch_device(ctx, 1, (size_t)1 << 30, am.p, (const uint64_t*)ao.p, a.msgs.size(), apk.p, an.p, ac.p, as.p, ar.p, ah.p, aok.p, s1) != 0);       // more than a chunk
            std::vector<uint8_t> ra(a.n), rb(b.n), ra2(a.n);
            REQUIRE(hipMemcpyAsync(rb.data(), bok.p, b.n, hipMemcpyDeviceToHost, s2) == hipSuccess);       // pageable destination: blocks until s2 got there
            REQUIRE(hipMemcpyAsync(ra.data(), aok.p, a.n, hipMemcpyDeviceToHost, s1) == hipSuccess);
            REQUIRE(hipMemcpyAsync(ra2.data(), aok2.p, a.n, hipMemcpyDeviceToHost, nullptr) == hipSuccess);   // NULL = the context's stream in the call; here the null stream ...
            REQUIRE(ra == a.ok_expect && rb == b.ok_expect);
            float ms[16]; const char* names[16]; const int nst = plume_last_stage_times(ctx, names, ms, 16);                      // ... so wait for the context's last call through its own API
            REQUIRE(nst >= 4);
            REQUIRE(hipMemcpy(ra2.data(), aok2.p, a.n, hipMemcpyDeviceToHost) == hipSuccess);
            REQUIRE(ra2 == a.ok_nonzk_expect);
        }
    }
    {
        g_what = "device-resident sign, then destroy with the call still queued";
        REQUIRE(plume_set_in_flight(ctx, 1) == 0);
        const Batch& q = b;
        DevArr m(q.msgs.data(), q.msgs.size()), o(q.off.data(), 8 * (q.n + 1)), sk(q.sk.data(), 32 * q.n), r(q.r.data(), 32 * q.n), pk(nullptr, 64 * q.n), nul(nullptr, 64 * q.n), c(nullptr, 32 * q.n), s(nullptr, 32 * q.n),
            rpt(nullptr, 64 * q.n), hr(nullptr, 64 * q.n), st(nullptr, q.n);
        REQUIRE(plume_sign_batch_device(ctx, 2, q.n, m.p, (const uint64_t*)o.p, q.msgs.size(), sk.p, r.p, nullptr, pk.p, nul.p, c.p, s.p, rpt.p, hr.p, st.p, s1) == 0);
        plume_destroy(ctx);                                   // has to wait for the call on s1: its workspace is what the queued kernels write
        REQUIRE(hipStreamSynchronize(s1) == hipSuccess);
        std::vector<uint8_t> got(64 * q.n), es(32 * q.n), enul(64 * q.n), d1(64 * q.n), d2(32 * q.n), d3(64 * q.n), d4(64 * q.n), d5(64 * q.n), d6(q.n);
        oracle_sign_batch(2, q.n, q.msgs.data(), q.off.data(), q.sk.data(), q.r.data(), nullptr, d1.data(), enul.data(), d2.data(), es.data(), d3.data(), d4.data(), d5.data(), d6.data(), 4);
        REQUIRE(hipMemcpy(got.data(), nul.p, 64 * q.n, hipMemcpyDeviceToHost) == hipSuccess && got == enul);
        got.resize(32 * q.n);
        REQUIRE(hipMemcpy(got.data(), s.p, 32 * q.n, hipMemcpyDeviceToHost) == hipSuccess && got == es);
    }
    REQUIRE(hipStreamDestroy(s1) == hipSuccess && hipStreamDestroy(s2) == hipSuccess);
}

// every device-resident entry point the other groups do not reach, against the host-pointer form of the same context (itself checked against the oracle by the other
// groups); verify and sign in the OVERLAPPED sub-batch order (a second stream of the context runs the stages in front of the multi-scalar kernel: pre stream, pre_ready events)
static void group_devapi(uint64_t seed) {
    plume_ctx* ctx = nullptr;
    setenv("PLUME_OVERLAP_MIN", "8", 1);                      // sub-batches overlap from 8 items on (default 2^17)
    REQUIRE(plume_init(&ctx, 4) == 0);
    unsetenv("PLUME_OVERLAP_MIN");
    REQUIRE(hipSetDevice(4) == hipSuccess);
    hipStream_t st;
    REQUIRE(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
    const Batch a = make_batch(1, 120 + seed % 5, true), v2 = make_batch(2, 60, true);
    auto dl = [&](const DevArr& d) { std::vector<uint8_t> h(d.bytes); REQUIRE(hipMemcpyAsync(h.data(), d.p, d.bytes, hipMemcpyDeviceToHost, st) == hipSuccess); return h; };   // pageable: blocks until st got there
    for (const Batch* bp : {&a, &v2}) {
        const Batch& b = *bp;
        const size_t n = b.n;
        const bool v1 = b.version == 1;
        g_what = "device-resident entry points, V" + std::to_string(b.version);
        DevArr m(b.msgs.data(), b.msgs.size()), o(b.off.data(), 8 * (n + 1)), pk(b.pk.data(), 64 * n), nul(b.nul.data(), 64 * n), c(b.c.data(), 32 * n), s(b.s.data(), 32 * n), rp(b.rpt.data(), 64 * n), hr(b.hr.data(), 64 * n);
        const uint64_t* off = (const uint64_t*)o.p;
        for (int sub : {3, 1}) {                               // overlapped order, then the serial one
            REQUIRE(plume_set_sub_batches(ctx, sub) == 0);
            {   // verify, SEC1 ingest
                const std::vector<uint8_t> pk33 = compress_all(b.pk, n), nul33 = compress_all(b.nul, n), r33 = compress_all(b.rpt, n), hr33 = compress_all(b.hr, n);
                std::vector<uint8_t> expect(n), okh(n);
                REQUIRE(plume_verify_batch_sec1(ctx, b.version, n, b.msgs.data(), b.off.data(), pk33.data(), nul33.data(), b.c.data(), b.s.data(), v1 ? r33.data() : nullptr, v1 ? hr33.data() : nullptr, expect.data()) == 0);
                DevArr dpk(pk33.data(), 33 * n), dnul(nul33.data(), 33 * n), dr(r33.data(), 33 * n), dhr(hr33.data(), 33 * n), ok(nullptr, n), ok64(nullptr, n);
                REQUIRE(plume_verify_batch_sec1_device(ctx, b.version, n, m.p, off, b.msgs.size(), dpk.p, dnul.p, c.p, s.p, v1 ? dr.p : nullptr, v1 ? dhr.p : nullptr, ok.p, st) == 0);
                REQUIRE(plume_verify_batch_device(ctx, b.version, n, m.p, off, b.msgs.size(), pk.p, nul.p, c.p, s.p, v1 ? rp.p : nullptr, v1 ? hr.p : nullptr, ok64.p, st) == 0);
                REQUIRE(dl(ok) == expect && dl(ok64) == b.ok_expect);
            }
            {   // sign, both output forms
                DevArr sk(b.sk.data(), 32 * n), r(b.r.data(), 32 * n), opk(nullptr, 64 * n), onul(nullptr, 64 * n), oc(nullptr, 32 * n), os(nullptr, 32 * n), orp(nullptr, 64 * n), ohr(nullptr, 64 * n), ost(nullptr, n);
                DevArr pk33(nullptr, 33 * n), nul33(nullptr, 33 * n), oc2(nullptr, 32 * n), os2(nullptr, 32 * n), r33(nullptr, 33 * n), hr33(nullptr, 33 * n), ost2(nullptr, n);
                REQUIRE(plume_sign_batch_device(ctx, b.version, n, m.p, off, b.msgs.size(), sk.p, r.p, nullptr, opk.p, onul.p, oc.p, os.p, orp.p, ohr.p, ost.p, st) == 0);
                REQUIRE(plume_sign_batch_sec1_device(ctx, b.version, n, m.p, off, b.msgs.size(), sk.p, r.p, nullptr, pk33.p, nul33.p, oc2.p, os2.p, r33.p, hr33.p, ost2.p, st) == 0);
                std::vector<uint8_t> epk(64 * n), enul(64 * n), ec(32 * n), es(32 * n), erp(64 * n), ehr(64 * n), eh(64 * n), est(n);
                oracle_sign_batch(b.version, n, b.msgs.data(), b.off.data(), b.sk.data(), b.r.data(), nullptr, epk.data(), enul.data(), ec.data(), es.data(), erp.data(), ehr.data(), eh.data(), est.data(), 4);
                REQUIRE(dl(opk) == epk && dl(onul) == enul && dl(oc) == ec && dl(os) == es && dl(orp) == erp && dl(ohr) == ehr && dl(ost) == est);
                REQUIRE(dl(pk33) == compress_all(epk, n) && dl(nul33) == compress_all(enul, n) && dl(r33) == compress_all(erp, n) && dl(hr33) == compress_all(ehr, n) && dl(os2) == es && dl(oc2) == ec);
            }
        }
        {   // hash_to_curve and its intermediates, hints, DER, registers, first occurrences
            std::vector<uint8_t> eh(64 * n), eu(64 * n), emap(128 * n), eq(128 * n), eh2(64 * n), ehint(192 * n), eder(109 * n), edst(n), efirst(n);
            uint64_t enu = 0;
            REQUIRE(plume_hash_to_curve_batch(ctx, n, b.msgs.data(), b.off.data(), b.pk.data(), eh.data()) == 0);
            REQUIRE(plume_h2c_intermediates_batch(ctx, n, b.msgs.data(), b.off.data(), b.pk.data(), 1, eu.data(), emap.data(), eq.data(), eh2.data()) == 0);
            REQUIRE(plume_h2c_hints_batch(ctx, n, b.msgs.data(), b.off.data(), b.pk.data(), 0, ehint.data()) == 0);
            REQUIRE(plume_scalars_to_sec1_der_batch(ctx, n, b.sk.data(), eder.data(), edst.data()) == 0);
            REQUIRE(plume_nullifier_first_occurrence(ctx, n, b.nul.data(), nullptr, nullptr, efirst.data(), &enu) == 0);
            DevArr h(nullptr, 64 * n), u(nullptr, 64 * n), mp(nullptr, 128 * n), q(nullptr, 128 * n), h2(nullptr, 64 * n), hint(nullptr, 192 * n), sk(b.sk.data(), 32 * n), der(nullptr, 109 * n), dst(nullptr, n),
                first(nullptr, n), nu(nullptr, 8), regs(nullptr, 64 * n);
            REQUIRE(plume_hash_to_curve_batch_device(ctx, n, m.p, off, b.msgs.size(), pk.p, h.p, st) == 0);
            REQUIRE(plume_h2c_intermediates_batch_device(ctx, n, m.p, off, b.msgs.size(), pk.p, 1, u.p, mp.p, q.p, h2.p, st) == 0);
            REQUIRE(plume_h2c_hints_batch_device(ctx, n, m.p, off, b.msgs.size(), pk.p, 0, hint.p, st) == 0);
            REQUIRE(plume_scalars_to_sec1_der_batch_device(ctx, n, sk.p, der.p, dst.p, st) == 0);
            REQUIRE(plume_nullifier_first_occurrence_device(ctx, n, nul.p, nullptr, nullptr, first.p, (uint64_t*)nu.p, st) == 0);
            REQUIRE(plume_registers_from_be_device(ctx, 2 * n, h.p, (uint64_t*)regs.p, st) == 0);
            REQUIRE(dl(h) == eh && dl(u) == eu && dl(mp) == emap && dl(q) == eq && dl(h2) == eh2 && dl(hint) == ehint && dl(der) == eder && dl(dst) == edst && dl(first) == efirst);
            std::vector<uint8_t> nub = dl(nu);
            REQUIRE(std::memcmp(nub.data(), &enu, 8) == 0);
            std::vector<uint64_t> eregs(8 * n);
            REQUIRE(plume_registers_from_be(2 * n, eh.data(), eregs.data()) == 0);
            REQUIRE(std::memcmp(dl(regs).data(), eregs.data(), 64 * n) == 0);
        }
        {   // the aggregate check: one call, and the same batch as two pieces with a carried record
            uint8_t seed32[32], eres[72];
            for (auto& x : seed32) x = (uint8_t)rng();
            const int mode = v1 ? 0 : 1;
            std::vector<uint8_t> ehok(n);
            REQUIRE(oracle_aggregate_check(b.version, mode, n, b.msgs.data(), b.off.data(), b.pk.data(), b.nul.data(), b.c.data(), b.s.data(), b.rpt.data(), b.hr.data(), seed32, 0, ehok.data(), eres, 4) == 0);
            DevArr hok(nullptr, n), res(nullptr, 72);
            REQUIRE(plume_aggregate_check_device(ctx, b.version, mode, n, m.p, off, b.msgs.size(), pk.p, nul.p, c.p, s.p, rp.p, hr.p, seed32, 0, hok.p, res.p, st) == 0);
            REQUIRE(dl(hok) == ehok);
            REQUIRE(std::memcmp(dl(res).data(), eres, 72) == 0);
        }
    }
    // (The OVERLAPPED order itself needs slices of at least 8192 items -- csrc/plume_host_logic.h sub_batch_bounds -- which is minutes of CPU under a sanitizer: the GPU suite runs
    // it, tests/test_gpu_round3.py::test_stage_times_serial_and_overlapped; here sub_batches = 3 exercises the knob's bookkeeping with one slice.)
    g_what = "argument errors of the device forms";
    REQUIRE(plume_hash_to_curve_batch_device(ctx, 3, nullptr, nullptr, 0, nullptr, nullptr, st) != 0);
    REQUIRE(plume_registers_from_be_device(ctx, 3, nullptr, nullptr, st) != 0);
    REQUIRE(std::string(plume_last_error()).size() > 0);
    plume_destroy(ctx);
    REQUIRE(hipStreamDestroy(st) == hipSuccess);
}

// allocation failures: the k-th hipMalloc / hipHostMalloc of a call fails, for every k the call makes.  The call must fail cleanly (an error code, no crash, nothing leaked or
// double-freed: ASan and the leak accounting at exit watch) and the context must work afterwards
static void group_faults(uint64_t seed) {
    const Batch b = make_batch(1, 40 + seed % 3, true), sg = make_batch(2, 24, false);
    struct Call { const char* name; std::function<int(plume_ctx*)> run; };
    std::vector<uint8_t> ok(b.n), opk(64 * sg.n), onul(64 * sg.n), oc(32 * sg.n), os(32 * sg.n), orp(64 * sg.n), ohr(64 * sg.n), ost(sg.n), res(72), hok(b.n);
    uint8_t seed32[32] = {1, 2, 3};
    const Call calls[] = {
        {"verify", [&](plume_ctx* c) { return plume_verify_batch(c, 1, b.n, b.msgs.data(), b.off.data(), b.pk.data(), b.nul.data(), b.c.data(), b.s.data(), b.rpt.data(), b.hr.data(), ok.data()); }},
        {"sign", [&](plume_ctx* c) { return plume_sign_batch(c, 2, sg.n, sg.msgs.data(), sg.off.data(), sg.sk.data(), sg.r.data(), nullptr, opk.data(), onul.data(), oc.data(), os.data(), orp.data(), ohr.data(), ost.data()); }},
        {"aggregate", [&](plume_ctx* c) { return plume_aggregate_check(c, 1, 0, b.n, b.msgs.data(), b.off.data(), b.pk.data(), b.nul.data(), b.c.data(), b.s.data(), b.rpt.data(), b.hr.data(), seed32, hok.data(), res.data()); }},
    };
    // pass 0: nothing else alive -- the generator's tables are allocated (and, on failure, given back) by the call under test; the first allocations of a verify call only,
    // a table build costs a second here.  Passes 1, 2: a keeper context per device holds the tables, every allocation of every call fails in turn (one device, three shards).
    plume_ctx* keeper[2] = {nullptr, nullptr};
    for (int pass = 0; pass < 3; pass++) {
        const int multi = pass == 2;
        if (pass == 1) {
            REQUIRE(plume_init(&keeper[0], 5) == 0 && plume_init(&keeper[1], 6) == 0);
            for (plume_ctx* kc : keeper) { REQUIRE(plume_set_eq1_short(kc, 3) == 0); for (const Call& cl : calls) REQUIRE(cl.run(kc) == 0); }
        }
        for (const Call& cl : calls) {
            if (pass == 0 && std::string(cl.name) != "verify") continue;
            if (pass == 2 && std::string(cl.name) == "sign") continue;      // (the shards' fan-out and join are the same code for every call: verify and the aggregate's combine step cover them)
            g_what = std::string("allocation failures in ") + cl.name + (multi ? " (three shards)" : pass == 0 ? " (tables not built yet)" : "");
            long total = -1, runs = 0;
            const auto t_begin = std::chrono::steady_clock::now();
            for (long k = 0; total < 0 || k < (pass == 0 ? std::min<long>(total, 10) : total); k++, runs++) {
                // a fresh context every time: the allocations of a first call (workspace, slots, tables) are the interesting ones
                plume_ctx* ctx = nullptr;
                int ids[3] = {5, 6, 5};
                REQUIRE((multi ? plume_init_multi(&ctx, ids, 3) : plume_init(&ctx, 6)) == 0);
                REQUIRE(plume_set_host_piece(ctx, 32) == 0 && plume_set_host_first_piece(ctx, 8) == 0 && plume_set_host_tail_piece(ctx, 8) == 0 && plume_set_eq1_short(ctx, 3) == 0);
                if (total < 0) {                               // how many allocations does the call make?
                    mockhip::fail_allocation(-1);
                    REQUIRE(cl.run(ctx) == 0);
                    total = mockhip::fail_allocation(-1);
                    REQUIRE(total > 3);
                    plume_destroy(ctx);
                    k = -1;
                    continue;
                }
                mockhip::fail_allocation(k);
                const int rc = cl.run(ctx);
                mockhip::fail_allocation(-1);
                // (a failed allocation may be survivable: the second host lane is optional, a registration is optional)
                if (rc == 0) { if (std::string(cl.name) == "verify") REQUIRE(ok == b.ok_expect); }
                else REQUIRE(rc == PLUME_ERR_HIP);
                REQUIRE(cl.run(ctx) == 0);                     // and the context is usable afterwards
                if (std::string(cl.name) == "verify") REQUIRE(ok == b.ok_expect);
                plume_destroy(ctx);
            }
            std::printf("  %s: %ld allocations per call, %ld failure points tried, %.1f s\n", g_what.c_str(), total, runs, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
        }
    }
    for (plume_ctx* kc : keeper) plume_destroy(kc);
    {   // the generator's tables themselves (the largest allocations by far): the window table of the verifier, then the comb of the short first equation
        g_what = "the generator's tables cannot be allocated";
        for (long skip = 0; skip < 2; skip++) {
            plume_ctx* ctx = nullptr;
            REQUIRE(plume_init(&ctx, 6) == 0 && plume_set_eq1_short(ctx, 3) == 0);
            mockhip::fail_allocation_of_at_least((size_t)1 << 20, skip);
            REQUIRE(calls[0].run(ctx) == PLUME_ERR_HIP);
            REQUIRE(calls[0].run(ctx) == 0 && ok == b.ok_expect);
            plume_destroy(ctx);
        }
    }
    {   // page-locked arrays: the call asks for a second lane of the context; when that lane cannot be created the call goes on with one
        g_what = "the second host lane cannot be created";
        keeper[0] = nullptr;
        REQUIRE(plume_init(&keeper[0], 6) == 0 && plume_set_eq1_short(keeper[0], 3) == 0 && calls[0].run(keeper[0]) == 0);
        Arr msgs(b.msgs.size(), 1), off(8 * (b.n + 1), 1), pk(64 * b.n, 1), nul(64 * b.n, 1), c(32 * b.n, 1), s(32 * b.n, 1), rpt(64 * b.n, 1), hr(64 * b.n, 1), okp(b.n, 1);
        msgs.set(b.msgs); std::memcpy(off.p, b.off.data(), 8 * (b.n + 1)); pk.set(b.pk); nul.set(b.nul); c.set(b.c); s.set(b.s); rpt.set(b.rpt); hr.set(b.hr);
        long total = -1;
        for (long k = 0; total < 0 || k < total; k++) {
            plume_ctx* ctx = nullptr;
            REQUIRE(plume_init(&ctx, 6) == 0);
            REQUIRE(plume_set_host_piece(ctx, 16) == 0 && plume_set_host_first_piece(ctx, 8) == 0 && plume_set_eq1_short(ctx, 3) == 0 && plume_set_host_lanes(ctx, 2) == 0);
            mockhip::fail_allocation(total < 0 ? -1 : k);
            const int rc = plume_verify_batch(ctx, 1, b.n, msgs.p, (const uint64_t*)off.p, pk.p, nul.p, c.p, s.p, rpt.p, hr.p, okp.p);
            const long made = mockhip::fail_allocation(-1);
            if (total < 0) { REQUIRE(rc == 0); total = made; k = -1; plume_destroy(ctx); continue; }
            REQUIRE(rc == 0 || rc == PLUME_ERR_HIP);
            if (rc == 0) REQUIRE(std::memcmp(okp.p, b.ok_expect.data(), b.n) == 0);
            REQUIRE(plume_verify_batch(ctx, 1, b.n, msgs.p, (const uint64_t*)off.p, pk.p, nul.p, c.p, s.p, rpt.p, hr.p, okp.p) == 0 && std::memcmp(okp.p, b.ok_expect.data(), b.n) == 0);
            plume_destroy(ctx);
        }
        plume_destroy(keeper[0]);
    }
}

// a machine whose GPU is not a gfx950 (PLUME_MOCK_ARCH): the library refuses it by name; and a machine without any device
static void group_noarch() {
    plume_ctx* ctx = nullptr;
    g_what = "another architecture";
    const int rc = plume_init(&ctx, 0);
    REQUIRE(rc == PLUME_ERR_NODEV && ctx == nullptr);
    REQUIRE(std::string(plume_last_error()).find(std::getenv("PLUME_MOCK_ARCH") ? "gfx950" : "no HIP device") != std::string::npos);
    int ids[2] = {0, 0};
    REQUIRE(plume_init_multi(&ctx, ids, 2) == PLUME_ERR_NODEV && ctx == nullptr);
}

static void group_misc(uint64_t seed) {
    plume_ctx* ctx = nullptr;
    REQUIRE(plume_init(&ctx, 7) == 0);
    Knobs k; k.piece = 32; k.first = 8; k.tail = 8;
    apply(ctx, k);
    const Batch clean = make_batch(1, 75 + seed % 4, false), spoiled = make_batch(1, 50, true), v2 = make_batch(2, 33, false), empty = make_batch(1, 0, false);
    g_what = "aggregate check, one device";
    for (int mk : {0, 1}) { check_aggregate(ctx, clean, mk, 0); check_aggregate(ctx, clean, mk, 1); check_aggregate(ctx, spoiled, mk, 0); check_aggregate(ctx, v2, mk, 1); check_aggregate(ctx, empty, mk, 0); }
    g_what = "hash_to_curve, DER, first occurrences, one device";
    check_h2c_and_friends(ctx, clean, 0);
    check_h2c_and_friends(ctx, v2, 1);
    check_h2c_and_friends(ctx, empty, 0);
    g_what = "two contexts on one device share the generator's tables; the first to go leaves them to the other";
    plume_ctx* other = nullptr;
    setenv("PLUME_INGEST_SPLIT_MAX", "20", 1);               // this context: calls of more than 20 items take the one-role ingest kernel and a scalar launch of its own,
    setenv("PLUME_MSM_PAIR_MAX", "30", 1);                   // calls of more than 30 the multi-scalar kernel of large batches (one lane per chain; the default here: two half chains)
    REQUIRE(plume_init(&other, 7) == 0);
    unsetenv("PLUME_INGEST_SPLIT_MAX");
    unsetenv("PLUME_MSM_PAIR_MAX");
    check_verify(other, v2, 0);
    plume_destroy(ctx);
    check_verify(other, clean, 1);
    check_sign(other, v2, 0, false);
    plume_destroy(other);
    plume_destroy(nullptr);
}

int main(int argc, char** argv) {
    const std::string group = argc > 1 ? argv[1] : "all";
    const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1;
    rng.seed(seed * 0x9E3779B97F4A7C15ull + 12345);
    std::printf("%s\n", plume_version());
    if (group == "verify" || group == "all") group_verify(seed);
    if (group == "sign" || group == "all") group_sign(seed);
    if (group == "multi" || group == "all") group_multi(seed);
    if (group == "benchseq" || group == "all") group_benchseq(seed);
    if (group == "device" || group == "all") group_device(seed);
    if (group == "misc" || group == "all") group_misc(seed);
    if (group == "devapi" || group == "all") group_devapi(seed);
    if (group == "faults" || group == "all") group_faults(seed);
    if (group == "noarch") group_noarch();
    // everything the library took from the runtime has gone back
    g_what = "leak check";
    REQUIRE(mockhip::outstanding(0) == 0);
    REQUIRE(mockhip::outstanding(1) == 0);
    REQUIRE(mockhip::outstanding(2) == 0);
    REQUIRE(mockhip::outstanding(3) == 0);
    long ops = 0, other = 0;
    for (int d = 0; d < 8; d++) { ops += mockhip::ops_run(d, false); other += mockhip::ops_run(d, true); }
    std::printf("pipeline_driver %s seed %llu: ok (%ld queued operations run, %ld of them pulled in through an event dependency)\n", group.c_str(), (unsigned long long)seed, ops, other);
    return 0;
}
