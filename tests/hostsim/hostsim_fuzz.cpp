// CPU-only fuzz harness for the host logic of libplume_hip.so (zk-nullifier-sig_amd/csrc/plume_host_logic.h: the same bodies plume_capi.hip calls).  Built by
// tests/test_sanitizers.py with -fsanitize=address,undefined and run as an executable: every buffer is sized EXACTLY (heap allocations, so an out-of-bounds access of one
// byte trips AddressSanitizer), every invariant the pipeline relies on is asserted.  Test infrastructure only.
//   hostsim_fuzz <seed> <cases>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <string>

#include "plume_host_logic.h"

using namespace plume_host;

#define REQUIRE(c)                                                                          \
    do {                                                                                    \
        if (!(c)) { std::fprintf(stderr, "hostsim_fuzz: %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(2); } \
    } while (0)

static std::mt19937_64 rng;
static uint64_t rnd(uint64_t lo, uint64_t hi) { return lo + rng() % (hi - lo + 1); }
// sizes that matter: powers of two, their neighbours, small, and anything
static size_t any_size(size_t cap) {
    switch (rnd(0, 4)) {
        case 0: return (size_t)rnd(0, 70);
        case 1: { size_t p = (size_t)1 << rnd(0, 26); size_t v = p + rnd(0, 2) - 1; return v > cap ? cap : v; }
        case 2: return (size_t)rnd(0, 4) * 65536 + rnd(0, 3);
        default: return (size_t)rnd(0, cap);
    }
}

static void fuzz_shards() {
    const size_t n = any_size(0xFFFFFFF0u), g = rnd(1, 64);
    size_t prev = 0;
    for (size_t d = 0; d < g; d++) {
        size_t lo, hi;
        shard_bounds(n, d, g, lo, hi);
        REQUIRE(lo == prev && lo <= hi && hi <= n);
        REQUIRE(hi - lo <= n / g + 1);                 // even
        prev = hi;
    }
    REQUIRE(prev == n);
}

static void fuzz_sub_batches() {
    const size_t n = any_size((size_t)1 << 26);
    const int k = (int)rnd(1, 64);
    const size_t omin = (size_t)1 << rnd(0, 20);
    const std::vector<size_t> b = sub_batch_bounds(n, k, omin);
    REQUIRE(!b.empty() && b.front() == 0 && b.back() == n);
    REQUIRE(b.size() - 1 <= (size_t)k || n == 0);
    for (size_t i = 1; i < b.size(); i++) {
        REQUIRE(b[i] > b[i - 1]);
        if (i + 1 < b.size()) REQUIRE((b[i] - b[i - 1]) % 1024 == 0);
    }
    if (n == 0) REQUIRE(b.size() == 1);
}

static void fuzz_pieces() {
    PieceKnobs kn;
    kn.chunk = rnd(0, 3) ? ((size_t)1 << rnd(0, 26)) : any_size((size_t)1 << 26) + 1;
    kn.host_piece = rnd(0, 3) ? ((size_t)1 << rnd(0, 26)) : any_size((size_t)1 << 26) + 1;
    kn.host_first_piece = rnd(0, 3) ? ((size_t)1 << rnd(0, 20)) : any_size((size_t)1 << 26) + 1;
    kn.host_tail_piece = rnd(0, 3) ? ((size_t)1 << rnd(0, 20)) : any_size((size_t)1 << 26) + 1;
    if (rnd(0, 3) == 0) kn = PieceKnobs{(size_t)1 << 20, (size_t)1 << 19, (size_t)1 << 16, (size_t)1 << 16};     // the defaults
    kn.out_lanes = (int)rnd(1, 2);
    const size_t cap0 = kn.host_piece < kn.chunk ? kn.host_piece : kn.chunk;
    size_t n = any_size(0x10000000u);
    if (n / cap0 > 3000) n = cap0 * rnd(1, 3000) + rnd(0, cap0 - 1);   // keep a case's piece count (and this harness's run time) bounded
    const bool heavy = rnd(0, 1);
    std::string list;
    const char* explicit_list = nullptr;
    if (rnd(0, 3) == 0) {                                          // an explicit list: valid, or mangled
        size_t rem = n;
        while (rem) { size_t c = rnd(1, rem < kn.chunk ? rem : kn.chunk); list += std::to_string(c); rem -= c; if (rem) list += ","; }
        switch (rnd(0, 5)) {
            case 0: list += ",7"; break;
            case 1: list += "x"; break;
            case 2: list = "," + list; break;
            case 3: list += ",99999999999999999999999999"; break;
            case 4: list = ""; break;
            default: break;
        }
        explicit_list = list.c_str();
    }
    const std::vector<size_t> s = piece_schedule(kn, n, heavy, explicit_list);
    const size_t cap = kn.host_piece < kn.chunk ? kn.host_piece : kn.chunk;
    size_t sum = 0;
    for (size_t c : s) { REQUIRE(c >= 1 && c <= kn.chunk); sum += c; }
    REQUIRE(sum == n);
    bool listed = false;                                           // pieces from the rules (not from a valid explicit list) also respect the largest piece
    if (explicit_list && !s.empty()) {
        std::string again;
        for (size_t i = 0; i < s.size(); i++) again += (i ? "," : "") + std::to_string(s[i]);
        listed = again == list;
    }
    if (!listed) for (size_t c : s) REQUIRE(c <= cap);
    REQUIRE(s.size() <= n / 1 + 1 && (n == 0) == s.empty());
    if (!listed && n) REQUIRE(s.size() <= 8 * (n / cap + 1) + 24);   // (3 (n / cap + 1) + 24 for the growing and tapered rules; the signer's uniform two-lane pieces are never below cap / 8)   // the number of pieces stays proportional to n / largest piece
}

static void fuzz_offsets() {
    const size_t total = rnd(1, 300), i0 = rnd(0, total - 1), cnt = rnd(0, total - 1 - i0);
    std::unique_ptr<uint64_t[]> off(new uint64_t[total + 1]);
    uint64_t at = rnd(0, 1) ? rnd(0, 1000) : rnd(0, 0xFFFFFFFFFFull);
    const int mode = (int)rnd(0, 5);
    for (size_t k = 0; k <= total; k++) { off[k] = at; at += mode == 1 ? rnd(0, 0x40000000u) : rnd(0, 100); }
    if (mode == 2 && total > 1) off[rnd(1, total)] = rnd(0, 50);              // a decreasing offset somewhere
    if (mode == 3) off[rnd(0, total)] = ~0ull - rnd(0, 5);
    std::unique_ptr<uint64_t[]> rel(new uint64_t[cnt + 1]);
    const int rc = rebase_offsets(off.get(), i0, cnt, rel.get());
    bool dec = false;
    for (size_t k = 1; k <= cnt; k++) dec = dec || off[i0 + k] < off[i0 + k - 1];
    if (dec) REQUIRE(rc == 1);
    if (rc == 0) {
        REQUIRE(!dec && rel[0] == 0 && rel[cnt] <= 0xFFFFFF00ull);
        for (size_t k = 0; k <= cnt; k++) REQUIRE(rel[k] == off[i0 + k] - off[i0]);
    }
    if (!dec && off[i0 + cnt] - off[i0] > 0xFFFFFF00ull) REQUIRE(rc == 2);
}

static void fuzz_der() {
    static const uint8_t head[7] = {0x30, 0x6b, 0x02, 0x01, 0x01, 0x04, 0x20}, mid[6] = {0xa1, 0x44, 0x03, 0x42, 0x00, 0x04};
    static const uint8_t order[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
                                      0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
    const size_t n = rnd(0, 40);
    std::unique_ptr<uint8_t[]> der(new uint8_t[kDerLen * n + (n ? 0 : 1)]), sc(new uint8_t[32 * n + (n ? 0 : 1)]), ok(new uint8_t[n + (n ? 0 : 1)]);
    std::vector<int> want(n);
    for (size_t i = 0; i < n; i++) {
        uint8_t* d = der.get() + kDerLen * i;
        for (size_t j = 0; j < kDerLen; j++) d[j] = (uint8_t)rng();
        const int kind = (int)rnd(0, 7);
        if (kind != 0) { std::memcpy(d, head, 7); std::memcpy(d + 39, mid, 6); }
        want[i] = kind != 0;
        if (kind == 2) { std::memset(d + 7, 0, 32); want[i] = 0; }                       // zero scalar
        if (kind == 3) { std::memcpy(d + 7, order, 32); want[i] = 0; }                   // = n
        if (kind == 4) { std::memcpy(d + 7, order, 32); d[38] -= 1; want[i] = 1; }       // n - 1
        if (kind == 5) { d[rnd(0, 6)] ^= (uint8_t)(1u << rnd(0, 7)); want[i] = 0; }      // a flipped header bit
        if (kind == 6) { d[39 + rnd(0, 5)] ^= (uint8_t)(1u << rnd(0, 7)); want[i] = 0; }
        if (want[i] && kind != 4 && std::memcmp(d + 7, order, 32) >= 0) want[i] = 0;     // a random scalar >= n (2^-128)
    }
    sec1_der_to_scalars(n, der.get(), sc.get(), ok.get());
    for (size_t i = 0; i < n; i++) {
        REQUIRE(ok[i] == want[i]);
        if (ok[i]) REQUIRE(std::memcmp(sc.get() + 32 * i, der.get() + kDerLen * i + 7, 32) == 0);
        else for (int j = 0; j < 32; j++) REQUIRE(sc[32 * i + j] == 0);
    }
}

static void fuzz_registers() {
    const size_t n = rnd(0, 50);
    std::unique_ptr<uint8_t[]> be(new uint8_t[32 * n + (n ? 0 : 1)]);
    std::unique_ptr<uint64_t[]> reg(new uint64_t[4 * n + (n ? 0 : 1)]);
    for (size_t i = 0; i < 32 * n; i++) be[i] = (uint8_t)rng();
    registers_from_be(n, be.get(), reg.get());
    for (size_t k = 0; k < n; k++)
        for (int j = 0; j < 4; j++)
            for (int b = 0; b < 8; b++) REQUIRE(((reg[4 * k + j] >> (8 * b)) & 0xFF) == be[32 * k + 31 - 8 * j - b]);
}

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1;
    const long cases = argc > 2 ? std::atol(argv[2]) : 20000;
    rng.seed(seed);
    for (long i = 0; i < cases; i++) {
        fuzz_shards(); fuzz_sub_batches(); fuzz_pieces(); fuzz_offsets(); fuzz_der(); fuzz_registers();
    }
    // the documented default schedules
    const PieceKnobs def{(size_t)1 << 20, (size_t)1 << 19, (size_t)1 << 16, (size_t)1 << 16};
    const size_t K = 1024;
    REQUIRE((piece_schedule(def, (size_t)1 << 20, false, nullptr) == std::vector<size_t>{64 * K, 192 * K, 512 * K, 256 * K}));
    REQUIRE((piece_schedule(def, (size_t)1 << 20, true, nullptr) == std::vector<size_t>{128 * K, 256 * K, 256 * K, 192 * K, 128 * K, 64 * K}));
    PieceKnobs def2 = def; def2.out_lanes = 2;                     // the signer on two lanes: uniform 2^16-item pieces from 2^18 items up; verify calls do not change
    REQUIRE((piece_schedule(def2, (size_t)1 << 20, true, nullptr) == std::vector<size_t>(16, 64 * K)));
    REQUIRE((piece_schedule(def2, (size_t)1 << 21, true, nullptr) == std::vector<size_t>(32, 64 * K)));
    REQUIRE((piece_schedule(def2, (size_t)1 << 22, true, nullptr) == std::vector<size_t>(32, 128 * K)));
    REQUIRE((piece_schedule(def2, ((size_t)1 << 18) + 5, true, nullptr) == std::vector<size_t>{64 * K, 64 * K, 64 * K, 64 * K, 5}));
    REQUIRE((piece_schedule(def2, ((size_t)1 << 18) - 1, true, nullptr) == piece_schedule(def, ((size_t)1 << 18) - 1, true, nullptr)));
    REQUIRE((piece_schedule(def2, (size_t)1 << 20, false, nullptr) == std::vector<size_t>{64 * K, 192 * K, 512 * K, 256 * K}));
    REQUIRE((piece_schedule(def, (size_t)1 << 17, false, nullptr) == std::vector<size_t>{128 * K}));
    REQUIRE((piece_schedule(def, (size_t)1 << 19, true, nullptr) == std::vector<size_t>{64 * K, 192 * K, 192 * K, 64 * K}));
    std::printf("hostsim_fuzz: seed %llu, %ld cases of 6 fuzzers: ok\n", (unsigned long long)seed, cases);
    return 0;
}
