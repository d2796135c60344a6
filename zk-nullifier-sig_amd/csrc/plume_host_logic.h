// Host-side logic of libplume_hip.so that touches no GPU: how a batch is cut (shards, sub-batches, pipeline pieces), how message offsets are rebased for a piece, and the
// byte-level parsers of the SEC1-DER and register forms.  No HIP dependency, so that a CPU-only harness (tests/hostsim) can compile exactly these bodies under
// AddressSanitizer + UBSan and fuzz them; plume_capi.hip includes this header and calls the same functions.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace plume_host {

// shard d of g owns [floor(d n / g), floor((d + 1) n / g)) of every array; n < 2^32 and g <= 64, so the product cannot overflow 64 bits
inline void shard_bounds(size_t n, size_t d, size_t g, size_t& lo, size_t& hi) { lo = n * d / g; hi = n * (d + 1) / g; }

// Device-resident calls: slice bounds for `sub_batches` slices of an n-item call (one slice below overlap_min items or when a slice would hold fewer than 8192 items);
// every slice but the last is a multiple of 1024 items (whole workgroups, aligned records).  Returns {0, ..., n}.
inline std::vector<size_t> sub_batch_bounds(size_t n, int sub_batches, size_t overlap_min) {
    size_t k = (sub_batches > 1 && n >= overlap_min) ? (size_t)sub_batches : 1;
    while (k > 1 && n / k < 8192) k--;
    size_t per = (n + k - 1) / k;
    per = (per + 1023) & ~(size_t)1023;
    std::vector<size_t> b{0};
    while (b.back() < n) b.push_back(b.back() + per < n ? b.back() + per : n);
    return b;
}

struct PieceKnobs {
    size_t chunk, host_piece, host_first_piece, host_tail_piece;
    int out_lanes = 1;        // lanes a call with large outputs (the signer) may spread its pieces over
};

// Host-pointer calls: the pieces of an n-item call, in order.  Every piece is in [1, min(host_piece, chunk)] and they add up to n.
// The first piece is small (nothing hides its upload), every following piece may be up to three times the one before it (an upload runs at ~7 ns per item, the kernels at
// ~20 ns per item, so piece k+1's upload still hides behind piece k's kernels) up to the largest piece.  `explicit_list`: PLUME_HOST_SCHEDULE, an experiment's
// comma-separated piece list, used only when it is well-formed, adds up to n and every entry fits the chunk.
inline std::vector<size_t> piece_schedule(const PieceKnobs& kn, size_t n, bool out_heavy, const char* explicit_list) {
    const size_t piece = kn.host_piece < kn.chunk ? kn.host_piece : kn.chunk;
    std::vector<size_t> sched;
    if (n == 0 || piece == 0) return sched;
    if (explicit_list) {
        size_t sum = 0;
        bool ok = true;
        for (const char* q = explicit_list; *q;) {
            char* end;
            const unsigned long long v = std::strtoull(q, &end, 10);
            if (end == q || v == 0 || v > kn.chunk || (*end && *end != ',')) { ok = false; break; }
            sched.push_back((size_t)v);
            sum += (size_t)v;
            if (sum > n) { ok = false; break; }
            q = *end ? end + 1 : end;
        }
        if (ok && sum == n) return sched;
        sched.clear();
    }
    const size_t tail = kn.host_tail_piece ? kn.host_tail_piece : 1;
    if (out_heavy && kn.out_lanes > 1 && tail <= piece && piece <= 8 * tail && n >= 4 * tail) {          // (a tail knob far below the largest piece keeps the rules below: the piece count stays O(n / piece))
        // The signer on TWO lanes (round 6): uniform pieces of the tail size (2^16), dealt out to the lanes in turn.  With the lanes' kernels really side by side (their streams on
        // different hardware queues) a piece's fixed cost -- launches, three serial inversions, ramps and tails -- hides beside the other lane's kernels, every download but the
        // last hides behind a kernel, and the last one is short: 2^20 signs from page-locked arrays 18.0 ms against 18.6-19.3 for the tapered one-lane schedule below on the same
        // box (tests/gpu_debug/r06_host_sched.py; device-resident: 16.2).  Pieces of 2^15 or 3 * 2^14 items lose 25 %: 2^16 is where the small-call kernels still apply and the
        // chip is half full.  (Verify calls, whose 1-byte-per-item downloads cost nothing, keep the growing schedule: 64k / 192k / 512k / 256k is still their best.)
        // Beyond 32 such pieces (2^21 items) the pieces are twice as large: 2^22 signs 66.8 ms against 69.1 with 2^16-item pieces and 67.3 on one lane (r06_t.sh).
        const size_t u = (n > 32 * tail && 2 * tail <= piece) ? 2 * tail : tail;
        for (size_t rem = n; rem;) { const size_t c = rem < u ? rem : u; sched.push_back(c); rem -= c; }
        return sched;
    }
    if (out_heavy && tail <= piece / 8 && n / 12 >= tail && n >= 2 * piece) {
        // Calls with large outputs (the signer: 96 bytes up, 320 down per item): every piece's download hides behind the NEXT piece's kernels and the last one behind nothing, so
        // the pieces taper towards the end (3t, 2t, t with t = the tail piece, 2^16) after a body of pieces of at most half the largest piece (2^18) behind a first piece of 2t.
        // Round 5, PLUME_HOST_TRACE timelines: the copies are fully hidden either way and a piece costs a fixed ~0.4 ms (its fourteen launches, three serial inversions, ramps
        // and tails) on top of 16.8 ns per item, so fewer, larger pieces in the middle and a short tail: 2^20 signs 19.99 ms against 20.42 for 64k / 192k / 512k / 192k / 64k
        // (tests/gpu_debug/host_sched_r05.py).
        const size_t big = piece / 2, body = n - 8 * tail;
        sched.push_back(2 * tail);
        const size_t k = (body + big - 1) / big;                 // body pieces, as even as possible, none above half the largest piece
        for (size_t j = 0; j < k; j++) sched.push_back(body / k + (j < body % k ? 1 : 0));
        sched.push_back(3 * tail); sched.push_back(2 * tail); sched.push_back(tail);
        return sched;
    }
    size_t rem = n, cur = (kn.host_first_piece && kn.host_first_piece < piece) ? kn.host_first_piece : piece;
    if (n <= piece && n <= 2 * cur) { sched.push_back(n); return sched; }   // small calls: one piece
    while (rem) {
        const size_t c = cur < rem ? cur : rem;
        sched.push_back(c);
        rem -= c;
        cur = (cur <= piece / 3) ? cur * 3 : piece;
    }
    if (out_heavy && sched.size() > 1 && sched.back() > 2 * tail) { const size_t last = sched.back(); sched.back() = last - tail; sched.push_back(tail); }
    return sched;
}

// Message offsets of the piece [i0, i0 + cnt) rebased to the piece's first byte: rel[k] = off[i0 + k] - off[i0], k = 0..cnt.  Returns 0, or 1 when the offsets decrease,
// 2 when the piece's bytes exceed what one pass addresses (4 GiB less a margin).  `rel` holds cnt + 1 entries.
inline int rebase_offsets(const uint64_t* off, size_t i0, size_t cnt, uint64_t* rel) {
    const uint64_t base = off[i0];
    for (size_t k = 0; k <= cnt; k++) {
        if (off[i0 + k] < base || (k && off[i0 + k] < off[i0 + k - 1])) return 1;
        rel[k] = off[i0 + k] - base;
    }
    return rel[cnt] > 0xFFFFFF00ull ? 2 : 0;
}

constexpr size_t kDerLen = 109;
// The STRUCTURE half of SecretKey::from_sec1_der for the fixed 109-byte form the wasm layer emits (javascript/src/lib.rs:98-110): ok[i] = 1 iff record i has that exact
// shape and its scalar is in [1, n-1]; the scalar is copied out (zeroed when rejected).
inline void sec1_der_to_scalars(size_t n, const uint8_t* der109, uint8_t* scalars, uint8_t* ok) {
    static const uint8_t head[7] = {0x30, 0x6b, 0x02, 0x01, 0x01, 0x04, 0x20}, mid[6] = {0xa1, 0x44, 0x03, 0x42, 0x00, 0x04};
    static const uint8_t order[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
                                      0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
    for (size_t i = 0; i < n; i++) {
        const uint8_t* d = der109 + kDerLen * i;
        bool good = std::memcmp(d, head, 7) == 0 && std::memcmp(d + 39, mid, 6) == 0;
        bool nz = false;
        for (int j = 0; j < 32; j++) nz = nz || d[7 + j] != 0;
        good = good && nz && std::memcmp(d + 7, order, 32) < 0;
        std::memcpy(scalars + 32 * i, d + 7, 32);
        if (!good) std::memset(scalars + 32 * i, 0, 32);
        ok[i] = good ? 1 : 0;
    }
}

// 32-byte big-endian values -> the circuit's 4 x 64-bit little-endian registers (circuits/circom/utils.ts:11-17): registers[4k + j] = bits 64j .. 64j+63
inline void registers_from_be(size_t nvalues, const uint8_t* be32, uint64_t* registers) {
    for (size_t k = 0; k < nvalues; k++)
        for (int j = 0; j < 4; j++) {
            uint64_t v = 0;
            for (int b = 0; b < 8; b++) v = (v << 8) | be32[32 * k + 8 * (3 - j) + b];
            registers[4 * k + j] = v;
        }
}

}  // namespace plume_host
