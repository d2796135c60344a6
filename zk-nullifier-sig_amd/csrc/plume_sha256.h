// SHA-256 (FIPS 180-4) for the PLUME hot path: one hash state per lane, fully unrolled compression with a
// rolling 16-word schedule (no 64-word array), byte streams assembled through a per-lane byte functor so that
// ragged messages and the 1-byte identity encoding need no scratch memory.
// Call sites in the reference: ExpandMsgXmd<Sha256> (rust-k256/src/utils.rs:15, semantics in
// rust-arkworks/src/fixed_hasher/expander.rs:89-134) and the c-hash (rust-k256/src/lib.rs:159-168).
#pragma once
#include "plume_field.h"

namespace plume {

PLUME_HD uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

PLUME_HD constexpr uint32_t sha256_k(int i) {
    constexpr uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    return K[i];
}

PLUME_HD void sha256_init(uint32_t st[8]) {
    st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
    st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
}
// state after absorbing one all-zero 64-byte block (the Z_pad of expand_message_xmd, expander.rs:108)
PLUME_HD void sha256_init_after_zero_block(uint32_t st[8]) {
    st[0] = 0xda5698be; st[1] = 0x17b9b469; st[2] = 0x62335799; st[3] = 0x779fbeca;
    st[4] = 0x8ce5d491; st[5] = 0xc0d26243; st[6] = 0xbafef9ea; st[7] = 0x1837a9d8;
}

// one compression; w[16] is consumed (used as the rolling schedule window)
PLUME_HD void sha256_compress(uint32_t st[8], uint32_t w[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    PLUME_UNROLL for (int i = 0; i < 64; i++) {
        uint32_t wi;
        if (i < 16) {
            wi = w[i];
        } else {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            w[i & 15] = wi;
        }
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + sha256_k(i) + wi;
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// Absorb `len` bytes given by byte(pos), 0 <= pos < len, then the FIPS padding for a total message length of
// prefix_len + len bytes (prefix_len must be a multiple of 64 and already absorbed into st).
template <class ByteFn>
PLUME_HD void sha256_absorb_pad(uint32_t st[8], uint32_t prefix_len, uint32_t len, ByteFn byte) {
    const uint32_t nblocks = (len + 9 + 63) >> 6;
    const uint32_t bits_lo = (prefix_len + len) << 3, bits_hi = (prefix_len + len) >> 29;
    PLUME_NOUNROLL for (uint32_t b = 0; b < nblocks; b++) {
        uint32_t w[16];
        PLUME_UNROLL for (int k = 0; k < 16; k++) {
            uint32_t word = 0;
            PLUME_UNROLL for (int q = 0; q < 4; q++) {
                uint32_t pos = b * 64 + 4 * k + q;
                uint32_t v = 0;
                if (pos < len) v = byte(pos);
                else if (pos == len) v = 0x80;
                word = (word << 8) | v;
            }
            w[k] = word;
        }
        if (b == nblocks - 1) { w[14] = bits_hi; w[15] = bits_lo; }
        sha256_compress(st, w);
    }
}

// byte k (0 = most significant) of a 256-bit big-endian integer held as little-endian limbs — select chain, no
// dynamic register indexing
PLUME_HD uint32_t be_byte_of_limbs(const uint32_t v[8], uint32_t k) {
    uint32_t limb = 7 - (k >> 2);
    uint32_t w = v[0];
    PLUME_UNROLL for (int i = 1; i < 8; i++) w = sel32(sel_mask(limb == (uint32_t)i), v[i], w);
    return (w >> (8 * (3 - (k & 3)))) & 0xFF;
}

}  // namespace plume
