// Nullifier-set post-processing: the application step after verification.  PLUME's purpose is ONE
// nullifier per (pk, message) (reference README.md:5; the nullifier field rust-k256/src/lib.rs:72-73), so the consumer of a
// verified batch has to find repeated nullifiers.  This stage marks, for every live item, whether it is the FIRST occurrence of
// its nullifier in the batch: first[i] = live[i] && no live j with nullifier[j] == nullifier[i] and id[j] < id[i]
// (id = position, or a caller-supplied global 64-bit id when the batch is one shard of a larger set).  The result is
// deterministic (smallest id wins), whatever order the lanes run in.
//
// Method: open-addressing hash table in HBM (2 slots per item, linear probing).  A slot is claimed with atomicCAS by the
// first lane that gets there; later lanes with the same 64-byte record (full compare against the claimant's record) share the
// slot, others probe on; atomicMin on the slot's id picks the winner; a second pass compares.  Per item: one 64-byte read of its
// own record, ~1.3 slot probes with one 64-byte gather each, two atomics: HBM-bound random access, microseconds per 2^20 items
// next to the 25 ms the verification takes.  Like every per-lane body in this directory it also compiles for the host (tests/devsim).
#pragma once
#include <stdint.h>

#include "plume_field.h"

namespace plume {

#define PLUME_DEDUP_EMPTY 0xFFFFFFFFu

struct DedupArgs {
    uint32_t n;
    const uint8_t* nul;              // 64 B / item (x || y big-endian; all-zero = identity), 16-byte aligned
    const uint8_t* live;             // optional, n bytes: 0 = the item does not take part (e.g. it failed verification)
    const uint64_t* ids;             // optional, n global ids (distinct); NULL: id = position
    uint8_t* first;                  // out, n bytes
    unsigned long long* n_unique;    // out (device word): number of first occurrences
    uint32_t* blockcnt;              // scratch: first occurrences per workgroup of the mark kernel (summed by one small kernel:
                                     // 16384 wavefronts adding to ONE counter serialise in L2 -- measured 0.2 ms of a 0.33 ms stage)
    // scratch
    uint32_t* slots;                 // mask+1 entries: index of the item that claimed the slot, or PLUME_DEDUP_EMPTY
    unsigned long long* minid;       // mask+1 entries: smallest id among the items sharing the slot
    uint32_t* myslot;                // n entries
    uint32_t mask;                   // table size - 1 (power of two >= 2n)
    uint32_t key[2];                 // per-call hash key (plume_capi.hip draws it with getrandom): records ground to collide under one key
                                     // do not collide under the next, so a crafted batch cannot pin the probe length at O(n).  Results do not depend on it.
};

PLUME_HD uint32_t dedup_table_size(uint32_t n) {
    uint32_t m = 64;
    while (m < 2u * n && m < 0x80000000u) m <<= 1;
    return m;
}
// 32-bit keyed mix of the 64-byte record (murmur3-style rounds; the key enters before each word's multiplication, so the word
// differences that cancel in the unkeyed function cannot be chosen without it).  Honest nullifiers are uniformly distributed group
// elements; the table stays correct -- only slower -- whatever the records are.
PLUME_HD uint32_t dedup_hash(const uint32_t* rec /* 16 words */, const uint32_t key[2]) {
    uint32_t h = 0x9747B28Cu ^ key[0];
    PLUME_UNROLL for (int i = 0; i < 16; i++) {
        const uint32_t kr = (key[1] << (i & 15)) | (key[1] >> ((32 - (i & 15)) & 31));
        uint32_t k = (rec[i] ^ kr) * 0xCC9E2D51u;
        k = (k << 15) | (k >> 17);
        k *= 0x1B873593u;
        h ^= k;
        h = (h << 13) | (h >> 19);
        h = h * 5u + 0xE6546B64u;
    }
    h ^= 64u; h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define PLUME_ATOMIC_CAS_U32(p, expect, desired) atomicCAS((p), (expect), (desired))
#define PLUME_ATOMIC_MIN_U64(p, v) atomicMin((p), (v))
#else
// host builds run the lanes one after another
PLUME_HD uint32_t plume_host_cas_u32(uint32_t* p, uint32_t expect, uint32_t desired) { uint32_t old = *p; if (old == expect) *p = desired; return old; }
PLUME_HD void plume_host_min_u64(unsigned long long* p, unsigned long long v) { if (v < *p) *p = v; }
#define PLUME_ATOMIC_CAS_U32(p, expect, desired) plume_host_cas_u32((p), (expect), (desired))
#define PLUME_ATOMIC_MIN_U64(p, v) plume_host_min_u64((p), (v))
#endif

PLUME_HD void dedup_clear(const DedupArgs& a, uint32_t slot) {
    a.slots[slot] = PLUME_DEDUP_EMPTY;
    a.minid[slot] = ~0ull;
}
PLUME_HD void dedup_insert(const DedupArgs& a, uint32_t i) {
    if (a.live && !a.live[i]) return;
    uint32_t rec[16];
    const uint32_t* mine = (const uint32_t*)(a.nul + 64 * (size_t)i);
    PLUME_UNROLL for (int k = 0; k < 16; k++) rec[k] = mine[k];
    uint32_t h = dedup_hash(rec, a.key) & a.mask;
    for (;;) {
        uint32_t owner = a.slots[h];
        if (owner == PLUME_DEDUP_EMPTY) {
            owner = PLUME_ATOMIC_CAS_U32(&a.slots[h], PLUME_DEDUP_EMPTY, i);
            if (owner == PLUME_DEDUP_EMPTY) break;                       // claimed
        }
        if (owner == i) break;
        const uint32_t* other = (const uint32_t*)(a.nul + 64 * (size_t)owner);
        uint32_t diff = 0;
        PLUME_UNROLL for (int k = 0; k < 16; k++) diff |= other[k] ^ rec[k];
        if (diff == 0) break;                                            // same nullifier: share the slot
        h = (h + 1) & a.mask;                                            // someone else's: probe on (the table is at most half full)
    }
    PLUME_ATOMIC_MIN_U64(&a.minid[h], a.ids ? (unsigned long long)a.ids[i] : (unsigned long long)i);
    a.myslot[i] = h;
}
// returns the flag; the caller (kernel: workgroup reduction into blockcnt, host: plain sum) accumulates n_unique
PLUME_HD bool dedup_mark(const DedupArgs& a, uint32_t i) {
    bool f = false;
    if (!a.live || a.live[i]) f = a.minid[a.myslot[i]] == (a.ids ? (unsigned long long)a.ids[i] : (unsigned long long)i);
    a.first[i] = f ? 1 : 0;
    return f;
}

}  // namespace plume
