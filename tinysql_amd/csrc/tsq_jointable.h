// tsq_jointable.h — the join hash table as seen by device code (shared by tsq_join.hip, the
// radix-partitioned probe in tsq_radix.h / tsq_ldsprobe.h and the partitioned build in tsq_buildpart.h).
//
//   keys[nbuckets][8]  uint64  TABLE WORDS w = mix64(key word), one 64-byte line per bucket
//   vals[nbuckets][8]  uint32  build row ids (RowPtr analogue, util/chunk/list.go:28-31)
// mix64 (the murmur3 finaliser) is a bijection on 64-bit words, so w1 == w2 <=> key word 1 == key word 2: storing
// the hashed word keeps equality exact (util/codec/codec.go:363-382) and every consumer that only routes a key —
// the radix partition, the LDS probe — works on w without hashing again.
//
// Geometry: the table is 2^tb SLICES of bs buckets.  slice(w) = top tb bits of w, local bucket =
// mulhi32(next 32 bits of w, bs); the global bucket slice*bs + local is MONOTONIC in w, so the table is range
// partitioned by the top bits of w for free.  A chain that runs off the end of its slice wraps to the slice's
// first bucket: every slice is a self-contained linear-probing table, which is what lets the partitioned build
// assemble a slice in LDS and the LDS probe (tsq_ldsprobe.h) answer a partition of probe keys from an LDS copy of
// a few slices.  tb = 0 is the plain table (one slice, wrap at nbuckets).
#ifndef TSQ_JOINTABLE_H
#define TSQ_JOINTABLE_H

#include <hip/hip_runtime.h>

#include "tsq_device.h"

#define TSQ_EMPTY_KEY 0x8080808080808080ULL
#define TSQ_BUCKET 8

struct JoinTable {
    uint64_t* keys;
    uint32_t* vals;
    uint64_t nbuckets;  // bs << tb
    uint32_t tb;        // slice bits (0: one slice)
    uint32_t bs;        // buckets per slice (< 2^32)
    const uint32_t* sent_rows;
    uint32_t sent_count;
    // CHAINED mode (a build key with tens of thousands of duplicates made the multimap's walks explode): ONE slot per distinct
    // table word, vals[slot] = the word's most recently inserted build row, next[row] = the next build row with the same word
    // (0xffffffff ends the chain) — rowHashMap's entry list (executor/hash_table.go:181-276): O(1) per inserted row whatever the
    // multiplicity.  nullptr: the multimap described above.
    const uint32_t* next;
};
#define TSQ_CHAIN_END 0xffffffffu



// the word a key word is stored and compared as
TSQ_HD uint64_t tsq_table_word(uint64_t kw) { return tsq_mix64(kw); }
// inverse of tsq_mix64 (murmur3 finaliser): the key word of a table word
TSQ_HD uint64_t tsq_unmix64(uint64_t w) {
    w ^= w >> 33;
    w *= 0x9CB4B2F8129337DBULL;  // inverse of 0xC4CEB9FE1A85EC53 mod 2^64
    w ^= w >> 33;
    w *= 0x4F74430C22A54005ULL;  // inverse of 0xFF51AFD7ED558CCD mod 2^64
    w ^= w >> 33;
    return w;
}
TSQ_HD uint32_t jt_slice(uint32_t tb, uint64_t w) { return tb ? (uint32_t)(w >> (64 - tb)) : 0u; }
TSQ_HD uint32_t jt_local(uint32_t tb, uint32_t bs, uint64_t w) {
    const uint32_t x = (uint32_t)((w << tb) >> 32);
    return (uint32_t)(((uint64_t)x * bs) >> 32);
}

// chained mode: the slot that holds table word w, or ~0 when no build row has it
__device__ __forceinline__ uint64_t jt_find_slot(const JoinTable& t, uint64_t w) {
    const uint64_t base = (uint64_t)jt_slice(t.tb, w) * t.bs;
    uint32_t lb = jt_local(t.tb, t.bs, w);
    for (uint32_t steps = 0; steps < t.bs; steps++) {
        const uint64_t bkt = base + lb;
        const ulonglong2* line = reinterpret_cast<const ulonglong2*>(t.keys + bkt * 8);
        const ulonglong2 a = line[0], b = line[1], c = line[2], d = line[3];
        const uint64_t k[8] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
        bool has_empty = false;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            if (k[s] == w) return bkt * 8 + (uint64_t)s;
            has_empty |= (k[s] == 0x8080808080808080ULL);
        }
        if (has_empty) break;
        lb = (lb + 1 == t.bs) ? 0 : lb + 1;
    }
    return ~0ull;
}

// Visits every slot of the multimap whose table word equals w: f(slot) for each.
// One 64-byte line (4 x dwordx4 loads, all issued before the first compare) per bucket; the walk
// ends at the first bucket that still has an EMPTY slot (nothing was ever pushed past it) and never
// leaves the slice of w.
template <class F>
__device__ __forceinline__ void for_each_slot_w(const JoinTable& t, uint64_t w, F&& f) {
    const uint64_t base = (uint64_t)jt_slice(t.tb, w) * t.bs;
    uint32_t lb = jt_local(t.tb, t.bs, w);
    for (uint32_t steps = 0; steps < t.bs; steps++) {
        const uint64_t bkt = base + lb;
        const ulonglong2* line = reinterpret_cast<const ulonglong2*>(t.keys + bkt * TSQ_BUCKET);
        const ulonglong2 a = line[0], b = line[1], c = line[2], d = line[3];
        const uint64_t k[8] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
        bool has_empty = false;
#pragma unroll
        for (int s = 0; s < TSQ_BUCKET; s++) {
            if (k[s] == w) f(bkt * TSQ_BUCKET + s);
            has_empty |= (k[s] == TSQ_EMPTY_KEY);
        }
        if (has_empty) break;
        lb = (lb + 1 == t.bs) ? 0 : lb + 1;
    }
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_xor_u64(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
    return v;
}
// exclusive prefix sum over the 64 lanes of a wave; *total = sum over all lanes
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t* total) {
    const int lane = threadIdx.x & 63;
    uint32_t x = v;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

#endif
