// tsq_jointable.h — the join hash table as seen by device code (shared by tsq_join.hip, the
// radix-partitioned probe in tsq_radix.h and tools/radix_ubench.hip).
//
//   keys[nbuckets][8]  uint64  key words, one 64-byte line per bucket
//   vals[nbuckets][8]  uint32  build row ids (RowPtr analogue, util/chunk/list.go:28-31)
// bucket(kw) = mulhi64(mix64(kw), nbuckets) is MONOTONIC in h = mix64(kw): the table is range
// partitioned by the top bits of h for free, which is what the radix probe exploits — partition
// p = h >> (64 - bits) only ever touches the contiguous bucket range
// [mulhi64(p << (64-bits), nbuckets), mulhi64((p+1) << (64-bits), nbuckets)] (+ spill-over buckets).
#ifndef TSQ_JOINTABLE_H
#define TSQ_JOINTABLE_H

#include <hip/hip_runtime.h>

#include "tsq_device.h"

#define TSQ_EMPTY_KEY 0x8080808080808080ULL
#define TSQ_BUCKET 8

struct JoinTable {
    uint64_t* keys;
    uint32_t* vals;
    uint64_t nbuckets;
    const uint32_t* sent_rows;
    uint32_t sent_count;
};

// Visits every slot of the multimap whose key word equals kw: f(slot) for each.
// One 64-byte line (4 x dwordx4 loads, all issued before the first compare) per bucket; the walk
// ends at the first bucket that still has an EMPTY slot (nothing was ever pushed past it).
template <class F>
__device__ __forceinline__ void for_each_slot(const JoinTable& t, uint64_t kw, F&& f) {
    uint64_t bkt = tsq_mulhi64(tsq_mix64(kw), t.nbuckets);
    for (;;) {
        const ulonglong2* line = reinterpret_cast<const ulonglong2*>(t.keys + bkt * TSQ_BUCKET);
        const ulonglong2 a = line[0], b = line[1], c = line[2], d = line[3];
        const uint64_t k[8] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
        bool has_empty = false;
#pragma unroll
        for (int s = 0; s < TSQ_BUCKET; s++) {
            if (k[s] == kw) f(bkt * TSQ_BUCKET + s);
            has_empty |= (k[s] == TSQ_EMPTY_KEY);
        }
        if (has_empty) break;
        bkt = (bkt + 1 == t.nbuckets) ? 0 : bkt + 1;
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
