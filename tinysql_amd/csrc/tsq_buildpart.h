// tsq_buildpart.h — partitioned build of the join hash table (device code; included by tsq_join.hip).
//
// Why: k_build_insert claims slots with one random 64-bit CAS per build row into a table of GBs: 9 ms per 1e8 rows,
// 0.03 of the HBM roofline (random device atomics run at 20-27 G/s, and every row also scatters a 4-byte row id).
// The table is 2^tb self-contained slices selected by the top tb bits of the table word w = mix64(key word)
// (tsq_jointable.h), so a slice can be ASSEMBLED IN LDS from the rows whose word falls into it and then written to HBM
// as whole lines:
//   pass 1  k_radix_partition<.., WITH_IDX, HASHED>  table words + row ids by the top b1 bits of w (tsq_radix.h)
//   pass 2  k_radix_subpartition              one workgroup per pass-1 partition splits it by the next b2 bits; the
//                                             workgroup owns its 2^b2 outputs, so cursors live in LDS (no device atomics)
//   pass 3  k_build_images                    one workgroup per sub-partition: LDS image of its nbuckets/2^(b1+b2)
//                                             buckets (<= 768 buckets = 72 KB: two workgroups per CU), filled with LDS
//                                             compare-and-swap, stored with 16-byte coalesced writes
// b1 + b2 = tb: a sub-partition IS a table slice; a chain that reaches the end of the slice wraps to its first bucket,
// exactly as k_build_insert and every probe walk do.  Rows of a skewed partition that overflow their pass-1 / pass-2
// region are appended to a row list and inserted afterwards by k_build_insert; a slice that cannot take all of its rows
// (heavily duplicated keys) raises the fail flag and the host rebuilds the table unsliced.  The resulting table is
// equivalent to the one k_build_insert builds (same buckets, slots filled front to back, possibly another slot order
// inside a chain).
//
// Replaces (reference): hashRowContainer.PutChunk + rowHashMap.Put (executor/hash_table.go:146-169,247-256).
#ifndef TSQ_BUILDPART_H
#define TSQ_BUILDPART_H

#include "tsq_radix.h"

#define TSQ_BP_MAXP2 256
#define TSQ_BP_MAX_SLICE 768  // buckets per LDS image (96 B each)
#define TSQ_BP_MAX_SLICE_ROWS 3200  // mean rows of a slice: k_build_images<512, 8> holds <= 4096 rows of a slice in registers

struct SubStore {
    uint64_t* keys;       // [Q * cap2] key words, Q = 2^(b1+b2)
    uint32_t* idx;        // [Q * cap2] build row ids
    uint32_t* count;      // [Q] rows stored per sub-partition
    uint32_t* ovf_rows;   // build row ids that found no room (inserted later with k_build_insert)
    uint32_t* ovf_count;
    uint32_t ovf_cap;
    uint32_t b1, b2, cap2;
};

// pass 2: partition p of a RadixStore (8 regions) -> 2^b2 sub-partitions
template <int NT, int K>
__global__ void __launch_bounds__(NT) k_radix_subpartition(RadixStore st, SubStore out) {
    constexpr int T = NT * K;
    __shared__ uint64_t s_keys[T];
    __shared__ uint32_t s_idx[T];
    __shared__ uint32_t s_hist[TSQ_BP_MAXP2];   // count, then (overflow flag | exclusive offset inside the tile)
    __shared__ uint32_t s_delta[TSQ_BP_MAXP2];  // global slot of the run minus its LDS offset
    __shared__ uint32_t s_cur[TSQ_BP_MAXP2];    // rows stored so far per sub-partition (this workgroup owns them)
    __shared__ uint32_t s_wsum[NT / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t P1 = 1u << st.bits, P2 = 1u << out.b2;
    const uint32_t shift = 64 - st.bits - out.b2, mask = P2 - 1;
    for (uint32_t p = blockIdx.x; p < P1; p += gridDim.x) {
        if (tid < P2) s_cur[tid] = 0;
        __syncthreads();
        for (uint32_t r = 0; r < st.R; r++) {
            const uint32_t region = p * st.R + r;
            const uint32_t len = radix_region_len(st, P1, p, r);
            const size_t rbase = (size_t)region * st.cap;
            for (uint32_t t0 = 0; t0 < len; t0 += T) {
                const uint32_t n = len - t0 < (uint32_t)T ? len - t0 : (uint32_t)T;
                uint64_t k[K];
                uint32_t id[K], pr[K];
                if (tid < P2) s_hist[tid] = 0;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const uint32_t pos = (uint32_t)j * NT + tid;
                    pr[j] = 0xffffffffu;
                    if (pos < n) {
                        k[j] = st.keys[rbase + t0 + pos];
                        id[j] = st.idx[rbase + t0 + pos];
                    }
                }
#pragma unroll
                for (int j = 0; j < K; j++) {
                    const uint32_t pos = (uint32_t)j * NT + tid;
                    if (pos < n) {
                        const uint32_t s = (uint32_t)(k[j] >> shift) & mask;  // k[] are table words
                        pr[j] = (s << 16) | atomicAdd(&s_hist[s], 1u);
                    }
                }
                __syncthreads();
                const uint32_t cnt = tid < P2 ? s_hist[tid] : 0u;
                uint32_t total;
                const uint32_t offs = block_excl_scan<NT>(cnt, s_wsum, &total);
                if (tid < P2) {
                    uint32_t flag = 0;
                    const uint32_t g = s_cur[tid];
                    if (g + cnt > out.cap2) flag = 1;  // skew: the whole run goes to the row list
                    else s_cur[tid] = g + cnt;
                    s_delta[tid] = ((p << out.b2) + tid) * out.cap2 + g - offs;
                    s_hist[tid] = offs | (flag << 31);
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < K; j++) {
                    if (pr[j] != 0xffffffffu) {
                        const uint32_t d = (s_hist[pr[j] >> 16] & 0x7fffffffu) + (pr[j] & 0xffffu);
                        s_keys[d] = k[j];
                        s_idx[d] = id[j];
                    }
                }
                __syncthreads();
                for (uint32_t i = tid; i < n; i += NT) {
                    const uint64_t key = s_keys[i];
                    const uint32_t s = (uint32_t)(key >> shift) & mask;
                    if (!(s_hist[s] >> 31)) {
                        const uint32_t d = s_delta[s] + i;  // 32-bit wrap-around: s_delta may be "negative"
                        out.keys[d] = key;
                        out.idx[d] = s_idx[i];
                    } else {
                        const uint32_t o = __hip_atomic_fetch_add(out.ovf_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (o < out.ovf_cap) out.ovf_rows[o] = s_idx[i];
                    }
                }
                __syncthreads();
            }
        }
        if (tid < P2) out.count[(p << out.b2) + tid] = s_cur[tid];
        __syncthreads();
    }
}

struct ImageArgs {
    SubStore in;
    JoinTable t;             // t.tb == b1 + b2
    uint32_t m;              // buckets per sub-partition = t.bs
    uint32_t* fail;          // [0] set when a slice cannot take all of its rows, [1] when two rows carry the same table word
    uint32_t* sent_rows;     // side list of the sentinel key word (capacity sent_cap)
    uint32_t sent_cap;
    uint32_t* sent_total;
    unsigned long long* inserted;  // += rows placed here (image or side list)
};

// pass 3: LDS image of one table slice.  Dynamic LDS: m*8 key words, then m*8 row ids.
// Persistent workgroups (two per CU), slices q = blockIdx.x, + gridDim.x, ...: the rows of the NEXT slice are loaded
// into registers before the current image is written out, so the only exposed latencies per slice are the LDS clear,
// the inserts and two barriers.  (One workgroup per slice, launched 32 Ki times, ran at a third of the HBM rate: 21 us
// per slice of launch -> count -> rows -> clear -> insert -> store, nothing overlapped; reading the bucket with wide LDS
// loads instead of slot by slot changed nothing — the walk was never the bottleneck.)
// Host guarantees cap2 <= NT * KPT.
template <int NT, int KPT>
__global__ void __launch_bounds__(NT) k_build_images(ImageArgs a) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    unsigned long long* s_keys = (unsigned long long*)s_dyn;
    uint32_t* s_vals = (uint32_t*)(s_dyn + (size_t)a.m * TSQ_BUCKET * 8);
    const uint32_t tid = threadIdx.x;
    const uint32_t Q = 1u << (a.in.b1 + a.in.b2);
    const uint32_t nslots = a.m * TSQ_BUCKET;
    __shared__ unsigned long long s_handled;
    if (tid == 0) s_handled = 0;
    uint32_t placed = 0;  // rows this thread put into an image or the side list (the row list is counted by k_build_insert)
    bool dup = false;
    uint64_t kw[KPT];
    uint32_t row[KPT];
    auto load_rows = [&](uint32_t q, uint32_t cnt) {
        const size_t src = (size_t)q * a.in.cap2;
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            const uint32_t i = (uint32_t)j * NT + tid;
            kw[j] = 0;
            row[j] = 0xffffffffu;
            if (i < cnt) {
                kw[j] = a.in.keys[src + i];
                row[j] = a.in.idx[src + i];
            }
        }
    };
    uint32_t q = blockIdx.x;
    uint32_t cnt = q < Q ? a.in.count[q] : 0u;
    if (q < Q) load_rows(q, cnt);
    while (q < Q) {
        const uint32_t qn = q + gridDim.x;
        const uint32_t cnt_n = qn < Q ? a.in.count[qn] : 0u;
        const uint64_t b0 = (uint64_t)q * a.m;
        {
            ulonglong2* z = (ulonglong2*)s_keys;
            for (uint32_t i = tid; i < nslots / 2; i += NT) z[i] = make_ulonglong2(TSQ_EMPTY_KEY, TSQ_EMPTY_KEY);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            if (row[j] == 0xffffffffu) continue;
            if (kw[j] == TSQ_EMPTY_KEY) {
                const uint32_t o = atomicAdd(a.sent_total, 1u);
                if (o < a.sent_cap) a.sent_rows[o] = row[j];
                placed++;
                continue;
            }
            uint32_t lb = jt_local(a.t.tb, a.m, kw[j]);  // kw[] are table words; the slice is q by construction
            bool done = false;
            for (uint32_t steps = 0; !done; steps++) {
                if (steps >= a.m) {  // the slice is full (heavily duplicated keys): the host rebuilds the table unsliced
                    __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                unsigned long long* b = s_keys + lb * TSQ_BUCKET;
#pragma unroll 1
                for (int sl = 0; sl < TSQ_BUCKET && !done; sl++) {
                    // a stale EMPTY is harmless (the CAS decides); non-EMPTY never reverts
                    unsigned long long cur = __hip_atomic_load(&b[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (cur == TSQ_EMPTY_KEY) {
                        cur = atomicCAS(&b[sl], (unsigned long long)TSQ_EMPTY_KEY, (unsigned long long)kw[j]);
                        if (cur == TSQ_EMPTY_KEY) {
                            s_vals[lb * TSQ_BUCKET + sl] = row[j];
                            done = true;
                            placed++;
                        }
                    }
                    if (!done && cur == kw[j]) dup = true;  // equal words walk the same buckets: the later one sees the earlier one's slot
                }
                lb = (lb + 1 == a.m) ? 0 : lb + 1;
            }
        }
        __syncthreads();
        if (qn < Q) load_rows(qn, cnt_n);  // in flight while the image is stored
        // slice image -> HBM, 16 bytes per lane (the slice starts on a 64-byte boundary: b0 * 64 B keys, b0 * 32 B row ids)
        {
            const ulonglong2* sk = (const ulonglong2*)s_keys;
            ulonglong2* dk = (ulonglong2*)(a.t.keys + b0 * TSQ_BUCKET);
            for (uint32_t i = tid; i < nslots / 2; i += NT) dk[i] = sk[i];
            const uint4* sv = (const uint4*)s_vals;
            uint4* dv = (uint4*)(a.t.vals + b0 * TSQ_BUCKET);
            for (uint32_t i = tid; i < nslots / 4; i += NT) dv[i] = sv[i];
        }
        __syncthreads();
        q = qn;
        cnt = cnt_n;
    }
    if (dup) __hip_atomic_store(a.fail + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t wsum = wave_sum_u64(placed);
    if ((tid & 63) == 0 && wsum) atomicAdd(&s_handled, (unsigned long long)wsum);
    __syncthreads();
    if (tid == 0 && s_handled) atomicAdd(a.inserted, s_handled);
}

// pass 3, counting variant: a bucket's slots are handed out by an LDS counter per bucket instead of a CAS walk over its
// slots.  A row takes ticket r = atomicAdd(count[home]); r < 8 is its slot, no compare, no retry (at load factor 0.75
// that is ~96 % of the rows); a row that finds its bucket sold out moves to the next bucket's counter, which preserves
// the probe's invariant (a row homed at b that lives in b+k implies b .. b+k-1 are full: a counter that reached 8 has
// handed out 8 slots and all of them are written before the image is stored).  Equal table words are found afterwards
// by re-reading the buckets home .. placed with wide LDS loads (the `unique` flag of the probe depends on it).
// Dynamic LDS: m*8 key words | m*8 row ids | m counters.
template <int NT, int KPT>
__global__ void __launch_bounds__(NT) k_build_images_cnt(ImageArgs a) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    unsigned long long* s_keys = (unsigned long long*)s_dyn;
    uint32_t* s_vals = (uint32_t*)(s_dyn + (size_t)a.m * TSQ_BUCKET * 8);
    uint32_t* s_cnt = (uint32_t*)(s_dyn + (size_t)a.m * TSQ_BUCKET * 12);
    const uint32_t tid = threadIdx.x;
    const uint32_t Q = 1u << (a.in.b1 + a.in.b2);
    const uint32_t nslots = a.m * TSQ_BUCKET;
    __shared__ unsigned long long s_handled;
    if (tid == 0) s_handled = 0;
    uint32_t placed = 0;
    bool dup = false;
    uint64_t kw[KPT];
    uint32_t row[KPT];
    uint32_t where[KPT];  // (home bucket << 16) | buckets walked past it; 0xffffffff = not in the image
    auto load_rows = [&](uint32_t q, uint32_t cnt) {
        const size_t src = (size_t)q * a.in.cap2;
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            const uint32_t i = (uint32_t)j * NT + tid;
            kw[j] = 0;
            row[j] = 0xffffffffu;
            if (i < cnt) {
                kw[j] = a.in.keys[src + i];
                row[j] = a.in.idx[src + i];
            }
        }
    };
    uint32_t q = blockIdx.x;
    uint32_t cnt = q < Q ? a.in.count[q] : 0u;
    if (q < Q) load_rows(q, cnt);
    while (q < Q) {
        const uint32_t qn = q + gridDim.x;
        const uint32_t cnt_n = qn < Q ? a.in.count[qn] : 0u;
        const uint64_t b0 = (uint64_t)q * a.m;
        {
            ulonglong2* z = (ulonglong2*)s_keys;
            for (uint32_t i = tid; i < nslots / 2; i += NT) z[i] = make_ulonglong2(TSQ_EMPTY_KEY, TSQ_EMPTY_KEY);
            uint4* zv = (uint4*)s_vals;
            for (uint32_t i = tid; i < nslots / 4; i += NT) zv[i] = make_uint4(0, 0, 0, 0);
            for (uint32_t i = tid; i < a.m; i += NT) s_cnt[i] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            where[j] = 0xffffffffu;
            if (row[j] == 0xffffffffu) continue;
            if (kw[j] == TSQ_EMPTY_KEY) {
                const uint32_t o = atomicAdd(a.sent_total, 1u);
                if (o < a.sent_cap) a.sent_rows[o] = row[j];
                placed++;
                continue;
            }
            const uint32_t home = jt_local(a.t.tb, a.m, kw[j]);
            uint32_t lb = home, steps = 0;
            for (;;) {
                const uint32_t r = atomicAdd(&s_cnt[lb], 1u);
                if (r < TSQ_BUCKET) {
                    s_keys[lb * TSQ_BUCKET + r] = kw[j];
                    s_vals[lb * TSQ_BUCKET + r] = row[j];
                    where[j] = (home << 16) | steps;
                    placed++;
                    break;
                }
                if (++steps >= a.m) {  // the slice is full (heavily duplicated keys): the host rebuilds the table unsliced
                    __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                lb = (lb + 1 == a.m) ? 0 : lb + 1;
            }
        }
        __syncthreads();
        // equal words share their home bucket and sit between it and the bucket the later one reached
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            if (where[j] == 0xffffffffu) continue;
            uint32_t lb = where[j] >> 16, eq = 0;
            for (uint32_t s = 0; s <= (where[j] & 0xffffu); s++) {
                const ulonglong2* b = (const ulonglong2*)(s_keys + lb * TSQ_BUCKET);
#pragma unroll
                for (int h = 0; h < TSQ_BUCKET / 2; h++) {
                    const ulonglong2 v = b[h];
                    eq += (v.x == kw[j]) + (v.y == kw[j]);
                }
                lb = (lb + 1 == a.m) ? 0 : lb + 1;
            }
            if (eq > 1) dup = true;
        }
        if (qn < Q) load_rows(qn, cnt_n);  // in flight while the image is stored
        {
            const ulonglong2* sk = (const ulonglong2*)s_keys;
            ulonglong2* dk = (ulonglong2*)(a.t.keys + b0 * TSQ_BUCKET);
            for (uint32_t i = tid; i < nslots / 2; i += NT) dk[i] = sk[i];
            const uint4* sv = (const uint4*)s_vals;
            uint4* dv = (uint4*)(a.t.vals + b0 * TSQ_BUCKET);
            for (uint32_t i = tid; i < nslots / 4; i += NT) dv[i] = sv[i];
        }
        __syncthreads();
        q = qn;
        cnt = cnt_n;
    }
    if (dup) __hip_atomic_store(a.fail + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t wsum = wave_sum_u64(placed);
    if ((tid & 63) == 0 && wsum) atomicAdd(&s_handled, (unsigned long long)wsum);
    __syncthreads();
    if (tid == 0 && s_handled) atomicAdd(a.inserted, s_handled);
}

#endif
