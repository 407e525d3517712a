// tsq_radix.h — LDS-staged radix partitioning of 64-bit key words and the partition-at-a-time
// probe of the join table (device code; included by tsq_join.hip and tools/radix_ubench.hip).
//
// Why: a probe of a 1.6 GB table is bound by the L2-miss request rate of the chip (~55 G random
// 64-byte lines/s measured, profiles/r01_probe_ubench.txt), i.e. 14 % of the HBM roofline in
// algorithmic bytes.  Random lines out of a table slice that fits ONE XCD's 4 MiB L2 arrive four
// times faster.  Because bucket(kw) is monotonic in h = mix64(kw) (tsq_jointable.h), partitioning
// the probe keys by the top `bits` bits of h makes every partition touch one contiguous slice of
// the unchanged table; the probe kernel then walks the partitions XCD by XCD.
//
// Replaces (reference): the worker dispatch of HashJoinExec — fetchOuterSideChunks handing outer
// chunks to `concurrency` join workers (executor/join.go:160-231) — with a data-dependent dispatch.
// The set of joined rows is unchanged: a probe row meets exactly the build rows with an equal key
// word (util/codec/codec.go:363-382) whichever partition it travels through.
//
// Layout of a partitioned store: P = 2^bits partitions x R = 8 regions x cap key slots.  Region r of
// partition p is appended to only by workgroups running on XCD r (HW_REG_XCC_ID), so the partially
// written line at each region's frontier stays in that XCD's L2 until it is complete (write combining
// in L2; 2^bits x 128 B of frontier per XCD).  Space is claimed with one returning atomic per
// (tile, PAIR of partitions): the cursors of region r are one row cursor[r * P + p], so two neighbouring
// partitions are one 64-bit word and one 64-bit atomic add claims both runs (a cursor never reaches 2^32,
// so the low half cannot carry into the high half).  The atomic rate (27 G/s measured) is what bounds the
// fan-out: with one 32-bit atomic per (tile, partition) 2^10 cost 0.36 ms per 1e8 keys and 2^11 0.59 ms.  Block-private regions without atomics were measured
// slower (0.54-0.65 ms): 8 MB of partially written lines per XCD do not survive in a 4 MiB L2.
// A run that does not fit its region goes to the overflow list (skewed keys); the overflow list is
// probed by a plain grid-stride kernel, so the result is exact for every key distribution.
#ifndef TSQ_RADIX_H
#define TSQ_RADIX_H

#include "tsq_jointable.h"

#define TSQ_RADIX_MIN_BITS 3
#define TSQ_RADIX_MAX_BITS 11
#define TSQ_RADIX_MAX_P (1 << TSQ_RADIX_MAX_BITS)
#define TSQ_RADIX_MAXSEG 2048

#define TSQ_RADIX_MAXV 2
struct RadixStore {
    uint64_t* keys;        // [P * R * cap] key words
    uint64_t* pay[TSQ_RADIX_MAXV];      // [P * R * cap] payload cells travelling with the key (optional)
    uint64_t* ovf_pay[TSQ_RADIX_MAXV];  // payload of the overflow list
    uint32_t* idx;         // [P * R * cap] source row ids (optional)
    uint32_t* cursor;      // [R][P] slots claimed per region (may exceed cap after an overflow); 8-byte aligned
    uint32_t* valid_end;   // [R][P] first slot that was NOT written (0xffffffff: none)
    uint64_t* ovf_keys;    // overflow list
    uint32_t* ovf_idx;
    uint32_t* ovf_count;
    unsigned long long* queue;  // 8 chunk-queue heads, TSQ_RADIX_QSTRIDE words apart (probe side)
    // multi-GPU redistribute (tsq_radix_split): partition = tsq_key_rank(key, rank_parts) instead of the top hash bits,
    // one exactly sized region per part starting at region_base[part] (cap is then unused: nothing can overflow)
    const uint32_t* region_base;
    uint32_t rank_parts;
    uint32_t ovf_cap;
    uint32_t bits, R, cap;
};
struct RadixSrc {  // one key column (+ up to two payload columns) of a device-resident chunk (util/chunk/column.go:28-34)
    const void* data;
    const uint8_t* nulls;
    int32_t type;
    int32_t skip_high;
    int64_t nrows;
    // key_kind 0: join key word (util/codec/codec.go:212-240); 1: GROUP BY key word (codec.go:713-746: reals by
    // their memcomparable image).  Rows with a NULL key or a NULL payload cell are not partitioned: kind 0 drops
    // them (inner join), kind 1 appends their row ids to exc_rows (the aggregate handles them row by row).
    int32_t key_kind;
    int32_t vtype[TSQ_RADIX_MAXV];
    const void* vdata[TSQ_RADIX_MAXV];
    const uint8_t* vnulls[TSQ_RADIX_MAXV];
    uint32_t* exc_rows;
    uint32_t* exc_count;
};
__device__ __forceinline__ uint64_t radix_src_key(const RadixSrc& src, int64_t row) {
    if (src.key_kind == 0) {
        uint32_t flag;
        return tsq_key_word(src.data, src.type, row, &flag);
    }
    if (src.type == TSQ_F32 || src.type == TSQ_F64) {  // util/codec/float.go:22-30
        const double f = src.type == TSQ_F32 ? (double)((const float*)src.data)[row] : ((const double*)src.data)[row];
        const uint64_t u = tsq_f64_bits(f);
        return f >= 0 ? (u | 0x8000000000000000ULL) : ~u;
    }
    return ((const uint64_t*)src.data)[row];
}

__device__ __forceinline__ uint32_t tsq_xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ uint32_t tsq_radix_part(uint64_t kw, uint32_t shift) { return (uint32_t)(tsq_mix64(kw) >> shift); }
// index of (partition p, region r) in cursor[] / valid_end[]; the DATA of the region sits at (p * R + r) * cap
__device__ __forceinline__ uint32_t radix_ctl(const RadixStore& st, uint32_t P, uint32_t p, uint32_t r) { return r * P + p; }
__device__ __forceinline__ uint32_t radix_region_len(const RadixStore& st, uint32_t P, uint32_t p, uint32_t r) {
    const uint32_t c = radix_ctl(st, P, p, r);
    uint32_t len = st.cursor[c];
    const uint32_t ve = st.valid_end[c];
    len = len < ve ? len : ve;
    return len < st.cap ? len : st.cap;
}

// exclusive prefix sum over the NT threads of a workgroup (contains one __syncthreads)
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wsum, uint32_t* total) {
    uint32_t wtot;
    const uint32_t ex = wave_excl_scan_u32(v, &wtot);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_wsum[w] = wtot;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        const uint32_t x = s_wsum[i];
        pre += i < w ? x : 0u;
        tot += x;
    }
    *total = tot;
    return pre + ex;
}

// K5a — radix partition.  One workgroup sorts a tile of NT*K key words by partition inside LDS
// (LDS histogram with returning ds_add -> block scan -> LDS scatter) and writes each partition's run
// with consecutive lanes on consecutive addresses.
// Algorithmic bytes: 8 B read + 8 B written per key (+8 B each way per payload column, +4 B with row ids).
// MINW = waves per SIMD the register allocation must leave room for (blocks/CU * NT / 256).
// HASHED: the store receives TABLE WORDS w = mix64(key word) (tsq_jointable.h) instead of key words, and the
// partition is the top bits of w — what the join paths use (the consumer never hashes again).
template <int NT, int K, int MINW, int V, bool WITH_IDX, bool HASHED = false>
__global__ void __launch_bounds__(NT, MINW) k_radix_partition(RadixSrc src, RadixStore st) {
    constexpr int T = NT * K;
    static_assert(V >= 0 && V <= TSQ_RADIX_MAXV, "payload columns");
    __shared__ uint64_t s_pay[V ? V : 1][V ? T : 1];
    constexpr int MAXPER = (TSQ_RADIX_MAX_P + NT - 1) / NT;
    static_assert(T <= 65536 && (K % 2) == 0, "tile");
    __shared__ uint64_t s_keys[T];
    __shared__ uint32_t s_idx[WITH_IDX ? T : 1];
    __shared__ uint32_t s_hist[TSQ_RADIX_MAX_P];   // per-partition count, then (overflow flag | exclusive offset)
    __shared__ uint32_t s_delta[TSQ_RADIX_MAX_P];  // global slot of the run minus its LDS offset
    __shared__ uint32_t s_wsum[NT / 64];
    __shared__ uint32_t s_flag;
    const uint32_t tid = threadIdx.x;
    const uint32_t P = st.rank_parts ? st.rank_parts : 1u << st.bits, shift = 64 - st.bits;
    const uint32_t r = st.rank_parts ? 0u : tsq_xcc_id();
    const uint32_t per = P >= (uint32_t)NT ? P / NT : 1u;
    if (tid == 0) s_flag = 0;
    const int64_t ntiles = (src.nrows + T - 1) / T;
    auto part_of = [&](uint64_t kw) -> uint32_t {
        if (HASHED) return (uint32_t)(kw >> shift);
        return st.rank_parts ? tsq_key_rank(kw, st.rank_parts) : tsq_radix_part(kw, shift);
    };
    bool wide = src.nulls == nullptr && src.type != TSQ_F32 && !src.skip_high && !(src.key_kind == 1 && src.type == TSQ_F64);
#pragma unroll
    for (int v = 0; v < V; v++) wide = wide && src.vnulls[v] == nullptr && src.vtype[v] != TSQ_F32;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * T;
        const int64_t rem = src.nrows - base;
        const uint32_t n = rem < T ? (uint32_t)rem : (uint32_t)T;
        uint64_t k[K];
        uint64_t pay[V ? V : 1][K];
        uint32_t pr[K];  // (partition << 16) | rank inside the tile, 0xffffffff = no key
        for (uint32_t p = tid; p < P; p += NT) s_hist[p] = 0;
        const bool full = wide && n == (uint32_t)T;
        if (full) {
            // every load of the tile first, the hashing afterwards: with tsq_table_word() written next to its load the compiler
            // waited for each 16-byte load before it issued the next one (K / 2 serial HBM round trips per tile)
            const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>((const uint64_t*)src.data + base);
#pragma unroll
            for (int j = 0; j < K / 2; j++) {
                const ulonglong2 v = s2[j * NT + tid];
                k[2 * j] = v.x;
                k[2 * j + 1] = v.y;
            }
#pragma unroll
            for (int vv = 0; vv < V; vv++) {
                const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>((const uint64_t*)src.vdata[vv] + base);
#pragma unroll
                for (int j = 0; j < K / 2; j++) {
                    const ulonglong2 v = p2[j * NT + tid];
                    pay[vv][2 * j] = v.x;
                    pay[vv][2 * j + 1] = v.y;
                }
            }
            if (HASHED) {
                __builtin_amdgcn_sched_barrier(0);  // nothing moves across: the loads above stay ahead of the multiplies below
#pragma unroll
                for (int j = 0; j < K; j++) k[j] = tsq_table_word(k[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const uint32_t pos = (uint32_t)j * NT + tid;
                pr[j] = 0xffffffffu;
                k[j] = 0;
#pragma unroll
                for (int vv = 0; vv < V; vv++) pay[vv][j] = 0;
                if (pos < n) {
                    bool isnull = tsq_is_null(src.nulls, base + pos);
#pragma unroll
                    for (int vv = 0; vv < V; vv++) isnull |= tsq_is_null(src.vnulls[vv], base + pos);
                    if (!isnull) {
                        k[j] = radix_src_key(src, base + pos);
                        if (!(src.skip_high && (k[j] >> 63))) pr[j] = 0;
                        if (HASHED) k[j] = tsq_table_word(k[j]);
#pragma unroll
                        for (int vv = 0; vv < V; vv++)
                            pay[vv][j] = src.vtype[vv] == TSQ_F32 ? (uint64_t)((const uint32_t*)src.vdata[vv])[base + pos]
                                                                 : ((const uint64_t*)src.vdata[vv])[base + pos];
                    } else if (src.key_kind == 1) {
                        const uint32_t e = __hip_atomic_fetch_add(src.exc_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        src.exc_rows[e] = (uint32_t)(base + pos);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K; j++) {
            if (full || pr[j] == 0) {
                const uint32_t p = part_of(k[j]);
                pr[j] = (p << 16) | atomicAdd(&s_hist[p], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of the histogram; thread t owns partitions [t*per, (t+1)*per)
        uint32_t c[MAXPER], sum = 0;
        const uint32_t p0 = tid * per;
#pragma unroll
        for (int q = 0; q < MAXPER; q++) {
            c[q] = ((uint32_t)q < per && p0 + q < P) ? s_hist[p0 + q] : 0u;
            sum += c[q];
        }
        uint32_t total;
        uint32_t run = block_excl_scan<NT>(sum, s_wsum, &total);
        // claim the runs: one 64-bit atomic per pair of neighbouring partitions (per is 1 or even)
        uint32_t g[MAXPER];
        if (per >= 2 && !st.rank_parts) {
#pragma unroll
            for (int q = 0; q < MAXPER; q += 2) {
                g[q] = 0;
                if (q + 1 < MAXPER) g[q + 1] = 0;
                if ((uint32_t)q < per && p0 + q < P && (c[q] | c[q + 1 < MAXPER ? q + 1 : q])) {
                    const uint32_t c1 = q + 1 < MAXPER ? c[q + 1] : 0u;
                    unsigned long long* cw = reinterpret_cast<unsigned long long*>(st.cursor + radix_ctl(st, P, p0 + q, r));
                    const unsigned long long old = atomicAdd(cw, (unsigned long long)c[q] | ((unsigned long long)c1 << 32));
                    g[q] = (uint32_t)old;
                    if (q + 1 < MAXPER) g[q + 1] = (uint32_t)(old >> 32);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < MAXPER; q++) {
                g[q] = 0;
                if ((uint32_t)q < per && p0 + q < P && c[q]) g[q] = atomicAdd(&st.cursor[radix_ctl(st, P, p0 + q, r)], c[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < MAXPER; q++) {
            if ((uint32_t)q < per && p0 + q < P) {
                const uint32_t p = p0 + q, cnt = c[q], offs = run;
                run += cnt;
                uint32_t flag = 0;
                if (cnt) {
                    const uint32_t region = p * st.R + r;
                    if (!st.region_base && g[q] + cnt > st.cap) {
                        flag = 1;
                        atomicMin(&st.valid_end[radix_ctl(st, P, p, r)], g[q]);
                    }
                    s_delta[p] = (st.region_base ? st.region_base[region] : region * st.cap) + g[q] - offs;
                    if (flag) s_flag = 1;
                }
                s_hist[p] = offs | (flag << 31);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K; j++) {
            if (full || pr[j] != 0xffffffffu) {
                const uint32_t d = (s_hist[pr[j] >> 16] & 0x7fffffffu) + (pr[j] & 0xffffu);
                s_keys[d] = k[j];
#pragma unroll
                for (int vv = 0; vv < V; vv++) s_pay[vv][d] = pay[vv][j];
                if (WITH_IDX) {
                    const uint32_t pos = full ? (((uint32_t)(j >> 1) * NT + tid) * 2 + (j & 1)) : ((uint32_t)j * NT + tid);
                    s_idx[d] = (uint32_t)base + pos;
                }
            }
        }
        __syncthreads();
        const bool any_ovf = s_flag != 0;
        for (uint32_t i = tid; i < total; i += NT) {
            const uint64_t key = s_keys[i];
            const uint32_t p = part_of(key);
            if (!any_ovf || !(s_hist[p] >> 31)) {
                const uint32_t d = s_delta[p] + i;
                st.keys[d] = key;
#pragma unroll
                for (int vv = 0; vv < V; vv++) st.pay[vv][d] = s_pay[vv][i];
                if (WITH_IDX) st.idx[d] = s_idx[i];
            }
        }
        if (any_ovf) {  // rare (skewed keys): runs that did not fit go to the overflow list, one slot at a time
            for (uint32_t i = tid; i < total; i += NT) {
                const uint64_t key = s_keys[i];
                const uint32_t p = part_of(key);
                if (s_hist[p] >> 31) {
                    const uint32_t o = __hip_atomic_fetch_add(st.ovf_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (o < st.ovf_cap) {
                        st.ovf_keys[o] = key;
#pragma unroll
                        for (int vv = 0; vv < V; vv++) st.ovf_pay[vv][o] = s_pay[vv][i];
                        if (WITH_IDX) st.ovf_idx[o] = s_idx[i];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ partition-at-a-time probe
struct RadixProbeArgs {
    RadixStore st;  // HASHED store: keys[] holds table words
    JoinTable t;
    unsigned long long* counters;  // [0] += joined rows
};

// matches of table word w in the buckets FOLLOWING global bucket bkt (the home bucket was full); the walk wraps inside the slice
static __device__ __noinline__ uint32_t radix_probe_spill(const JoinTable& t, uint64_t w, uint64_t bkt) {
    uint32_t c = 0;
    const uint64_t base = (uint64_t)jt_slice(t.tb, w) * t.bs;
    uint32_t lb = (uint32_t)(bkt - base);
    for (uint32_t steps = 1; steps < t.bs; steps++) {
        lb = (lb + 1 == t.bs) ? 0 : lb + 1;
        const ulonglong2* line = reinterpret_cast<const ulonglong2*>(t.keys + (base + lb) * TSQ_BUCKET);
        const ulonglong2 a = line[0], b = line[1], cc = line[2], d = line[3];
        const uint64_t k[8] = {a.x, a.y, b.x, b.y, cc.x, cc.y, d.x, d.y};
        bool has_empty = false;
#pragma unroll
        for (int s = 0; s < TSQ_BUCKET; s++) {
            c += k[s] == w ? 1u : 0u;
            has_empty |= k[s] == TSQ_EMPTY_KEY;
        }
        if (has_empty) return c;
    }
    return c;
}

// K3r — COUNT(*) probe over a partitioned key store.
//
// Ordering is what makes the L2 work: workgroup b serves virtual XCD vx = b & 7 (the dispatcher places
// workgroup b on XCD b % 8 — observed on gfx950, used for speed only) and draws chunk tickets from
// queue[vx].  Tickets enumerate the chunks (256*U keys) of partitions vx, vx+8, vx+16, ... in order, so
// the keys in flight on one XCD always belong to a window of 2-3 consecutive partitions and their table
// slices (contiguous, ~1.5 MB each) stay in that XCD's 4 MiB L2 (measured: TCC hit rate 25 % with a
// static assignment that lets workgroups drift apart, 84 % with the ordered queue — profiles/).
// Software pipeline: thread 0 (scout) has the ticket atomic of chunk i+2 in flight and decodes chunk
// i+1's descriptor into LDS while chunk i's table lines are in flight; the key loads of chunk i+1 are
// issued after the table loads of chunk i so waiting for the lines does not wait for them.
// Quad probing: the 4 x 16-byte pieces of a 64-byte bucket are read by 4 adjacent lanes with ONE
// dwordx4 load (16 distinct lines per wave instruction instead of 64: the texture-address unit is
// charged per line — tools/ta_ubench.hip).  A streaming prefetch of the next slice was measured and
// dropped (1.28 ms vs 1.04 ms without).  A key whose home bucket is full is parked in LDS and
// resolved later by full waves (deferred spill) instead of stalling its wave on a dependent load.
// Algorithmic bytes: 8 B key + one 16 B slot per probe row.
#define TSQ_RADIX_QSTRIDE 64  // queue heads 512 B apart: each on its own L2 channel
template <int U, int MINW = 6>
__global__ void __launch_bounds__(256, MINW) k_radix_probe_count(RadixProbeArgs a) {
    constexpr uint32_t CH = 256 * U;
    constexpr uint32_t END = 0xffffffffu;
    constexpr uint32_t SPCAP = 2 * CH > 1024 ? 2 * CH : 1024;
    __shared__ uint32_t s_len[TSQ_RADIX_MAXSEG];               // [pi * 8 + r] keys stored in region r of partition 8 pi + vx
    __shared__ uint32_t s_cstart[TSQ_RADIX_MAX_P / 8 + 1];     // chunks before partition pi
    __shared__ uint32_t s_wsum[4];
    __shared__ uint64_t s_dsrc[2];
    __shared__ uint32_t s_dn[2];
    __shared__ uint64_t s_spk[SPCAP], s_spb[SPCAP];
    __shared__ uint32_t s_spn;
    __shared__ unsigned long long s_total;
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t vx = blockIdx.x & 7u;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3, cap = a.st.cap;
    if (tid == 0) {
        s_spn = 0;
        s_total = 0;
    }
    for (uint32_t i = tid; i < NP * 8; i += 256) s_len[i] = radix_region_len(a.st, P, (i >> 3) * 8 + vx, i & 7);
    __syncthreads();
    {
        uint32_t nch = 0;
        if (tid < NP)
            for (int r = 0; r < 8; r++) nch += ((uint32_t)s_len[tid * 8 + r] + CH - 1) / CH;
        uint32_t total;
        const uint32_t ex = block_excl_scan<256>(nch, s_wsum, &total);
        if (tid < NP) s_cstart[tid] = ex;
        if (tid == 0) s_cstart[NP] = total;
    }
    __syncthreads();
    const uint32_t nchunks = s_cstart[NP];
    auto take = [&]() -> uint32_t {
        return (uint32_t)__hip_atomic_fetch_add(&a.st.queue[vx * TSQ_RADIX_QSTRIDE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto decode = [&](uint32_t t, int slot) {  // thread 0 only: ticket -> (partition, region, chunk)
        if (t >= nchunks) {
            s_dn[slot] = END;
            return;
        }
        uint32_t lo = 0, hi = NP;  // s_cstart[lo] <= t < s_cstart[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_cstart[mid] <= t) lo = mid;
            else hi = mid;
        }
        const uint32_t pi = lo;
        uint32_t c = t - s_cstart[pi], r = 0;
        for (; r < 7; r++) {
            const uint32_t cr = ((uint32_t)s_len[pi * 8 + r] + CH - 1) / CH;
            if (c < cr) break;
            c -= cr;
        }
        const uint32_t p = pi * 8 + vx, len = s_len[pi * 8 + r];
        s_dn[slot] = len - c * CH < CH ? len - c * CH : CH;
        s_dsrc[slot] = (uint64_t)(a.st.keys + (size_t)(p * 8 + r) * cap + (size_t)c * CH);
    };
    uint64_t cnt = 0;   // per lane (drain path)
    uint64_t scnt = 0;  // per wave, lives in scalar registers
    uint32_t tk_pending = 0;
    if (tid == 0) {
        const uint32_t tA = take(), tB = take();
        tk_pending = take();
        decode(tA, 0);
        decode(tB, 1);
    }
    __syncthreads();
    uint64_t kn[U];
    uint32_t nn = s_dn[0];
#pragma unroll
    for (int u = 0; u < U; u++) {
        kn[u] = 0;
        const uint32_t i = u * 256 + tid;
        if (nn != END && i < nn) kn[u] = __builtin_nontemporal_load((const uint64_t*)s_dsrc[0] + i);
    }
    for (uint32_t it = 0;; it++) {
        const uint32_t n = nn;
        if (n == END) break;
        uint64_t k[U], bkt[U];
        ulonglong2 L[U][4];  // L[u][g]: 16 bytes (lane & 3) of the bucket of the key held by lane 16 g + (lane >> 2)
#pragma unroll
        for (int u = 0; u < U; u++) {
            k[u] = kn[u];
            bkt[u] = (uint32_t)(u * 256) + tid < n ? (uint64_t)jt_slice(a.t.tb, k[u]) * a.t.bs + jt_local(a.t.tb, a.t.bs, k[u]) : 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint64_t bq = __shfl(bkt[u], g * 16 + (lane >> 2), 64);
                L[u][g] = reinterpret_cast<const ulonglong2*>(a.t.keys + bq * TSQ_BUCKET)[lane & 3];
            }
        }
        const int ns = (it + 1) & 1;
        nn = s_dn[ns];
#pragma unroll
        for (int u = 0; u < U; u++) {
            kn[u] = 0;
            const uint32_t i = u * 256 + tid;
            if (nn != END && i < nn) kn[u] = __builtin_nontemporal_load((const uint64_t*)s_dsrc[ns] + i);
        }
        if (tid == 0) {  // chunk it+2 -> slot it & 1 (its readers finished before the previous barrier)
            decode(tk_pending, it & 1);
            tk_pending = take();
        }
        // Matches are counted on the scalar unit: the kernel is bound by vector-instruction issue (SQ counters:
        // ACTIVE_INST_ANY x waves per SIMD > 1), so every compare writes a 64-bit lane mask and the validity /
        // sentinel / "quad has an EMPTY slot" logic, the popcounts and the adds run as SALU on those masks.
        const int n_u = (int)__builtin_amdgcn_readfirstlane(n), wave0 = (int)__builtin_amdgcn_readfirstlane(tid & ~63u);
#pragma unroll
        for (int u = 0; u < U; u++) {
            int nv = n_u - (u * 256 + wave0);  // keys of this wave in this round: source lanes [0, nv)
            nv = nv < 0 ? 0 : (nv > 64 ? 64 : nv);
            const uint64_t src_sent = __ballot(k[u] == TSQ_EMPTY_KEY);  // source lanes whose probe key is the sentinel word (rare)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int q = 4 * (nv - 16 * g);  // lane l reads the bucket of source lane 16 g + (l >> 2)
                q = q < 0 ? 0 : (q > 64 ? 64 : q);
                const uint64_t vmask = q >= 64 ? ~0ull : ((1ull << q) - 1ull);
                if (vmask == 0) continue;
                const int srcl = g * 16 + (lane >> 2);
                const uint64_t kw = __shfl(k[u], srcl, 64);
                const uint64_t x = L[u][g].x, y = L[u][g].y;
                uint64_t m_live = vmask;
                if ((src_sent >> (16 * g)) & 0xffffull) {  // a sentinel probe key in this group of 16: it matches the side list only
                    const uint64_t m_sent = __ballot(kw == TSQ_EMPTY_KEY) & vmask;
                    m_live = vmask & ~m_sent;
                    scnt += (uint64_t)__popcll(m_sent & 0x1111111111111111ull) * a.t.sent_count;
                }
                scnt += (uint64_t)__popcll(__ballot(x == kw) & m_live) + (uint64_t)__popcll(__ballot(y == kw) & m_live);
                // slots are claimed in order (k_build_insert takes the first EMPTY one), so a bucket still has an EMPTY
                // slot iff its LAST slot is EMPTY: only the y word of lane 4q+3 has to be looked at
                const uint64_t quad_e = (__ballot(y == TSQ_EMPTY_KEY) >> 3) & 0x1111111111111111ull;
                const uint64_t park = m_live & 0x1111111111111111ull & ~quad_e;  // home bucket full: deferred spill
                if (park) {
                    const uint64_t bq = __shfl(bkt[u], srcl, 64);
                    if ((park >> lane) & 1ull) {
                        const uint32_t sl = atomicAdd(&s_spn, 1u);
                        s_spk[sl] = kw;
                        s_spb[sl] = bq;
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t spn = s_spn;
        if (nn == END || spn + CH > SPCAP) {  // drain the parked keys with full waves
            for (uint32_t i = tid; i < spn; i += 256) cnt += radix_probe_spill(a.t, s_spk[i], s_spb[i]);
            __syncthreads();
            if (tid == 0) s_spn = 0;
            __syncthreads();
        }
    }
    cnt = wave_sum_u64(cnt) + scnt;
    if ((tid & 63) == 0 && cnt) atomicAdd(&s_total, (unsigned long long)cnt);
    __syncthreads();
    if (tid == 0 && s_total) atomicAdd(&a.counters[0], s_total);  // one device atomic per workgroup
}


// the overflow list (runs that did not fit their region): plain grid-stride probe
static __global__ void __launch_bounds__(256) k_radix_probe_ovf(RadixProbeArgs a) {
    uint32_t n = *a.st.ovf_count;
    n = n < a.st.ovf_cap ? n : a.st.ovf_cap;
    uint64_t cnt = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint64_t w = a.st.ovf_keys[i];  // table words (HASHED store)
        if (w == TSQ_EMPTY_KEY) cnt += a.t.sent_count;
        else for_each_slot_w(a.t, w, [&](uint64_t) { cnt++; });
    }
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
}

#endif
