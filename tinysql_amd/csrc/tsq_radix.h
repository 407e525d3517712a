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
// Layout of a partitioned store: P = 2^bits partitions x R regions x cap key slots.
//   * shared regions  (R = 8)     : region r of partition p is appended to only by workgroups
//     running on XCD r (HW_REG_XCC_ID), so the partially written line at each region's frontier
//     stays in that XCD's L2 until it is complete (write combining in L2; 2^bits x 128 B per XCD).
//     Space is claimed with one returning atomic per (tile, partition).
//   * private regions (R = grid)  : region r belongs to workgroup r; the cursors live in LDS, no
//     global atomics at all.
// A run that does not fit its region goes to the overflow list (skewed keys); the overflow list is
// probed by a plain grid-stride kernel, so the result is exact for every key distribution.
#ifndef TSQ_RADIX_H
#define TSQ_RADIX_H

#include "tsq_jointable.h"

#define TSQ_RADIX_MIN_BITS 3
#define TSQ_RADIX_MAX_BITS 11
#define TSQ_RADIX_MAX_P (1 << TSQ_RADIX_MAX_BITS)
#define TSQ_RADIX_MAXSEG 2048

struct RadixStore {
    uint64_t* keys;        // [P * R * cap] key words
    uint32_t* idx;         // [P * R * cap] source row ids (optional)
    uint32_t* cursor;      // [P * R] slots claimed per region (may exceed cap after an overflow)
    uint32_t* valid_end;   // [P * R] first slot that was NOT written (0xffffffff: none)
    uint64_t* ovf_keys;    // overflow list
    uint32_t* ovf_idx;
    uint32_t* ovf_count;
    uint32_t ovf_cap;
    uint32_t bits, R, cap;
};
struct RadixSrc {  // one key column of a device-resident chunk (util/chunk/column.go:28-34)
    const void* data;
    const uint8_t* nulls;
    int32_t type;
    int32_t skip_high;
    int64_t nrows;
};

__device__ __forceinline__ uint32_t tsq_xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ uint32_t tsq_radix_part(uint64_t kw, uint32_t shift) { return (uint32_t)(tsq_mix64(kw) >> shift); }

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
// Algorithmic bytes: 8 B read + 8 B written per key (+4 B with row ids).
// MINW = waves per SIMD the register allocation must leave room for (blocks/CU * NT / 256).
template <int NT, int K, int MINW, bool PRIVATE, bool WITH_IDX>
__global__ void __launch_bounds__(NT, MINW) k_radix_partition(RadixSrc src, RadixStore st) {
    constexpr int T = NT * K;
    constexpr int MAXPER = (TSQ_RADIX_MAX_P + NT - 1) / NT;
    static_assert(T <= 65536 && (K % 2) == 0, "tile");
    __shared__ uint64_t s_keys[T];
    __shared__ uint32_t s_idx[WITH_IDX ? T : 1];
    __shared__ uint32_t s_hist[TSQ_RADIX_MAX_P];   // per-partition count, then (overflow flag | exclusive offset)
    __shared__ uint32_t s_delta[TSQ_RADIX_MAX_P];  // global slot of the run minus its LDS offset
    __shared__ uint32_t s_cur[PRIVATE ? TSQ_RADIX_MAX_P : 1];
    __shared__ uint32_t s_wsum[NT / 64];
    __shared__ uint32_t s_flag;
    const uint32_t tid = threadIdx.x;
    const uint32_t P = 1u << st.bits, shift = 64 - st.bits;
    const uint32_t r = PRIVATE ? blockIdx.x : tsq_xcc_id();
    const uint32_t per = P >= (uint32_t)NT ? P / NT : 1u;
    if (PRIVATE)
        for (uint32_t p = tid; p < P; p += NT) s_cur[p] = 0;
    if (tid == 0) s_flag = 0;
    const int64_t ntiles = (src.nrows + T - 1) / T;
    const bool wide = src.nulls == nullptr && src.type != TSQ_F32 && !src.skip_high;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * T;
        const int64_t rem = src.nrows - base;
        const uint32_t n = rem < T ? (uint32_t)rem : (uint32_t)T;
        uint64_t k[K];
        uint32_t pr[K];  // (partition << 16) | rank inside the tile, 0xffffffff = no key
        for (uint32_t p = tid; p < P; p += NT) s_hist[p] = 0;
        const bool full = wide && n == (uint32_t)T;
        if (full) {
            const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>((const uint64_t*)src.data + base);
#pragma unroll
            for (int j = 0; j < K / 2; j++) {
                const ulonglong2 v = s2[j * NT + tid];
                k[2 * j] = v.x;
                k[2 * j + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const uint32_t pos = (uint32_t)j * NT + tid;
                pr[j] = 0xffffffffu;
                k[j] = 0;
                if (pos < n && !tsq_is_null(src.nulls, base + pos)) {
                    uint32_t flag;
                    k[j] = tsq_key_word(src.data, src.type, base + pos, &flag);
                    if (!(src.skip_high && (k[j] >> 63))) pr[j] = 0;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K; j++) {
            if (full || pr[j] == 0) {
                const uint32_t p = tsq_radix_part(k[j], shift);
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
#pragma unroll
        for (int q = 0; q < MAXPER; q++) {
            if ((uint32_t)q < per && p0 + q < P) {
                const uint32_t p = p0 + q, cnt = c[q], offs = run;
                run += cnt;
                uint32_t flag = 0;
                if (cnt) {
                    const uint32_t region = p * st.R + r;
                    uint32_t g;
                    if (PRIVATE) {
                        g = s_cur[p];
                        if (g + cnt <= st.cap) s_cur[p] = g + cnt;
                        else flag = 1;
                    } else {
                        g = atomicAdd(&st.cursor[region], cnt);
                        if (g + cnt > st.cap) {
                            flag = 1;
                            atomicMin(&st.valid_end[region], g);
                        }
                    }
                    s_delta[p] = region * st.cap + g - offs;
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
            const uint32_t p = tsq_radix_part(key, shift);
            if (!any_ovf || !(s_hist[p] >> 31)) {
                const uint32_t d = s_delta[p] + i;
                st.keys[d] = key;
                if (WITH_IDX) st.idx[d] = s_idx[i];
            } else {
                const uint32_t o = atomicAdd(st.ovf_count, 1u);
                if (o < st.ovf_cap) {
                    st.ovf_keys[o] = key;
                    if (WITH_IDX) st.ovf_idx[o] = s_idx[i];
                }
            }
        }
        __syncthreads();
    }
    if (PRIVATE)
        for (uint32_t p = tid; p < P; p += NT) st.cursor[p * st.R + r] = s_cur[p];
}

// ------------------------------------------------------------------ partition-at-a-time probe
struct RadixProbeArgs {
    RadixStore st;
    JoinTable t;
    unsigned long long* counters;  // [0] += joined rows
};

// matches of kw in the buckets FOLLOWING bkt (the home bucket was full)
__device__ __noinline__ uint32_t radix_probe_spill(const JoinTable& t, uint64_t kw, uint64_t bkt) {
    uint32_t c = 0;
    for (;;) {
        bkt = (bkt + 1 == t.nbuckets) ? 0 : bkt + 1;
        const ulonglong2* line = reinterpret_cast<const ulonglong2*>(t.keys + bkt * TSQ_BUCKET);
        const ulonglong2 a = line[0], b = line[1], cc = line[2], d = line[3];
        const uint64_t k[8] = {a.x, a.y, b.x, b.y, cc.x, cc.y, d.x, d.y};
        bool has_empty = false;
#pragma unroll
        for (int s = 0; s < TSQ_BUCKET; s++) {
            c += k[s] == kw ? 1u : 0u;
            has_empty |= k[s] == TSQ_EMPTY_KEY;
        }
        if (has_empty) return c;
    }
}

// K3r — COUNT(*) probe over a partitioned key store.  Workgroup b serves virtual XCD b & 7 (the
// dispatcher places workgroup b on XCD b % 8 — observed, used for speed only) and walks the
// partitions p = 8*pi + (b & 7) in order, taking the same 1/J share of every partition, so all
// workgroups of an XCD sweep the same table slice at the same time and the slice stays in that
// XCD's L2.  U keys per lane are in flight, the next U are prefetched.
// Algorithmic bytes: 8 B key + one 16 B slot per probe row.
template <int U>
__global__ void __launch_bounds__(256) k_radix_probe_count(RadixProbeArgs a) {
    __shared__ uint32_t s_base[TSQ_RADIX_MAXSEG], s_n[TSQ_RADIX_MAXSEG];
    const uint32_t tid = threadIdx.x;
    const uint32_t vx = blockIdx.x & 7u, j = blockIdx.x >> 3, J = gridDim.x >> 3;
    const uint32_t P = 1u << a.st.bits, NP = P >> 3, R = a.st.R, cap = a.st.cap;
    const uint32_t nseg_p = R >= J ? R / J : 1u;  // regions of one partition served by this workgroup
    const uint32_t S = R >= J ? 1u : J / R;       // or: slices per region
    const uint32_t nsegs = NP * nseg_p;
    for (uint32_t sg = tid; sg < nsegs; sg += 256) {
        const uint32_t pi = sg / nseg_p, m = sg % nseg_p, p = pi * 8 + vx;
        const uint32_t r = R >= J ? j + m * J : j % R, s = R >= J ? 0u : j / R;
        const uint32_t region = p * R + r;
        uint32_t len = a.st.cursor[region];
        const uint32_t ve = a.st.valid_end[region];
        len = len < ve ? len : ve;
        len = len < cap ? len : cap;
        const uint32_t lo = (uint32_t)((uint64_t)len * s / S), hi = (uint32_t)((uint64_t)len * (s + 1) / S);
        s_base[sg] = region * cap + lo;
        s_n[sg] = hi - lo;
    }
    __syncthreads();
    uint32_t seg = 0, off = 0;
    uint64_t cnt = 0;
    bool alive = true;
    auto fetch = [&](uint64_t& k) -> bool {
        while (seg < nsegs && off >= s_n[seg]) {
            seg++;
            off = 0;
        }
        if (seg >= nsegs) {
            alive = false;
            return false;
        }
        const uint32_t i = off + tid;
        off += 256;
        if (i < s_n[seg]) {
            k = a.st.keys[(size_t)s_base[seg] + i];
            return true;
        }
        return false;
    };
    uint64_t kn[U];
    bool vn[U];
#pragma unroll
    for (int u = 0; u < U; u++) { kn[u] = 0; vn[u] = fetch(kn[u]); }
    bool more = alive || vn[0];
    while (more) {
        uint64_t k[U];
        bool v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { k[u] = kn[u]; v[u] = vn[u]; }
        bool first_alive = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            kn[u] = 0;
            vn[u] = fetch(kn[u]);
            if (u == 0) first_alive = alive;
        }
        more = first_alive;
        uint64_t bkt[U];
        ulonglong2 L[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                bkt[u] = tsq_mulhi64(tsq_mix64(k[u]), a.t.nbuckets);
                const ulonglong2* line = reinterpret_cast<const ulonglong2*>(a.t.keys + bkt[u] * TSQ_BUCKET);
                L[u][0] = line[0]; L[u][1] = line[1]; L[u][2] = line[2]; L[u][3] = line[3];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (v[u]) {
                const uint64_t kw = k[u];
                if (kw == TSQ_EMPTY_KEY) {
                    cnt += a.t.sent_count;
                } else {
                    const uint64_t w[8] = {L[u][0].x, L[u][0].y, L[u][1].x, L[u][1].y, L[u][2].x, L[u][2].y, L[u][3].x, L[u][3].y};
                    uint32_t c = 0;
                    bool has_empty = false;
#pragma unroll
                    for (int s = 0; s < TSQ_BUCKET; s++) {
                        c += w[s] == kw ? 1u : 0u;
                        has_empty |= w[s] == TSQ_EMPTY_KEY;
                    }
                    if (!has_empty) c += radix_probe_spill(a.t, kw, bkt[u]);
                    cnt += c;
                }
            }
        }
    }
    cnt = wave_sum_u64(cnt);
    if ((tid & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
}

// the overflow list (runs that did not fit their region): plain grid-stride probe
__global__ void __launch_bounds__(256) k_radix_probe_ovf(RadixProbeArgs a) {
    uint32_t n = *a.st.ovf_count;
    n = n < a.st.ovf_cap ? n : a.st.ovf_cap;
    uint64_t cnt = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint64_t kw = a.st.ovf_keys[i];
        if (kw == TSQ_EMPTY_KEY) cnt += a.t.sent_count;
        else for_each_slot(a.t, kw, [&](uint64_t) { cnt++; });
    }
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&a.counters[0], (unsigned long long)cnt);
}

#endif
