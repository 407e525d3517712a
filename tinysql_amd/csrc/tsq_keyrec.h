// tsq_keyrec.h — COUNT(*) of an inner join on SEVERAL key columns and on STRING keys, partitioned (round 5; device code, included by
// tsq_join.hip; the aggregate's dictionary of group keys, tsq_keydict.h, partitions its rows with the same passes).
//
// The reference's own join benchmark keys on (bigint, varstring) — `keyIdx: []int{0, 1}`, executor/benchmark_test.go:352-360 — and that
// shape took the direct route: a hash of the key cells finds a slot of the 64-bit table in HBM, then the build row's cells are fetched
// and compared (codec.EqualChunkRow, util/codec/codec.go:363-382).  Per probe row that is one random table line and, on a hit, three to
// four more random lines (row id, offsets, bytes, the integer cell) over hundreds of megabytes: the address translation of random
// 64-byte reads bounds it at ~7e9 lines/s — 2.9e9 probe rows/s at 1e7 x 1e7, 90 times below the packed integer route.
//
// Here no probe leaves the chip's caches.  A row's key cells are packed into a KEY RECORD of 32 bytes — per key column its flag
// byte (8 / 9 / 5: the classes of tsq_key_word; 2 for a string, as codec.go:233-235) followed by the 8-byte word or by a length byte and
// the string's bytes — so two rows have equal keys iff their records are equal byte for byte (the flag makes a string never equal a
// number and an UNSIGNED cell above 2^63 never equal a negative one, codec.go:219-224).  Records that do not fit 32 bytes: a build
// side that has one keeps the direct route; a probe row that has one cannot match any build row.  NULL key cells drop the row on both
// sides (hash_table.go:161-163, join.go:344).  Then
//   k_kr_hist     both sides: records -> 64-bit mix -> partition (top bits); an LDS histogram per workgroup (a contiguous chunk of rows)
//   k_kr_offsets  + k_kr_scan: a partition's records are contiguous, every workgroup's share inside it too
//   k_kr_scatter  records written to their places: 32 B per row, two 16-byte stores
//   k_kr_probe    one workgroup per partition: an open-addressed index over the partition's build records (<= TSQ_KR_CAP: 18-bit tag +
//                 record number per entry) in LDS, the records themselves in the XCD's L2; the probe records stream past: slot walk in
//                 LDS, on a tag match the four words compared.  Duplicate build keys are separate entries (the multimap of rowHashMap).
// Every byte moves in streams: key cells read once per side and pass, 32 B written and 32 B read per row — ~100 B per row pair instead
// of ~5 random lines.  Algorithmic bytes per probe row (SURVEY.md 8d pricing for this key shape): the key cells + one 16-byte slot.
// The build side's records are made once per build (first eligible probe batch) and kept.
#ifndef TSQ_KEYREC_H
#define TSQ_KEYREC_H

#include "tsq_keyrec_dp.h"

#define TSQ_KR_CAP 12288     // build records of one partition the probe kernel indexes in LDS ...
#define TSQ_KR_SLOTS 16384   // ... with one 4-byte entry each (18-bit tag | 14-bit record number): 64 KB, two workgroups per CU
#define TSQ_KR_NT 1024       // threads of the hist / scatter passes
#define TSQ_KR_PNT 512       // threads of the probe kernel
#define TSQ_KR_MAXP 16384    // partitions: one LDS counter each in the hist / scatter passes (64 KB)
#define TSQ_KR_MAXWG 256     // workgroups of the hist / scatter passes = contiguous row chunks
#define TSQ_KR_FILL 8192     // build records per partition, on average (the host picks P for it; + 45 sigma stays below TSQ_KR_CAP)

struct KrArgs {
    KrSrc src;
    uint32_t pbits;          // log2(partitions)
    uint32_t n_wg;           // workgroups of the pass
    int64_t rows_per_wg;     // rows [wg * rows_per_wg, + rows_per_wg) belong to workgroup wg (a multiple of TSQ_KR_NT)
    uint32_t* counts;        // [n_wg][P] rows of (workgroup, partition); after k_kr_offsets: the pair's first place INSIDE its partition
    uint32_t* pstart;        // [P + 1] first record of every partition; [P] = all records (k_kr_offsets: the totals, then scanned)
    unsigned long long* rec; // scatter: [records][4] in partition order
    uint32_t* ids;           // scatter: the source row of every record (nullptr: not kept — COUNT(*) needs no row numbers)
    uint32_t* flags;         // [0] |= 1: a row's record does not fit (build side: the route is off)
    // the aggregate (tsq_keydict.h): 8-byte argument cells travel with the records, rows without a record are listed
    int32_t n_pay;
    const uint64_t* pay_src[TSQ_KR_MAXPAY];
    const uint8_t* pay_nulls[TSQ_KR_MAXPAY];
    uint64_t* pay_dst[TSQ_KR_MAXPAY];
    uint8_t* pay_nn;         // [records] bit v: travelling cell v is NOT NULL (nullptr: no travelling column is nullable)
    // ... as ONE 64-byte slot per row instead (the aggregate with <= 3 travelling columns): words 0-3 the record, 4-6 the cells, word 7 =
    // source row | NOT-NULL bits << 32.  One full 64-byte piece per row instead of three partial lines in three arrays (record 32 B,
    // row id 4 B, cell 8 B: the counters saw ~145 B written per row)
    unsigned long long* slot;
    uint32_t* norec;         // rows whose cells do not fit a record (nullptr: not kept) ...
    int32_t norec_all;       // ... and, for the outer side of an outer join, every other row without a key (NULL cell, selected == 0)
    unsigned long long* norec_count;
};

// the digests of a string column's cells (KrSrc.digest): WAVE = one wave per row (long cells: the lanes take the cell's words in turn,
// 512 contiguous bytes per step), else one lane per row
struct KrDigestArgs {
    const uint8_t* data;
    const int64_t* offs;
    const uint8_t* nulls;
    int64_t nrows;
    uint64_t* out;
    int32_t weak;  // tests (TSQ_KNOB_KEYREC = 3): the digest says nothing beyond the length — every two cells of one length collide and the byte comparison decides
};
template <bool WAVE>
static __global__ void __launch_bounds__(256) k_kr_digest(KrDigestArgs a) {
    const uint32_t lane = WAVE ? (threadIdx.x & 63u) : 0u, step = WAVE ? 64u : 1u;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
    for (int64_t r = WAVE ? (t >> 6) : t; r < a.nrows; r += WAVE ? (nt >> 6) : nt) {
        if (tsq_is_null(a.nulls, r)) {
            if (lane == 0) a.out[r] = 0;
            continue;
        }
        const int64_t o = a.offs[r], n = a.offs[r + 1] - o;
        const uint8_t* p = a.data + o;
        const uint32_t nw = (uint32_t)((n + 7) >> 3);
        uint64_t sum = 0;
        for (uint32_t i = lane; i < nw; i += step) {
            const uint32_t m = (uint64_t)n - 8ull * i < 8ull ? (uint32_t)((uint64_t)n - 8ull * i) : 8u;
            uint64_t v = kr_load8(p + 8ull * i, m);
            if (m < 8u) v &= (1ull << (8u * m)) - 1ull;
            sum += kr_digest_word(v, i);
        }
        if (WAVE) sum = wave_sum_u64(sum);
        if (lane == 0) a.out[r] = kr_digest_finish(a.weak ? 0ull : sum, (uint64_t)n);
    }
}
// are the n bytes at x and y the same?  Asked by a whole wave with the same arguments (the matches of digest records are checked byte for
// byte — the lengths are equal already): lane l compares words l, l + 64, l + 128, l + 192 of each round of 2 KiB
__device__ __forceinline__ bool kr_wave_bytes_equal(const uint8_t* x, const uint8_t* y, uint64_t n, uint32_t lane) {
    const uint64_t nw = (n + 7) >> 3;
    for (uint64_t i0 = 0; i0 < nw; i0 += 256) {
        bool diff = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = i0 + 64u * u + lane;
            if (i < nw) {
                const uint32_t m = n - 8 * i < 8 ? (uint32_t)(n - 8 * i) : 8u;
                uint64_t a = kr_load8(x + 8 * i, m), b = kr_load8(y + 8 * i, m);
                if (m < 8u) {
                    const uint64_t mask = (1ull << (8u * m)) - 1ull;
                    a &= mask;
                    b &= mask;
                }
                diff = diff || a != b;
            }
        }
        if (__ballot(diff)) return false;
    }
    return true;
}

// counts[wg][p]: how many rows of workgroup wg's chunk belong to partition p (an LDS histogram, written out coalesced)
static __global__ void __launch_bounds__(TSQ_KR_NT) k_kr_hist(KrArgs a) {
    extern __shared__ uint32_t s_hist[];
    const uint32_t wg = blockIdx.x, P = 1u << a.pbits;
    for (uint32_t i = threadIdx.x; i < P; i += TSQ_KR_NT) s_hist[i] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)wg * a.rows_per_wg, hi = lo + a.rows_per_wg < a.src.nrows ? lo + a.rows_per_wg : a.src.nrows;
    uint32_t bad = 0;
    for (int64_t row = lo + threadIdx.x; row < hi; row += TSQ_KR_NT) {
        uint64_t w[4];
        bool toolong;
        if (!kr_record(a.src, row, w, &toolong)) {
            bad |= toolong ? 1u : 0u;
            continue;
        }
        const uint32_t p = a.pbits ? (uint32_t)(kr_hash(w) >> (64 - a.pbits)) : 0u;
        atomicAdd(&s_hist[p], 1u);
    }
    // (one atomic per workgroup: with every cell too long for a record — the reference's 5 KiB benchmark key — a lane-by-lane atomicOr was
    // 1e5 atomics on one word, 1.1 ms of the pass that only finds out that digests are needed)
    if (__syncthreads_or((int)bad) && threadIdx.x == 0) atomicOr(a.flags, 1u);
    for (uint32_t i = threadIdx.x; i < P; i += TSQ_KR_NT) a.counts[(size_t)wg * P + i] = s_hist[i];
}
// thread p: the prefix of partition p's counts over the workgroups (coalesced across p), its total -> pstart[p]
static __global__ void __launch_bounds__(256) k_kr_offsets(KrArgs a) {
    const uint32_t P = 1u << a.pbits;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    uint32_t run = 0;
    for (uint32_t w0 = 0; w0 < a.n_wg; w0 += 16) {  // 16 loads in flight: one round trip per 16 workgroups instead of one per workgroup
        uint32_t c[16];
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] = w0 + i < a.n_wg ? a.counts[(size_t)(w0 + i) * P + p] : 0u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (w0 + i < a.n_wg) a.counts[(size_t)(w0 + i) * P + p] = run;
            run += c[i];
        }
    }
    a.pstart[p] = run;
}
// exclusive scan of pstart[0 .. P) in place, pstart[P] = total; flags[1] = the largest partition (one workgroup of 1024 threads)
static __global__ void __launch_bounds__(1024) k_kr_scan(uint32_t* v, uint32_t n, uint32_t* flags) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_run, s_max;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_run = 0; s_max = 0; }
    __syncthreads();
    uint32_t mx = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t c = i < n ? v[i] : 0u;
        mx = c > mx ? c : mx;
        uint32_t x = c;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t before = s_run;
        for (uint32_t w = 0; w < wave; w++) before += s_w[w];
        if (i < n) v[i] = before + x - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = before + x;
        __syncthreads();
    }
    atomicMax(&s_max, mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        v[n] = s_run;
        atomicMax(&flags[1], s_max);
    }
}
// the records to their places: pstart[p] + counts[wg][p] + the row's rank among the workgroup's rows of p (an LDS cursor)
static __global__ void __launch_bounds__(TSQ_KR_NT) k_kr_scatter(KrArgs a) {
    extern __shared__ uint32_t s_cur[];
    const uint32_t wg = blockIdx.x, P = 1u << a.pbits;
    for (uint32_t i = threadIdx.x; i < P; i += TSQ_KR_NT) s_cur[i] = a.pstart[i] + a.counts[(size_t)wg * P + i];
    __syncthreads();
    const int64_t lo = (int64_t)wg * a.rows_per_wg, hi = lo + a.rows_per_wg < a.src.nrows ? lo + a.rows_per_wg : a.src.nrows;
    for (int64_t row = lo + threadIdx.x; row < hi; row += TSQ_KR_NT) {
        uint64_t w[4];
        bool toolong;
        if (!kr_record(a.src, row, w, &toolong)) {
            if ((toolong || a.norec_all) && a.norec) a.norec[atomicAdd(a.norec_count, 1ull)] = (uint32_t)row;  // (rare: one device atomic per such row)
            continue;
        }
        const uint32_t p = a.pbits ? (uint32_t)(kr_hash(w) >> (64 - a.pbits)) : 0u;
        const uint64_t pos = atomicAdd(&s_cur[p], 1u);
        if (a.slot) {
            uint64_t c[3] = {0, 0, 0};
            uint32_t m = 0;
#pragma unroll
            for (int v = 0; v < 3; v++)
                if (v < a.n_pay) {
                    c[v] = a.pay_src[v][row];
                    m |= tsq_is_null(a.pay_nulls[v], row) ? 0u : (1u << v);
                }
            ulonglong2* d = reinterpret_cast<ulonglong2*>(a.slot + pos * 8);
            d[0] = make_ulonglong2(w[0], w[1]);
            d[1] = make_ulonglong2(w[2], w[3]);
            d[2] = make_ulonglong2(c[0], c[1]);
            d[3] = make_ulonglong2(c[2], (unsigned long long)(uint32_t)row | ((unsigned long long)m << 32));
            continue;
        }
        // (measured: NON-TEMPORAL stores of the record, the row id and the travelling cells make this pass 1.8x slower — the scattered pieces
        // of a line do meet in L2 often enough)
        ulonglong2* d = reinterpret_cast<ulonglong2*>(a.rec + pos * 4);
        d[0] = make_ulonglong2(w[0], w[1]);
        d[1] = make_ulonglong2(w[2], w[3]);
        if (a.ids) a.ids[pos] = (uint32_t)row;
        if (a.n_pay) {
            uint32_t m = 0;
#pragma unroll
            for (int v = 0; v < TSQ_KR_MAXPAY; v++)
                if (v < a.n_pay) {
                    a.pay_dst[v][pos] = a.pay_src[v][row];
                    m |= tsq_is_null(a.pay_nulls[v], row) ? 0u : (1u << v);
                }
            if (a.pay_nn) a.pay_nn[pos] = (uint8_t)m;
        }
    }
}

struct KrProbeArgs {
    const unsigned long long* brec;  // build records, partition order
    const uint32_t* bstart;          // [P + 1] first build record of every partition
    const unsigned long long* prec;
    const uint32_t* pstart;
    uint32_t P;
    unsigned long long* counters;    // [0] += joined rows (COUNT(*) mode); EMIT mode: counters[0] is untouched
    // materialising (EMIT): every joined (probe row, build row) pair -> pairs[cursor++] (probe row | build row << 32, the form k_gather_cols
    // takes); batch_count: += joined rows of this launch (the sizing launch runs with pairs == nullptr)
    const uint32_t* bids;
    const uint32_t* pids;
    unsigned long long* pairs;
    unsigned long long* part_cnt;   // [P + 1] sizing launch: joined rows of every partition; (k_kr_scan64) -> the first output row of every partition,
                                    // [P] = all of them; emit launch: positions = part_cnt[p] + an LDS cursor (one SHARED device cursor cost ~11 ns per joined
                                    // row chip-wide: 55 ms for 5e6 rows)
    uint32_t* flags;                 // [0] |= 2: a partition with more than TSQ_KR_CAP build records (cannot happen after the host's check)
    int32_t outer;                   // materialising an outer join: a probe record without a joined build row makes one pair (probe row, TSQ_KR_MISS)
    // digest records (long string keys): equal records are candidates — the cells of these string columns are compared byte for byte
    // (build row bids[..], probe row pids[..]: both kept in this mode)
    int32_t n_verify;
    const uint8_t* vb_data[TSQ_MAX_KEYS];
    const int64_t* vb_offs[TSQ_MAX_KEYS];
    const uint8_t* vp_data[TSQ_MAX_KEYS];
    const int64_t* vp_offs[TSQ_MAX_KEYS];
    // the materialising form probes twice (sizing launch, emit launch): the sizing launch notes in vmask[record] whether EVERY candidate of
    // the probe record passed its byte comparison (anything else is a digest collision, 2^-64) — the emit launch then takes the candidates of
    // such a record as they are, whatever order its index hands them out in, and compares bytes only for the others: a 5 KiB key is read
    // once, not twice.  vmode 0: compare, 1: compare and note, 2: use the notes
    uint32_t* vmask;
    int32_t vmode;
};
#define TSQ_KR_MISS 0xffffffffull
// One workgroup per partition.  The build records of the partition stay where the scatter pass put them (a contiguous window of
// <= 384 KB: it is read once to build the index and then served by the XCD's L2); LDS holds an open-addressed index over them — per
// record one entry (tag = 18 bits of the mix that neither chose the partition nor the slot, 14-bit record number).  A probe record
// walks the slots from its home until an empty one; on a tag match the build record's four words are compared (a false tag match
// costs one more 32-byte read, 2^-18 per slot looked at).  Duplicate build keys are separate entries (the multimap of rowHashMap).
template <bool VERIFY>
static __global__ void __launch_bounds__(TSQ_KR_PNT) k_kr_probe(KrProbeArgs a) {
    __shared__ uint32_t s_tab[TSQ_KR_SLOTS];
    __shared__ unsigned long long s_cnt;
    __shared__ unsigned long long s_pcnt;  // joined rows of the current partition (sizing) / its output cursor (emit)
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;
    unsigned long long mine = 0;
    for (uint32_t p = blockIdx.x; p < a.P; p += gridDim.x) {
        const uint64_t b0 = a.bstart[p], b1 = a.bstart[p + 1];
        const uint64_t p0 = a.pstart[p], p1 = a.pstart[p + 1];
        if (p1 == p0 || (b1 == b0 && !a.outer)) continue;  // (block-uniform)
        uint32_t nb = (uint32_t)(b1 - b0);
        if (nb > TSQ_KR_CAP) {
            if (tid == 0) atomicOr(a.flags, 2u);
            nb = TSQ_KR_CAP;
        }
        __syncthreads();  // the previous partition's probes are done with the index
        if (tid == 0) s_pcnt = 0;
        for (uint32_t i = tid; i < TSQ_KR_SLOTS; i += TSQ_KR_PNT) s_tab[i] = 0xffffffffu;
        __syncthreads();
        for (uint32_t i = tid; i < nb; i += TSQ_KR_PNT) {
            const ulonglong2* s = reinterpret_cast<const ulonglong2*>(a.brec + (b0 + i) * 4);
            const ulonglong2 x = s[0], y = s[1];
            const uint64_t w[4] = {x.x, x.y, y.x, y.y};
            const uint64_t h = kr_hash(w);
            const uint32_t entry = (((uint32_t)(h >> 14) & 0x3ffffu) << 14) | i;
            uint32_t slot = (uint32_t)h & (TSQ_KR_SLOTS - 1);
            while (atomicCAS(&s_tab[slot], 0xffffffffu, entry) != 0xffffffffu) slot = (slot + 1) & (TSQ_KR_SLOTS - 1);
        }
        __syncthreads();
        if (!VERIFY) {
            for (uint64_t r = p0 + tid; r < p1; r += TSQ_KR_PNT) {
                const ulonglong2* s = reinterpret_cast<const ulonglong2*>(a.prec + r * 4);
                const ulonglong2 x = s[0], y = s[1];
                const uint64_t w[4] = {x.x, x.y, y.x, y.y};
                const uint64_t h = kr_hash(w);
                const uint32_t tag = (uint32_t)(h >> 14) & 0x3ffffu;
                uint32_t slot = (uint32_t)h & (TSQ_KR_SLOTS - 1);
                bool any = false;
                for (;;) {
                    const uint32_t e = s_tab[slot];
                    if (e == 0xffffffffu) break;
                    if ((e >> 14) == tag) {
                        const ulonglong2* bq = reinterpret_cast<const ulonglong2*>(a.brec + (b0 + (e & 0x3fffu)) * 4);
                        const ulonglong2 bx = bq[0], by = bq[1];
                        bool same = bx.x == w[0] && bx.y == w[1] && by.x == w[2] && by.y == w[3];
                        if (same) {
                            any = true;
                            mine++;
                            if (a.part_cnt) {  // materialising: count per partition (sizing), or the pair at the partition's next output row (emit)
                                const unsigned long long k = atomicAdd(&s_pcnt, 1ull);
                                if (a.pairs) a.pairs[a.part_cnt[p] + k] = (unsigned long long)a.pids[r] | ((unsigned long long)a.bids[b0 + (e & 0x3fffu)] << 32);
                            }
                        }
                    }
                    slot = (slot + 1) & (TSQ_KR_SLOTS - 1);
                }
                if (a.outer && !any) {  // onMissMatch: the outer row once, NULL-padded
                    const unsigned long long k = atomicAdd(&s_pcnt, 1ull);
                    if (a.pairs) a.pairs[a.part_cnt[p] + k] = (unsigned long long)a.pids[r] | (TSQ_KR_MISS << 32);
                }
            }
        } else {
            // digest records: a record match says "same lengths, same digests" — the bytes are compared by the whole wave, one candidate
            // at a time (64 lanes x 8 bytes of each side per load: a 5 KiB key is ten loads, not 640 dependent ones of one lane), so
            // every lane of the wave walks its slots in step with the others
            const uint32_t lane = tid & 63u;
            for (uint64_t rb = p0 + (tid & ~63u); rb < p1; rb += TSQ_KR_PNT) {  // (wave-uniform)
                const uint64_t r = rb + lane;
                const bool valid = r < p1;
                uint64_t w[4] = {0, 0, 0, 0};
                if (valid) {
                    const ulonglong2* s = reinterpret_cast<const ulonglong2*>(a.prec + r * 4);
                    const ulonglong2 x = s[0], y = s[1];
                    w[0] = x.x, w[1] = x.y, w[2] = y.x, w[3] = y.y;
                }
                const uint64_t h = kr_hash(w);
                const uint32_t tag = (uint32_t)(h >> 14) & 0x3ffffu;
                uint32_t slot = (uint32_t)h & (TSQ_KR_SLOTS - 1);
                const uint32_t prow = valid ? a.pids[r] : 0u;
                bool any = false, walking = valid;
                const bool trusted = a.vmode == 2 && valid && a.vmask[r] != 0u;  // (every candidate of this record passed in the sizing launch)
                bool all_passed = true;
                while (__ballot(walking)) {
                    bool cand = false;
                    uint32_t bi = 0, brow = 0;
                    if (walking) {
                        const uint32_t e = s_tab[slot];
                        if (e == 0xffffffffu) {
                            walking = false;
                        } else if ((e >> 14) == tag) {
                            bi = e & 0x3fffu;
                            const ulonglong2* bq = reinterpret_cast<const ulonglong2*>(a.brec + (b0 + bi) * 4);
                            const ulonglong2 bx = bq[0], by = bq[1];
                            cand = bx.x == w[0] && bx.y == w[1] && by.x == w[2] && by.y == w[3];
                            if (cand) brow = a.bids[b0 + bi];
                        }
                    }
                    bool same = cand;
                    for (uint64_t need = __ballot(cand && !trusted); need; need &= need - 1) {
                        const int L = __builtin_ctzll(need);
                        const uint64_t vb = (uint32_t)__shfl((int)brow, L), vp = (uint32_t)__shfl((int)prow, L);
                        bool eq = true;
                        for (int v = 0; v < a.n_verify && eq; v++) {
                            const int64_t bo = a.vb_offs[v][vb], po = a.vp_offs[v][vp];
                            eq = kr_wave_bytes_equal(a.vb_data[v] + bo, a.vp_data[v] + po, (uint64_t)(a.vb_offs[v][vb + 1] - bo), lane);
                        }
                        if ((int)lane == L) same = eq;
                    }
                    if (cand && !same) all_passed = false;
                    if (same) {
                        any = true;
                        mine++;
                        if (a.part_cnt) {
                            const unsigned long long k = atomicAdd(&s_pcnt, 1ull);
                            if (a.pairs) a.pairs[a.part_cnt[p] + k] = (unsigned long long)prow | ((unsigned long long)brow << 32);
                        }
                    }
                    if (walking) slot = (slot + 1) & (TSQ_KR_SLOTS - 1);
                }
                if (a.vmode == 1 && valid) a.vmask[r] = all_passed ? 1u : 0u;
                if (a.outer && valid && !any) {
                    const unsigned long long k = atomicAdd(&s_pcnt, 1ull);
                    if (a.pairs) a.pairs[a.part_cnt[p] + k] = (unsigned long long)prow | (TSQ_KR_MISS << 32);
                }
            }
        }
        if (a.part_cnt && !a.pairs) {
            __syncthreads();
            if (tid == 0) a.part_cnt[p] = s_pcnt;
        }
    }
    mine = wave_sum_u64(mine);
    __syncthreads();
    if ((tid & 63u) == 0 && mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (tid == 0 && s_cnt && !a.part_cnt) atomicAdd(&a.counters[0], s_cnt);
}
// outer join: the outer rows that have no key at all (listed by the scatter pass), NULL-padded
static __global__ void __launch_bounds__(256) k_kr_miss_pairs(const uint32_t* rows, int64_t n, unsigned long long* pairs) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) pairs[i] = (unsigned long long)rows[i] | (TSQ_KR_MISS << 32);
}
// exclusive scan of v[0 .. n) in place, v[n] = total (one workgroup of 1024 threads, 64-bit counts)
static __global__ void __launch_bounds__(1024) k_kr_scan64(unsigned long long* v, uint32_t n) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_run;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long c = i < n ? v[i] : 0ull;
        unsigned long long x = c;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        unsigned long long before = s_run;
        for (uint32_t w = 0; w < wave; w++) before += s_w[w];
        if (i < n) v[i] = before + x - c;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) v[n] = s_run;
}

#endif
