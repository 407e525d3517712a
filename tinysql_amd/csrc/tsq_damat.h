// tsq_damat.h — the materialising packed join with the build side in LDS (round 6; device code, included by tsq_join.hip).
//
// HashJoinExec.Next always materialises the joined rows (executor/join.go:290-323, joiner.go:145-150, util/chunk/chunk.go:334-356).
// Round 4's route for it (tsq_dajoin.h: k_da_partition_cols -> k_da_emit_cols) keeps the build columns SORTED BY WORD in HBM and reads
// a build cell with one random 8-byte load from its partition's window: 390 KB per partition for 1e8 build rows, 512 partitions in
// flight — the windows outgrow the L2 (4 MB per XCD) and every read fetches a line: the emit kernel's counters show 5.6 GB fetched for
// 1.9 GB of algorithmic reads (profiles/r06_mat_base_pmc.txt), 2.1 ms per 1e8 joined rows, plus 1.2 ms once per build for the sort.
//
// Here a partition's build rows sit in LDS, so a build cell costs an LDS read.  That needs partitions of a few thousand build rows —
// 2^14 of them for 1e8 — and one partition pass cannot fan out that far (2^13 partitions: 1.5 ms per 1e8 (k, v) rows against 0.7 ms at
// 2^11, tools/mat_ubench.hip: every XCD keeps 2^13 x 2 frontier lines open and its L2 stops merging the 2- and 8-byte stores).  So:
//   level 1  k_da_partition_cols (unchanged): 2^11 partitions x 8 XCD regions, entries + travelling columns
//   level 2  k_dm_split: ONE workgroup per level-1 partition splits it S ways (S <= 8) by the next bits of the word into FINAL partitions
//            q = p * S + s whose rows are contiguous; an inner join drops the probe rows whose word is not in the build side's bitmap on
//            the way (a semi-join filter: they would be read and thrown away by the emit kernel), so the counts it leaves are the
//            output rows per final partition — the sizing pass of the old route comes for free
//   emit     k_dm_emit: per final partition the build rows become a RANKED table in LDS (presence bits + popcount prefix + the payload
//            cells in word order), the probe rows stream through, and since the build side has no duplicate keys every probe row makes
//            exactly one output row at (base of q) + (its index in q): no compaction, every lane writes two consecutive rows of a
//            column with one 16-byte store on a 16-byte boundary (a wave: 1 KB contiguous)
// Eligible: what the travelling-columns route takes, with a build side WITHOUT duplicate keys (the images kernel knows) and no run
// that overflowed its level-1 region in this batch (skewed keys: the old route keeps the batch).  Left / right outer joins: nothing is
// dropped at level 2, the emit kernel pads the rows whose word has no bit.
// Replaces (reference): joiner.tryToMatchInners / onMissMatch + Chunk.AppendRow (joiner.go:145-410, chunk.go:334-356), as K4e did.
// Bytes per probe row and travelling column: 8 read + 8 written (level 1), 8 + 8 (level 2), 8 read (emit) + 8 per output cell.
#ifndef TSQ_DAMAT_H
#define TSQ_DAMAT_H

#include "tsq_dajoin.h"

#define TSQ_DM_MAXS 8
struct DmStore {                        // the FINAL partitions of one side: q = p * S + s; the rows of q are contiguous
    uint16_t* ent;                      // [P1 * cap1] entries (the word's low ebits2 bits); level-1 partition p owns [p * cap1, (p + 1) * cap1)
    uint64_t* pay[TSQ_DA_MAXCOLS];      // travelling columns, same slots
    uint8_t* nnmask;                    // NOT-NULL bits of a row's travelling cells (null: no travelling column is nullable)
    uint32_t* off;                      // [Q] first slot of q (a multiple of 8: 16-byte loads of entries)
    unsigned long long* cnt;            // [Q + 1] rows of q (the probe side's are scanned into output bases: rows of q = cnt[q + 1] - cnt[q])
    uint32_t cap1;                      // slots per level-1 partition
    uint32_t sbits;                     // log2 S
    uint32_t ebits2;                    // entry bits of a final partition = ebits1 - sbits
};
struct DmSplitArgs {
    DaColStore src;                     // level 1
    int32_t n_cols;
    DmStore dst;
    const uint32_t* bitmap;             // FILTER: one bit per word of the build side's domain (bit index = the word)
};

// region r of a level-1 partition holds pairs [s_pbase[r], s_pbase[r + 1]) of the partition's pairs of rows (regions start on 128-byte
// lines, so a pair of entries is one aligned 4-byte word and a pair of cells one aligned 16-byte unit)
__device__ __forceinline__ uint32_t dm_region_of(const uint32_t* s_base, uint32_t x) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 1; k < 8; k++) r += x >= s_base[k] ? 1u : 0u;
    return r;
}

template <int NT, bool FILTER>
__global__ void __launch_bounds__(NT) k_dm_split(DmSplitArgs a) {
    constexpr int K = 4;                // pairs of rows per thread and tile
    constexpr int T = NT * K * 2;       // rows per tile
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ uint32_t s_len[8], s_ubase[9], s_pbase[9];
    __shared__ uint32_t s_cnt[TSQ_DM_MAXS], s_run[TSQ_DM_MAXS], s_tcnt[TSQ_DM_MAXS], s_tpre[TSQ_DM_MAXS + 1];
    uint64_t* s_pay = reinterpret_cast<uint64_t*>(s_dyn);
    uint16_t* s_w = reinterpret_cast<uint16_t*>(s_dyn + (size_t)T * 8);
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_dyn + (size_t)T * 10);
    const DaStore& st = a.src.st;
    const uint32_t tid = threadIdx.x;
    const uint32_t P1 = 1u << st.bits, ebits1 = st.ebits, S = 1u << a.dst.sbits, ebits2 = a.dst.ebits2, emask2 = (1u << ebits2) - 1u;
    const uint16_t* const ent1 = reinterpret_cast<const uint16_t*>(st.ent);
    for (uint32_t p = blockIdx.x; p < P1; p += gridDim.x) {
        __syncthreads();
        if (tid < 8) s_len[tid] = da_region_len(st, P1, p, tid);
        if (tid < TSQ_DM_MAXS) s_cnt[tid] = 0;
        if (FILTER) {
            const uint4* src = reinterpret_cast<const uint4*>(a.bitmap + (((size_t)p << ebits1) >> 5));
            uint4* dst = reinterpret_cast<uint4*>(s_bits);
            for (uint32_t i = tid; i < (1u << ebits1) / 128u; i += NT) dst[i] = src[i];
            if ((1u << ebits1) < 128u)
                for (uint32_t i = tid; i < (1u << ebits1) / 32u; i += NT) s_bits[i] = a.bitmap[(((size_t)p << ebits1) >> 5) + i];
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t u = 0, q = 0;
            for (int r = 0; r < 8; r++) {
                s_ubase[r] = u;
                s_pbase[r] = q;
                u += (s_len[r] + 7u) >> 3;
                q += (s_len[r] + 1u) >> 1;
            }
            s_ubase[8] = u;
            s_pbase[8] = q;
        }
        __syncthreads();
        const uint32_t n_units = s_ubase[8], n_pairs = s_pbase[8];
        // ---- pass A: the rows of every final partition (units of 8 entries; sixteen-bit counters packed into two words per thread)
        {
            uint64_t c0 = 0, c1 = 0;
            for (uint32_t x = tid; x < n_units; x += NT) {
                const uint32_t r = dm_region_of(s_ubase, x), q = x - s_ubase[r], len = s_len[r];
                const uint4 ev = reinterpret_cast<const uint4*>(ent1 + (size_t)(p * 8u + r) * st.cap)[q];
                const uint32_t ew[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    const uint32_t e = (ew[i >> 1] >> ((i & 1u) * 16u)) & 0xffffu;
                    bool hit = q * 8u + i < len;
                    if (FILTER) hit = hit && ((s_bits[e >> 5] >> (e & 31u)) & 1u);
                    const uint32_t sub = e >> ebits2;
                    const uint64_t one = hit ? (1ull << ((sub & 3u) * 16u)) : 0ull;
                    c0 += (sub & 4u) ? 0ull : one;
                    c1 += (sub & 4u) ? one : 0ull;
                }
            }
            // (a thread sees at most cap1 / NT + 8 rows: the fields of a wave's sum stay below 2^16 while cap1 <= 2^15 * NT / 64)
            c0 = wave_sum_u64(c0);
            c1 = wave_sum_u64(c1);
            if ((tid & 63u) == 0) {
#pragma unroll
                for (uint32_t s = 0; s < TSQ_DM_MAXS; s++) {
                    const uint32_t c = (uint32_t)(((s & 4u) ? c1 : c0) >> ((s & 3u) * 16u)) & 0xffffu;
                    if (c) atomicAdd(&s_cnt[s], c);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t run = p * a.dst.cap1;
            for (uint32_t s = 0; s < S; s++) {
                s_run[s] = run;
                a.dst.off[p * S + s] = run;
                a.dst.cnt[p * S + s] = s_cnt[s];
                run += (s_cnt[s] + 7u) & ~7u;
            }
        }
        // ---- pass B: tiles of NT * K pairs, sorted by sub-partition in LDS, written out as runs
        for (uint32_t t0 = 0; t0 < n_pairs; t0 += NT * K) {
            __syncthreads();
            if (tid < TSQ_DM_MAXS) s_tcnt[tid] = 0;
            uint32_t e[2 * K], d[2 * K];
            bool hit[2 * K];
            size_t at[K];
#pragma unroll
            for (int j = 0; j < K; j++) {
                const uint32_t x = t0 + (uint32_t)j * NT + tid;
                const bool act = x < n_pairs;
                const uint32_t r = dm_region_of(s_pbase, act ? x : 0u), q = (act ? x : 0u) - s_pbase[r], len = s_len[r];
                at[j] = (size_t)(p * 8u + r) * st.cap + (size_t)q * 2u;
                const uint32_t ew = *reinterpret_cast<const uint32_t*>(ent1 + at[j]);
                e[2 * j] = ew & 0xffffu;
                e[2 * j + 1] = ew >> 16;
                hit[2 * j] = act && q * 2u < len;
                hit[2 * j + 1] = act && q * 2u + 1u < len;
                if (FILTER) {
                    hit[2 * j] = hit[2 * j] && ((s_bits[e[2 * j] >> 5] >> (e[2 * j] & 31u)) & 1u);
                    hit[2 * j + 1] = hit[2 * j + 1] && ((s_bits[e[2 * j + 1] >> 5] >> (e[2 * j + 1] & 31u)) & 1u);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2 * K; i++) d[i] = hit[i] ? atomicAdd(&s_tcnt[e[i] >> ebits2], 1u) : 0u;
            __syncthreads();
            if (tid == 0) {
                uint32_t run = 0;
                for (uint32_t s = 0; s < S; s++) {
                    s_tpre[s] = run;
                    run += s_tcnt[s];
                }
                s_tpre[S] = run;
            }
            __syncthreads();
            const uint32_t tile_rows = s_tpre[S];
#pragma unroll
            for (int i = 0; i < 2 * K; i++)
                if (hit[i]) {
                    d[i] += s_tpre[e[i] >> ebits2];
                    s_w[d[i]] = (uint16_t)e[i];
                }
            __syncthreads();
            for (uint32_t i = tid; i < tile_rows; i += NT) {
                const uint32_t w = s_w[i], sub = w >> ebits2;
                a.dst.ent[s_run[sub] + (i - s_tpre[sub])] = (uint16_t)(w & emask2);
            }
            for (int v = 0; v < a.n_cols; v++) {
                ulonglong2 c[K];
#pragma unroll
                for (int j = 0; j < K; j++) c[j] = *reinterpret_cast<const ulonglong2*>(a.src.pay[v] + at[j]);
                __syncthreads();  // the previous column's write-out has read the buffer
#pragma unroll
                for (int j = 0; j < K; j++) {
                    if (hit[2 * j]) s_pay[d[2 * j]] = c[j].x;
                    if (hit[2 * j + 1]) s_pay[d[2 * j + 1]] = c[j].y;
                }
                __syncthreads();
                uint64_t* dst = a.dst.pay[v];
                for (uint32_t i = tid; i < tile_rows; i += NT) {
                    const uint32_t sub = (uint32_t)s_w[i] >> ebits2;
                    dst[s_run[sub] + (i - s_tpre[sub])] = s_pay[i];
                }
            }
            if (a.src.nnmask) {
                uint8_t* s_mask = reinterpret_cast<uint8_t*>(s_pay);
                uint32_t m[K];
#pragma unroll
                for (int j = 0; j < K; j++) m[j] = *reinterpret_cast<const uint16_t*>(a.src.nnmask + at[j]);
                __syncthreads();
#pragma unroll
                for (int j = 0; j < K; j++) {
                    if (hit[2 * j]) s_mask[d[2 * j]] = (uint8_t)(m[j] & 0xffu);
                    if (hit[2 * j + 1]) s_mask[d[2 * j + 1]] = (uint8_t)(m[j] >> 8);
                }
                __syncthreads();
                for (uint32_t i = tid; i < tile_rows; i += NT) {
                    const uint32_t sub = (uint32_t)s_w[i] >> ebits2;
                    a.dst.nnmask[s_run[sub] + (i - s_tpre[sub])] = s_mask[i];
                }
            }
            __syncthreads();
            if (tid < TSQ_DM_MAXS) s_run[tid] += s_tcnt[tid];
        }
    }
}

// exclusive scan of the rows per final partition, in place (n = Q + 1 <= 2^14 + 1: the last slot arrives as 0 and leaves as the total).
// One workgroup, the values staged in LDS so that HBM sees two coalesced sweeps (k_scan_blocks walks 17 dependent loads per thread
// at this size: 43 us against ~8 here — 2 % of a probe pass)
static __global__ void __launch_bounds__(1024) k_dm_scan(unsigned long long* v, uint32_t n) {
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ unsigned long long s_w[16];
    unsigned long long* s_v = reinterpret_cast<unsigned long long*>(s_dyn);
    const uint32_t tid = threadIdx.x, per = (n + 1023u) / 1024u, lo = tid * per;
    for (uint32_t i = tid; i < n; i += 1024u) s_v[i] = v[i];
    __syncthreads();
    unsigned long long sum = 0;
    for (uint32_t i = lo; i < lo + per && i < n; i++) sum += s_v[i];
    unsigned long long x = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(x, o, 64);
        if ((tid & 63u) >= (uint32_t)o) x += y;
    }
    if ((tid & 63u) == 63u) s_w[tid >> 6] = x;
    __syncthreads();
    unsigned long long run = x - sum;
    for (uint32_t w = 0; w < (tid >> 6); w++) run += s_w[w];
    for (uint32_t i = lo; i < lo + per && i < n; i++) {
        const unsigned long long c = s_v[i];
        s_v[i] = run;
        run += c;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024u) v[i] = s_v[i];
}

struct DmEmitArgs {
    DmStore bst, pst;                   // final partitions of the build / the probe side (same geometry)
    DaDomain dm;
    uint32_t pbits;                     // log2 of the final partitions
    unsigned long long row0;            // output rows before the partitions' rows (the exception rows come first)
    uint32_t tab_rows;                  // rows of the LDS payload table (>= the build rows of the largest final partition; a multiple of 32)
    // the output batch, as in DaEmitColsArgs: the key columns are not moved, a row's key is kmin + unmix(q : e)
    uint64_t* out_pkey;
    uint8_t* out_pkey_nn;               // (null: the probe key column cannot hold NULLs; a row that reached a partition never has a NULL key: ones)
    uint64_t* out_bkey;
    uint8_t* out_bkey_nn;               // (always set for an outer join)
    int32_t n_probe, n_build;           // travelling probe columns / build columns in the table
    uint64_t* out_probe[TSQ_DA_MAXCOLS];
    uint8_t* out_probe_nn[TSQ_DA_MAXCOLS];   // NOT-NULL byte flags: the kernel writes the flag of every row (null: the column cannot hold NULLs)
    uint64_t* out_build[TSQ_DA_MAXCOLS];
    uint8_t* out_build_nn[TSQ_DA_MAXCOLS];   // (always set for an outer join: the padded rows are NULL)
};
// LDS of k_dm_emit: presence bits | popcount prefix per 32 cells | n_build payload tables | their NOT-NULL bits
__host__ __device__ inline size_t dm_emit_lds(uint32_t ebits2, uint32_t tab_rows, int n_build, bool build_nulls) {
    const size_t words = ((size_t)1 << ebits2) / 32 + 1;
    return words * 8 + (size_t)tab_rows * 8 * (size_t)(n_build > 0 ? n_build : 0) + (build_nulls ? (size_t)n_build * (tab_rows / 8) : 0) + 16;
}

template <int NT, bool OUTER>
__global__ void __launch_bounds__(NT) k_dm_emit(DmEmitArgs a) {
    constexpr int U = 4;                // pairs of probe rows in flight per thread
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ uint32_t s_wsum[NT / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t Q = 1u << a.pbits, ebits2 = a.pst.ebits2, cells = 1u << ebits2, W = cells >= 32u ? cells / 32u : 1u;
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_dyn);
    uint32_t* s_coarse = s_bits + W;
    uint64_t* s_tab = reinterpret_cast<uint64_t*>(s_dyn + (((size_t)W * 8 + 15) & ~(size_t)15));
    uint32_t* s_tnn = reinterpret_cast<uint32_t*>(s_tab + (size_t)a.tab_rows * (size_t)(a.n_build > 0 ? a.n_build : 0));
    const bool bnulls = a.bst.nnmask != nullptr;
    const uint32_t nnw = a.tab_rows / 32u;  // NOT-NULL words per build column
    const uint32_t wpt = (W + NT - 1) / NT;
    for (uint32_t q = blockIdx.x; q < Q; q += gridDim.x) {
        const uint32_t bcnt = (uint32_t)a.bst.cnt[q], boff = a.bst.off[q];
        const unsigned long long pb0 = a.pst.cnt[q];
        const uint32_t pcnt = (uint32_t)(a.pst.cnt[q + 1] - pb0), poff = a.pst.off[q];
        if (pcnt == 0) continue;  // (block-uniform)
        __syncthreads();  // the previous partition's probe rows are done with the table
        for (uint32_t i = tid; i < W; i += NT) s_bits[i] = 0;
        if (bnulls)
            for (uint32_t i = tid; i < nnw * (uint32_t)a.n_build; i += NT) s_tnn[i] = 0xffffffffu;
        __syncthreads();
        // ---- the build rows' presence bits (units of 8 entries)
        {
            const uint4* eb = reinterpret_cast<const uint4*>(a.bst.ent + boff);
            for (uint32_t x = tid; x < (bcnt + 7u) >> 3; x += NT) {
                const uint4 ev = eb[x];
                const uint32_t ew[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    const uint32_t e = (ew[i >> 1] >> ((i & 1u) * 16u)) & 0xffffu;
                    if (x * 8u + i < bcnt) atomicOr(&s_bits[e >> 5], 1u << (e & 31u));
                }
            }
        }
        __syncthreads();
        // ---- popcount prefix: s_coarse[w] = build rows with a smaller cell than w * 32
        {
            uint32_t sum = 0;
            for (uint32_t k = 0; k < wpt; k++) {
                const uint32_t w = tid * wpt + k;
                sum += w < W ? (uint32_t)__popc(s_bits[w]) : 0u;
            }
            uint32_t total;
            uint32_t run = block_excl_scan<NT>(sum, s_wsum, &total);
            for (uint32_t k = 0; k < wpt; k++) {
                const uint32_t w = tid * wpt + k;
                if (w < W) {
                    s_coarse[w] = run;
                    run += (uint32_t)__popc(s_bits[w]);
                }
            }
        }
        __syncthreads();
        auto rank_of = [&](uint32_t e) -> uint32_t { return s_coarse[e >> 5] + (uint32_t)__popc(s_bits[e >> 5] & ((1u << (e & 31u)) - 1u)); };
        // ---- the build rows' cells into word order (pairs of rows: 4 bytes of entries, 16 bytes of a column per lane)
        if (a.n_build > 0) {
            for (uint32_t x = tid; x < (bcnt + 1u) >> 1; x += NT) {
                const uint32_t ew = *reinterpret_cast<const uint32_t*>(a.bst.ent + boff + (size_t)x * 2u);
                const bool v1 = x * 2u + 1u < bcnt;
                const uint32_t r0 = rank_of(ew & 0xffffu), r1 = v1 ? rank_of(ew >> 16) : 0u;
                for (int v = 0; v < a.n_build; v++) {
                    const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(a.bst.pay[v] + boff + (size_t)x * 2u);
                    s_tab[(size_t)v * a.tab_rows + r0] = c.x;
                    if (v1) s_tab[(size_t)v * a.tab_rows + r1] = c.y;
                }
                if (bnulls) {
                    const uint32_t m = *reinterpret_cast<const uint16_t*>(a.bst.nnmask + boff + (size_t)x * 2u);
                    for (int v = 0; v < a.n_build; v++) {
                        if (!((m >> v) & 1u)) atomicAnd(&s_tnn[(uint32_t)v * nnw + (r0 >> 5)], ~(1u << (r0 & 31u)));
                        if (v1 && !((m >> (8 + v)) & 1u)) atomicAnd(&s_tnn[(uint32_t)v * nnw + (r1 >> 5)], ~(1u << (r1 & 31u)));
                    }
                }
            }
        }
        __syncthreads();
        // ---- the probe rows: row i of the partition is output row base + i.  A lane takes the two rows that share a 16-byte unit of
        // the OUTPUT columns (an odd base shifts the pairing by one row), U pairs per step with all their loads issued first
        const unsigned long long base = a.row0 + pb0;
        const uint32_t sh = (uint32_t)base & 1u;
        const uint32_t npairs = (pcnt + sh + 1u) >> 1;
        const uint16_t* pe = a.pst.ent + poff;
        for (uint32_t x0 = tid; x0 < npairs; x0 += NT * U) {
            uint32_t e0[U], e1[U], mm[U];
            bool v0[U], v1[U], h0[U], h1[U];
            int64_t i0[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t x = x0 + (uint32_t)u * NT;
                i0[u] = (int64_t)x * 2 - (int64_t)sh;  // rows i0, i0 + 1 of the partition -> output rows base + i0 (even), + 1
                v0[u] = x < npairs && i0[u] >= 0;
                v1[u] = x < npairs && i0[u] + 1 < (int64_t)pcnt;
                e0[u] = v0[u] ? (uint32_t)pe[i0[u]] : 0u;
                e1[u] = v1[u] ? (uint32_t)pe[i0[u] + 1] : 0u;
                mm[u] = 0xffffu;
                if (a.pst.nnmask) mm[u] = (v0[u] ? (uint32_t)a.pst.nnmask[poff + i0[u]] : 0xffu) | ((v1[u] ? (uint32_t)a.pst.nnmask[poff + i0[u] + 1] : 0xffu) << 8);
            }
            // one column after the other: the U pairs' loads (or table reads) of a column first, then its U stores
            auto put = [&](int u, uint64_t* col, uint64_t c0, uint64_t c1) {
                const unsigned long long g = base + (unsigned long long)i0[u];
                if (v0[u] && v1[u]) {
                    tsq_v2u64 y;
                    y.x = c0;
                    y.y = c1;
                    __builtin_nontemporal_store(y, reinterpret_cast<tsq_v2u64*>(col + g));
                } else if (v0[u]) TSQ_EMIT_STORE(&col[g], c0);
                else if (v1[u]) TSQ_EMIT_STORE(&col[g + 1], c1);
            };
            // NOT-NULL byte flags: EVERY row's flag is written (both rows of a pair with one 2-byte store) — the arrays arrive uninitialised
            // (a hipMemset of 1e8 flag bytes per nullable column cost 0.3 ms: 1.2 of the 3.3 ms of a nullable LEFT OUTER probe pass)
            auto put_null = [&](int u, uint8_t* nn, bool n0, bool n1) {
                const unsigned long long g = base + (unsigned long long)i0[u];
                if (v0[u] && v1[u]) *reinterpret_cast<uint16_t*>(nn + g) = (uint16_t)((n0 ? 0u : 1u) | (n1 ? 0u : 0x100u));
                else if (v0[u]) nn[g] = n0 ? 0 : 1;
                else if (v1[u]) nn[g + 1] = n1 ? 0 : 1;
            };
            for (int v = 0; v < a.n_probe; v++) {
                uint64_t c0[U], c1[U];
                const uint64_t* src = a.pst.pay[v] + poff;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    c0[u] = v0[u] ? src[i0[u]] : 0ull;
                    c1[u] = v1[u] ? src[i0[u] + 1] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    put(u, a.out_probe[v], c0[u], c1[u]);
                    if (a.out_probe_nn[v]) put_null(u, a.out_probe_nn[v], !((mm[u] >> v) & 1u), !((mm[u] >> (8 + v)) & 1u));
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                h0[u] = v0[u];
                h1[u] = v1[u];
                if (OUTER) {  // (an inner join's rows all have a build row: level 2 dropped the others)
                    h0[u] = h0[u] && ((s_bits[e0[u] >> 5] >> (e0[u] & 31u)) & 1u);
                    h1[u] = h1[u] && ((s_bits[e1[u] >> 5] >> (e1[u] & 31u)) & 1u);
                }
            }
            if (a.out_pkey) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint64_t k0 = a.dm.kmin + (uint64_t)tsq_da_unmix((q << ebits2) | e0[u], a.dm.s, a.dm.mask);
                    const uint64_t k1 = a.dm.kmin + (uint64_t)tsq_da_unmix((q << ebits2) | e1[u], a.dm.s, a.dm.mask);
                    put(u, a.out_pkey, k0, k1);
                    put(u, a.out_bkey, (!OUTER || h0[u]) ? k0 : 0ull, (!OUTER || h1[u]) ? k1 : 0ull);
                    if (a.out_pkey_nn) put_null(u, a.out_pkey_nn, false, false);
                    if (a.out_bkey_nn) put_null(u, a.out_bkey_nn, OUTER && !h0[u], OUTER && !h1[u]);
                }
            }
            if (a.n_build > 0 || OUTER) {
                uint32_t r0[U], r1[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    r0[u] = h0[u] ? rank_of(e0[u]) : 0u;
                    r1[u] = h1[u] ? rank_of(e1[u]) : 0u;
                }
                for (int v = 0; v < a.n_build; v++) {
                    const uint64_t* tab = s_tab + (size_t)v * a.tab_rows;
                    uint64_t c0[U], c1[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        c0[u] = h0[u] ? tab[r0[u]] : 0ull;
                        c1[u] = h1[u] ? tab[r1[u]] : 0ull;
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        put(u, a.out_build[v], c0[u], c1[u]);
                        if (a.out_build_nn[v]) {
                            const uint32_t* tn = s_tnn + (uint32_t)v * nnw;
                            put_null(u, a.out_build_nn[v], !h0[u] || (bnulls && !((tn[r0[u] >> 5] >> (r0[u] & 31u)) & 1u)),
                                     !h1[u] || (bnulls && !((tn[r1[u] >> 5] >> (r1[u] & 31u)) & 1u)));
                        }
                    }
                }
            }
        }
    }
}

#endif
