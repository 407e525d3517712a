// tsq_aggfast.h — LDS pre-aggregation for HashAggExec (device code, included by tsq_agg.hip).
//
// Why: the group-table upsert of tsq_agg.hip pays one CAS probe + one or two device-scope atomics per
// input row in HBM; random device atomics run at ~20-27 G/s on MI355X (profiles/r01_probe_ubench.txt),
// i.e. SELECT k, SUM(v), COUNT(*) GROUP BY k over 1e9 rows / 1e6 groups took 94 ms = 2 % of the HBM
// roofline.  LDS atomics are two orders of magnitude faster, so rows are first aggregated inside LDS
// and only the per-workgroup partial groups touch the table in HBM.  This is the reference's own
// two-phase shape — partial workers -> shuffle by group key -> final workers
// (executor/aggregate.go:96-133, 332-356, 424-457) — with LDS as the partial workers' map:
//   L (few groups, <= S/2): every workgroup streams its share of the input columns into a private LDS
//     table and emits its partial groups (PartialResult rows);
//   H (up to ~2 M groups): rows are radix partitioned by the group-key hash (tsq_radix.h, payload =
//     the argument cells), then ONE workgroup aggregates one partition whose groups fit its LDS table;
//     the partitioned store holds TABLE WORDS w = mix64(key word) (the HASHED store of the join): the one
//     hash a row pays selects its partition (top bits) and its LDS slot (low bits), and w identifies the
//     group as well as the key does (mix64 is a bijection; the key comes back with tsq_unmix64 when a
//     partial group is emitted);
//   the partial groups are merged into the HBM group table by k_agg_merge (= the final workers'
//     consumeIntermData + MergePartialResult, aggfuncs/*.go).
// Rows the LDS stage cannot take (NULL key or NULL argument cell, table full, sentinel key) are handed
// to the row-at-a-time upsert, so every aggregate keeps its exact NULL protocol.
#ifndef TSQ_AGGFAST_H
#define TSQ_AGGFAST_H

#include "tsq_radix.h"

#define TSQ_AF_MAXW 5     /* 64-bit LDS words per group besides the key */
#define TSQ_AF_NT 1024    /* one workgroup of 16 waves per CU */
#define TSQ_AF_EMPTY 0x8080808080808080ULL

struct AfAgg {
    int32_t func, type;  // TSQ_AGG_*, argument type
    int32_t v;           // payload cell index, -1 = no argument cell (COUNT(*), FIRSTROW(key))
    int32_t w;           // first LDS word, -1 = none
};
struct AfPlan {
    int32_t n_aggs;
    AfAgg f[TSQ_MAX_AGGS];  // same order as the handle's aggregates
    int32_t W, V;
    int32_t key_col, key_type;
    int32_t vcol[TSQ_RADIX_MAXV], vtype[TSQ_RADIX_MAXV];
    unsigned long long init[TSQ_AF_MAXW];  // initial value of every word (MIN starts at all ones)
    // what a row does to LDS word k, decoded once per kernel instead of walking f[] for every row: bits 0-2 = AF_W_* kind,
    // bit 3 = which argument cell, bits 4-5 = the cell's TSQ_* type (tsq_agg_create fills it from f[])
    uint32_t wdesc[TSQ_AF_MAXW];
};
enum { AF_W_ADD1 = 0, AF_W_ADD_REAL = 1, AF_W_ADD_LO32 = 2, AF_W_ADD_HI32 = 3, AF_W_MAX = 4, AF_W_MIN = 5 };
inline uint32_t af_wdesc(int kind, int v, int type) { return (uint32_t)kind | ((uint32_t)(v > 0 ? 1 : 0) << 3) | ((uint32_t)type << 4); }
// fills wdesc[] from f[] (host side, after W / V / f[].w / f[].v are known)
inline void af_fill_wdesc(AfPlan& p) {
    for (int k = 0; k < TSQ_AF_MAXW; k++) p.wdesc[k] = 0;
    for (int i = 0; i < p.n_aggs; i++) {
        const AfAgg& f = p.f[i];
        if (f.w < 0) continue;
        const bool real = f.type == TSQ_F32 || f.type == TSQ_F64;
        switch (f.func) {
            case TSQ_AGG_COUNT: p.wdesc[f.w] = af_wdesc(AF_W_ADD1, 0, 0); break;
            case TSQ_AGG_SUM:
            case TSQ_AGG_AVG: {
                int w = f.w;
                if (real) p.wdesc[w++] = af_wdesc(AF_W_ADD_REAL, f.v, f.type);
                else {
                    p.wdesc[w++] = af_wdesc(AF_W_ADD_LO32, f.v, f.type);
                    p.wdesc[w++] = af_wdesc(AF_W_ADD_HI32, f.v, f.type);
                }
                if (f.func == TSQ_AGG_AVG) p.wdesc[w] = af_wdesc(AF_W_ADD1, 0, 0);
                break;
            }
            case TSQ_AGG_MAX: p.wdesc[f.w] = af_wdesc(AF_W_MAX, f.v, f.type); break;
            case TSQ_AGG_MIN: p.wdesc[f.w] = af_wdesc(AF_W_MIN, f.v, f.type); break;
        }
    }
}
// slot of a key word inside an LDS table of 2^log2s slots (L mode: the rows come straight from the input columns).  Not mix64:
// its two 64-bit multiplies are eight quarter-rate v_mul_lo/hi_u32; a 32-bit multiplicative hash (top bits) is enough for a table
// that is at most half full.
__device__ __forceinline__ uint32_t af_slot_hash(uint64_t tag, uint32_t log2s) {
    const uint32_t lo = (uint32_t)tag, hi = (uint32_t)(tag >> 32);
    return ((lo ^ (hi * 0x85EBCA6Bu)) * 0x9E3779B1u) >> (32u - log2s);
}
struct AfPartials {  // partial groups in HBM, structure of arrays
    unsigned long long* key;
    unsigned long long* w[TSQ_AF_MAXW];
    uint32_t* count;  // records appended (may exceed cap: then the batch is redone row by row)
    uint32_t cap;
};

__device__ __forceinline__ uint64_t af_ord_image(uint64_t cell, int32_t type) {  // order preserving image for min/max on uint64
    switch (type) {
        case TSQ_I64: return cell ^ 0x8000000000000000ULL;
        case TSQ_U64: return cell;
        default: {
            const double f = type == TSQ_F32 ? (double)tsq_bits_f32((uint32_t)cell) : tsq_bits_f64(cell);
            const uint64_t u = tsq_f64_bits(f);
            return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
        }
    }
}
__device__ __forceinline__ double af_real(uint64_t cell, int32_t type) {
    return type == TSQ_F32 ? (double)tsq_bits_f32((uint32_t)cell) : tsq_bits_f64(cell);
}
__device__ __forceinline__ bool af_is_real(int32_t t) { return t == TSQ_F32 || t == TSQ_F64; }

// the words of a partial group that consists of ONE row
__device__ __forceinline__ void af_row_words(const AfPlan& pl, const uint64_t* cells, unsigned long long* w) {
    for (int i = 0; i < pl.W; i++) w[i] = pl.init[i];
    for (int i = 0; i < pl.n_aggs; i++) {
        const AfAgg f = pl.f[i];
        if (f.w < 0) continue;
        switch (f.func) {
            case TSQ_AGG_COUNT: w[f.w] = 1; break;
            case TSQ_AGG_SUM:
            case TSQ_AGG_AVG:
                if (af_is_real(f.type)) {
                    w[f.w] = tsq_f64_bits(af_real(cells[f.v], f.type));
                    if (f.func == TSQ_AGG_AVG) w[f.w + 1] = 1;
                } else {
                    w[f.w] = cells[f.v];
                    w[f.w + 1] = ((int64_t)cells[f.v] < 0) ? ~0ull : 0ull;
                    if (f.func == TSQ_AGG_AVG) w[f.w + 2] = 1;
                }
                break;
            case TSQ_AGG_MAX:
            case TSQ_AGG_MIN: w[f.w] = af_ord_image(cells[f.v], f.type); break;
        }
    }
}

struct AfLdsArgs {
    AfPlan plan;
    AfPartials out;
    // MODE 0: straight from the input columns
    tsq_colset in;
    int64_t nrows;
    uint32_t* exc_rows;   // rows with a NULL key / NULL argument cell
    uint32_t* exc_count;
    // MODE 1: one workgroup per partition of a partitioned store; MODE 2: its overflow list
    RadixStore st;
};

// K7a — LDS pre-aggregation (updatePartialResult of one partial worker, aggregate.go:332-350, into LDS).
template <int MODE, int W>
__global__ void __launch_bounds__(TSQ_AF_NT) k_agg_lds(AfLdsArgs a) {
    constexpr uint32_t S = W <= 3 ? 4096u : 2048u, LOG2S = W <= 3 ? 12u : 11u;
    constexpr bool HASHED_TAGS = MODE != 0;  // MODE 1 / 2 read a HASHED store: tags are table words
    uint32_t wd[W];  // wave uniform: what a row adds to each word
#pragma unroll
    for (int k = 0; k < W; k++) wd[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.plan.wdesc[k]);
    __shared__ unsigned long long s_key[S];
    __shared__ unsigned long long s_w[W][S];
    __shared__ uint32_t s_used;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < S; i += TSQ_AF_NT) {
        s_key[i] = TSQ_AF_EMPTY;
#pragma unroll
        for (int k = 0; k < W; k++) s_w[k][i] = a.plan.init[k];
    }
    if (tid == 0) s_used = 0;
    __syncthreads();

    auto spill = [&](uint64_t tag, const uint64_t* cells) {  // a one-row partial group straight to HBM
        unsigned long long w[TSQ_AF_MAXW];
        af_row_words(a.plan, cells, w);
        const uint32_t o = __hip_atomic_fetch_add(a.out.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o < a.out.cap) {
            a.out.key[o] = HASHED_TAGS ? tsq_unmix64(tag) : tag;
#pragma unroll
            for (int k = 0; k < W; k++) a.out.w[k][o] = w[k];
        }
    };
    auto apply = [&](uint64_t tag, uint64_t c0, uint64_t c1) {  // cells by value: a dynamically indexed local array would live in scratch
        if (tag == TSQ_AF_EMPTY) {  // the sentinel key lives in a special slot of the HBM table
            const uint64_t cells[TSQ_RADIX_MAXV] = {c0, c1};
            spill(tag, cells);
            return;
        }
        uint32_t slot = HASHED_TAGS ? ((uint32_t)tag & (S - 1)) : af_slot_hash(tag, LOG2S);
        // (testing the home slot once before the loop — "the group is usually already there" — was measured and dropped: 0.66 ->
        // 0.71 ms per 1e8 rows; nearly every wave has a lane that needs the loop anyway, and the extra LDS read is not free)
        bool found = false;
        for (int probe = 0; probe < 64 && !found; probe++) {
            unsigned long long cur = s_key[slot];
            if (cur == TSQ_AF_EMPTY) {
                if (s_used >= S - S / 8) break;  // keep the table probe-able: the row is spilled instead
                cur = atomicCAS(&s_key[slot], (unsigned long long)TSQ_AF_EMPTY, (unsigned long long)tag);
                if (cur == TSQ_AF_EMPTY) {
                    atomicAdd(&s_used, 1u);
                    found = true;
                    break;
                }
            }
            if (cur == tag) found = true;
            else slot = (slot + 1) & (S - 1);
        }
        if (!found) {
            const uint64_t cells[TSQ_RADIX_MAXV] = {c0, c1};
            spill(tag, cells);
            return;
        }
        // one non-returning LDS atomic per word.  int64 sums: the low and the (signed) high 32-bit halves are summed separately in
        // 64 bits — neither can overflow in < 2^31 rows — and recombined into the 128-bit (lo, hi) pair when the group is emitted
        // (func_sum.go:133-137 is exact in 128 bits)
#pragma unroll
        for (int k = 0; k < W; k++) {
            const uint32_t d = wd[k];
            const uint64_t cell = (d & 8u) ? c1 : c0;
            const int32_t type = (int32_t)(d >> 4);
            switch (d & 7u) {
                case AF_W_ADD1: atomicAdd(&s_w[k][slot], 1ull); break;
                case AF_W_ADD_REAL: atomicAdd(reinterpret_cast<double*>(&s_w[k][slot]), af_real(cell, type)); break;
                case AF_W_ADD_LO32: atomicAdd(&s_w[k][slot], (unsigned long long)(cell & 0xffffffffull)); break;
                case AF_W_ADD_HI32: atomicAdd(&s_w[k][slot], (unsigned long long)((long long)cell >> 32)); break;
                case AF_W_MAX: atomicMax(&s_w[k][slot], (unsigned long long)af_ord_image(cell, type)); break;
                default: atomicMin(&s_w[k][slot], (unsigned long long)af_ord_image(cell, type)); break;
            }
        }
    };

    constexpr int U = 4;  // rows in flight per lane: the loop is bound by HBM latency at 16 waves per CU otherwise
    if (MODE == 0) {
        const int kc = a.plan.key_col;
        RadixSrc ks;
        ks.data = a.in.data[kc];
        ks.type = a.in.type[kc];
        ks.key_kind = 1;
        const int64_t stride = (int64_t)gridDim.x * TSQ_AF_NT * U;
        for (int64_t r0 = (int64_t)blockIdx.x * TSQ_AF_NT * U + tid; r0 < a.nrows; r0 += stride) {
            uint64_t tag[U], cells[U][TSQ_RADIX_MAXV];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t r = r0 + (int64_t)u * TSQ_AF_NT;
                ok[u] = false;
                tag[u] = 0;
                cells[u][0] = cells[u][1] = 0;
                if (r < a.nrows) {
                    bool isnull = tsq_is_null(a.in.nulls[kc], r);
                    for (int v = 0; v < a.plan.V; v++) isnull |= tsq_is_null(a.in.nulls[a.plan.vcol[v]], r);
                    if (isnull) {
                        const uint32_t e = __hip_atomic_fetch_add(a.exc_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a.exc_rows[e] = (uint32_t)r;
                    } else {
                        ok[u] = true;
                        tag[u] = radix_src_key(ks, r);
#pragma unroll
                        for (int v = 0; v < TSQ_RADIX_MAXV; v++) {
                            if (v < a.plan.V) {
                                const int c = a.plan.vcol[v];
                                cells[u][v] = a.in.type[c] == TSQ_F32 ? (uint64_t)((const uint32_t*)a.in.data[c])[r] : ((const uint64_t*)a.in.data[c])[r];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (ok[u]) apply(tag[u], cells[u][0], cells[u][1]);
        }
    } else if (MODE == 1) {
        const uint32_t p = blockIdx.x;
        for (uint32_t r = 0; r < a.st.R; r++) {
            const uint32_t region = p * a.st.R + r;
            const uint32_t len = radix_region_len(a.st, 1u << a.st.bits, p, r);
            const size_t base = (size_t)region * a.st.cap;
            for (uint32_t i0 = tid; i0 < len; i0 += TSQ_AF_NT * U) {
                uint64_t tag[U], cells[U][TSQ_RADIX_MAXV];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t i = i0 + (uint32_t)u * TSQ_AF_NT;
                    tag[u] = 0;
                    cells[u][0] = cells[u][1] = 0;
                    if (i < len) {
                        tag[u] = a.st.keys[base + i];
#pragma unroll
                        for (int v = 0; v < TSQ_RADIX_MAXV; v++)
                            if (v < a.plan.V) cells[u][v] = a.st.pay[v][base + i];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (i0 + (uint32_t)u * TSQ_AF_NT < len) apply(tag[u], cells[u][0], cells[u][1]);
            }
        }
    } else {
        uint32_t n = *a.st.ovf_count;
        n = n < a.st.ovf_cap ? n : a.st.ovf_cap;
        for (uint32_t i = blockIdx.x * TSQ_AF_NT + tid; i < n; i += gridDim.x * TSQ_AF_NT) {
            uint64_t cells[TSQ_RADIX_MAXV] = {0, 0};
#pragma unroll
            for (int v = 0; v < TSQ_RADIX_MAXV; v++)
                if (v < a.plan.V) cells[v] = a.st.ovf_pay[v][i];
            apply(a.st.ovf_keys[i], cells[0], cells[1]);
        }
    }
    __syncthreads();
    // emit the partial groups of this workgroup: ONE returning atomic per workgroup claims s_used records (a
    // same-address atomic per wave costs ~11 ns each chip-wide: 65 K of them were 0.6 ms of a 1 ms kernel),
    // then a block scan per 1024-slot pass places every occupied slot
    __shared__ uint32_t s_base, s_wsum[TSQ_AF_NT / 64];
    if (tid == 0) s_base = s_used ? __hip_atomic_fetch_add(a.out.count, s_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    __syncthreads();
    uint32_t running = s_base;
    for (uint32_t i0 = 0; i0 < S; i0 += TSQ_AF_NT) {
        const uint32_t i = i0 + tid;
        const unsigned long long key = s_key[i];
        const bool occ = key != TSQ_AF_EMPTY;
        uint32_t total;
        const uint32_t ex = block_excl_scan<TSQ_AF_NT>(occ ? 1u : 0u, s_wsum, &total);
        const uint32_t o = running + ex;
        running += total;
        __syncthreads();  // s_wsum is reused by the next pass
        if (occ) {
            if (o < a.out.cap) {
                a.out.key[o] = HASHED_TAGS ? tsq_unmix64(key) : key;
                unsigned long long w[W];
#pragma unroll
                for (int k = 0; k < W; k++) w[k] = s_w[k][i];
                for (int q = 0; q < a.plan.n_aggs; q++) {  // split int64 sums -> (lo, hi) of the 128-bit value
                    const AfAgg f = a.plan.f[q];
                    if (f.w < 0 || (f.func != TSQ_AGG_SUM && f.func != TSQ_AGG_AVG) || af_is_real(f.type)) continue;
#pragma unroll
                    for (int k = 0; k + 1 < W; k++) {
                        if (k == f.w) {
                            const unsigned long long lo32 = w[k], hi32 = w[k + 1];
                            const unsigned long long lo = (hi32 << 32) + lo32;
                            w[k] = lo;
                            w[k + 1] = (unsigned long long)((long long)hi32 >> 32) + (lo < lo32 ? 1ull : 0ull);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < W; k++) a.out.w[k][o] = w[k];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K7p — PARTITIONED GROUPS (round 6): a GROUP BY with about as many groups as rows (SELECT .. GROUP BY l_orderkey, .. of Q3: 1.3e7
// groups of 3e7 rows; BenchmarkAggGroupByNDV at NDV = 1e7, executor/benchmark_test.go:296-305).  H mode gives such an input up — its
// partial groups do not shrink the batch, and every one of them is a random upsert into the table in HBM (k_agg_merge: 7e9 a second,
// the row path's speed) — and the row path took 136 ms for 1e7 rows of 6.3e6 groups.  Here the operator's group table IS a set of
// LDS-sized sub-tables: 2^bits partitions (the radix partition of H mode) x 2^sbits sub-tables of S slots each, kept in HBM between the
// batches.  A batch is partitioned as in H mode; ONE workgroup per partition loads sub-table t into LDS, lets the partition's rows whose
// word belongs to t find or insert their group there (the same CAS + one LDS atomic per word as k_agg_lds), stores the sub-table back,
// and goes on to t + 1 (the partition's rows are read 2^sbits times: from the L2).  A word that finds its sub-table full is SPILLED as a
// one-row partial group (k_agg_merge puts it into the table in HBM): a sub-table never shrinks, so a key lives either in its sub-table
// from some batch on or in the HBM table for good — one home per group.  At the end the groups come straight out of the sub-tables
// (k_dense_finalize, pg form) when the HBM table stayed empty; otherwise they are emitted as partial groups and merged (k_pg_emit).
// Replaces (reference): getGroupKey + the partial-result map + shuffle + final map (executor/aggregate.go:332-356, 424-457).
struct AfPgArgs {
    AfPlan plan;
    AfPartials out;                       // spilled one-row partial groups
    RadixStore st;
    unsigned long long* pg_key;           // [P << sbits][S] table words (TSQ_AF_EMPTY: free)
    unsigned long long* pg_w[TSQ_AF_MAXW];
    uint32_t* pg_used;                    // [P << sbits] occupied slots
    uint32_t sbits;
};
template <int W>
__global__ void __launch_bounds__(TSQ_AF_NT) k_agg_pg(AfPgArgs a) {
    constexpr uint32_t S = W <= 3 ? 4096u : 2048u;
    uint32_t wd[W];
#pragma unroll
    for (int k = 0; k < W; k++) wd[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.plan.wdesc[k]);
    __shared__ unsigned long long s_key[S];
    __shared__ unsigned long long s_w[W][S];
    __shared__ uint32_t s_used;
    const uint32_t tid = threadIdx.x;
    const uint32_t P = 1u << a.st.bits, NS = 1u << a.sbits, sshift = 64u - a.st.bits - a.sbits;
    auto spill = [&](uint64_t tag, uint64_t c0, uint64_t c1) {
        const uint64_t cells[TSQ_RADIX_MAXV] = {c0, c1};
        unsigned long long w[TSQ_AF_MAXW];
        af_row_words(a.plan, cells, w);
        const uint32_t o = __hip_atomic_fetch_add(a.out.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o < a.out.cap) {
            a.out.key[o] = tsq_unmix64(tag);
#pragma unroll
            for (int k = 0; k < W; k++) a.out.w[k][o] = w[k];
        }
    };
    auto apply = [&](uint64_t tag, uint64_t c0, uint64_t c1) {
        if (tag == TSQ_AF_EMPTY) { spill(tag, c0, c1); return; }  // (the sentinel word lives in a slot of its own of the HBM table)
        uint32_t slot = (uint32_t)tag & (S - 1);
        bool found = false;
        for (int probe = 0; probe < 64 && !found; probe++) {
            unsigned long long cur = s_key[slot];
            if (cur == TSQ_AF_EMPTY) {
                if (s_used >= S - S / 8) break;  // full (for good): the row is spilled
                cur = atomicCAS(&s_key[slot], (unsigned long long)TSQ_AF_EMPTY, (unsigned long long)tag);
                if (cur == TSQ_AF_EMPTY) {
                    atomicAdd(&s_used, 1u);
                    found = true;
                    break;
                }
            }
            if (cur == tag) found = true;
            else slot = (slot + 1) & (S - 1);
        }
        if (!found) { spill(tag, c0, c1); return; }
#pragma unroll
        for (int k = 0; k < W; k++) {
            const uint32_t d = wd[k];
            const uint64_t cell = (d & 8u) ? c1 : c0;
            const int32_t type = (int32_t)(d >> 4);
            switch (d & 7u) {
                case AF_W_ADD1: atomicAdd(&s_w[k][slot], 1ull); break;
                case AF_W_ADD_REAL: atomicAdd(reinterpret_cast<double*>(&s_w[k][slot]), af_real(cell, type)); break;
                case AF_W_ADD_LO32: atomicAdd(&s_w[k][slot], (unsigned long long)(cell & 0xffffffffull)); break;
                case AF_W_ADD_HI32: atomicAdd(&s_w[k][slot], (unsigned long long)((long long)cell >> 32)); break;
                case AF_W_MAX: atomicMax(&s_w[k][slot], (unsigned long long)af_ord_image(cell, type)); break;
                default: atomicMin(&s_w[k][slot], (unsigned long long)af_ord_image(cell, type)); break;
            }
        }
    };
    constexpr int U = 4;
    // a work item = one sub-table (partition p, part t): two workgroups per CU (W <= 1) load / apply / store at different times, and the
    // NS parts of a partition run next to each other — its rows come from HBM once and from the L2 for the other parts
    for (uint32_t item = blockIdx.x; item < P * NS; item += gridDim.x) {
        const uint32_t p = item / NS, t = item % NS;
        // the partition's rows = its (at most 8) regions one after the other: ONE loop over all of them — a loop per region was eight
        // dependent round trips to memory for the ~400 rows each holds when a batch brings about one row per slot (Q3: 0.98 ms a pass)
        uint32_t off[9];
        off[0] = 0;
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) off[r + 1] = off[r] + (r < a.st.R ? radix_region_len(a.st, P, p, r) : 0u);
        const uint32_t rows_p = off[8];
        if (rows_p == 0) continue;  // (block-uniform)
        {
            const size_t sub = (size_t)p * NS + t, tb = sub * S;
            __syncthreads();  // the previous sub-table has been stored
            for (uint32_t i = tid; i < S; i += TSQ_AF_NT) {
                s_key[i] = a.pg_key[tb + i];
#pragma unroll
                for (int k = 0; k < W; k++) s_w[k][i] = a.pg_w[k][tb + i];
            }
            if (tid == 0) s_used = a.pg_used[sub];
            __syncthreads();
            const size_t pbase = (size_t)p * a.st.R * a.st.cap;
            for (uint32_t i0 = tid; i0 < rows_p; i0 += TSQ_AF_NT * U) {
                uint64_t tag[U], cells[U][TSQ_RADIX_MAXV];
                bool mine[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t i = i0 + (uint32_t)u * TSQ_AF_NT;
                    tag[u] = 0;
                    cells[u][0] = cells[u][1] = 0;
                    mine[u] = false;
                    if (i < rows_p) {
                        uint32_t r = 0;
#pragma unroll
                        for (uint32_t q = 1; q < 8; q++) r += i >= off[q] ? 1u : 0u;
                        uint32_t o = off[0];
#pragma unroll
                        for (uint32_t q = 1; q < 8; q++) o = r == q ? off[q] : o;
                        const size_t at = pbase + (size_t)r * a.st.cap + (i - o);
                        tag[u] = a.st.keys[at];
                        mine[u] = a.sbits == 0 || (uint32_t)((tag[u] >> sshift) & (NS - 1)) == t;
#pragma unroll
                        for (int v = 0; v < TSQ_RADIX_MAXV; v++)
                            if (v < a.plan.V && mine[u]) cells[u][v] = a.st.pay[v][at];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (mine[u]) apply(tag[u], cells[u][0], cells[u][1]);
            }
            __syncthreads();
            for (uint32_t i = tid; i < S; i += TSQ_AF_NT) {
                a.pg_key[tb + i] = s_key[i];
#pragma unroll
                for (int k = 0; k < W; k++) a.pg_w[k][tb + i] = s_w[k][i];
            }
            if (tid == 0) a.pg_used[sub] = s_used;
        }
    }
}
// the overflow list of the partition pass (skewed keys: runs that missed their region) and anything else that must not wait: every row
// as a one-row partial group for k_agg_merge.  (Such a key may live in a sub-table as well: the end then takes the merging way.)
struct AfPgOvfArgs {
    AfPlan plan;
    AfPartials out;
    RadixStore st;
};
static __global__ void __launch_bounds__(256) k_pg_ovf(AfPgOvfArgs a) {
    uint32_t n = *a.st.ovf_count;
    n = n < a.st.ovf_cap ? n : a.st.ovf_cap;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint64_t cells[TSQ_RADIX_MAXV] = {0, 0};
#pragma unroll
        for (int v = 0; v < TSQ_RADIX_MAXV; v++)
            if (v < a.plan.V) cells[v] = a.st.ovf_pay[v][i];
        unsigned long long w[TSQ_AF_MAXW];
        af_row_words(a.plan, cells, w);
        const uint32_t o = __hip_atomic_fetch_add(a.out.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o < a.out.cap) {
            a.out.key[o] = tsq_unmix64(a.st.ovf_keys[i]);
            for (int k = 0; k < a.plan.W; k++) a.out.w[k][o] = w[k];
        }
    }
}
// the groups of slots [lo, lo + n) of the sub-tables as partial groups (int64 sums: (low halves, high halves) -> the 128-bit (lo, hi)
// pair, as k_agg_lds emits them): the merging way out, taken when the table in HBM holds groups as well
struct AfPgEmitArgs {
    AfPlan plan;
    AfPartials out;
    const unsigned long long* pg_key;
    const unsigned long long* pg_w[TSQ_AF_MAXW];
    uint64_t lo, n;
};
static __global__ void __launch_bounds__(256) k_pg_emit(AfPgEmitArgs a) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (uint64_t)gridDim.x * 256) {
        const unsigned long long key = a.pg_key[a.lo + i];
        if (key == TSQ_AF_EMPTY) continue;
        unsigned long long w[TSQ_AF_MAXW];
        for (int k = 0; k < a.plan.W; k++) w[k] = a.pg_w[k][a.lo + i];
        for (int q = 0; q < a.plan.n_aggs; q++) {
            const AfAgg f = a.plan.f[q];
            if (f.w < 0 || (f.func != TSQ_AGG_SUM && f.func != TSQ_AGG_AVG) || af_is_real(f.type)) continue;
            const unsigned long long lo32 = w[f.w], hi32 = w[f.w + 1];
            const unsigned long long lo = (hi32 << 32) + lo32;
            w[f.w] = lo;
            w[f.w + 1] = (unsigned long long)((long long)hi32 >> 32) + (lo < lo32 ? 1ull : 0ull);
        }
        const uint32_t o = __hip_atomic_fetch_add(a.out.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o < a.out.cap) {
            a.out.key[o] = tsq_unmix64(key);
            for (int k = 0; k < a.plan.W; k++) a.out.w[k][o] = w[k];
        }
    }
}
// how many distinct keys does the input hold?  8192 keys spread over the batch go into one LDS set; d of them were there already:
// N ~ n^2 / (2 d) (the birthday bound, good to a factor of two up to ~1e9 keys) — enough to tell "about as many groups as rows" from
// what H mode takes, without putting a prefix of the batch into the table (which would then hold groups the sub-tables hold too)
struct AfPgSampleArgs {
    RadixSrc src;
    uint32_t* out;  // [0] keys sampled (not NULL), [1] duplicates among them
};
static __global__ void __launch_bounds__(1024) k_pg_sample(AfPgSampleArgs a) {
    constexpr uint32_t S = 16384, N = 8192;
    __shared__ unsigned long long s_set[S];
    __shared__ uint32_t s_n, s_dup;
    for (uint32_t i = threadIdx.x; i < S; i += 1024) s_set[i] = TSQ_AF_EMPTY;
    if (threadIdx.x == 0) s_n = s_dup = 0;
    __syncthreads();
    const int64_t stride = a.src.nrows / N > 0 ? a.src.nrows / N : 1;
    for (uint32_t i = threadIdx.x; i < N; i += 1024) {
        const int64_t r = (int64_t)i * stride;
        if (r >= a.src.nrows || tsq_is_null(a.src.nulls, r)) continue;
        const unsigned long long w = tsq_mix64(radix_src_key(a.src, r));
        if (w == TSQ_AF_EMPTY) continue;
        atomicAdd(&s_n, 1u);
        uint32_t slot = (uint32_t)w & (S - 1);
        for (;;) {
            unsigned long long cur = s_set[slot];
            if (cur == TSQ_AF_EMPTY) cur = atomicCAS(&s_set[slot], (unsigned long long)TSQ_AF_EMPTY, w);
            if (cur == TSQ_AF_EMPTY) break;
            if (cur == w) { atomicAdd(&s_dup, 1u); break; }
            slot = (slot + 1) & (S - 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { a.out[0] = s_n; a.out[1] = s_dup; }
}

#endif
