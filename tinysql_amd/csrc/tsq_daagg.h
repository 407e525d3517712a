// tsq_daagg.h — PACKED-KEY pre-aggregation for HashAggExec (device code, included by tsq_agg.hip).
//
// The H mode of tsq_aggfast.h moves 48 B per 16-byte row (partition: 16 B read + 16 B written, LDS pass: 16 B read) and its LDS
// pass claims group slots with a compare-and-swap walk (148 scalar instructions per 64 rows of exec-mask bookkeeping).  When the
// group key is ONE integer column whose values span few bits — the first large batch says so: [kmin, kmin + 2^b) with b <= 23 —
//   d = key - kmin, u = bijective mix of d on b bits (tsq_dajoin.h), partition = top bits of u, ENTRY e = the low bits,
// the partitioned store carries 2-byte entries next to the argument cells (10 B instead of 16 B per row each way) and the LDS
// pass is DIRECT ADDRESSED: the accumulators of entry e are words [k][e] of the workgroup's LDS — no tag, no claim, no walk,
// one non-returning LDS atomic per word and row.  u is a bijection of the key, so a cell IS a group (util/codec/codec.go:713-746:
// equal key cells, equal group); the key of a partial group comes back as kmin + unmix(u).
// Rows outside the range (later batches may bring new keys), NULL keys and NULL argument cells go to the exception list and
// take the row-at-a-time upsert, exactly like the rows tsq_aggfast.h cannot take.
// Round 4: (1) NARROW ARGUMENT CELLS — one integer argument column whose sampled values are all in [0, 2^16) or [0, 2^32) travels
// as 2- or 4-byte cells (DaAggStore.paybytes): 4..6 B per row each way instead of 10; a value that does not fit is an exception.
// (2) DENSE PARTIAL STATE — one key column: the LDS tables are folded into accumulators [word][u] in HBM (u = the b-bit packed
// word: 2^b x W x 8 B, 24 MB for C3) instead of being appended as partial groups and merged into the hash table after every batch;
// the touched cells become partial groups ONCE, when the operator finishes (k_daagg_dense_emit), and take the usual merge then.
//
// Replaces (reference): HashAggPartialWorker.updatePartialResult + getGroupKey + getPartialResult
// (executor/aggregate.go:332-410) with LDS as the partial worker's map; the partial groups are merged by k_agg_merge
// (consumeIntermData, aggregate.go:424-427).  Algorithmic bytes: 16 B per row.
#ifndef TSQ_DAAGG_H
#define TSQ_DAAGG_H

#include "tsq_aggfast.h"
#include "tsq_dajoin.h"
#include "tsq_wavescan.h"

#define TSQ_DAAGG_MAX_BITS 23
#define TSQ_DAAGG_PACK_SHIFT 40  /* SIG 3 (daagg_apply): count << 40 | sum of 16-bit values */

struct DaAggStore {
    uint16_t* ent;        // [P * 8 * cap] entries
    uint64_t* pay[TSQ_RADIX_MAXV];  // [P * 8 * cap] argument cells travelling with the entry
    uint32_t* cursor;     // [8][P]
    uint32_t* valid_end;  // [8][P]
    // overflow store (skewed keys: a tile's run that did not fit its region): the whole words u and the argument cells, appended run
    // by run (one device atomic per run) and aggregated by k_daagg_ovf into the dense state.  ovf_u == nullptr (no dense state), or
    // the store is full: those rows go to the exception list instead and are aggregated row by row
    uint32_t* ovf_u;
    uint64_t* ovf_pay[TSQ_RADIX_MAXV];
    uint32_t* ovf_count;
    uint32_t ovf_cap;
    uint32_t bits, ebits, cap;
    uint32_t paybytes;    // width of pay[0]'s cells in the store: 8, or 4 / 2 when V == 1 and the values fit (zero-extended)
};
// A group key of SEVERAL integer columns as one word: field i holds key_i - kmin_i (or the NULL code of a nullable column) at bit
// shift[i]; a cell outside its field's window makes the row an exception, like a key outside [kmin, kmin + 2^b) of the one-column
// route.  Equal words <=> equal cells in every column, NULL = NULL (the group key of getGroupKey, aggregate.go:359-394, is the
// concatenation of the encoded cells: util/codec/codec.go:713-746).  n == 1: the DaDomain alone describes the key.
#define TSQ_DAAGG_MAXK 4
#define TSQ_DAAGG_NO_NULL 0xffffffffu
struct DaAggKeys {
    int32_t n;
    uint64_t kmin[TSQ_DAAGG_MAXK];
    uint32_t maxd[TSQ_DAAGG_MAXK];      // largest key - kmin the field holds
    uint32_t nullcode[TSQ_DAAGG_MAXK];  // field value of a NULL cell; TSQ_DAAGG_NO_NULL: a NULL cell is an exception
    uint32_t shift[TSQ_DAAGG_MAXK], width[TSQ_DAAGG_MAXK];
    int8_t fr_key[TSQ_MAX_AGGS];        // aggregate i is FIRSTROW(key column fr_key[i]) (-1: it is not)
};
struct DaAggSrc {
    const void* kdata;
    const uint8_t* knulls;
    const void* mkdata[TSQ_DAAGG_MAXK];   // several key columns (DaAggKeys.n > 1)
    const uint8_t* mknulls[TSQ_DAAGG_MAXK];
    const void* vdata[TSQ_RADIX_MAXV];
    const uint8_t* vnulls[TSQ_RADIX_MAXV];
    int32_t vtype[TSQ_RADIX_MAXV];
    int64_t nrows;
    uint32_t* exc_rows;   // rows the LDS stage cannot take (NULL key / NULL argument / key outside the packed range)
    uint32_t* exc_count;  // [0] the exception rows; [1] those among them whose argument did not fit the narrow cells (DaAggStore.paybytes)
};
// the fields of row `row` as one number d, or TSQ_DA_NONE (a cell outside its window, a NULL without a code)
__device__ __forceinline__ uint32_t daagg_fields(const DaAggKeys& ks, const DaAggSrc& src, int64_t row) {
    uint32_t d = 0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < TSQ_DAAGG_MAXK; k++) {
        if (k < ks.n) {
            uint32_t f;
            if (tsq_is_null(src.mknulls[k], row)) {
                f = ks.nullcode[k];
                ok = ok && f != TSQ_DAAGG_NO_NULL;
            } else {
                const uint64_t diff = ((const uint64_t*)src.mkdata[k])[row] - ks.kmin[k];
                ok = ok && diff <= (uint64_t)ks.maxd[k];
                f = (uint32_t)diff;
            }
            d |= f << ks.shift[k];
        }
    }
    return ok ? d : TSQ_DA_NONE;
}
// field k of d back into (cell, is NULL)
__device__ __forceinline__ uint64_t daagg_field_cell(const DaAggKeys& ks, uint32_t d, int k, bool* isnull) {
    const uint32_t f = (d >> ks.shift[k]) & (ks.width[k] >= 32 ? 0xffffffffu : ((1u << ks.width[k]) - 1u));
    *isnull = ks.nullcode[k] != TSQ_DAAGG_NO_NULL && f == ks.nullcode[k];
    return *isnull ? 0ull : ks.kmin[k] + (uint64_t)f;
}
__device__ __forceinline__ uint32_t daagg_region_len(const DaAggStore& st, uint32_t P, uint32_t p, uint32_t r) {
    const uint32_t c = r * P + p;
    uint32_t len = st.cursor[c];
    const uint32_t ve = st.valid_end[c];
    len = len < ve ? len : ve;
    return len < st.cap ? len : st.cap;
}

// Round 5 — HOT KEYS.  A skewed batch (the Zipf variant of C3: ONE key holds 5 % of the rows, fifteen keys 20 %) sends the rows of a hot key
// to one partition: they overflow its region, travel through the overflow store and meet again in one LDS cell (k_daagg_ovf 4.6 ms +
// k_agg_da 1.9 instead of 0.4 ms per 2.5e8-row batch).  Those rows need not travel at all: a sample of the batch (k_daagg_hot_sample:
// 64 Ki rows, every key seen >= TSQ_DAAGG_HOT_MIN times, at most TSQ_DAAGG_HOT_MAX of them) names the hot keys, and the partition kernel
// accumulates their rows in a small LDS table of its own (the words of the plan, exactly as k_agg_da would) and adds that table to
// the dense state when it ends — one device atomic per (workgroup, hot key, word).  What is left for the store has no key above
// ~0.04 % of the rows.  A uniform batch has no hot key: the kernels pay one LDS read of the count.
#define TSQ_DAAGG_HOT_MAX 256
#define TSQ_DAAGG_HOT_SLOTS 1024 /* DIRECT-MAPPED LDS table of the hot keys inside the partition kernel: one read per row, no probe loop (the 8 lookups of a
                                    lane are independent loads); a hot key whose slot is taken stays an ordinary key */
#define TSQ_DAAGG_HOT_MIN 12     /* occurrences among 65536 sampled rows: 0.018 % of the batch */
#define TSQ_DAAGG_HOT_MAXW 3     /* words per hot slot: plans with more LDS words per group do not use the feature (3 x 1024 x 8 B of LDS) */
#define TSQ_DAAGG_HOT_SAMPLE 65536
struct DaAggHot {
    uint32_t* keys;   // [TSQ_DAAGG_HOT_MAX] packed words u of the batch's hot keys; nullptr: the feature is off for this launch
    uint32_t* n;      // [1]
    unsigned long long* dense_w[TSQ_AF_MAXW];
    uint32_t* dense_touch;
    unsigned long long init[TSQ_AF_MAXW];
    uint32_t wdesc[TSQ_AF_MAXW];
    int32_t W;
};
struct DaAggHotSampleArgs {
    const uint64_t* kdata;
    const uint8_t* knulls;
    int64_t nrows;
    DaDomain dm;
    uint32_t* table;  // [2 * TSQ_DAAGG_HOT_TABLE]: words, counts (zeroed / set to TSQ_DA_NONE by the host before the launch)
    uint32_t* keys;
    uint32_t* n;
};
#define TSQ_DAAGG_HOT_TABLE 16384
// one sampled row per thread (64 workgroups: the loads are in flight together — one workgroup walking 64 samples per thread took
// 0.65 ms, as long as half a partition pass), counted in a hash table in HBM
__global__ void __launch_bounds__(1024) k_daagg_hot_sample(DaAggHotSampleArgs a) {
    constexpr uint32_t S = TSQ_DAAGG_HOT_TABLE;
    const int64_t ns = a.nrows < TSQ_DAAGG_HOT_SAMPLE ? a.nrows : (int64_t)TSQ_DAAGG_HOT_SAMPLE;
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i >= ns) return;
    // workgroup b reads 1024 CONSECUTIVE rows at b / 64 of the batch (coalesced: 64 Ki rows spread one by one over the batch cost
    // 105 us in TLB and line misses).  Clustered input makes a block see few keys often: such a key is absorbed although it may
    // not be hot over the whole batch — that costs nothing but its LDS slot.
    const int64_t nblk = (ns + 1023) / 1024;
    const int64_t row = (a.nrows / nblk) * (int64_t)blockIdx.x + threadIdx.x;
    if (row >= a.nrows) return;
    if (tsq_is_null(a.knulls, row)) return;
    const uint32_t u = da_word(a.dm, a.kdata[row]);
    if (u == TSQ_DA_NONE) return;
    uint32_t* keys = a.table;
    uint32_t* cnt = a.table + S;
    uint32_t slot = (u * 0x9E3779B1u) >> 18;
    // (uniform keys: 64 Ki distinct words meet 16 Ki slots — a row that finds no place within 4 steps is not counted.  A hot key shows up in
    // the first blocks and owns its slot long before the table fills; 32 steps made the kernel cost 105 us on a uniform batch)
    for (int step = 0; step < 4; step++) {
        uint32_t cur = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == TSQ_DA_NONE) cur = atomicCAS(&keys[slot], TSQ_DA_NONE, u);
        if (cur == TSQ_DA_NONE || cur == u) { atomicAdd(&cnt[slot], 1u); return; }
        slot = (slot + 1) & (S - 1);
    }
}
// ... and the keys seen >= TSQ_DAAGG_HOT_MIN times become the batch's hot keys (one workgroup)
__global__ void __launch_bounds__(1024) k_daagg_hot_list(DaAggHotSampleArgs a) {
    constexpr uint32_t S = TSQ_DAAGG_HOT_TABLE;
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    // the hottest first: when more keys pass the threshold than the list holds, the ones left out are the coolest
    for (uint32_t lo = TSQ_DAAGG_HOT_MIN * 8u, hi = 0xffffffffu; lo >= TSQ_DAAGG_HOT_MIN; hi = lo, lo >>= 1) {
        for (uint32_t i = threadIdx.x; i < S; i += 1024) {
            const uint32_t c = a.table[S + i];
            if (a.table[i] != TSQ_DA_NONE && c >= lo && c < hi) {
                const uint32_t o = atomicAdd(&s_n, 1u);
                if (o < TSQ_DAAGG_HOT_MAX) a.keys[o] = a.table[i];
            }
        }
        __syncthreads();
        if (s_n >= TSQ_DAAGG_HOT_MAX) break;
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) *a.n = s_n < TSQ_DAAGG_HOT_MAX ? s_n : (uint32_t)TSQ_DAAGG_HOT_MAX;
}
// what one row does to the words of hot slot h (the generic form of daagg_apply: W and the descriptors are run-time values here)
__device__ __forceinline__ void daagg_hot_apply(const DaAggHot& hot, unsigned long long (*s_hw)[TSQ_DAAGG_HOT_SLOTS], uint32_t h, uint64_t c0, uint64_t c1) {
#pragma unroll
    for (int k = 0; k < TSQ_DAAGG_HOT_MAXW; k++) {
        if (k >= hot.W) break;
        const uint32_t d = hot.wdesc[k];
        const uint64_t cell = (d & 8u) ? c1 : c0;
        const int32_t type = (int32_t)(d >> 4);
        switch (d & 7u) {
            case AF_W_ADD1: atomicAdd(&s_hw[k][h], 1ull); break;
            case AF_W_ADD_REAL: atomicAdd(reinterpret_cast<double*>(&s_hw[k][h]), af_real(cell, type)); break;
            case AF_W_ADD_LO32: atomicAdd(&s_hw[k][h], (unsigned long long)(cell & 0xffffffffull)); break;
            case AF_W_ADD_HI32:
                if ((long long)cell >> 32) atomicAdd(&s_hw[k][h], (unsigned long long)((long long)cell >> 32));
                break;
            case AF_W_MAX: atomicMax(&s_hw[k][h], (unsigned long long)af_ord_image(cell, type)); break;
            default: atomicMin(&s_hw[k][h], (unsigned long long)af_ord_image(cell, type)); break;
        }
    }
}

// K5e — partition of (packed key entry, argument cells).  The structure of k_da_partition (tsq_dajoin.h) with V payload columns
// staged through LDS next to the words; a run that does not fit its region sends its ROWS to the exception list (they are
// aggregated row by row: exact under any skew).
template <int PB> struct da_pay_t { typedef uint64_t type; };
template <> struct da_pay_t<4> { typedef uint32_t type; };
template <> struct da_pay_t<2> { typedef uint16_t type; };
// WITH_ROW = false (round 4): the tile does not keep its source rows in LDS — only legal with an overflow store that holds a whole
// batch (then no row of a run ever needs the exception list) — which brings a 512-thread tile of 4096 rows down to 40..64 KB of LDS
// and 127 VGPRs: TWO workgroups per CU, one loading while the other scatters (what k_da_partition2 did for the join: the
// 1024-thread version's waves are parked more than half of their cycles).
template <int NT, int K, int V, int PB = 8, bool WITH_ROW = true>
__global__ void __launch_bounds__(NT, WITH_ROW ? 1 : 4) k_daagg_partition(DaAggSrc src, DaDomain dm, DaAggStore st, DaAggKeys ks, DaAggHot hot) {
    constexpr int T = NT * K;
    static_assert(PB == 8 || (V == 1 && (PB == 4 || PB == 2)), "narrow cells: one argument column");
    typedef typename da_pay_t<PB>::type PT;
    constexpr int MAXPER = (TSQ_RADIX_MAX_P + NT - 1) / NT;
    static_assert(T <= 65536 && (K % 2) == 0 && V >= 0 && V <= TSQ_RADIX_MAXV, "tile");
    __shared__ uint32_t s_u[T];
    __shared__ uint32_t s_row[WITH_ROW ? T : 1];
    __shared__ PT s_pay[V ? V : 1][V ? T : 1];
    __shared__ uint32_t s_hist[TSQ_RADIX_MAX_P];
    __shared__ uint32_t s_delta[TSQ_RADIX_MAX_P];
    __shared__ uint32_t s_wsum[NT / 64];
    __shared__ uint32_t s_flag, s_obase;
    // hot keys of the batch (round 5): their rows are aggregated here and never reach the store
    // (only the two-workgroups-per-CU variants have the LDS for it: the 1024-thread tiles fill a CU's 160 KB)
    constexpr int HS = WITH_ROW ? 32 : TSQ_DAAGG_HOT_SLOTS;
    __shared__ uint32_t s_hk[HS];
    __shared__ unsigned long long s_hw[WITH_ROW ? 1 : TSQ_DAAGG_HOT_MAXW][WITH_ROW ? 32 : TSQ_DAAGG_HOT_SLOTS];
    __shared__ uint32_t s_ht[HS / 32];  // slot received a row
    __shared__ uint32_t s_hn;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_hn = (!WITH_ROW && hot.keys) ? *hot.n : 0u;
    __syncthreads();
    const uint32_t hot_n = WITH_ROW ? 0u : s_hn;
    if constexpr (!WITH_ROW) if (hot_n) {
        for (uint32_t i = tid; i < TSQ_DAAGG_HOT_SLOTS; i += NT) {
            s_hk[i] = TSQ_DA_NONE;
#pragma unroll
            for (int k = 0; k < TSQ_DAAGG_HOT_MAXW; k++) s_hw[k][i] = hot.init[k];
        }
        if (tid < TSQ_DAAGG_HOT_SLOTS / 32) s_ht[tid] = 0;
        __syncthreads();
        if (tid < hot_n) {  // two slots per key (a key that finds both taken is simply not absorbed: 256 keys, 1024 slots)
            const uint32_t u = hot.keys[tid];
            if (atomicCAS(&s_hk[(u * 0x9E3779B1u) >> 22], TSQ_DA_NONE, u) != TSQ_DA_NONE) atomicCAS(&s_hk[(u * 0x85EBCA6Bu) >> 22], TSQ_DA_NONE, u);
        }
        __syncthreads();
    }
    const uint32_t P = 1u << st.bits, ebits = st.ebits, emask = (1u << ebits) - 1u;
    auto fits = [](uint64_t cell) -> bool { return PB == 8 || (cell >> (PB == 8 ? 0 : 8 * PB)) == 0; };
    uint32_t misfits = 0;
    const uint32_t r = tsq_xcc_id();
    const uint32_t per = P >= (uint32_t)NT ? P / NT : 1u;
    if (tid == 0) s_flag = 0;
    const int64_t ntiles = (src.nrows + T - 1) / T;
    const bool mk = ks.n > 1;  // several key columns
    bool wide = mk || src.knulls == nullptr;
#pragma unroll
    for (int k = 0; k < TSQ_DAAGG_MAXK; k++)
        if (mk && k < ks.n) wide = wide && src.mknulls[k] == nullptr;
#pragma unroll
    for (int v = 0; v < V; v++) wide = wide && src.vnulls[v] == nullptr && src.vtype[v] != TSQ_F32;
    auto except = [&](uint32_t row) {
        const uint32_t e = __hip_atomic_fetch_add(src.exc_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        src.exc_rows[e] = row;
    };
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * T;
        const int64_t rem = src.nrows - base;
        const uint32_t n = rem < T ? (uint32_t)rem : (uint32_t)T;
        uint32_t u[K], rk[K];
        uint64_t pay[V ? V : 1][K];
        for (uint32_t p = tid; p < P; p += NT) s_hist[p] = 0;
        const bool full = wide && n == (uint32_t)T;
        if (full) {
            uint64_t k[K];
            uint32_t dsum[K];  // several key columns: the fields so far, TSQ_DA_NONE once a cell fell outside its window
            if (mk) {
#pragma unroll
                for (int j = 0; j < K; j++) dsum[j] = 0;
                for (int kk = 0; kk < ks.n; kk++) {  // one key column at a time (its 16-byte loads in flight together)
                    const ulonglong2* c2 = reinterpret_cast<const ulonglong2*>((const uint64_t*)src.mkdata[kk] + base);
#pragma unroll
                    for (int j = 0; j < K / 2; j++) {
                        const ulonglong2 x = c2[j * NT + tid];
                        k[2 * j] = x.x;
                        k[2 * j + 1] = x.y;
                    }
                    const uint64_t kmin = ks.kmin[kk], maxd = ks.maxd[kk];
                    const uint32_t sh = ks.shift[kk];
#pragma unroll
                    for (int j = 0; j < K; j++) {
                        const uint64_t diff = k[j] - kmin;
                        dsum[j] = (diff <= maxd && dsum[j] != TSQ_DA_NONE) ? (dsum[j] | ((uint32_t)diff << sh)) : TSQ_DA_NONE;
                    }
                }
            } else {
                const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>((const uint64_t*)src.kdata + base);
#pragma unroll
                for (int j = 0; j < K / 2; j++) {
                    const ulonglong2 x = s2[j * NT + tid];
                    k[2 * j] = x.x;
                    k[2 * j + 1] = x.y;
                }
            }
#pragma unroll
            for (int v = 0; v < V; v++) {
                const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>((const uint64_t*)src.vdata[v] + base);
#pragma unroll
                for (int j = 0; j < K / 2; j++) {
                    const ulonglong2 x = p2[j * NT + tid];
                    pay[v][2 * j] = x.x;
                    pay[v][2 * j + 1] = x.y;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < K; j++) {
                if (mk) u[j] = dsum[j] == TSQ_DA_NONE ? TSQ_DA_NONE : tsq_da_mix(dsum[j], dm.s, dm.mask);
                else u[j] = da_word(dm, k[j]);
                if (PB != 8 && u[j] != TSQ_DA_NONE && !fits(pay[0][j])) {
                    u[j] = TSQ_DA_NONE;
                    misfits++;
                }
                if (u[j] == TSQ_DA_NONE) except((uint32_t)base + ((uint32_t)(j >> 1) * NT + tid) * 2 + (j & 1));
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const uint32_t pos = (uint32_t)j * NT + tid;
                u[j] = TSQ_DA_NONE;
#pragma unroll
                for (int v = 0; v < V; v++) pay[v][j] = 0;
                if (pos < n) {
                    bool isnull = !mk && tsq_is_null(src.knulls, base + pos);
#pragma unroll
                    for (int v = 0; v < V; v++) isnull |= tsq_is_null(src.vnulls[v], base + pos);
                    if (!isnull) {
                        if (mk) {
                            const uint32_t d = daagg_fields(ks, src, base + pos);
                            if (d != TSQ_DA_NONE) u[j] = tsq_da_mix(d, dm.s, dm.mask);
                        } else {
                            u[j] = da_word(dm, ((const uint64_t*)src.kdata)[base + pos]);
                        }
                    }
                    if (u[j] != TSQ_DA_NONE) {
#pragma unroll
                        for (int v = 0; v < V; v++)
                            pay[v][j] = src.vtype[v] == TSQ_F32 ? (uint64_t)((const uint32_t*)src.vdata[v])[base + pos] : ((const uint64_t*)src.vdata[v])[base + pos];
                        if (PB != 8 && !fits(pay[0][j])) {
                            u[j] = TSQ_DA_NONE;
                            misfits++;
                        }
                    }
                    if (u[j] == TSQ_DA_NONE) except((uint32_t)base + pos);
                }
            }
        }
        if constexpr (!WITH_ROW) if (hot_n) {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const uint32_t h1 = (u[j] * 0x9E3779B1u) >> 22, h2 = (u[j] * 0x85EBCA6Bu) >> 22;
                const uint32_t k1 = s_hk[h1], k2 = s_hk[h2];  // (two independent LDS reads: no probe loop)
                if (u[j] == TSQ_DA_NONE || (k1 != u[j] && k2 != u[j])) continue;
                const uint32_t hs = k1 == u[j] ? h1 : h2;
                daagg_hot_apply(hot, s_hw, hs, V > 0 ? pay[0][j] : 0ull, V > 1 ? pay[V > 1 ? 1 : 0][j] : 0ull);
                if (!((s_ht[hs >> 5] >> (hs & 31u)) & 1u)) atomicOr(&s_ht[hs >> 5], 1u << (hs & 31u));
                u[j] = TSQ_DA_NONE;  // aggregated: nothing of this row travels
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K; j++) {
            rk[j] = 0;
            if (u[j] != TSQ_DA_NONE) rk[j] = atomicAdd(&s_hist[u[j] >> ebits], 1u);
        }
        __syncthreads();
        uint32_t c[MAXPER], sum = 0;
        const uint32_t p0 = tid * per;
#pragma unroll
        for (int q = 0; q < MAXPER; q++) {
            c[q] = ((uint32_t)q < per && p0 + q < P) ? s_hist[p0 + q] : 0u;
            sum += c[q];
        }
        uint32_t total;
        uint32_t run = block_excl_scan<NT>(sum, s_wsum, &total);
        uint32_t g[MAXPER];
        if (per >= 2) {
#pragma unroll
            for (int q = 0; q < MAXPER; q += 2) {
                g[q] = 0;
                if (q + 1 < MAXPER) g[q + 1] = 0;
                if ((uint32_t)q < per && p0 + q < P && (c[q] | c[q + 1 < MAXPER ? q + 1 : q])) {
                    const uint32_t c1 = q + 1 < MAXPER ? c[q + 1] : 0u;
                    unsigned long long* cw = reinterpret_cast<unsigned long long*>(st.cursor + (r * P + p0 + q));
                    const unsigned long long old = atomicAdd(cw, (unsigned long long)c[q] | ((unsigned long long)c1 << 32));
                    g[q] = (uint32_t)old;
                    if (q + 1 < MAXPER) g[q + 1] = (uint32_t)(old >> 32);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < MAXPER; q++) {
                g[q] = 0;
                if ((uint32_t)q < per && p0 + q < P && c[q]) g[q] = atomicAdd(&st.cursor[r * P + p0 + q], c[q]);
            }
        }
        uint32_t oc[MAXPER], ooff[MAXPER];  // this thread's runs that did not fit their regions: rows, offset in the tile
#pragma unroll
        for (int q = 0; q < MAXPER; q++) {
            oc[q] = ooff[q] = 0;
            if ((uint32_t)q < per && p0 + q < P) {
                const uint32_t p = p0 + q, cnt = c[q], offs = run;
                run += cnt;
                uint32_t flag = 0;  // 2: the run goes to the overflow store, 3: to the exception list
                if (cnt) {
                    s_delta[p] = (p * 8u + r) * st.cap + g[q] - offs;
                    if (g[q] + cnt > st.cap) {
                        flag = 3;
                        atomicMin(&st.valid_end[r * P + p], g[q]);
                        s_flag = 1;
                        oc[q] = cnt;
                        ooff[q] = offs;
                    }
                }
                s_hist[p] = offs | (flag << 30);
            }
        }
        __syncthreads();
        // skewed keys: the tile's overflowing runs get consecutive places in the overflow store behind ONE device atomic per tile (a
        // same-address atomic costs ~11 ns chip-wide: one per run — 6e5 per batch on the Zipf variant of C3 — was 5 ms of the kernel)
        if (s_flag != 0 && st.ovf_u != nullptr) {
            uint32_t osum = 0;
#pragma unroll
            for (int q = 0; q < MAXPER; q++) osum += oc[q];
            uint32_t ototal;
            uint32_t orun = block_excl_scan<NT>(osum, s_wsum, &ototal);
            if (tid == 0) s_obase = ototal ? __hip_atomic_fetch_add(st.ovf_count, ototal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            __syncthreads();
            const uint32_t ob = s_obase;
            if (ototal && ob <= st.ovf_cap && ototal <= st.ovf_cap - ob) {
#pragma unroll
                for (int q = 0; q < MAXPER; q++)
                    if (oc[q]) {
                        s_delta[p0 + q] = ob + orun - ooff[q];
                        s_hist[p0 + q] = ooff[q] | (2u << 30);
                        orun += oc[q];
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < K; j++)
            if (u[j] != TSQ_DA_NONE) {
                const uint32_t d = (s_hist[u[j] >> ebits] & 0x3fffffffu) + rk[j];
                s_u[d] = u[j];
                if (WITH_ROW) s_row[d] = (uint32_t)base + (full ? (((uint32_t)(j >> 1) * NT + tid) * 2 + (j & 1)) : ((uint32_t)j * NT + tid));
#pragma unroll
                for (int v = 0; v < V; v++) s_pay[v][d] = (PT)pay[v][j];
            }
        __syncthreads();
        const bool any_ovf = s_flag != 0;
        for (uint32_t i = tid; i < total; i += NT) {
            const uint32_t w = s_u[i], p = w >> ebits;
            const uint32_t how = any_ovf ? (s_hist[p] >> 30) : 0u;
            if (how == 0) {
                const uint32_t d = s_delta[p] + i;
                st.ent[d] = (uint16_t)(w & emask);
#pragma unroll
                for (int v = 0; v < V; v++) reinterpret_cast<PT*>(st.pay[v])[d] = s_pay[v][i];
            } else if (how == 2) {  // skewed keys: the run did not fit its region — overflow store (k_daagg_ovf)
                const uint32_t d = s_delta[p] + i;
                st.ovf_u[d] = w;
#pragma unroll
                for (int v = 0; v < V; v++) reinterpret_cast<PT*>(st.ovf_pay[v])[d] = s_pay[v][i];
            } else if (WITH_ROW) {
                except(s_row[i]);  // no overflow store / store full: the row is aggregated row by row
            } else {
                atomicOr(src.exc_count + 2, 1u);  // cannot happen (the store holds a whole batch); the host fails the batch if it does
            }
        }
        __syncthreads();
    }
    if constexpr (!WITH_ROW) if (hot_n) {  // this workgroup's share of the hot keys' groups -> the dense state (words in their LDS form, like daagg_fold_dense)
        __syncthreads();
        for (uint32_t i = tid; i < TSQ_DAAGG_HOT_SLOTS; i += NT) {
            const uint32_t u = s_hk[i];
            if (u == TSQ_DA_NONE || !((s_ht[i >> 5] >> (i & 31u)) & 1u)) continue;
#pragma unroll
            for (int k = 0; k < TSQ_DAAGG_HOT_MAXW; k++) {
                if (k >= hot.W) break;
                const unsigned long long v = s_hw[k][i];
                unsigned long long* g = hot.dense_w[k] + u;
                switch (hot.wdesc[k] & 7u) {
                    case AF_W_ADD_REAL: atomicAdd(reinterpret_cast<double*>(g), tsq_bits_f64(v)); break;
                    case AF_W_MAX: atomicMax(g, v); break;
                    case AF_W_MIN: atomicMin(g, v); break;
                    default: if (v) atomicAdd(g, v); break;
                }
            }
            atomicOr(&hot.dense_touch[u >> 5], 1u << (u & 31u));
        }
    }
    if (PB != 8) {  // one device atomic per wave that saw a value too wide for the cells: the host widens the cells when there are many
        for (int o = 32; o > 0; o >>= 1) misfits += __shfl_xor(misfits, o, 64);
        if ((tid & 63u) == 0 && misfits) atomicAdd(src.exc_count + 1, misfits);
    }
}

// K7d — direct-addressed LDS pre-aggregation of one partition: words [k][e], one non-returning LDS atomic per word and row.
// A cell that received a row is marked in a bitmap; the partial groups (key = kmin + unmix(p : e), or the field word d of a
// several-column key) leave through ONE returning device atomic per workgroup and a block scan, as in k_agg_lds.
struct DaAggLdsArgs {
    AfPlan plan;
    AfPartials out;
    DaAggStore st;
    DaDomain dm;
    uint32_t nsplit;  // workgroups per partition (1, 2, 4 or 8: each takes every nsplit-th XCC region)
    // dense partial state (one key column): word k of packed word u accumulates in dense_w[k][u], bit u of dense_touch says the
    // cell received a row; nullptr: the workgroup appends its touched cells to `out` as partial groups
    unsigned long long* dense_w[TSQ_AF_MAXW];
    uint32_t* dense_touch;
    uint32_t concurrent;  // the partition kernel of the NEXT batch runs beside this launch (side stream) and adds its hot keys to the dense state: device atomics
};
// what one row does to the accumulators of cell e
// SIG: the two commonest plans with their update descriptors known at compile time — 1: SUM(BIGINT cell 0) + COUNT(*) (words lo32,
// hi32, count), 2: SUM(DOUBLE cell 0) + COUNT(*) — instead of a wave-uniform switch per word and row (scalar compares and
// branches: k_agg_da<3,4096> issued as many scalar as vector instructions, profiles/r03_bench_sq.txt); 0: any plan.
// 3 (round 4): plan 1 with 2-BYTE argument cells and a dense state: value < 2^16 and fewer than 2^24 rows per partition and batch
// (host), so the sum (< 2^40) and the count share ONE LDS word — `count << 40 | sum`, one atomic per row instead of two; the fold
// into the dense state takes the word apart again (TSQ_DAAGG_PACK_SHIFT).
template <int W, int CELLS, int SIG = 0>
__device__ __forceinline__ void daagg_apply(const uint32_t (&wd)[W], unsigned long long (*s_w)[CELLS], uint32_t* s_touch, uint32_t e, uint64_t c0, uint64_t c1) {
    if (SIG == 3 && W == 3) {  // (no touch bit: the packed word counts its rows — a cell is touched iff its word is not zero, daagg_fold_dense)
        atomicAdd(&s_w[0][e], (1ull << TSQ_DAAGG_PACK_SHIFT) + (unsigned long long)c0);
        return;
    }
    // (a plain LDS read first: after its first row a cell's bit is set, and a returning-or-not LDS atomic costs more than a read)
    if (!((s_touch[e >> 5] >> (e & 31u)) & 1u)) atomicOr(&s_touch[e >> 5], 1u << (e & 31u));
    if (SIG == 1 && W == 3) {
        atomicAdd(&s_w[0][e], (unsigned long long)(c0 & 0xffffffffull));
        if ((long long)c0 >> 32) atomicAdd(&s_w[1][e], (unsigned long long)((long long)c0 >> 32));
        atomicAdd(&s_w[W - 1][e], 1ull);
        return;
    }
    if (SIG == 2 && W == 2) {
        atomicAdd(reinterpret_cast<double*>(&s_w[0][e]), tsq_bits_f64(c0));
        atomicAdd(&s_w[W - 1][e], 1ull);
        return;
    }
#pragma unroll
    for (int k = 0; k < W; k++) {
        const uint32_t d = wd[k];
        const uint64_t cell = (d & 8u) ? c1 : c0;
        const int32_t type = (int32_t)(d >> 4);
        switch (d & 7u) {
            case AF_W_ADD1: atomicAdd(&s_w[k][e], 1ull); break;
            case AF_W_ADD_REAL: atomicAdd(reinterpret_cast<double*>(&s_w[k][e]), af_real(cell, type)); break;
            case AF_W_ADD_LO32: atomicAdd(&s_w[k][e], (unsigned long long)(cell & 0xffffffffull)); break;
            case AF_W_ADD_HI32:  // adding zero changes nothing: a non-negative value below 2^32 (most counters, prices, ids) skips the atomic
                if ((long long)cell >> 32) atomicAdd(&s_w[k][e], (unsigned long long)((long long)cell >> 32));
                break;
            case AF_W_MAX: atomicMax(&s_w[k][e], (unsigned long long)af_ord_image(cell, type)); break;
            default: atomicMin(&s_w[k][e], (unsigned long long)af_ord_image(cell, type)); break;
        }
    }
}
// Round 5 — the same for a WAVE whose lanes hold consecutive rows of the store: runs of lanes with the same cell (a hot key: 5 % of a
// Zipf batch is ONE key, ~93 % of its partition's rows) are reduced with segmented shuffle scans first and only the last lane of a run
// touches LDS — 64 same-address LDS atomics are 64 serial operations (k_daagg_ovf spent 4.4 ms per 2.5e8-row Zipf batch in them, the hot
// partitions of k_agg_da as long again).  Wave-uniform choice: fewer than TSQ_DAAGG_COMBINE_MIN lanes repeating their neighbour's
// cell -> the plain per-row path (uniform keys pay one ballot).  Every lane of the wave must call; `live` = the lane holds a row.
#define TSQ_DAAGG_COMBINE_MIN 8
// returns: the wave combined (enough lanes repeated their neighbour's cell) — the caller's hint for the rows that follow in the same lanes
template <int W, int CELLS, int SIG = 0>
__device__ __forceinline__ bool daagg_apply_wave(const uint32_t (&wd)[W], unsigned long long (*s_w)[CELLS], uint32_t* s_touch, uint32_t e, uint64_t c0, uint64_t c1,
                                                 bool live) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t eprev = (uint32_t)__shfl_up((int)e, 1, 64);
    const bool lprev = __shfl_up((int)live, 1, 64) != 0;
    const bool head = !live || lane == 0 || !lprev || e != eprev;
    const unsigned long long hm = __ballot(head);
    if (64 - __popcll(hm) < TSQ_DAAGG_COMBINE_MIN) {
        if (live) daagg_apply<W, CELLS, SIG>(wd, s_w, s_touch, e, c0, c1);
        return false;
    }
    const bool tail = live && (lane == 63 || ((hm >> (lane + 1)) & 1ull));
    const uint32_t cm = sa_cond_mask(head, lane);
    if (SIG == 3 && W == 3) {
        const uint64_t v = sa_scan_add(live ? (1ull << TSQ_DAAGG_PACK_SHIFT) + c0 : 0ull, cm);
        if (tail) atomicAdd(&s_w[0][e], (unsigned long long)v);
        return true;
    }
    if (tail && !((s_touch[e >> 5] >> (e & 31u)) & 1u)) atomicOr(&s_touch[e >> 5], 1u << (e & 31u));
#pragma unroll
    for (int k = 0; k < W; k++) {
        const uint32_t d = wd[k];
        const uint64_t cell = (d & 8u) ? c1 : c0;
        const int32_t type = (int32_t)(d >> 4);
        switch (d & 7u) {
            case AF_W_ADD1: {
                const uint64_t v = sa_scan_add(live ? 1ull : 0ull, cm);
                if (tail) atomicAdd(&s_w[k][e], (unsigned long long)v);
                break;
            }
            case AF_W_ADD_REAL: {
                const double v = sa_scan_addf(live ? af_real(cell, type) : 0.0, cm);
                if (tail) atomicAdd(reinterpret_cast<double*>(&s_w[k][e]), v);
                break;
            }
            case AF_W_ADD_LO32: {
                const uint64_t v = sa_scan_add(live ? (cell & 0xffffffffull) : 0ull, cm);
                if (tail) atomicAdd(&s_w[k][e], (unsigned long long)v);
                break;
            }
            case AF_W_ADD_HI32: {
                const uint64_t v = sa_scan_add(live ? (uint64_t)((long long)cell >> 32) : 0ull, cm);
                if (tail && v) atomicAdd(&s_w[k][e], (unsigned long long)v);
                break;
            }
            case AF_W_MAX: {
                const uint64_t v = sa_scan_max(live ? af_ord_image(cell, type) : 0ull, cm);
                if (tail) atomicMax(&s_w[k][e], (unsigned long long)v);
                break;
            }
            default: {
                const uint64_t v = sa_scan_min(live ? af_ord_image(cell, type) : ~0ull, cm);
                if (tail) atomicMin(&s_w[k][e], (unsigned long long)v);
                break;
            }
        }
    }
    return true;
}
// the touched cells of the workgroup's table -> partial groups; key_of(cell) = the 64-bit key word of the record
template <int W, int CELLS, class KeyOf>
__device__ __forceinline__ void daagg_emit_cells(const AfPlan& plan, const AfPartials& out, unsigned long long (*s_w)[CELLS], const uint32_t* s_touch,
                                                 uint32_t* s_base, uint32_t* s_wsum, KeyOf key_of) {
    const uint32_t tid = threadIdx.x;
    uint32_t mine = 0;
    for (uint32_t i = tid; i < (uint32_t)CELLS / 32; i += TSQ_AF_NT) mine += (uint32_t)__popc(s_touch[i]);
    uint32_t used;
    (void)block_excl_scan<TSQ_AF_NT>(mine, s_wsum, &used);
    __syncthreads();
    if (tid == 0) *s_base = used ? __hip_atomic_fetch_add(out.count, used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    __syncthreads();
    uint32_t running = *s_base;
    for (uint32_t i0 = 0; i0 < (uint32_t)CELLS; i0 += TSQ_AF_NT) {
        const uint32_t i = i0 + tid;
        const bool occ = (s_touch[i >> 5] >> (i & 31u)) & 1u;
        uint32_t total;
        const uint32_t ex = block_excl_scan<TSQ_AF_NT>(occ ? 1u : 0u, s_wsum, &total);
        const uint32_t o = running + ex;
        running += total;
        __syncthreads();  // s_wsum is reused by the next pass
        if (occ && o < out.cap) {
            out.key[o] = key_of(i);
            unsigned long long w[W];
#pragma unroll
            for (int k = 0; k < W; k++) w[k] = s_w[k][i];
            for (int q = 0; q < plan.n_aggs; q++) {  // split int64 sums -> (lo, hi) of the 128-bit value
                const AfAgg f = plan.f[q];
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
            for (int k = 0; k < W; k++) out.w[k][o] = w[k];
        }
    }
}
// the workgroup's table folded into the dense accumulators of partition p (cells [p << ebits, (p + 1) << ebits)).  `shared`: other
// workgroups of this launch fold into the same cells (nsplit > 1) — device atomics; otherwise the cells are this workgroup's alone
// until the kernel ends and a read-modify-write is enough.  The words stay in their LDS form (a BIGINT sum as lo32 / hi32 sums:
// the host flushes the state before 2^31 rows went into it, so neither can wrap).
template <int W, int CELLS, int SIG = 0>
__device__ __forceinline__ void daagg_fold_dense(const uint32_t (&wd)[W], const DaAggLdsArgs& a, unsigned long long (*s_w)[CELLS], const uint32_t* s_touch, uint32_t p) {
    const uint32_t tid = threadIdx.x;
    const uint32_t ncell = 1u << a.st.ebits;  // <= CELLS, >= 32 (host)
    const size_t cbase = (size_t)p << a.st.ebits;
    const bool shared = a.nsplit > 1 || a.concurrent != 0;
    for (uint32_t i = tid; i < ncell; i += TSQ_AF_NT) {
        if (SIG == 3 && W == 3) {  // word 0 = count << 40 | sum: into the dense lo32 sum (word 0) and count (word 2); the hi32 sum stays
            const unsigned long long v = s_w[0][i];
            // the touch bits of these 64 cells (ncell is a multiple of 32, i of the wave's first lane one of 64: one or two words)
            const unsigned long long tm = __ballot(v != 0ull);
            if ((tid & 63u) == 0) {
                const size_t wbit = (cbase + i) >> 5;
                if ((uint32_t)tm) atomicOr(&a.dense_touch[wbit], (uint32_t)tm);
                if ((uint32_t)(tm >> 32)) atomicOr(&a.dense_touch[wbit + 1], (uint32_t)(tm >> 32));
            }
            if (v == 0ull) continue;
            const unsigned long long sum = v & ((1ull << TSQ_DAAGG_PACK_SHIFT) - 1ull), cnt = v >> TSQ_DAAGG_PACK_SHIFT;
            unsigned long long* g0 = a.dense_w[0] + cbase + i;
            unsigned long long* g2 = a.dense_w[W - 1] + cbase + i;
            if (shared) {
                atomicAdd(g0, sum);
                atomicAdd(g2, cnt);
            } else {
                *g0 += sum;
                *g2 += cnt;
            }
            continue;
        }
        if (!((s_touch[i >> 5] >> (i & 31u)) & 1u)) continue;
#pragma unroll
        for (int k = 0; k < W; k++) {
            unsigned long long* g = a.dense_w[k] + cbase + i;
            const unsigned long long v = s_w[k][i];
            switch (wd[k] & 7u) {
                case AF_W_ADD_REAL:
                    if (shared) atomicAdd(reinterpret_cast<double*>(g), tsq_bits_f64(v));
                    else *g = tsq_f64_bits(tsq_bits_f64(*g) + tsq_bits_f64(v));
                    break;
                case AF_W_MAX:
                    if (shared) atomicMax(g, v);
                    else if (v > *g) *g = v;
                    break;
                case AF_W_MIN:
                    if (shared) atomicMin(g, v);
                    else if (v < *g) *g = v;
                    break;
                default:  // counts, lo32 / hi32 sums: 64-bit adds
                    if (shared) atomicAdd(g, v);
                    else *g += v;
                    break;
            }
        }
    }
    if (!(SIG == 3 && W == 3))
        for (uint32_t i = tid; i < ncell / 32; i += TSQ_AF_NT)
            if (s_touch[i]) atomicOr(&a.dense_touch[(cbase >> 5) + i], s_touch[i]);
}
template <int W, int CELLS, int SIG = 0>
__global__ void __launch_bounds__(TSQ_AF_NT) k_agg_da(DaAggLdsArgs a) {
    uint32_t wd[W];
#pragma unroll
    for (int k = 0; k < W; k++) wd[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.plan.wdesc[k]);
    __shared__ unsigned long long s_w[W][CELLS];
    __shared__ uint32_t s_touch[CELLS / 32];
    __shared__ uint32_t s_base, s_wsum[TSQ_AF_NT / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t P = 1u << a.st.bits;
    // a work item = (partition, share of its 8 XCC regions): a narrow key range has few partitions (2^5 for 17 bits) — with one
    // workgroup per partition 32 of 256 CUs worked (3.5 ms per 2.5e8 rows); nsplit workgroups per partition each take every
    // nsplit-th region and emit their own partial groups (the merge adds them up)
    const uint32_t nsplit = a.nsplit;
    for (uint32_t item = blockIdx.x; item < P * nsplit; item += gridDim.x) {
        const uint32_t p = item / nsplit, r0 = item % nsplit;
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)CELLS; i += TSQ_AF_NT) {
#pragma unroll
            for (int k = 0; k < W; k++) s_w[k][i] = a.plan.init[k];
        }
        for (uint32_t i = tid; i < (uint32_t)CELLS / 32; i += TSQ_AF_NT) s_touch[i] = 0;
        __syncthreads();
        auto regions = [&](auto pb_tag) {  // the width of the travelling argument cells is the same for the whole launch
            constexpr int PB = decltype(pb_tag)::value;
            for (uint32_t r = r0; r < 8; r += nsplit) {
                const uint32_t len = daagg_region_len(a.st, P, p, r);
                const size_t base = (size_t)(p * 8u + r) * a.st.cap;
                // Round 6: EIGHT consecutive rows per lane and load — the entries as one 16-byte vector, the argument cells as one (2-byte
                // cells), two (4-byte) or four per column (8-byte).  The 2-byte loads of round 3 kept 16 KB in flight per CU (16 waves x 4
                // rows x two 128-byte wave loads): the kernel waited for memory, not for its LDS atomics (one atomic instead of two per
                // row changed 2 % — profiles/r04_ab_measurements.txt).  The regions start on 128-byte boundaries (cap % 64 == 0) and a
                // lane's rows on 16-byte ones; the last vector of a region may reach past `len` but not past the region (`live`).
                for (uint32_t i0 = tid * 8u; i0 < len; i0 += TSQ_AF_NT * 8u) {
                    const uint4 ev = *reinterpret_cast<const uint4*>(a.st.ent + base + i0);
                    uint64_t cells[8][TSQ_RADIX_MAXV];
                    if (PB == 2) {
                        const uint4 cv = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.st.pay[0]) + base + i0);
                        const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                        for (int x = 0; x < 8; x++) {
                            cells[x][0] = (uint64_t)((cw[x >> 1] >> ((x & 1) * 16)) & 0xffffu);
                            cells[x][1] = 0ull;
                        }
                    } else if (PB == 4) {
                        const uint4* cp = reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(a.st.pay[0]) + base + i0);
                        const uint4 c0 = cp[0], c1 = cp[1];
                        const uint32_t cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                        for (int x = 0; x < 8; x++) {
                            cells[x][0] = (uint64_t)cw[x];
                            cells[x][1] = 0ull;
                        }
                    } else {
#pragma unroll
                        for (int v = 0; v < TSQ_RADIX_MAXV; v++) {
                            if (v < a.plan.V) {
                                const ulonglong2* cp = reinterpret_cast<const ulonglong2*>(a.st.pay[v] + base + i0);
#pragma unroll
                                for (int x = 0; x < 4; x++) {
                                    const ulonglong2 c = cp[x];
                                    cells[2 * x][v] = c.x;
                                    cells[2 * x + 1][v] = c.y;
                                }
                            } else {
#pragma unroll
                                for (int x = 0; x < 8; x++) cells[x][v] = 0ull;
                            }
                        }
                    }
                    const uint32_t ew[4] = {ev.x, ev.y, ev.z, ev.w};
                    // (the loop bound is not wave-uniform: lanes past the end of the region fall out of the last iteration, so the wave
                    // variant — every lane calls — is entered only when the whole wave is still inside the loop; `live` covers the rows
                    // past the end.  The lanes of a wave hold rows 8 apart: a run of one key long enough to be worth combining — the
                    // rows of a hot key that overflowed — still puts the same cell into neighbouring lanes)
                    const bool whole_wave = (i0 - (tid & 63u) * 8u) + 63u * 8u < len;
                    // ... and a wave whose first rows did not combine skips the test for its other seven (neighbour shuffles, a ballot and a
                    // population count per row are a fifth of the instructions of a row whose keys do not repeat; either way is exact)
                    bool combine = whole_wave;
#pragma unroll
                    for (int x = 0; x < 8; x++) {
                        const uint32_t e = (ew[x >> 1] >> ((x & 1) * 16)) & 0xffffu;
                        const bool live = i0 + (uint32_t)x < len;
                        if (combine) combine = daagg_apply_wave<W, CELLS, SIG>(wd, s_w, s_touch, e, cells[x][0], cells[x][1], live);
                        else if (live) daagg_apply<W, CELLS, SIG>(wd, s_w, s_touch, e, cells[x][0], cells[x][1]);
                    }
                }
            }
        };
        if (a.st.paybytes == 4) regions(std::integral_constant<int, 4>{});
        else if (a.st.paybytes == 2) regions(std::integral_constant<int, 2>{});
        else regions(std::integral_constant<int, 8>{});
        __syncthreads();
        if (a.dense_touch != nullptr) {
            daagg_fold_dense<W, CELLS, SIG>(wd, a, s_w, s_touch, p);
            continue;
        }
        daagg_emit_cells<W, CELLS>(a.plan, a.out, s_w, s_touch, &s_base, s_wsum, [&](uint32_t i) -> unsigned long long {
            return a.dm.kmin + (uint64_t)tsq_da_unmix((p << a.st.ebits) | i, a.dm.s, a.dm.mask);
        });
    }
}

// K7g — the overflow store of a skewed batch into the dense state.  The rows of a hot key fill the store (its partition's region
// holds lambda x 1.08 rows; a key with 5 % of the rows brings twenty times that): every workgroup takes a stripe, pre-aggregates in
// an LDS table HASHED by the word u (the hot keys claim their slots at once, the same one-atomic-per-word updates as k_agg_da), and
// folds the table into the dense accumulators with device atomics at the end.  A row that finds no slot within 16 steps goes to
// its dense cell directly: different addresses, so no same-address serialisation (what made the row-at-a-time upsert take 2.9 s
// on the Zipf variant of C3: 1.7e8 rows behind twenty slots).
struct DaAggOvfArgs {
    AfPlan plan;
    DaAggStore st;
    unsigned long long* dense_w[TSQ_AF_MAXW];
    uint32_t* dense_touch;
};
template <int W>
__device__ __forceinline__ void daagg_words_to_dense(const uint32_t (&wd)[W], unsigned long long* const* dense_w, uint32_t* dense_touch, uint32_t u,
                                                     const unsigned long long (&v)[W]) {
#pragma unroll
    for (int k = 0; k < W; k++) {
        unsigned long long* g = dense_w[k] + u;
        switch (wd[k] & 7u) {
            case AF_W_ADD_REAL: atomicAdd(reinterpret_cast<double*>(g), tsq_bits_f64(v[k])); break;
            case AF_W_MAX: atomicMax(g, v[k]); break;
            case AF_W_MIN: atomicMin(g, v[k]); break;
            default: if (v[k]) atomicAdd(g, v[k]); break;
        }
    }
    atomicOr(&dense_touch[u >> 5], 1u << (u & 31u));
}
template <int W>
__global__ void __launch_bounds__(TSQ_AF_NT) k_daagg_ovf(DaAggOvfArgs a) {
    constexpr uint32_t S = W <= 3 ? 4096u : 2048u, LOG2S = W <= 3 ? 12u : 11u;
    constexpr uint32_t EMPTY = 0xffffffffu;
    uint32_t n = *a.st.ovf_count;
    n = n < a.st.ovf_cap ? n : a.st.ovf_cap;
    if (n == 0) return;
    uint32_t wd[W];
#pragma unroll
    for (int k = 0; k < W; k++) wd[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.plan.wdesc[k]);
    __shared__ uint32_t s_key[S];
    __shared__ unsigned long long s_w[W][S];
    __shared__ uint32_t s_touch[S / 32];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < S; i += TSQ_AF_NT) {
        s_key[i] = EMPTY;
#pragma unroll
        for (int k = 0; k < W; k++) s_w[k][i] = a.plan.init[k];
    }
    for (uint32_t i = tid; i < S / 32; i += TSQ_AF_NT) s_touch[i] = 0;
    __syncthreads();
    auto rows = [&](auto pb_tag) {
        constexpr int PB = decltype(pb_tag)::value;
        typedef typename da_pay_t<PB>::type PT;
        for (uint32_t i = blockIdx.x * TSQ_AF_NT + tid; i < n; i += gridDim.x * TSQ_AF_NT) {
            const uint32_t u = a.st.ovf_u[i];
            uint64_t c0 = 0, c1 = 0;
            if (PB != 8) c0 = (uint64_t)reinterpret_cast<const PT*>(a.st.ovf_pay[0])[i];
            else {
                if (a.plan.V > 0) c0 = a.st.ovf_pay[0][i];
                if (a.plan.V > 1) c1 = a.st.ovf_pay[1][i];
            }
            uint32_t slot = (u * 0x9E3779B1u) >> (32u - LOG2S);
            bool found = false;
            for (int step = 0; step < 16 && !found; step++) {
                uint32_t cur = s_key[slot];
                if (cur == EMPTY) cur = atomicCAS(&s_key[slot], EMPTY, u);
                if (cur == EMPTY || cur == u) found = true;
                else slot = (slot + 1) & (S - 1);
            }
            // consecutive rows of the store are a tile's run of ONE partition — mostly one hot key: the wave combines neighbours that
            // found the same slot before it touches LDS (daagg_apply_wave; the last, partial wave of the stripe goes row by row)
            const bool whole_wave = (i - (tid & 63u)) + 63u < n;
            if (whole_wave) daagg_apply_wave<W, (int)S, 0>(wd, s_w, s_touch, slot, c0, c1, found);
            else if (found) daagg_apply<W, (int)S, 0>(wd, s_w, s_touch, slot, c0, c1);
            if (!found) {  // the words of a one-row group, straight into the row's dense cell
                unsigned long long v[W];
#pragma unroll
                for (int k = 0; k < W; k++) {
                    const uint64_t cell = (wd[k] & 8u) ? c1 : c0;
                    const int32_t type = (int32_t)(wd[k] >> 4);
                    switch (wd[k] & 7u) {
                        case AF_W_ADD1: v[k] = 1ull; break;
                        case AF_W_ADD_REAL: v[k] = tsq_f64_bits(af_real(cell, type)); break;
                        case AF_W_ADD_LO32: v[k] = cell & 0xffffffffull; break;
                        case AF_W_ADD_HI32: v[k] = (unsigned long long)((long long)cell >> 32); break;
                        default: v[k] = af_ord_image(cell, type); break;
                    }
                }
                daagg_words_to_dense<W>(wd, a.dense_w, a.dense_touch, u, v);
            }
        }
    };
    if (a.st.paybytes == 4) rows(std::integral_constant<int, 4>{});
    else if (a.st.paybytes == 2) rows(std::integral_constant<int, 2>{});
    else rows(std::integral_constant<int, 8>{});
    __syncthreads();
    for (uint32_t i = tid; i < S; i += TSQ_AF_NT) {
        const uint32_t u = s_key[i];
        if (u == EMPTY) continue;
        unsigned long long v[W];
#pragma unroll
        for (int k = 0; k < W; k++) v[k] = s_w[k][i];
        daagg_words_to_dense<W>(wd, a.dense_w, a.dense_touch, u, v);
    }
}

// K7f — the dense partial state becomes partial groups (once per operator, or before 2^31 rows went into it): every touched cell u
// leaves as (key = kmin + unmix(u), words in the partial-group form of daagg_emit_cells) and is reset to the words' initial values.
struct DaAggDenseArgs {
    AfPlan plan;
    AfPartials out;
    DaDomain dm;
    unsigned long long* dense_w[TSQ_AF_MAXW];
    uint32_t* dense_touch;
    uint64_t ncells;
    int32_t init_only;  // 1: set every cell to the initial words, emit nothing (first use)
};
// Output places: a workgroup takes TSQ_DENSE_CHUNK consecutive cells, counts the touched ones (their touch words), reserves its range of
// the partial-group list with ONE device atomic and hands places out from an LDS cursor.  (One returning device atomic per wave on the
// list's single counter cost ~11 ns each, chip-wide: 1.5 ms of a 4.9 ms aggregate with 4.3e6 groups over a 2^23-cell state.)
#define TSQ_DENSE_CHUNK 4096
__global__ void __launch_bounds__(256) k_daagg_dense_emit(DaAggDenseArgs a) {
    __shared__ uint32_t s_cnt, s_cur;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nchunks = (a.ncells + TSQ_DENSE_CHUNK - 1) / TSQ_DENSE_CHUNK;
    for (uint64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const uint64_t lo = ch * TSQ_DENSE_CHUNK;
        if (a.init_only) {
            for (uint32_t i = threadIdx.x; i < TSQ_DENSE_CHUNK; i += 256)
                if (lo + i < a.ncells)
                    for (int k = 0; k < a.plan.W; k++) a.dense_w[k][lo + i] = a.plan.init[k];
            if (threadIdx.x < TSQ_DENSE_CHUNK / 32 && lo + 32ull * threadIdx.x < a.ncells) a.dense_touch[(lo >> 5) + threadIdx.x] = 0;
            continue;
        }
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        if (threadIdx.x < TSQ_DENSE_CHUNK / 32 && lo + 32ull * threadIdx.x < a.ncells) {  // (cells beyond ncells have no touch bit set)
            const uint32_t c = (uint32_t)__popc(a.dense_touch[(lo >> 5) + threadIdx.x]);
            if (c) atomicAdd(&s_cnt, c);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cur = s_cnt ? __hip_atomic_fetch_add(a.out.count, s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        __syncthreads();
        if (s_cnt == 0) continue;  // (block-uniform)
        for (uint32_t i0 = 0; i0 < TSQ_DENSE_CHUNK; i0 += 256) {
            const uint64_t u = lo + i0 + threadIdx.x;
            const bool occ = u < a.ncells && ((a.dense_touch[u >> 5] >> (u & 31u)) & 1u);
            const unsigned long long m = __ballot(occ);
            if (m == 0) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_cur, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (!occ) continue;
            const uint32_t o = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            unsigned long long w[TSQ_AF_MAXW];
            for (int k = 0; k < a.plan.W; k++) {
                w[k] = a.dense_w[k][u];
                a.dense_w[k][u] = a.plan.init[k];
            }
            for (int q = 0; q < a.plan.n_aggs; q++) {  // split int64 sums -> (lo, hi) of the 128-bit value, as daagg_emit_cells does
                const AfAgg f = a.plan.f[q];
                if (f.w < 0 || (f.func != TSQ_AGG_SUM && f.func != TSQ_AGG_AVG) || af_is_real(f.type)) continue;
                const unsigned long long lo32 = w[f.w], hi32 = w[f.w + 1];
                const unsigned long long lo64 = (hi32 << 32) + lo32;
                w[f.w] = lo64;
                w[f.w + 1] = (unsigned long long)((long long)hi32 >> 32) + (lo64 < lo32 ? 1ull : 0ull);
            }
            if (o < a.out.cap) {
                a.out.key[o] = a.dm.kmin + (uint64_t)tsq_da_unmix((uint32_t)u, a.dm.s, a.dm.mask);
                for (int k = 0; k < a.plan.W; k++) a.out.w[k][o] = w[k];
            }
        }
        __syncthreads();  // every touch bit of the chunk has been read
        if (threadIdx.x < TSQ_DENSE_CHUNK / 32 && lo + 32ull * threadIdx.x < a.ncells) a.dense_touch[(lo >> 5) + threadIdx.x] = 0;
    }
}

// the value range of up to two columns over a SAMPLE of the batch — of every `every` consecutive 256-row blocks one is read — for the
// set-up of the packed route: a key or an argument the sample did not show is an exception row later, never a wrong result.
struct DaAggRangeArgs {
    const uint64_t* data[2];
    const uint8_t* nulls[2];
    uint64_t flip[2];          // order image of the column's type: x ^ flip compares unsigned
    int32_t ncols;
    int64_t nrows;
    int64_t every;
    unsigned long long* out;   // per column c: [3c] min image (preset ~0), [3c + 1] max image (preset 0), [3c + 2] rows seen
};
__global__ void __launch_bounds__(256) k_daagg_sample_range(DaAggRangeArgs a) {
    uint64_t lo[2] = {~0ull, ~0ull}, hi[2] = {0, 0}, n[2] = {0, 0};
    const int64_t nblocks = (a.nrows + 255) / 256;
    for (int64_t sb = blockIdx.x; sb * a.every < nblocks; sb += gridDim.x) {
        const int64_t row = sb * a.every * 256 + threadIdx.x;
        if (row >= a.nrows) continue;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (c >= a.ncols || tsq_is_null(a.nulls[c], row)) continue;
            const uint64_t x = a.data[c][row] ^ a.flip[c];
            lo[c] = x < lo[c] ? x : lo[c];
            hi[c] = x > hi[c] ? x : hi[c];
            n[c]++;
        }
    }
    __shared__ uint64_t s_lo[2][4], s_hi[2][4], s_n[2][4];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t l2 = __shfl_xor(lo[c], o, 64), h2 = __shfl_xor(hi[c], o, 64);
            lo[c] = l2 < lo[c] ? l2 : lo[c];
            hi[c] = h2 > hi[c] ? h2 : hi[c];
            n[c] += __shfl_xor(n[c], o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            s_lo[c][threadIdx.x >> 6] = lo[c];
            s_hi[c][threadIdx.x >> 6] = hi[c];
            s_n[c][threadIdx.x >> 6] = n[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 && (int)threadIdx.x < a.ncols) {
        const int c = threadIdx.x;
        uint64_t l = s_lo[c][0], h = s_hi[c][0], m = s_n[c][0];
        for (int w = 1; w < 4; w++) {
            l = s_lo[c][w] < l ? s_lo[c][w] : l;
            h = s_hi[c][w] > h ? s_hi[c][w] : h;
            m += s_n[c][w];
        }
        if (m) {
            atomicMin(&a.out[3 * c], (unsigned long long)l);
            atomicMax(&a.out[3 * c + 1], (unsigned long long)h);
            atomicAdd(&a.out[3 * c + 2], (unsigned long long)m);
        }
    }
}

// K7e — the same accumulators WITHOUT a partition pass, for a several-column key whose field word fits one LDS table (d < CELLS):
// every workgroup takes a stripe of the input columns, cell d of its table is the group.  Rows with a NULL argument, a cell outside
// its field's window or a NULL key cell without a code go to the exception list.  16 B per row read once and
// nothing written but the partial groups.
struct DaAggLowArgs {
    AfPlan plan;
    AfPartials out;
    DaAggSrc src;
    DaAggKeys ks;
};
template <int W, int CELLS>
__global__ void __launch_bounds__(TSQ_AF_NT) k_agg_da_low(DaAggLowArgs a) {
    uint32_t wd[W];
#pragma unroll
    for (int k = 0; k < W; k++) wd[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.plan.wdesc[k]);
    __shared__ unsigned long long s_w[W][CELLS];
    __shared__ uint32_t s_touch[CELLS / 32];
    __shared__ uint32_t s_base, s_wsum[TSQ_AF_NT / 64];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < (uint32_t)CELLS; i += TSQ_AF_NT) {
#pragma unroll
        for (int k = 0; k < W; k++) s_w[k][i] = a.plan.init[k];
    }
    for (uint32_t i = tid; i < (uint32_t)CELLS / 32; i += TSQ_AF_NT) s_touch[i] = 0;
    __syncthreads();
    // a contiguous stripe of rows per workgroup (the tail of a 64-row bitmap word belongs to one workgroup)
    const int64_t per = ((a.src.nrows + gridDim.x - 1) / gridDim.x + 63) & ~(int64_t)63;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < a.src.nrows ? lo + per : a.src.nrows;
    constexpr int U = 4;  // rows per thread in flight
    constexpr uint32_t SKIP = 0xfffffffeu;
    for (int64_t r0 = lo + tid; r0 < hi; r0 += (int64_t)TSQ_AF_NT * U) {
        uint32_t d[U];
        uint64_t c[U][TSQ_RADIX_MAXV];
#pragma unroll
        for (int x = 0; x < U; x++) {
            const int64_t row = r0 + (int64_t)x * TSQ_AF_NT;
            d[x] = SKIP;
#pragma unroll
            for (int v = 0; v < TSQ_RADIX_MAXV; v++) c[x][v] = 0;
            if (row < hi) {
                bool isnull = false;
#pragma unroll
                for (int v = 0; v < TSQ_RADIX_MAXV; v++)
                    if (v < a.plan.V) isnull |= tsq_is_null(a.src.vnulls[v], row);
                d[x] = isnull ? TSQ_DA_NONE : daagg_fields(a.ks, a.src, row);
#pragma unroll
                for (int v = 0; v < TSQ_RADIX_MAXV; v++)
                    if (v < a.plan.V) c[x][v] = a.src.vtype[v] == TSQ_F32 ? (uint64_t)((const uint32_t*)a.src.vdata[v])[row] : ((const uint64_t*)a.src.vdata[v])[row];
            }
        }
#pragma unroll
        for (int x = 0; x < U; x++) {
            if (d[x] == SKIP) continue;
            if (d[x] == TSQ_DA_NONE || d[x] >= (uint32_t)CELLS) {
                const uint32_t i = __hip_atomic_fetch_add(a.src.exc_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.src.exc_rows[i] = (uint32_t)(r0 + (int64_t)x * TSQ_AF_NT);
                continue;
            }
            daagg_apply<W, CELLS>(wd, s_w, s_touch, d[x], c[x][0], c[x][1]);
        }
    }
    __syncthreads();
    daagg_emit_cells<W, CELLS>(a.plan, a.out, s_w, s_touch, &s_base, s_wsum, [&](uint32_t i) -> unsigned long long { return (unsigned long long)i; });
}

#endif
