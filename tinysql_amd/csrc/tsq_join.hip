// tsq_join.hip — hash join for gfx950 (MI355X).  Replaces executor/join.go + hash_table.go +
// joiner.go of the reference (citations at each piece).
//
// Data layout in HBM
//   build columns : one contiguous device array per column (+ optional null bitmap), appended
//                   chunk by chunk — the device analogue of chunk.List (util/chunk/list.go:22-38).
//   hash table    : bucketised open addressing, structure of arrays (tsq_jointable.h)
//                     keys[nbuckets][8]  uint64  table words w = mix64(key word), one 64-byte line per bucket
//                     vals[nbuckets][8]  uint32  build row ids (RowPtr analogue, list.go:28-31)
//                   2^tb self-contained slices of bs buckets (chains wrap inside their slice).
//                   A probe touches exactly one 64 B line of `keys` (plus the next bucket only if
//                   that one is full); `vals` is touched only for matches that are materialised.
//                   Duplicate keys simply occupy further slots: the table is a multimap like
//                   rowHashMap (hash_table.go:181-276); insertion order inside a key is not kept
//                   (row order is unspecified across join workers in the reference too).
//   EMPTY sentinel: 0x8080808080808080 (memset-able).  Build rows whose TABLE word equals the
//                   sentinel go to a small side list so every int64 value remains a legal key.
// Equality: a key cell is (flag, word) as in util/codec/codec.go:212-240; single-column keys store
// the word itself (exact), multi-column keys store a 64-bit mix and verify against the build
// columns through the row id.
#include "tsq_stage.h"
#include "tsq_jointable.h"
#include "tsq_buildpart.h"
#include "tsq_ldsprobe.h"
#include "tsq_keyrec.h"
#include "tsq_dajoin.h"
#include "tsq_damat.h"

#include <chrono>
#include <deque>
#include <memory>

struct KeySpec {
    int32_t n_keys;
    int32_t bidx[TSQ_MAX_KEYS];
    int32_t pidx[TSQ_MAX_KEYS];
    int32_t skip_high;  // single int key with mixed signedness: cells with the top bit set never match
};

// ------------------------------------------------------------------ device helpers
template <bool MULTI>
__device__ __forceinline__ bool load_kw(const tsq_colset& cs, const int32_t* idx, int n_keys, bool skip_high,
                                        int64_t row, uint64_t& kw) {
    if (!MULTI) {
        const int c = idx[0];
        if (tsq_is_null(cs.nulls[c], row)) return false;
        uint32_t flag;
        kw = tsq_key_word(cs.data[c], cs.type[c], row, &flag);
        if (skip_high && (kw >> 63)) return false;
        return true;
    } else {
        uint64_t h = 0x6A09E667F3BCC908ULL;
        for (int k = 0; k < n_keys; k++) {
            const int c = idx[k];
            if (tsq_is_null(cs.nulls[c], row)) return false;
            uint32_t flag;
            uint64_t w;
            if (cs.type[c] == TSQ_BYTES) {  // a string key cell is (compactBytesFlag, its bytes) (codec.go:233-235): any hash of the bytes will do,
                                            // keys_equal compares the bytes themselves
                const int64_t o0 = cs.offs[c][row], n = cs.offs[c][row + 1] - o0;
                w = tsq_hash_bytes((const uint8_t*)cs.data[c] + o0, n);
                flag = 2;
            } else {
                w = tsq_key_word(cs.data[c], cs.type[c], row, &flag);
            }
            h = tsq_splitmix64(h ^ w) + flag;
        }
        kw = h;
        return true;
    }
}
// util/codec/codec.go:363-382 EqualChunkRow for multi-column keys
__device__ __forceinline__ bool keys_equal(const tsq_colset& b, const tsq_colset& p, const KeySpec& ks, int64_t brow,
                                           int64_t prow) {
    for (int k = 0; k < ks.n_keys; k++) {
        const int cb = ks.bidx[k], cp = ks.pidx[k];
        if (b.type[cb] == TSQ_BYTES) {  // bytes.Equal(b1, b2) (codec.go:377); a string and a number never share a flag (the host sets never_match)
            const int64_t ob = b.offs[cb][brow], nb = b.offs[cb][brow + 1] - ob, op = p.offs[cp][prow], np = p.offs[cp][prow + 1] - op;
            if (nb != np) return false;
            const uint8_t* x = (const uint8_t*)b.data[cb] + ob;
            const uint8_t* y = (const uint8_t*)p.data[cp] + op;
            for (int64_t i = 0; i < nb; i++)
                if (x[i] != y[i]) return false;
            continue;
        }
        uint32_t f1, f2;
        uint64_t w1 = tsq_key_word(b.data[cb], b.type[cb], brow, &f1);
        uint64_t w2 = tsq_key_word(p.data[cp], p.type[cp], prow, &f2);
        if (f1 != f2 || w1 != w2) return false;
    }
    return true;
}

// ------------------------------------------------------------------ K2: build insert
// Replaces hashRowContainer.PutChunk (executor/hash_table.go:146-169) + rowHashMap.Put (:247-256).
// One build row per lane; claim the first EMPTY slot of the home bucket with a 64-bit CAS, spill to
// the next bucket when full.  NULL keys are never inserted (:161-163).
struct BuildArgs {
    tsq_colset b;
    KeySpec ks;
    JoinTable t;
    int64_t row0, nrows;
    uint32_t* sent_rows;   // side list (capacity sent_cap)
    uint32_t sent_cap;
    uint32_t* sent_total;  // number of sentinel-key rows seen
    unsigned long long* inserted;
    const uint32_t* row_list;  // optional: insert only these build rows (rows the partitioned build handed back)
    uint32_t* fail;            // [0] set when a row found no slot within TSQ_MAX_WALK buckets (slice full / key duplicated too often)
                               // [1] set when a row walked past a slot holding its own table word (the build side has duplicate keys)
};
#define TSQ_MAX_WALK 2048u
template <bool MULTI>
__global__ void __launch_bounds__(256) k_build_insert(BuildArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t ins = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.nrows; r += stride) {
        const int64_t row = a.row_list ? (int64_t)a.row_list[r] : a.row0 + r;
        uint64_t kw;
        if (!load_kw<MULTI>(a.b, a.ks.bidx, a.ks.n_keys, a.ks.skip_high, row, kw)) continue;
        ins++;
        const uint64_t w = tsq_table_word(kw);
        if (w == TSQ_EMPTY_KEY) {
            uint32_t i = atomicAdd(a.sent_total, 1u);
            if (i < a.sent_cap) a.sent_rows[i] = (uint32_t)row;
            continue;
        }
        const uint64_t base_b = (uint64_t)jt_slice(a.t.tb, w) * a.t.bs;
        uint32_t lb = jt_local(a.t.tb, a.t.bs, w);
        const uint32_t max_steps = a.t.bs < TSQ_MAX_WALK ? a.t.bs : TSQ_MAX_WALK;
        bool done = false, dup = false;
        for (uint32_t steps = 0; !done; steps++) {
            if (steps >= max_steps) {  // a full slice (skewed keys) or one key duplicated beyond reason: the host decides
                __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            const uint64_t bkt = base_b + lb;
            unsigned long long* base = (unsigned long long*)(a.t.keys + bkt * TSQ_BUCKET);
#pragma unroll 1
            for (int s = 0; s < TSQ_BUCKET && !done; s++) {
                // a stale EMPTY is harmless (the CAS decides); non-EMPTY never reverts
                unsigned long long cur = __hip_atomic_load(base + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == TSQ_EMPTY_KEY) {
                    cur = atomicCAS(base + s, (unsigned long long)TSQ_EMPTY_KEY, (unsigned long long)w);
                    if (cur == TSQ_EMPTY_KEY) {
                        a.t.vals[bkt * TSQ_BUCKET + s] = (uint32_t)row;
                        done = true;
                    }
                }
                // equal words walk the same buckets in the same order, so the later of two always sees the earlier one's slot
                if (!done && cur == w) dup = true;
            }
            lb = (lb + 1 == a.t.bs) ? 0 : lb + 1;
        }
        if (dup) __hip_atomic_store(a.fail + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!done && __hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // the build is void: stop early
    }
    uint64_t tot = wave_sum_u64(ins);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(a.inserted, (unsigned long long)tot);
}
// K2c — chained insert (JoinTable.next): find-or-claim the word's ONE slot, then push the row at the head of its chain
// (rowHashMap.Put, hash_table.go:247-256).  Walks depend on the number of DISTINCT words only.
struct ChainArgs {
    BuildArgs b;
    uint32_t* next;  // [build rows]
};
template <bool MULTI>
__global__ void __launch_bounds__(256) k_build_insert_chained(ChainArgs ca) {
    const BuildArgs& a = ca.b;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t ins = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.nrows; r += stride) {
        const int64_t row = a.row0 + r;
        uint64_t kw;
        if (!load_kw<MULTI>(a.b, a.ks.bidx, a.ks.n_keys, a.ks.skip_high, row, kw)) continue;
        ins++;
        const uint64_t w = tsq_table_word(kw);
        if (w == TSQ_EMPTY_KEY) {
            uint32_t i = atomicAdd(a.sent_total, 1u);
            if (i < a.sent_cap) a.sent_rows[i] = (uint32_t)row;
            continue;
        }
        const uint64_t base_b = (uint64_t)jt_slice(a.t.tb, w) * a.t.bs;
        uint32_t lb = jt_local(a.t.tb, a.t.bs, w);
        const uint32_t max_steps = a.t.bs < TSQ_MAX_WALK ? a.t.bs : TSQ_MAX_WALK;
        uint64_t slot = ~0ull;
        for (uint32_t steps = 0; slot == ~0ull; steps++) {
            if (steps >= max_steps) {
                __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            const uint64_t bkt = base_b + lb;
            unsigned long long* base = (unsigned long long*)(a.t.keys + bkt * TSQ_BUCKET);
#pragma unroll 1
            for (int s = 0; s < TSQ_BUCKET && slot == ~0ull; s++) {
                unsigned long long cur = __hip_atomic_load(base + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == TSQ_EMPTY_KEY) cur = atomicCAS(base + s, (unsigned long long)TSQ_EMPTY_KEY, (unsigned long long)w);
                if (cur == TSQ_EMPTY_KEY || cur == w) slot = bkt * TSQ_BUCKET + (uint64_t)s;  // claimed it, or the word lives here
            }
            lb = (lb + 1 == a.t.bs) ? 0 : lb + 1;
        }
        if (slot == ~0ull) {
            if (__hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            continue;
        }
        ca.next[row] = atomicExch(&a.t.vals[slot], (uint32_t)row);  // vals start at TSQ_CHAIN_END
    }
    uint64_t tot = wave_sum_u64(ins);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(a.inserted, (unsigned long long)tot);
}
// second chance for the sentinel side list when it overflowed its first capacity
template <bool MULTI>
__global__ void __launch_bounds__(256) k_collect_sentinel(BuildArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.nrows; r += stride) {
        uint64_t kw;
        if (!load_kw<MULTI>(a.b, a.ks.bidx, a.ks.n_keys, a.ks.skip_high, a.row0 + r, kw)) continue;
        if (tsq_table_word(kw) != TSQ_EMPTY_KEY) continue;
        uint32_t i = atomicAdd(a.sent_total, 1u);
        if (i < a.sent_cap) a.sent_rows[i] = (uint32_t)(a.row0 + r);
    }
}

// ------------------------------------------------------------------ K3/K4: probe
struct ProbeArgs {
    tsq_colset p;          // probe chunk(s), device resident
    tsq_colset b;          // build columns
    KeySpec ks;
    JoinTable t;
    int64_t nrows;
    const uint8_t* selected;          // optional, one byte per probe row (join.go:328 result)
    const tsq_expr_prog* filters;     // outerSideFilter over probe schema (device), may be null
    int32_t n_filters;
    const tsq_expr_prog* conds;       // OtherConditions over lhs||rhs (device), may be null
    int32_t n_conds;
    int32_t join_type;
    int32_t probe_is_left;            // output order: left||right
    unsigned long long* counters;     // [0]=joined rows [1]=checksum sum [2]=checksum xor [3]=err word [4]=div0 [5]=out cursor
    // materialising mode: contiguous rows per workgroup so that output positions need no device atomics.  The sizing
    // pass (K3) leaves every workgroup's match count in block_base[], k_scan_blocks turns them into exclusive bases,
    // the emit pass (K4) walks the same rows and hands out positions from an LDS cursor.  (One returning device atomic
    // per wave on a single cursor was 4.4 ms of the 6 ms emit kernel for 25 M rows: ~11 ns each, chip-wide.)
    unsigned long long* block_base;
    int64_t rows_per_block;
    // emit only: joined rows as (probe row | build row << 32) pairs, build row 0xffffffff = no match (outer join)
    unsigned long long* pairs;
    // sizing pass -> emit pass: per probe row (first joined build row | output rows << 32).  A probe row with one output
    // row (the common case) is emitted from this word alone; only rows with several matches walk the table again.
    unsigned long long* first_cnt;
    // ordered mode (tsq_join_set_ordered): joined rows come out in probe-row order, the matches of one probe row in
    // build-row order — MergeJoinExec's output order (executor/merge_join.go:257-310) when its inputs are sorted
    int32_t ordered;
};

// probe-side eligibility of row k: selected && outer filter && non-NULL key (join.go:344)
template <bool MULTI, bool GEN>
__device__ __forceinline__ bool probe_row_valid(const ProbeArgs& a, int64_t k, uint64_t& kw, uint64_t& errw, uint32_t& div0) {
    if (GEN) {
        if (a.selected && !a.selected[k]) return false;
        if (a.n_filters > 0) {
            tsq_chunk_src src{&a.p, k};
            bool sel = false, isnull = false;
            int ec = 0, en = 0, d0 = 0;
            tsq_status s = tsq_filter_row(a.filters, a.n_filters, src, &sel, &isnull, &ec, &en, &d0);
            div0 += (uint32_t)d0;
            if (s != TSQ_OK) {
                uint64_t w = tsq_errword(ec, en, (uint64_t)k, s);
                errw = w < errw ? w : errw;
                return false;
            }
            if (!sel) return false;
        }
    }
    return load_kw<MULTI>(a.p, a.ks.pidx, a.ks.n_keys, a.ks.skip_high, k, kw);
}

// is the (probe row k, build row brow) pair a joined row?  verify multi-column keys, then the
// OtherConditions on the joined row (joiner.go:155-167).
template <bool MULTI, bool GEN>
__device__ __forceinline__ bool pair_matches(const ProbeArgs& a, int64_t k, uint32_t brow, uint64_t& errw, uint32_t& div0) {
    if (MULTI && !keys_equal(a.b, a.p, a.ks, brow, k)) return false;
    if (GEN && a.n_conds > 0) {
        tsq_joined_src src;
        if (a.probe_is_left) { src.left = &a.p; src.right = &a.b; src.lrow = k; src.rrow = brow; }
        else { src.left = &a.b; src.right = &a.p; src.lrow = brow; src.rrow = k; }
        bool sel = false, isnull = false;
        int ec = 0, en = 0, d0 = 0;
        tsq_status s = tsq_filter_row(a.conds, a.n_conds, src, &sel, &isnull, &ec, &en, &d0);
        div0 += (uint32_t)d0;
        if (s != TSQ_OK) {
            // conditions are evaluated per probe row batch in the reference; order by probe row
            uint64_t w = tsq_errword(ec, en, (uint64_t)k, s);
            errw = w < errw ? w : errw;
            return false;
        }
        return sel;
    }
    return true;
}

__device__ __forceinline__ uint64_t joined_rowhash(const ProbeArgs& a, int64_t k, int64_t brow /* <0: NULL build side */) {
    uint64_t h = TSQ_ROWHASH_SEED;
    const tsq_colset& L = a.probe_is_left ? a.p : a.b;
    const tsq_colset& R = a.probe_is_left ? a.b : a.p;
    const int64_t lrow = a.probe_is_left ? k : brow, rrow = a.probe_is_left ? brow : k;
    uint32_t c = 0;
    for (int i = 0; i < L.n; i++, c++) {
        uint64_t v = (lrow < 0 || tsq_is_null(L.nulls[i], lrow)) ? TSQ_ROWHASH_NULL : tsq_cell_raw(L, i, lrow);
        h = tsq_rowhash_step(h, v, c);
    }
    for (int i = 0; i < R.n; i++, c++) {
        uint64_t v = (rrow < 0 || tsq_is_null(R.nulls[i], rrow)) ? TSQ_ROWHASH_NULL : tsq_cell_raw(R, i, rrow);
        h = tsq_rowhash_step(h, v, c);
    }
    return h;
}

// visits every build row joined with probe row k
template <bool MULTI, bool GEN, class F>
__device__ __forceinline__ void for_each_match(const ProbeArgs& a, int64_t k, uint64_t kw, uint64_t& errw, uint32_t& div0, F&& f) {
    const uint64_t w = tsq_table_word(kw);
    if (w == TSQ_EMPTY_KEY) {
        for (uint32_t j = 0; j < a.t.sent_count; j++) {
            uint32_t brow = a.t.sent_rows[j];
            if (pair_matches<MULTI, GEN>(a, k, brow, errw, div0)) f(brow);
        }
        return;
    }
    if (a.t.next) {  // chained table: one slot per word, then the word's rows
        const uint64_t slot = jt_find_slot(a.t, w);
        if (slot == ~0ull) return;
        for (uint32_t brow = a.t.vals[slot]; brow != TSQ_CHAIN_END; brow = a.t.next[brow]) {
            if (!MULTI && !(GEN && a.n_conds > 0)) f(brow);
            else if (pair_matches<MULTI, GEN>(a, k, brow, errw, div0)) f(brow);
        }
        return;
    }
    for_each_slot_w(a.t, w, [&](uint64_t slot) {
        if (!MULTI && !(GEN && a.n_conds > 0)) {
            f(a.t.vals[slot]);  // the load is dead-code-eliminated when f ignores the row id
        } else {
            uint32_t brow = a.t.vals[slot];
            if (pair_matches<MULTI, GEN>(a, k, brow, errw, div0)) f(brow);
        }
    });
}

#define TSQ_PAIR_MISS 0xffffffffu

// K3 — COUNT(*) probe (+ optional fused row checksum).  Replaces the per-row loop of join2Chunk
// (executor/join.go:343-360) + GetMatchedRows (hash_table.go:110-134) when no row is materialised.
// Fast path (inner join, single key, no filters): reads 8 B of probe key + one 64 B table line.
template <bool MULTI, bool GEN, bool CHK>
__global__ void __launch_bounds__(256) k_probe_count(ProbeArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t cnt = 0, csum = 0, cxor = 0, errw = TSQ_ERRWORD_NONE;
    uint32_t div0 = 0;
    const bool outer = a.join_type != TSQ_JOIN_INNER;
    const bool chunked = a.block_base != nullptr;
    const int64_t k_first = chunked ? (int64_t)blockIdx.x * a.rows_per_block + threadIdx.x : (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k_last = chunked ? (((int64_t)blockIdx.x + 1) * a.rows_per_block < a.nrows ? ((int64_t)blockIdx.x + 1) * a.rows_per_block : a.nrows) : a.nrows;
    const int64_t k_step = chunked ? (int64_t)blockDim.x : stride;
    for (int64_t k = k_first; k < k_last; k += k_step) {
        uint64_t kw = 0;
        uint32_t c = 0, first = TSQ_PAIR_MISS;
        if (probe_row_valid<MULTI, GEN>(a, k, kw, errw, div0)) {
            if (!MULTI && !GEN && !CHK && !a.first_cnt && !a.t.next) {
                // leanest form: keys only, never touches vals
                const uint64_t w = tsq_table_word(kw);
                if (w == TSQ_EMPTY_KEY) c = a.t.sent_count;
                else for_each_slot_w(a.t, w, [&](uint64_t) { c++; });
            } else {
                for_each_match<MULTI, GEN>(a, k, kw, errw, div0, [&](uint32_t brow) {
                    if (c == 0) first = brow;
                    c++;
                    if (CHK) {
                        uint64_t h = joined_rowhash(a, k, (int64_t)brow);
                        csum += h;
                        cxor ^= h;
                    }
                });
            }
        }
        if (GEN && outer && c == 0) {  // onMissMatch (joiner.go:274-277,337-340)
            c = 1;
            if (CHK) {
                uint64_t h = joined_rowhash(a, k, -1);
                csum += h;
                cxor ^= h;
            }
        }
        if (a.first_cnt) a.first_cnt[k] = (unsigned long long)first | ((unsigned long long)c << 32);  // coalesced
        cnt += c;
    }
    cnt = wave_sum_u64(cnt);
    if (CHK) { csum = wave_sum_u64(csum); cxor = wave_xor_u64(cxor); }
    {   // one device atomic per workgroup on the shared row counter (per wave they were ~11 ns each, chip-wide)
        __shared__ unsigned long long s_block;
        if (threadIdx.x == 0) s_block = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s_block, (unsigned long long)cnt);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (chunked) a.block_base[blockIdx.x] = s_block;  // this workgroup's output rows, for the emit pass
            if (s_block) atomicAdd(&a.counters[0], s_block);
        }
    }
    if (CHK && (threadIdx.x & 63) == 0) {
        atomicAdd(&a.counters[1], (unsigned long long)csum);
        atomicXor(&a.counters[2], (unsigned long long)cxor);
    }
    if (GEN) {
        if (errw != TSQ_ERRWORD_NONE) atomicMin(&a.counters[3], (unsigned long long)errw);
        if (div0) atomicAdd(&a.counters[4], (unsigned long long)div0);
    }
}

// K4 — materialising probe, in two kernels.  Replaces joiner.tryToMatchInners / makeJoinRowToChunk / onMissMatch
// (executor/joiner.go:145-150,220-410) and Chunk.AppendRow (util/chunk/chunk.go:334-356).
// K4a (k_probe_emit) walks the probe rows again and writes only WHICH rows join: one 8-byte (probe row, build row) pair
// per output row at the position the sizing pass reserved.  K4b (k_gather_cols) then copies the columns, one launch
// dimension per output column, eight consecutive output rows per thread.  Copying the columns from inside the probe
// loop was latency bound: with a selective join only a few lanes of a wave have a match, and they ran ~2 dependent
// loads per column one after the other (1.4 ms for 8 M probe rows -> 0.8 M x 10 columns).  In K4b every lane is
// active, the eight gathers of a thread are independent, the stores are 64 contiguous bytes per thread, and the
// null bitmap byte of the eight rows is written directly (no byte flags + pack pass).
__device__ __forceinline__ void write_pair(const ProbeArgs& a, uint64_t pos, int64_t k, uint32_t brow) {
    a.pairs[pos] = (unsigned long long)(uint32_t)k | ((unsigned long long)brow << 32);
}

struct GatherCol {
    const void* src;          // source column data
    const uint8_t* src_nulls; // source null bitmap (bit 1 = NOT NULL) or null
    void* dst;
    uint8_t* dst_bitmap;      // packed output bitmap or null (column cannot hold NULLs)
    int32_t es;               // 4 or 8
    int32_t from_probe;
};
struct GatherArgs {
    const unsigned long long* pairs;
    int64_t rows;
    GatherCol col[2 * TSQ_MAX_COLS];
};
__global__ void __launch_bounds__(256) k_gather_cols(GatherArgs a) {
    const GatherCol c = a.col[blockIdx.y];
    if (c.es == 0) return;  // a var-len column: k_varlen_* handle it
    const int64_t groups = (a.rows + 7) >> 3;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r0 = g << 3;
        const int n = a.rows - r0 < 8 ? (int)(a.rows - r0) : 8;
        uint32_t idx[8];
        if (n == 8) {
            const ulonglong2* pp = (const ulonglong2*)(a.pairs + r0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const ulonglong2 v = pp[q];
                idx[2 * q] = c.from_probe ? (uint32_t)v.x : (uint32_t)(v.x >> 32);
                idx[2 * q + 1] = c.from_probe ? (uint32_t)v.y : (uint32_t)(v.y >> 32);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned long long v = i < n ? a.pairs[r0 + i] : 0ull;
                idx[i] = c.from_probe ? (uint32_t)v : (uint32_t)(v >> 32);
            }
        }
        uint32_t nn = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            bool ok = i < n && (c.from_probe || idx[i] != TSQ_PAIR_MISS);
            if (ok && c.src_nulls) ok = (c.src_nulls[idx[i] >> 3] >> (idx[i] & 7)) & 1;
            nn |= ok ? (1u << i) : 0u;
        }
        if (c.es == 8) {
            uint64_t v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (nn >> i) & 1 ? ((const uint64_t*)c.src)[idx[i]] : 0ull;
            uint64_t* d = (uint64_t*)c.dst + r0;
            if (n == 8) {
#pragma unroll
                for (int q = 0; q < 4; q++) ((ulonglong2*)d)[q] = make_ulonglong2(v[2 * q], v[2 * q + 1]);
            } else {
                for (int i = 0; i < n; i++) d[i] = v[i];
            }
        } else {
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (nn >> i) & 1 ? ((const uint32_t*)c.src)[idx[i]] : 0u;
            uint32_t* d = (uint32_t*)c.dst + r0;
            if (n == 8) {
                ((uint4*)d)[0] = make_uint4(v[0], v[1], v[2], v[3]);
                ((uint4*)d)[1] = make_uint4(v[4], v[5], v[6], v[7]);
            } else {
                for (int i = 0; i < n; i++) d[i] = v[i];
            }
        }
        if (c.dst_bitmap) c.dst_bitmap[g] = (uint8_t)nn;
    }
}

// K4c — var-len output columns (util/chunk/column.go:28-34; Chunk.AppendRow of a var-len cell, chunk.go:334-356): the lengths of
// the gathered cells (k_varlen_len), their exclusive scan = the output offsets (tsq_launch_scan64), then the bytes.
struct VarGatherArgs {
    const unsigned long long* pairs;
    int64_t rows;
    const int64_t* src_offs;
    const uint8_t* src_data;
    const uint8_t* src_nulls;
    int32_t from_probe;
    int64_t* out_offs;    // [rows + 1]: k_varlen_len leaves the lengths, the scan turns them into offsets
    uint8_t* out_data;
    uint8_t* out_bitmap;  // packed, or null when the column cannot hold NULLs
};
__device__ __forceinline__ uint32_t var_src_row(const VarGatherArgs& a, int64_t r, bool* valid) {
    const unsigned long long v = a.pairs[r];
    const uint32_t idx = a.from_probe ? (uint32_t)v : (uint32_t)(v >> 32);
    bool ok = a.from_probe || idx != TSQ_PAIR_MISS;
    if (ok && a.src_nulls) ok = (a.src_nulls[idx >> 3] >> (idx & 7)) & 1;
    *valid = ok;
    return idx;
}
__global__ void __launch_bounds__(256) k_varlen_len(VarGatherArgs a) {
    const int64_t groups = (a.rows + 7) >> 3;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r0 = g << 3;
        uint32_t nn = 0;
        for (int i = 0; i < 8 && r0 + i < a.rows; i++) {
            bool ok;
            const uint32_t idx = var_src_row(a, r0 + i, &ok);
            a.out_offs[r0 + i] = ok ? a.src_offs[idx + 1] - a.src_offs[idx] : 0;  // a NULL cell has no bytes
            nn |= ok ? (1u << i) : 0u;
        }
        if (a.out_bitmap) a.out_bitmap[g] = (uint8_t)nn;
    }
}
// short cells: one row per lane
__global__ void __launch_bounds__(256) k_varlen_copy_rows(VarGatherArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.rows; r += (int64_t)gridDim.x * blockDim.x) {
        bool ok;
        const uint32_t idx = var_src_row(a, r, &ok);
        if (!ok) continue;
        const uint8_t* s = a.src_data + a.src_offs[idx];
        uint8_t* d = a.out_data + a.out_offs[r];
        const int64_t n = a.out_offs[r + 1] - a.out_offs[r];
        tsq_copy_cell(d, s, n);
    }
}
// long cells (the reference's join benchmark carries a 5 KiB payload, executor/benchmark_test.go:328): one row per wave, 64 lanes on
// consecutive bytes
__global__ void __launch_bounds__(256) k_varlen_copy_wave(VarGatherArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < a.rows; r += nwaves) {
        bool ok;
        const uint32_t idx = var_src_row(a, r, &ok);
        if (!ok) continue;
        const uint8_t* s = a.src_data + a.src_offs[idx];
        uint8_t* d = a.out_data + a.out_offs[r];
        const int64_t n = a.out_offs[r + 1] - a.out_offs[r];
        // head up to an 8-byte boundary of the destination, then 8 bytes per lane, then the tail
        int64_t head = (8 - ((uintptr_t)d & 7)) & 7;
        head = head < n ? head : n;
        if (lane < head) d[lane] = s[lane];
        const int64_t words = (n - head) >> 3;
        for (int64_t w = lane; w < words; w += 64) {
            uint64_t x;
            memcpy(&x, s + head + w * 8, 8);  // the source is not aligned with the destination
            *reinterpret_cast<uint64_t*>(d + head + w * 8) = x;
        }
        const int64_t done = head + words * 8;
        if (done + lane < n) d[done + lane] = s[done + lane];
    }
}

// in-place sort of one probe row's pairs by build row id (the high word): insertion sort for the usual short match lists,
// heapsort beyond 32 so that a heavily duplicated key costs n log n, not n^2, per probe row
__device__ __noinline__ void sort_pairs_by_build_row(unsigned long long* p, uint32_t n) {
    if (n <= 32) {
        for (uint32_t i = 1; i < n; i++) {
            const unsigned long long v = p[i];
            uint32_t q = i;
            while (q > 0 && (p[q - 1] >> 32) > (v >> 32)) {
                p[q] = p[q - 1];
                q--;
            }
            p[q] = v;
        }
        return;
    }
    auto sift = [&](uint32_t root, uint32_t end) {  // max-heap on the build row id over p[0, end)
        const unsigned long long v = p[root];
        for (;;) {
            uint32_t child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && (p[child + 1] >> 32) > (p[child] >> 32)) child++;
            if ((p[child] >> 32) <= (v >> 32)) break;
            p[root] = p[child];
            root = child;
        }
        p[root] = v;
    };
    for (uint32_t i = n / 2; i-- > 0;) sift(i, n);
    for (uint32_t end = n - 1; end > 0; end--) {
        const unsigned long long t = p[0];
        p[0] = p[end];
        p[end] = t;
        sift(0, end);
    }
}

template <bool MULTI, bool GEN>
__global__ void __launch_bounds__(256) k_probe_emit(ProbeArgs a) {
    __shared__ unsigned long long s_cur;
    uint64_t errw = TSQ_ERRWORD_NONE;
    uint32_t div0 = 0;
    const int64_t k_begin = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t k_last = k_begin + a.rows_per_block < a.nrows ? k_begin + a.rows_per_block : a.nrows;
    if (k_last < k_begin) k_last = k_begin;  // workgroups past the end of the batch
    const int64_t k_round = k_begin + ((k_last - k_begin + 63) & ~(int64_t)63);  // wave-uniform trip count for the collectives
    if (threadIdx.x == 0) s_cur = a.block_base[blockIdx.x];
    __syncthreads();
    if (a.ordered) {
        // positions in probe-row order: block-wide exclusive scan of the per-row output counts, 256 consecutive rows per
        // step, on top of the workgroup's base (the workgroups own consecutive row ranges, k_scan_blocks ordered them)
        __shared__ uint32_t s_wsum[4];
        unsigned long long run = a.block_base[blockIdx.x];
        const int64_t k_round256 = k_begin + ((k_last - k_begin + 255) & ~(int64_t)255);
        for (int64_t k = k_begin + threadIdx.x; k < k_round256; k += 256) {
            const unsigned long long fc = k < k_last ? a.first_cnt[k] : 0ull;
            const uint32_t first = (uint32_t)fc, n_out = (uint32_t)(fc >> 32);
            uint32_t total;
            const uint32_t ex = block_excl_scan<256>(n_out, s_wsum, &total);
            const uint64_t pos = run + ex;
            run += total;
            if (n_out) write_pair(a, pos, k, first);
            if (n_out > 1) {
                uint64_t kw = 0;
                uint32_t seen = 0;
                if (probe_row_valid<MULTI, GEN>(a, k, kw, errw, div0))
                    for_each_match<MULTI, GEN>(a, k, kw, errw, div0, [&](uint32_t brow) { write_pair(a, pos + seen++, k, brow); });
                // the slots of a chain are filled in whatever order the build's CAS won: put this row's matches into
                // build-row (= insertion) order
                sort_pairs_by_build_row(a.pairs + pos, n_out);
            }
            __syncthreads();  // s_wsum is reused by the next step
        }
        return;
    }
    for (int64_t k = k_begin + threadIdx.x; k < k_round; k += blockDim.x) {
        const bool active = k < k_last;
        // what the sizing pass found for this probe row: no second walk of the table for rows with one output row
        const unsigned long long fc = active ? a.first_cnt[k] : 0ull;
        const uint32_t first = (uint32_t)fc, n_out = (uint32_t)(fc >> 32);
        uint32_t total;
        uint32_t prefix = wave_excl_scan_u32(n_out, &total);
        unsigned long long base = 0;
        if ((threadIdx.x & 63) == 0 && total) base = atomicAdd(&s_cur, (unsigned long long)total);  // LDS cursor of this workgroup
        base = __shfl(base, 0, 64);
        uint64_t pos = base + prefix;
        if (n_out) write_pair(a, pos, k, first);  // the first match, or the NULL-padded row of an outer join (first = MISS)
        if (n_out > 1) {  // duplicates: the remaining matches, in walk order (same filters / conditions as the sizing pass,
                          // whose errors and warnings were reported there)
            uint64_t kw = 0;
            uint32_t seen = 0;
            if (probe_row_valid<MULTI, GEN>(a, k, kw, errw, div0))
                for_each_match<MULTI, GEN>(a, k, kw, errw, div0, [&](uint32_t brow) {
                    if (seen++) write_pair(a, pos + seen - 1, k, brow);
                });
        }
    }
}

// exclusive scan of the per-workgroup match counts (n <= a few thousand): one workgroup
__global__ void __launch_bounds__(1024) k_scan_blocks(unsigned long long* v, int n) {
    __shared__ unsigned long long s_w[16];
    const int per = (n + 1023) / 1024, lo = threadIdx.x * per;
    unsigned long long sum = 0;
    for (int i = lo; i < lo + per && i < n; i++) sum += v[i];
    unsigned long long x = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(x, o, 64);
        if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned long long pre = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre += s_w[w];
    unsigned long long run = pre + x - sum;
    for (int i = lo; i < lo + per && i < n; i++) {
        const unsigned long long c = v[i];
        v[i] = run;
        run += c;
    }
}

// ====================================================================== host side
namespace {

struct ResultBatch {  // one probe batch worth of joined rows
    int64_t rows = 0, cursor = 0;
    std::vector<DevBuf> data;        // per output column, device
    std::vector<DevBuf> notnull;     // per output column, device byte flags (cap 0 => column has no NULLs)
    std::vector<DevBuf> bitmap;      // packed null bitmaps (device)
    std::vector<DevBuf> offs;        // var-len output columns: offsets[rows + 1] (device)
    std::vector<int64_t> nbytes;     // ... and their data bytes
    bool on_host = false;
    std::vector<PinnedBuf> hdata, hbitmap, hoffs;
    // host pushes (round 6): the D2H copies of the batch run on the operator's copy stream beside the kernels of the next batch; `ready`
    // is recorded behind them and the device columns are kept until it has fired (settle_batch)
    hipEvent_t ready = nullptr;
    bool pending = false;
    void release() {
        if (ready) {
            if (pending) (void)hipEventSynchronize(ready);
            (void)hipEventDestroy(ready);
            ready = nullptr;
            pending = false;
        }
        for (auto& b : data) b.release();
        for (auto& b : notnull) b.release();
        for (auto& b : bitmap) b.release();
        for (auto& b : offs) b.release();
        for (auto& b : hdata) b.release();
        for (auto& b : hbitmap) b.release();
        for (auto& b : hoffs) b.release();
    }
};

}  // namespace

struct tsq_join {
    tsq_handle_hdr hdr;
    tsq_ctx* ctx = nullptr;
    tsq_join_cfg cfg;
    std::vector<tsq_expr_prog> conds_h, filters_h;
    DevBuf conds_d, filters_d;
    std::atomic<int> cancelled{0};

    // build side
    std::vector<ColStore> bcols;
    bool build_done = false;
    bool table_ready = false;  // the 64-bit table exists (tsq_join_build_finish, or the first probe batch that needed it)
    bool never_match = false;  // key classes differ (int vs float): no row can ever match
    bool multi = false;
    KeySpec ks{};
    DevBuf tkeys, tvals, sent, tnext;
    bool chained = false;     // the table holds one slot per distinct word + row chains (JoinTable.next)
    uint64_t nbuckets = 0;  // bs << tb
    uint32_t tb = 0, bs = 0;  // slice geometry (tsq_jointable.h)
    bool unique = false;      // no two table slots hold the same word (the build compares every slot it walks past)
    uint32_t sent_count = 0;
    int64_t build_inserted = 0;
    int64_t build_handed_back = 0;  // partitioned build: rows that went through the row list (skew, chains crossing a slice end)

    // host staging (shared by build and probe pushes; one side is active at a time)
    HostStage stage;

    // probe side device batch (for host pushes)
    std::vector<ColStore> pcols;
    DevBuf psel;
    bool probe_done = false;
    bool host_mode = true;  // result placement follows the first probe push
    bool general = false;   // needs the GEN kernels (outer join / filters / conditions / selected)
    bool general_cfg = false;  // ... by its configuration alone (outer join / filters / conditions): what the packed routes ask
    bool count_only = false, checksum = false, ordered = false;
    DevBuf counters;        // 8 x u64 on device
    int64_t total_out = 0;  // emit mode: rows produced so far
    std::deque<std::unique_ptr<ResultBatch>> results;

    // radix probe path of the COUNT(*) fast path (tsq_radix.h)
    int32_t radix_mode = TSQ_RADIX_AUTO;
    DevBuf rkeys, rctl, rvend, rovf;  // partitioned keys | cursor + queue heads + overflow count | valid_end | overflow keys
    DevBuf rpay[TSQ_LDS_MAXPAY], rovfpay[TSQ_LDS_MAXPAY];  // materialising radix path: probe payload columns travelling with the key
    DevBuf tkcnt;                                          // ... joined rows per ticket, then their exclusive scan
    DevBuf tpay[TSQ_LDS_MAXPAY];                           // ... build payload columns in table-slot order (k_table_payload)
    bool tpay_ready = false;
    DevBuf bbase;                     // per-workgroup output bases of the materialising probe
    DevBuf pairs;                     // (probe row, build row) of every joined row of the current slice
    DevBuf firstcnt;                  // per probe row of the slice: first joined build row | output rows << 32
    // packed-key route (tsq_dajoin.h): key range of the build side + direct-address images, made by the first eligible probe batch
    int32_t packing_mode = TSQ_RADIX_AUTO;
    int da_state = 0;                 // 0: not tried yet, 1: usable, -1: not usable for this build side
    DaDomain da_dm{};
    uint32_t da_pbits = 0, da_ebits = 0;
    bool da_unique = false;
    bool da_bits = false;             // bit cells: a unique build side whose keys span 29..31 bits (COUNT(*) route)
    bool da_multi = false;            // several integer key columns composed into one key column (k_da_compose)
    DaFields da_fields{};
    DevBuf da_ckey, rckey;            // composite key column of the build side | of the probe batch in flight
    DevBuf da_img;                    // 2^b one-byte cells (bit cells: 2^b / 8 bytes)
    double da_build_ms = 0;
    // a build side SHARDED over the ranks of a communicator whose packed images were summed across the ranks
    // (tsq_join_build_finish_shared): the handle answers COUNT(*) for LOCAL probe rows, no 64-bit table exists
    std::vector<uint8_t> used_out;    // per output column: 0 = the parent never reads it (tsq_join_set_used_columns); empty: all are used
    double last_sampled_hit_ratio = -1.0;  // of the last probe batch whose materialising route was chosen by a sample (k_da_sample)
    int wide_state = 0;               // several integer key columns of 29..63 bits: COUNT(*) through a single-key child join (wide_prepare)
    tsq_join* wide = nullptr;
    // key records (tsq_keyrec.h): COUNT(*) on several key columns / string keys whose cells fit 32 bytes, partitioned
    int kr_state = 0;                 // 0: not tried, 1: the build side's records are in place, -1: not usable for this build side
    uint32_t kr_pbits = 0;
    DevBuf kr_brec, kr_bstart, kr_counts, kr_prec, kr_pstart, kr_flags, kr_bids, kr_pids, kr_pcnt, kr_norec;
    // long string keys (round 6): the string key columns enter the records as (length, digest of the bytes); matches are verified byte for byte
    bool kr_digest = false;
    DevBuf kr_bdig[TSQ_MAX_KEYS], kr_pdig[TSQ_MAX_KEYS];
    DevBuf kr_vmask;                  // ... the outcomes of the sizing launch's byte comparisons, one word per probe record (k_kr_probe: vmode)
    bool filters_folded = false;      // this batch: the outer-side filters are already in the selected[] flags the packed routes take (fold_outer_filters)
    DevBuf fflags;                    //   ... those flags
    DevBuf heads;                     // da_emit_cols: first-candidate flags of a batch (outer join + conditions + duplicate build keys)
    int64_t direct_batches = 0;       // batches that went through the direct route
    int64_t div0_packed = 0;          // division-by-zero warnings of conditions evaluated over materialised batches (da_post_conditions)
    bool shared = false;
    int64_t shared_image_bytes = 0, shared_usable_local = 0;
    double shared_allreduce_ms = 0;
    int da_bitrows_state = 0;         // bit-cell pairs route (unique build side, 4-byte entries): bit image + coarse popcounts + sorted build rows
    DevBuf da_bitimg;                 // ... the bit form of byte cells (b <= 28)
    int da_rows_state = 0;            // materialising packed route: build rows sorted by word (CSR over the images)
    DevBuf da_coarse, da_pstart, da_brows;
    DevBuf da_coarse_c, da_pstart_c;  // the travelling-columns route keeps its OWN coarse counts / partition starts: AUTO may prepare both routes in one join, and a give-up of one must not free what the other reads
    DevBuf ridx, rovfidx, rmiss;      // ... probe rows travelling with the entries | of the overflow list | that cannot match (outer joins)
    int da_cols_state = 0;            // travelling-columns route: the build columns sorted by word (+ NOT-NULL bytes)
    DevBuf da_bsorted[TSQ_DA_MAXCOLS], da_bsorted_nn[TSQ_DA_MAXCOLS];
    DevBuf rcols[TSQ_DA_MAXCOLS], rnnmask;  // ... the probe columns travelling with the entries, their NOT-NULL bits
    // materialising packed route with the build side in LDS (round 6, tsq_damat.h): two partition levels, ranked payload tables
    int dm_state = 0;                 // 0: not tried yet, 1: the build side's final partitions are ready, -1: not usable for this build side
    uint32_t dm_sbits = 0, dm_tab_rows = 0, dm_cap1_b = 0;
    int dm_nb = 0, dm_bcol_of[TSQ_DA_MAXCOLS] = {};  // the build columns in the tables (all but the key column)
    bool dm_bnulls = false;
    DevBuf dm_bent, dm_bpay[TSQ_DA_MAXCOLS], dm_bnn, dm_boff, dm_bcnt, dm_bitmap;  // build side: final partitions | one bit per word of the domain
    DevBuf dm_pent, dm_ppay[TSQ_DA_MAXCOLS], dm_pnn, dm_poff, dm_pcnt;             // the probe batch in flight: final partitions
    // level 1 of the build side WITH its columns, made by da_prepare in place of its key-only partition pass when the join materialises
    // (the images are assembled from the same entries): dm_prepare_build starts from it — one pass over the build side saved
    bool dm_l1_ready = false, dm_l1_nulls = false;
    uint32_t dm_l1_cap = 0;
    DevBuf dm_l1ent, dm_l1ctl, dm_l1vend, dm_l1nn, dm_l1pay[TSQ_DA_MAXCOLS];
    static constexpr int RING = 32;   // HIP events of the most recent radix batches: [slot][0..2] = start, after partition, end
    hipEvent_t rev[RING][3] = {};

    // stats
    tsq_stats st{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;  // host pushes: result batches leave for pinned memory here (deliver_batch)
    hipEvent_t ev_emit = nullptr;       // main stream: the columns of the batch being delivered are written
    hipEvent_t ev_h2d = nullptr;        // main stream: the staged rows of a flush have left pinned memory
    int64_t stage_sent = 0;             // probe rows of the staging buffers whose H2D copies are queued already (probe_stage_early)
    bool have_build_ev = false, have_probe_ev = false;
    double probe_ms_acc = 0;
};

static tsq_status build_table(tsq_join* j);  // (defined with tsq_join_build_finish)

namespace {

tsq_status check_cancel(tsq_join* j) {
    if (j->cancelled.load()) return tsq_fail(&j->hdr, TSQ_ERR_CANCELLED, "join cancelled");
    return TSQ_OK;
}

bool is_int_class(int32_t t) { return t == TSQ_I64 || t == TSQ_U64; }

tsq_status build_flush(tsq_join* j) {
    HostStage& sg = j->stage;
    if (sg.staged == 0) return TSQ_OK;
    DevBuf tmp, tmp2;
    for (size_t c = 0; c < j->bcols.size(); c++) {
        tsq_status s = sg.append_to(j->ctx, &j->hdr, (int)c, j->bcols[c], tmp, tmp2);
        if (s != TSQ_OK) { tmp.release(); tmp2.release(); return s; }
        j->st.h2d_bytes += j->bcols[c].type == TSQ_BYTES ? sg.nbytes[c] + sg.staged * 8 : sg.staged * j->bcols[c].elem();
    }
    hipError_t e = hipStreamSynchronize(j->ctx->stream);  // staging memory is reused
    tmp.release();
    tmp2.release();
    sg.reset();
    if (e != hipSuccess) return tsq_fail(&j->hdr, TSQ_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return TSQ_OK;
}

template <bool MULTI>
tsq_status launch_build(tsq_join* j, BuildArgs& a) {
    int grid = tsq_grid_for(j->ctx, a.nrows, 256);
    hipLaunchKernelGGL(k_build_insert<MULTI>, dim3(grid), dim3(256), 0, j->ctx->stream, a);
    TSQ_HIP(&j->hdr, hipGetLastError());
    j->st.kernel_launches++;
    return TSQ_OK;
}

void fill_table(tsq_join* j, JoinTable& t) {
    t.keys = j->tkeys.as<uint64_t>();
    t.vals = j->tvals.as<uint32_t>();
    t.nbuckets = j->nbuckets;
    t.tb = j->tb;
    t.bs = j->bs;
    t.sent_rows = j->sent.as<uint32_t>();
    t.sent_count = j->sent_count;
    t.next = j->chained ? j->tnext.as<uint32_t>() : nullptr;
}

// decode the device error word into a status (first offending node, then row)
tsq_status status_from_errword(tsq_join* j, uint64_t w) {
    if (w == TSQ_ERRWORD_NONE) return TSQ_OK;
    tsq_status s = (tsq_status)(w & 15);
    char buf[160];
    snprintf(buf, sizeof buf, "expression error %d in join condition/filter (conjunct %d, node %d, probe row %llu)", (int)s,
             (int)(w >> 58), (int)((w >> 52) & 63), (unsigned long long)((w >> 4) & 0xffffffffffffULL));
    return tsq_fail(&j->hdr, s, buf);
}

tsq_status reset_counters(tsq_join* j, bool only_batch) {
    // counters[3] (err word) starts at all ones; [5] (emit cursor) is per batch
    unsigned long long init[8] = {0, 0, 0, TSQ_ERRWORD_NONE, 0, 0, 0, 0};
    if (only_batch) {
        TSQ_HIP(&j->hdr, hipMemsetAsync((char*)j->counters.p + 5 * 8, 0, 8, j->ctx->stream));
        return TSQ_OK;
    }
    memcpy(j->ctx->pinned, init, sizeof init);
    TSQ_HIP(&j->hdr, hipMemcpyAsync(j->counters.p, j->ctx->pinned, sizeof init, hipMemcpyHostToDevice, j->ctx->stream));
    TSQ_HIP(&j->hdr, hipStreamSynchronize(j->ctx->stream));
    return TSQ_OK;
}

tsq_status read_counters(tsq_join* j, unsigned long long* out8) {
    TSQ_HIP(&j->hdr, hipMemcpyAsync(j->ctx->pinned, j->counters.p, 8 * 8, hipMemcpyDeviceToHost, j->ctx->stream));
    TSQ_HIP(&j->hdr, hipStreamSynchronize(j->ctx->stream));
    memcpy(out8, j->ctx->pinned, 64);
    return TSQ_OK;
}

template <bool MULTI, bool GEN>
tsq_status launch_count(tsq_join* j, ProbeArgs& a, bool chk) {
    int grid = tsq_grid_for(j->ctx, a.nrows, 256);
    if (chk) hipLaunchKernelGGL((k_probe_count<MULTI, GEN, true>), dim3(grid), dim3(256), 0, j->ctx->stream, a);
    else hipLaunchKernelGGL((k_probe_count<MULTI, GEN, false>), dim3(grid), dim3(256), 0, j->ctx->stream, a);
    TSQ_HIP(&j->hdr, hipGetLastError());
    j->st.kernel_launches++;
    return TSQ_OK;
}
tsq_status dispatch_count(tsq_join* j, ProbeArgs& a, bool chk) {
    if (j->multi) return j->general ? launch_count<true, true>(j, a, chk) : launch_count<true, false>(j, a, chk);
    return j->general ? launch_count<false, true>(j, a, chk) : launch_count<false, false>(j, a, chk);
}
tsq_status dispatch_emit(tsq_join* j, ProbeArgs& a) {
    int grid = tsq_grid_for(j->ctx, a.nrows, 256);
    if (j->multi) {
        if (j->general) hipLaunchKernelGGL((k_probe_emit<true, true>), dim3(grid), dim3(256), 0, j->ctx->stream, a);
        else hipLaunchKernelGGL((k_probe_emit<true, false>), dim3(grid), dim3(256), 0, j->ctx->stream, a);
    } else {
        if (j->general) hipLaunchKernelGGL((k_probe_emit<false, true>), dim3(grid), dim3(256), 0, j->ctx->stream, a);
        else hipLaunchKernelGGL((k_probe_emit<false, false>), dim3(grid), dim3(256), 0, j->ctx->stream, a);
    }
    TSQ_HIP(&j->hdr, hipGetLastError());
    j->st.kernel_launches++;
    return TSQ_OK;
}


// the D2H copies of a host-mode batch are done: its device columns go back to the pool (stream order protects pooled buffers on the main
// stream only, so they were kept until the copy stream was through with them)
tsq_status settle_batch(tsq_join* j, ResultBatch* rb) {
    if (!rb->pending) return TSQ_OK;
    TSQ_HIP(&j->hdr, hipEventSynchronize(rb->ready));
    rb->pending = false;
    for (auto& b : rb->data) b.release();
    for (auto& b : rb->notnull) b.release();
    for (auto& b : rb->bitmap) b.release();
    for (auto& b : rb->offs) b.release();
    return TSQ_OK;
}

// hand a materialised batch (device columns) to tsq_join_pull: host pushes get it in pinned host memory (pulls are then plain memcpy)
tsq_status deliver_batch(tsq_join* j, std::unique_ptr<ResultBatch> rb, const std::vector<bool>& may_null_v) {
    tsq_ctx* ctx = j->ctx;
    const int nout = j->cfg.n_probe_cols + j->cfg.n_build_cols;
    const bool probe_is_left = j->cfg.build_is_right != 0;
    const int nl = probe_is_left ? j->cfg.n_probe_cols : j->cfg.n_build_cols;
    const int64_t out_rows = rb->rows;
    if (j->host_mode) {
        // round 6: the copies run on the operator's copy stream behind an event of the main stream, the call does not wait for them — the
        // next batch's staging, H2D and kernels run beside them, tsq_join_pull waits for `ready` (TSQ_KNOB_HOST_OVERLAP = 0: one stream
        // and a wait here, as before)
        const bool overlap = (tsq_knob(ctx, TSQ_KNOB_HOST_OVERLAP, 3) & 1) != 0;
        hipStream_t cs = ctx->stream;
        if (overlap) {
            if (!j->copy_stream) TSQ_HIP(&j->hdr, hipStreamCreateWithFlags(&j->copy_stream, hipStreamNonBlocking));
            if (!j->ev_emit) TSQ_HIP(&j->hdr, hipEventCreateWithFlags(&j->ev_emit, hipEventDisableTiming));
            TSQ_HIP(&j->hdr, hipEventCreateWithFlags(&rb->ready, hipEventDisableTiming));
            cs = j->copy_stream;
            TSQ_HIP(&j->hdr, hipEventRecord(j->ev_emit, ctx->stream));
            TSQ_HIP(&j->hdr, hipStreamWaitEvent(cs, j->ev_emit, 0));
        }
        rb->hdata.resize(nout);
        rb->hbitmap.resize(nout);
        rb->hoffs.resize(nout);
        for (int oc = 0; oc < nout; oc++) {
            const bool from_probe = probe_is_left ? oc < nl : oc >= nl;
            const int sc = oc < nl ? oc : oc - nl;
            const int32_t type = from_probe ? j->cfg.probe_types[sc] : j->cfg.build_types[sc];
            if (!rb->data[oc].p && !(type == TSQ_BYTES && rb->offs[oc].p)) continue;  // a column the parent does not use: never materialised
            size_t bytes = type == TSQ_BYTES ? (size_t)rb->nbytes[oc] : (size_t)out_rows * tsq_elem_size(type);
            tsq_status s = rb->hdata[oc].reserve(&j->hdr, bytes + 16);
            if (s == TSQ_OK && type == TSQ_BYTES) s = rb->hoffs[oc].reserve(&j->hdr, ((size_t)out_rows + 1) * 8 + 16);
            if (s != TSQ_OK) { if (overlap) (void)hipStreamSynchronize(cs); rb->release(); return s; }  // (copies of earlier columns may be on their way into the buffers being handed back)
            if (bytes) TSQ_HIP(&j->hdr, hipMemcpyAsync(rb->hdata[oc].p, rb->data[oc].p, bytes, hipMemcpyDeviceToHost, cs));
            if (type == TSQ_BYTES) {
                TSQ_HIP(&j->hdr, hipMemcpyAsync(rb->hoffs[oc].p, rb->offs[oc].p, ((size_t)out_rows + 1) * 8, hipMemcpyDeviceToHost, cs));
                bytes += ((size_t)out_rows + 1) * 8;
            }
            j->st.d2h_bytes += bytes;
            if (may_null_v[oc]) {
                s = rb->hbitmap[oc].reserve(&j->hdr, tsq_bitmap_bytes(out_rows) + 16);
                if (s != TSQ_OK) { if (overlap) (void)hipStreamSynchronize(cs); rb->release(); return s; }  // (copies of earlier columns may be on their way into the buffers being handed back)
                TSQ_HIP(&j->hdr, hipMemcpyAsync(rb->hbitmap[oc].p, rb->bitmap[oc].p, tsq_bitmap_bytes(out_rows), hipMemcpyDeviceToHost, cs));
            }
        }
        if (overlap) {
            TSQ_HIP(&j->hdr, hipEventRecord(rb->ready, cs));
            rb->pending = true;
            rb->on_host = true;
            // at most three batches in flight behind the one being pulled (their device columns are alive until their copies are done)
            int inflight = 0;
            for (auto& r : j->results) inflight += r->pending ? 1 : 0;
            if (inflight >= 3)
                for (auto& r : j->results)
                    if (r->pending) { TSQ_TRY(settle_batch(j, r.get())); break; }
        } else {
        TSQ_HIP(&j->hdr, hipStreamSynchronize(ctx->stream));
        for (auto& b : rb->data) b.release();
        for (auto& b : rb->notnull) b.release();
        for (auto& b : rb->bitmap) b.release();
        for (auto& b : rb->offs) b.release();
        rb->on_host = true;
        }
    } else {
        TSQ_HIP(&j->hdr, hipStreamSynchronize(ctx->stream));
        for (auto& b : rb->notnull) b.release();
    }
    j->total_out += out_rows;
    j->st.out_rows += out_rows;
    j->results.push_back(std::move(rb));
    return TSQ_OK;
}

// ---------------------------------------------------------------- radix probe path (host side)
// Eligible: COUNT(*) without checksum, inner join, one key column, no filters / conditions / selected[].
// AUTO takes it when both the table and the batch are big enough to pay for a partition pass.
bool radix_eligible(const tsq_join* j, int64_t nrows, const uint8_t* selected_dev) {
    if (j->radix_mode == TSQ_RADIX_OFF) return false;
    if (!j->count_only || j->checksum || j->multi || j->general_cfg || j->never_match || j->chained) return false;
    if (nrows <= 0 || nrows > 0x7fffffffLL) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE) return true;
    // table slices of ~1.5 MB per partition need >= 8 partitions to be worth it; batch >= 4 Mi rows
    return j->nbuckets * 64 >= ((uint64_t)12 << 20) && nrows >= (4 << 20);
}

// COUNT(*) over a join on several integer key columns: no 64-bit radix route exists for it, only the packed one (da_multi_ok is
// defined with the packed route below)
bool da_multi_ok(const tsq_join* j);
bool da_multi_count_eligible(const tsq_join* j, int64_t nrows, const uint8_t* selected_dev) {
    if (j->radix_mode == TSQ_RADIX_OFF || j->packing_mode == TSQ_RADIX_OFF || !j->multi || !da_multi_ok(j) || j->da_state < 0) return false;
    if (!j->count_only || j->checksum || j->general_cfg || j->never_match || j->chained) return false;
    if (nrows <= 0 || nrows > 0x7fffffffLL) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE || j->packing_mode == TSQ_RADIX_FORCE) return true;
    return nrows >= (4 << 20);
}

// Geometry of a radix probe batch.  LDS route (sliced table): P = 2^pb partitions of 2^(tb - pb) table slices each, read as S
// images of nf slices (<= 128 KB); L2 route (plain table, or TSQ_RADIX_KERNEL=l2): table slices of ~1.5 MB per partition.
struct RadixPlan {
    bool lds;
    uint32_t bits, S, nf;
};
RadixPlan radix_plan_for(const tsq_join* j) {
    RadixPlan pl{false, TSQ_RADIX_MIN_BITS, 1, 1};
    // test / experiment knobs: TSQ_RADIX_KERNEL=l2 keeps the L2 route, TSQ_LDS_NF_MAX caps the slices per image (forces S > 1 on
    // small tables), TSQ_RADIX_PB_MAX caps log2(partitions)
    const int64_t nf_env = tsq_knob(j->ctx, TSQ_KNOB_LDS_NF_MAX, 0), pb_env = tsq_knob(j->ctx, TSQ_KNOB_RADIX_PB_MAX, 0);
    const bool want_lds = tsq_knob(j->ctx, TSQ_KNOB_RADIX_KERNEL_L2, 0) == 0;
    if (want_lds && j->tb >= TSQ_RADIX_MIN_BITS && (uint64_t)j->bs * 64 <= TSQ_LDS_IMAGE_MAX) {
        uint32_t nf_max = (uint32_t)(TSQ_LDS_IMAGE_MAX / ((uint64_t)j->bs * 64));
        if (nf_env >= 1 && (uint32_t)nf_env < nf_max) nf_max = (uint32_t)nf_env;
        uint32_t k = 0;
        while ((2u << k) <= nf_max) k++;  // largest power of two <= nf_max
        int pb = (int)j->tb - (int)k;
        if (pb < TSQ_RADIX_MIN_BITS) pb = TSQ_RADIX_MIN_BITS;
        if (pb > TSQ_RADIX_MAX_BITS) pb = TSQ_RADIX_MAX_BITS;
        if (pb_env >= TSQ_RADIX_MIN_BITS && pb_env < pb) pb = (int)pb_env;
        const uint32_t fpp = 1u << (j->tb - (uint32_t)pb);
        pl.lds = true;
        pl.bits = (uint32_t)pb;
        pl.nf = fpp < nf_max ? fpp : nf_max;
        pl.S = (fpp + pl.nf - 1) / pl.nf;
        return pl;
    }
    const double slice_target = 1.5 * 1024 * 1024;  // two to three slices stay resident in a 4 MiB L2
    const double parts = (double)j->nbuckets * 64.0 / slice_target;
    uint32_t bits = TSQ_RADIX_MIN_BITS;
    while (bits < 10 && (double)(1u << bits) < parts) bits++;
    if (j->tb && bits > j->tb) bits = j->tb;  // a partition is a whole number of slices
    pl.bits = bits;
    return pl;
}

tsq_status radix_probe(tsq_join* j, const tsq_colset& pcs, int64_t nrows) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    RadixStore st;
    memset(&st, 0, sizeof st);
    const RadixPlan pl = radix_plan_for(j);
    st.bits = pl.bits;
    st.R = 8;
    const uint32_t P = 1u << st.bits;
    constexpr int NT = 1024, K = 16, T = NT * K;
    const double lam = (double)nrows / ((double)P * 8.0);
    st.cap = (uint32_t)(lam * 1.08 + 8.0 * sqrt(lam) + 2.0 * T / 64.0 + 64.0);
    st.cap = (st.cap + 15u) & ~15u;  // regions start on 128-byte lines
    const size_t nregions = (size_t)P * 8;
    if (nregions * st.cap >= 0xffffffffULL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "radix probe batch too large");
    const size_t ctl_words32 = nregions + 16;                       // cursors + overflow count (+pad)
    const size_t ctl_bytes = ((ctl_words32 * 4 + 511) & ~(size_t)511) + 8 * TSQ_RADIX_QSTRIDE * 8;
    TSQ_TRY(j->rkeys.reserve(ctx, h, nregions * st.cap * 8 + 256));
    TSQ_TRY(j->rctl.reserve(ctx, h, ctl_bytes));
    TSQ_TRY(j->rvend.reserve(ctx, h, nregions * 4));
    TSQ_TRY(j->rovf.reserve(ctx, h, (size_t)nrows * 8 + 64));
    st.keys = j->rkeys.as<uint64_t>();
    st.cursor = j->rctl.as<uint32_t>();
    st.ovf_count = st.cursor + nregions;
    st.queue = (unsigned long long*)((char*)j->rctl.p + ((ctl_words32 * 4 + 511) & ~(size_t)511));
    st.valid_end = j->rvend.as<uint32_t>();
    st.ovf_keys = j->rovf.as<uint64_t>();
    st.ovf_cap = (uint32_t)nrows;
    TSQ_HIP(h, hipMemsetAsync(j->rctl.p, 0, ctl_bytes, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->rvend.p, 0xff, nregions * 4, ctx->stream));

    RadixSrc src;
    memset(&src, 0, sizeof src);
    const int kc = j->ks.pidx[0];
    src.data = pcs.data[kc];
    src.nulls = pcs.nulls[kc];
    src.type = pcs.type[kc];
    src.skip_high = j->ks.skip_high;
    src.nrows = nrows;
    const int64_t ntiles = (nrows + T - 1) / T;
    const int pgrid = (int)std::min<int64_t>(ntiles, ctx->num_cus);  // 148 KB of LDS: one workgroup per CU
    hipEvent_t* re = j->rev[j->st.radix_batches % tsq_join::RING];
    for (int e = 0; e < 3; e++)
        if (!re[e]) TSQ_HIP(h, hipEventCreate(&re[e]));
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(re[0], ctx->stream));
    hipLaunchKernelGGL((k_radix_partition<NT, K, 4, 0, false, true>), dim3(pgrid), dim3(NT), 0, ctx->stream, src, st);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipEventRecord(re[1], ctx->stream));
    RadixProbeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.st = st;
    fill_table(j, pa.t);
    pa.counters = j->counters.as<unsigned long long>();
    if (pl.lds) {
        constexpr int LNT = 1024;
        LdsProbeArgs la;
        memset(&la, 0, sizeof la);
        la.st = st;
        la.t = pa.t;
        la.S = pl.S;
        la.nf = pl.nf;
        la.unique = j->unique ? 1u : 0u;
        la.counters = pa.counters;
        const size_t lds = (size_t)pl.nf * j->bs * 64 + (size_t)(LNT / 64) * TSQ_LDS_RING_BYTES;
        const int lgrid = std::max(1, ctx->num_cus / 8) * 8;
        if (tsq_knob(ctx, TSQ_KNOB_LDS_PROF, 0) != 0) {  // experiment: per-phase shader cycles of the LDS probe, printed per batch (synchronises)
            DevBuf pb;
            TSQ_TRY(pb.reserve(ctx, h, 64));
            la.prof = pb.as<unsigned long long>();
            TSQ_HIP(h, hipMemsetAsync(pb.p, 0, 64, ctx->stream));
            TSQ_HIP(h, hipFuncSetAttribute((const void*)k_lds_probe_count<LNT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_lds_probe_count<LNT, true>), dim3(lgrid), dim3(LNT), lds, ctx->stream, la);
            unsigned long long pf[8];
            TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 24, pb.p, 64, hipMemcpyDeviceToHost, ctx->stream));
            TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
            memcpy(pf, ctx->pinned + 24, 64);
            pb.release();
            const double nw = (double)lgrid * (LNT / 64);
            fprintf(stderr, "[lds-prof] P=2^%u S=%u nf=%u bs=%u waves=%.0f | per wave kcycles: imageA %.1f loadwait %.1f step %.1f probe %.1f drainB %.1f total %.1f | probes/wave %.0f\n",
                    st.bits, pl.S, pl.nf, j->bs, nw, pf[0] / nw / 1e3, pf[1] / nw / 1e3, pf[2] / nw / 1e3, pf[3] / nw / 1e3, pf[4] / nw / 1e3, pf[5] / nw / 1e3, pf[6] / nw);
        } else {
            TSQ_HIP(h, hipFuncSetAttribute((const void*)k_lds_probe_count<LNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_lds_probe_count<LNT>), dim3(lgrid), dim3(LNT), lds, ctx->stream, la);
        }
    } else {
        const int per_xcd = std::max(1, ctx->num_cus / 8) * 6;  // 6 workgroups per CU (measured optimum 5-6)
        hipLaunchKernelGGL((k_radix_probe_count<2>), dim3(per_xcd * 8), dim3(256), 0, ctx->stream, pa);
    }
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_radix_probe_ovf, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
    j->have_probe_ev = true;
    j->st.kernel_launches += 3;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)st.bits;
    j->st.probe_route = pl.lds ? TSQ_ROUTE_RADIX_LDS : TSQ_ROUTE_RADIX_L2;
    return TSQ_OK;
}

// ---------------------------------------------------------------- packed-key route (host side; tsq_dajoin.h)
// Eligible (on top of radix_eligible): ONE integer key column on both sides whose build-side range fits TSQ_DA_MAX_BITS bits and is
// dense enough (>= 1 build row per 32 cells), no build key with more than 255 duplicates.  The range and the images are made by
// the first probe batch that asks for them (the build side does not know yet whether the probe will only count).
struct DaGeom {
    uint32_t P, cap;
    size_t nregions, ent_bytes, ctl_bytes;
};
DaGeom da_geometry(uint32_t pbits, uint32_t ebits, int64_t nrows, int T) {
    DaGeom g;
    g.P = 1u << pbits;
    // region (p, r) takes the tiles that ran on XCD r: ceil(tiles / 8) of them — for a batch of a few tiles that is far more than
    // rows / (8 P), and a region sized by the average would send most of a small batch to the overflow list
    const double tiles = ceil((double)nrows / T);
    const double lam = std::max((double)nrows / ((double)g.P * 8.0), ceil(tiles / 8.0) * std::min<double>((double)T, (double)nrows) / (double)g.P);
    g.cap = (uint32_t)(lam * 1.08 + 8.0 * sqrt(lam) + 2.0 * T / 64.0 + 64.0);
    g.cap = (g.cap + 63u) & ~63u;  // regions start on 128-byte lines also with 2-byte entries
    g.nregions = (size_t)g.P * 8;
    g.ent_bytes = g.nregions * g.cap * (ebits > 16 ? 4 : 2) + 256;
    g.ctl_bytes = ((g.nregions + 16) * 4 + 511) & ~(size_t)511;
    return g;
}
tsq_status da_launch_partition(tsq_join* j, const DaSrc& src, const DaStore& st, bool with_idx = false, bool miss = false, const DaMk* mk = nullptr) {
    constexpr int NT = 1024, K = 16, T = NT * K;
    const int64_t ntiles = (src.nrows + T - 1) / T;
    const dim3 grid((unsigned)std::min<int64_t>(ntiles, j->ctx->num_cus));
    if (mk) {  // several key columns composed inside the kernel (da_mk_usable: COUNT(*), 2-byte entries, no bitmaps / flags)
        const dim3 grid2((unsigned)std::min<int64_t>(ntiles, (int64_t)j->ctx->num_cus * 2));
        if (st.ebits > 16) hipLaunchKernelGGL((k_da_partition2<512, 8, 4, true, uint32_t, false, true>), grid2, dim3(512), 0, j->ctx->stream, src, j->da_dm, st, *mk);
        else hipLaunchKernelGGL((k_da_partition2<512, 8, 4, true, uint16_t, false, true>), grid2, dim3(512), 0, j->ctx->stream, src, j->da_dm, st, *mk);
        TSQ_HIP(&j->hdr, hipGetLastError());
        j->st.kernel_launches++;
        return TSQ_OK;
    }
    if (with_idx && st.ebits > 16) {  // (4-byte entries with row ids: the bit-cell pairs route; 8 keys per thread — the row ids share the LDS)
        constexpr int K8 = 8, T8 = NT * K8;
        const dim3 grid8((unsigned)std::min<int64_t>((src.nrows + T8 - 1) / T8, j->ctx->num_cus));
        if (miss) hipLaunchKernelGGL((k_da_partition<NT, K8, uint32_t, true, true>), grid8, dim3(NT), 0, j->ctx->stream, src, j->da_dm, st);
        else hipLaunchKernelGGL((k_da_partition<NT, K8, uint32_t, true, false>), grid8, dim3(NT), 0, j->ctx->stream, src, j->da_dm, st);
    } else if (with_idx && miss) hipLaunchKernelGGL((k_da_partition<NT, K, uint16_t, true, true>), grid, dim3(NT), 0, j->ctx->stream, src, j->da_dm, st);
    else if (with_idx) hipLaunchKernelGGL((k_da_partition<NT, K, uint16_t, true, false>), grid, dim3(NT), 0, j->ctx->stream, src, j->da_dm, st);
    else {
        // COUNT(*) routes: two 512-thread workgroups per CU (k_da_partition2: one loads while the other scatters) with non-temporal key
        // loads (3 x A/B in one session, profiles/r03_partition_nt_ab.txt: step 0.362 vs 0.373 ms, and the probe kernel that follows
        // finds more of the entries in cache).  TSQ_KNOB_DA_PARTITION = 1 keeps one 1024-thread workgroup per CU, TSQ_KNOB_DA_NT_LOADS = 0
        // plain loads (A/B measurements).
        const bool wide = st.ebits > 16;
        const int64_t variant = tsq_knob(j->ctx, TSQ_KNOB_DA_PARTITION, 0);
        const bool two = variant == 0 ? !wide : variant == 2;  // (4-byte entries: the default stays the 1024-thread kernel until the A/B below is in)
        const bool ntl = tsq_knob(j->ctx, TSQ_KNOB_DA_NT_LOADS, 1) != 0;
        const dim3 grid2((unsigned)std::min<int64_t>(ntiles, (int64_t)j->ctx->num_cus * 2));
        const bool flags = src.nulls != nullptr || src.sel != nullptr;  // (NULL bitmap / selection flags: the FLAGS instantiations keep the 16-byte-load path)
        if (two && flags && ntl) {
            if (wide) hipLaunchKernelGGL((k_da_partition2<512, 8, 4, true, uint32_t, true>), grid2, dim3(512), 0, j->ctx->stream, src, j->da_dm, st);
            else hipLaunchKernelGGL((k_da_partition2<512, 8, 4, true, uint16_t, true>), grid2, dim3(512), 0, j->ctx->stream, src, j->da_dm, st);
        } else if (two && wide) {
            if (ntl) hipLaunchKernelGGL((k_da_partition2<512, 8, 4, true, uint32_t>), grid2, dim3(512), 0, j->ctx->stream, src, j->da_dm, st);
            else hipLaunchKernelGGL((k_da_partition2<512, 8, 4, false, uint32_t>), grid2, dim3(512), 0, j->ctx->stream, src, j->da_dm, st);
        } else if (two) {
            if (ntl) hipLaunchKernelGGL((k_da_partition2<512, 8, 4, true>), grid2, dim3(512), 0, j->ctx->stream, src, j->da_dm, st);
            else hipLaunchKernelGGL((k_da_partition2<512, 8, 4>), grid2, dim3(512), 0, j->ctx->stream, src, j->da_dm, st);
        } else if (wide) hipLaunchKernelGGL((k_da_partition<NT, K, uint32_t>), grid, dim3(NT), 0, j->ctx->stream, src, j->da_dm, st);
        else hipLaunchKernelGGL((k_da_partition<NT, K, uint16_t>), grid, dim3(NT), 0, j->ctx->stream, src, j->da_dm, st);
    }
    TSQ_HIP(&j->hdr, hipGetLastError());
    j->st.kernel_launches++;
    return TSQ_OK;
}

// several key columns ride the packed routes as ONE composite column (tsq_dajoin.h, k_da_compose) when all of them are integers
bool da_multi_ok(const tsq_join* j) {
    if (!j->multi || j->ks.n_keys > TSQ_DA_MAXKEYS) return false;
    for (int k = 0; k < j->ks.n_keys; k++)
        if (!is_int_class(j->cfg.build_types[j->ks.bidx[k]]) || !is_int_class(j->cfg.probe_types[j->ks.pidx[k]])) return false;
    return true;
}
// the key column the packed kernels partition: the build side's ...
void da_build_key(const tsq_join* j, DaSrc& src) {
    memset(&src, 0, sizeof src);
    const int kc = j->ks.bidx[0];
    src.nrows = j->bcols[kc].rows;
    if (j->da_multi) {
        src.data = j->da_ckey.as<uint64_t>();
        return;
    }
    src.data = j->bcols[kc].data.as<uint64_t>();
    src.nulls = j->bcols[kc].has_nulls ? j->bcols[kc].nulls.as<uint8_t>() : nullptr;
}
// ... and a probe batch's (composed into j->rckey first when the key has several columns)
// may the partition kernel compose the key columns itself?  The COUNT(*) kernel with two workgroups per CU, columns
// without NULL bitmaps on 16-byte boundaries, no selection flags; knob DA_PARTITION = 1 (the 1024-thread kernel) keeps k_da_compose
bool da_mk_usable(const tsq_join* j, const tsq_colset& pcs, const DaStore& st, const uint8_t* sel, DaMk& mk) {
    (void)st;
    if (!j->da_multi || sel != nullptr || tsq_knob(j->ctx, TSQ_KNOB_DA_PARTITION, 0) != 0 || tsq_knob(j->ctx, TSQ_KNOB_DA_NT_LOADS, 1) == 0) return false;
    memset(&mk, 0, sizeof mk);
    mk.f = j->da_fields;
    for (int k = 0; k < j->ks.n_keys; k++) {
        const int c = j->ks.pidx[k];
        if (pcs.nulls[c] != nullptr || ((uintptr_t)pcs.data[c] & 15u)) return false;
        mk.col[k] = (const uint64_t*)pcs.data[c];
    }
    return true;
}
tsq_status da_probe_key(tsq_join* j, const tsq_colset& pcs, int64_t nrows, DaSrc& src, const uint8_t* sel = nullptr) {
    memset(&src, 0, sizeof src);
    src.nrows = nrows;
    src.sel = sel;
    if (!j->da_multi) {
        const int kc = j->ks.pidx[0];
        src.data = (const uint64_t*)pcs.data[kc];
        src.nulls = pcs.nulls[kc];
        return TSQ_OK;
    }
    TSQ_TRY(j->rckey.reserve(j->ctx, &j->hdr, (size_t)nrows * 8 + 64));
    DaComposeArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.f = j->da_fields;
    for (int k = 0; k < j->ks.n_keys; k++) {
        ca.col[k] = (const uint64_t*)pcs.data[j->ks.pidx[k]];
        ca.nulls[k] = pcs.nulls[j->ks.pidx[k]];
    }
    ca.nrows = nrows;
    ca.out = j->rckey.as<uint64_t>();
    hipLaunchKernelGGL(k_da_compose, dim3(tsq_grid_for(j->ctx, nrows, 256)), dim3(256), 0, j->ctx->stream, ca);
    TSQ_HIP(&j->hdr, hipGetLastError());
    j->st.kernel_launches++;
    src.data = j->rckey.as<uint64_t>();
    return TSQ_OK;
}
// the fields of a several-column key from the build side's columns, and the build side's composite column
tsq_status da_compose_build(tsq_join* j, bool* ok, uint32_t max_total_bits = TSQ_DA_MAX_BITS) {
    *ok = false;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const int nk = j->ks.n_keys;
    const int64_t nb = j->bcols[j->ks.bidx[0]].rows;
    DaFields& f = j->da_fields;
    memset(&f, 0, sizeof f);
    f.n = nk;
    uint32_t total = 0;
    for (int k = 0; k < nk; k++) {
        const int c = j->ks.bidx[k];
        const int32_t bt = j->cfg.build_types[c], pt = j->cfg.probe_types[j->ks.pidx[k]];
        DaMinMaxArgs ma;
        memset(&ma, 0, sizeof ma);
        ma.src.data = j->bcols[c].data.as<uint64_t>();
        ma.src.nulls = j->bcols[c].has_nulls ? j->bcols[c].nulls.as<uint8_t>() : nullptr;
        ma.src.nrows = nb;
        ma.flip = (bt == TSQ_I64 && pt == TSQ_I64) ? 0x8000000000000000ULL : 0ULL;
        ma.skip_high = bt != pt ? 1 : 0;  // BIGINT against BIGINT UNSIGNED: cells >= 2^63 never match (codec.go:219-224)
        ma.out = (unsigned long long*)(ctx->dscratch + 48);
        ctx->pinned[48] = ~0ULL;
        ctx->pinned[49] = 0;
        ctx->pinned[50] = 0;
        TSQ_HIP(h, hipMemcpyAsync(ctx->dscratch + 48, ctx->pinned + 48, 24, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_da_minmax, dim3(tsq_grid_for(ctx, nb, 256)), dim3(256), 0, ctx->stream, ma);
        TSQ_HIP(h, hipGetLastError());
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 48, ctx->dscratch + 48, 24, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        j->st.kernel_launches++;
        if (ctx->pinned[50] == 0) return TSQ_OK;  // no usable cell in this column: nothing can match, the direct route says so
        const uint64_t kmin = ctx->pinned[48] ^ ma.flip, range = (ctx->pinned[49] ^ ma.flip) - kmin;
        if (max_total_bits < 64 && (range >> max_total_bits)) return TSQ_OK;
        uint32_t w = 0;
        while (w < 64 && (range >> w) != 0) w++;
        f.kmin[k] = kmin;
        f.maxd[k] = range;
        f.shift[k] = total;
        f.skip_high[k] = ma.skip_high;
        total += w;
        if (total > max_total_bits) return TSQ_OK;
    }
    TSQ_TRY(j->da_ckey.reserve(ctx, h, (size_t)nb * 8 + 64));
    DaComposeArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.f = f;
    for (int k = 0; k < nk; k++) {
        ca.col[k] = j->bcols[j->ks.bidx[k]].data.as<uint64_t>();
        ca.nulls[k] = j->bcols[j->ks.bidx[k]].has_nulls ? j->bcols[j->ks.bidx[k]].nulls.as<uint8_t>() : nullptr;
    }
    ca.nrows = nb;
    ca.out = j->da_ckey.as<uint64_t>();
    hipLaunchKernelGGL(k_da_compose, dim3(tsq_grid_for(ctx, nb, 256)), dim3(256), 0, ctx->stream, ca);
    TSQ_HIP(h, hipGetLastError());
    j->st.kernel_launches++;
    *ok = true;
    return TSQ_OK;
}

// sc != nullptr: the build side is SHARDED over the ranks of a communicator (tsq_join_build_finish_shared) — the key range is the
// range over all ranks, every rank assembles the images of ITS rows over that range, and the images are summed across the ranks
// (one all-reduce, once per build side): afterwards every rank holds the images of the WHOLE build side and probes its own probe
// rows locally.  Every rank takes the same decisions: they depend on the configuration and on all-reduced values only.
bool dm_cols_shape(const tsq_join* j);
tsq_status da_prepare(tsq_join* j, tsq_comm* sc = nullptr, tsq_status pre_status = TSQ_OK, bool want_cols = false) {
    if (j->da_state) return TSQ_OK;
    j->da_state = -1;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const bool env_off = tsq_knob(ctx, TSQ_KNOB_PACKED_KEYS, 1) == 0;
    if (env_off || j->packing_mode == TSQ_RADIX_OFF || (j->multi && (sc || !da_multi_ok(j))) || j->never_match) return TSQ_OK;
    const int kc = j->ks.bidx[0];
    const int32_t bt = j->cfg.build_types[kc], pt = j->cfg.probe_types[j->ks.pidx[0]];
    if (!is_int_class(bt) || !is_int_class(pt)) return TSQ_OK;
    const int64_t nb = j->bcols[kc].rows;
    if (!sc && (nb <= 0 || nb >= 0xffffffffLL)) return TSQ_OK;
    const bool force = j->packing_mode == TSQ_RADIX_FORCE;
    const int64_t min_build = tsq_knob(ctx, TSQ_KNOB_DA_MIN_BUILD_ROWS, (int64_t)(1 << 20));  // (experiment knob)
    if (!sc && !force && nb < min_build) return TSQ_OK;
    if (j->multi) {  // several key columns: one composite column, unsigned, ~0 = cannot match
        bool ok = false;
        TSQ_TRY(da_compose_build(j, &ok));
        if (!ok) { j->da_ckey.release(); return TSQ_OK; }
        j->da_multi = true;
    }
    // ---- key range of the build side
    DaMinMaxArgs ma;
    memset(&ma, 0, sizeof ma);
    da_build_key(j, ma.src);
    ma.flip = (!j->da_multi && bt == TSQ_I64 && pt == TSQ_I64) ? 0x8000000000000000ULL : 0ULL;  // mixed signedness: only cells < 2^63 are usable, both orders agree
    ma.skip_high = j->da_multi ? 1 : j->ks.skip_high;
    ma.out = (unsigned long long*)(ctx->dscratch + 48);
    ctx->pinned[48] = ~0ULL;
    ctx->pinned[49] = 0;
    ctx->pinned[50] = 0;
    ctx->pinned[51] = 0;  // [51]: the two flag words of the images kernel
    // a shared build is a COLLECTIVE: a rank that cannot take part (too many rows, a failure before this call — pre_status —, a HIP
    // error below) still joins every all-reduce, and every rank returns an error once the flags have been agreed on (a rank
    // that returned early left the others waiting in RCCL)
    bool local_fail = sc && (nb >= 0xffffffffLL || pre_status != TSQ_OK);
    hipError_t e_mm = hipSuccess;
    if (nb > 0 && !local_fail) {
        e_mm = hipMemcpyAsync(ctx->dscratch + 48, ctx->pinned + 48, 32, hipMemcpyHostToDevice, ctx->stream);
        if (e_mm == hipSuccess) {
            hipLaunchKernelGGL(k_da_minmax, dim3(tsq_grid_for(ctx, nb, 256)), dim3(256), 0, ctx->stream, ma);
            e_mm = hipGetLastError();
        }
        if (e_mm == hipSuccess) e_mm = hipMemcpyAsync(ctx->pinned + 48, ctx->dscratch + 48, 24, hipMemcpyDeviceToHost, ctx->stream);
        if (e_mm == hipSuccess) e_mm = hipStreamSynchronize(ctx->stream);
        j->st.kernel_launches++;
        if (e_mm != hipSuccess) {
            if (!sc) return tsq_fail(h, TSQ_ERR_HIP, std::string("packed-key range: ") + hipGetErrorString(e_mm));
            local_fail = true;
            ctx->pinned[48] = ~0ULL;  // (an empty range: this rank adds nothing to the agreed one)
            ctx->pinned[49] = 0;
            ctx->pinned[50] = 0;
        }
    }
    uint64_t lo_img = ctx->pinned[48], hi_img = ctx->pinned[49], usable = ctx->pinned[50];
    const uint64_t usable_local = usable;
    if (sc) {  // the range and the usable rows over all ranks (images compare unsigned: through int64 with the top bit flipped)
        int64_t mm[2] = {(int64_t)(lo_img ^ 0x8000000000000000ULL), ~(int64_t)(hi_img ^ 0x8000000000000000ULL)};
        TSQ_TRY(tsq_comm_allreduce_host_i64(sc, mm, 2, 2));
        int64_t su[1] = {(int64_t)usable};
        TSQ_TRY(tsq_comm_allreduce_host_i64(sc, su, 1, 0));
        lo_img = (uint64_t)mm[0] ^ 0x8000000000000000ULL;
        hi_img = (uint64_t)(~mm[1]) ^ 0x8000000000000000ULL;
        usable = (uint64_t)su[0];
        if (!force && (int64_t)usable < min_build) return TSQ_OK;
    }
    if (usable == 0) return TSQ_OK;
    const int pb_env = (int)tsq_knob(ctx, TSQ_KNOB_DA_PBITS, -1);  // (experiment knob)
    // 29..31 bits: one BIT per cell instead of one byte (k_da_build_bits) — only a build side WITHOUT duplicate keys fits that
    // (the images kernel finds out); COUNT(*) route only.  The arithmetic is host-only: tsq_da_plan (tsq_dapack.h)
    // (bit cells also serve a materialising join of a unique build side: the bit-cell pairs route, da_emit_bits)
    const DaPlan pl = tsq_da_plan(lo_img ^ ma.flip, hi_img ^ ma.flip, usable, ma.skip_high, true, force, pb_env);
    if (!pl.ok) return TSQ_OK;
    const bool bits_mode = pl.bit_cells != 0;
    j->da_bits = bits_mode;
    j->da_pbits = pl.pbits;
    j->da_ebits = pl.ebits;
    j->da_dm = pl.dm;
    // ---- partition the build keys, assemble the images
    const size_t img_bytes = (size_t)tsq_da_image_bytes(pl);
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nb, 1024 * 16);
    if (g.nregions * g.cap >= 0xffffffffULL) {
        if (!sc) return TSQ_OK;
        local_fail = true;
    }
    DevBuf ent, ctl, vend, ovf, ovfi, nnm, pay[TSQ_DA_MAXCOLS];
    auto release_all = [&]() {
        for (DevBuf* x : {&ent, &ctl, &vend, &ovf, &ovfi, &nnm}) x->release();
        for (auto& b : pay) b.release();
    };
    // a materialising join whose build side may go into LDS (tsq_damat.h): the partition pass takes the build COLUMNS along and its
    // store is kept for dm_prepare_build (k_da_partition_cols in place of k_da_partition2: + 16 B per row here, - 26 B per row there)
    const bool cols_l1 = want_cols && !sc && !bits_mode && nb > 0 && pl.ebits <= 16 && pl.ebits >= 7 && dm_cols_shape(j);
    int l1_cols[TSQ_DA_MAXCOLS], l1_n = 0;
    bool l1_nulls = false;
    if (cols_l1) {
        const int kb = j->da_multi ? -1 : kc;
        for (int c = 0; c < j->cfg.n_build_cols; c++)
            if (c != kb) {
                l1_cols[l1_n++] = c;
                l1_nulls = l1_nulls || j->bcols[c].has_nulls;
            }
    }
    tsq_status s = TSQ_OK;
    hipError_t e = hipSuccess;
    if (!local_fail) {
        s = j->da_img.reserve(ctx, h, img_bytes + 64);
        if (s == TSQ_OK && nb > 0) s = ent.reserve(ctx, h, g.ent_bytes);
        if (s == TSQ_OK && nb > 0) s = ctl.reserve(ctx, h, g.ctl_bytes);
        if (s == TSQ_OK && nb > 0) s = vend.reserve(ctx, h, g.nregions * 4);
        if (s == TSQ_OK && nb > 0) s = ovf.reserve(ctx, h, (size_t)nb * 4 + 64);
        if (cols_l1) {
            if (s == TSQ_OK) s = ovfi.reserve(ctx, h, (size_t)nb * 4 + 64);
            for (int v = 0; v < l1_n && s == TSQ_OK; v++) s = pay[v].reserve(ctx, h, g.nregions * g.cap * 8 + 256);
            if (s == TSQ_OK && l1_nulls) s = nnm.reserve(ctx, h, g.nregions * g.cap + 256);
        }
        if (s != TSQ_OK && !sc) { release_all(); j->da_img.release(); return s; }
    }
    if (!local_fail && s == TSQ_OK && nb == 0) {  // (a rank of a shared build side without rows: its images are zeros)
        e = hipMemsetAsync(j->da_img.p, 0, img_bytes, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    } else if (!local_fail && s == TSQ_OK) {
        DaStore st;
        memset(&st, 0, sizeof st);
        st.ent = ent.p;
        st.cursor = ctl.as<uint32_t>();
        st.ovf_count = st.cursor + g.nregions;
        st.valid_end = vend.as<uint32_t>();
        st.ovf = ovf.as<uint32_t>();
        st.ovf_cap = (uint32_t)nb;
        st.bits = j->da_pbits;
        st.ebits = j->da_ebits;
        st.cap = g.cap;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(ctl.p, 0, g.ctl_bytes, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(vend.p, 0xff, g.nregions * 4, ctx->stream);
        if (e == hipSuccess && cols_l1) {
            DaColStore cs;
            memset(&cs, 0, sizeof cs);
            cs.st = st;
            cs.st.miss_count = st.cursor + g.nregions + 1;
            cs.st.ovf_idx = ovfi.as<uint32_t>();
            for (int v = 0; v < l1_n; v++) cs.pay[v] = pay[v].as<uint64_t>();
            cs.nnmask = l1_nulls ? nnm.as<uint8_t>() : nullptr;
            DaColSrc src;
            memset(&src, 0, sizeof src);
            src.key = ma.src;
            src.n_cols = l1_n;
            src.any_nulls = l1_nulls ? 1 : 0;
            for (int v = 0; v < l1_n; v++) {
                src.col[v] = j->bcols[l1_cols[v]].data.as<uint64_t>();
                src.nulls[v] = j->bcols[l1_cols[v]].has_nulls ? j->bcols[l1_cols[v]].nulls.as<uint8_t>() : nullptr;
            }
            constexpr int TC = 1024 * 8;
            hipLaunchKernelGGL((k_da_partition_cols<1024, 8, false>), dim3((unsigned)std::min<int64_t>((nb + TC - 1) / TC, ctx->num_cus)), dim3(1024), 0, ctx->stream, src, j->da_dm, cs);
            e = hipGetLastError();
            j->st.kernel_launches++;
        } else if (e == hipSuccess) s = da_launch_partition(j, ma.src, st);
        DaImageArgs ia;
        memset(&ia, 0, sizeof ia);
        ia.st = st;
        ia.img = j->da_img.as<uint8_t>();
        ia.flags = (uint32_t*)(ctx->dscratch + 51);
        const size_t img_lds = bits_mode ? ((size_t)1 << j->da_ebits) / 8 : ((size_t)1 << j->da_ebits);
        if (e == hipSuccess && s == TSQ_OK) {
            if (bits_mode) {
                e = hipFuncSetAttribute((const void*)k_da_build_bits<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds);
                if (e == hipSuccess) hipLaunchKernelGGL((k_da_build_bits<1024>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus)), dim3(1024), img_lds, ctx->stream, ia);
            } else if (j->da_ebits > 16) {
                e = hipFuncSetAttribute((const void*)k_da_build_images<1024, uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds);
                if (e == hipSuccess) hipLaunchKernelGGL((k_da_build_images<1024, uint32_t>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus)), dim3(1024), img_lds, ctx->stream, ia);
            } else {
                e = hipFuncSetAttribute((const void*)k_da_build_images<512, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds);
                if (e == hipSuccess) hipLaunchKernelGGL((k_da_build_images<512, uint16_t>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus * 2)), dim3(512), img_lds, ctx->stream, ia);
            }
            if (e == hipSuccess) e = hipGetLastError();
        }
        if (e == hipSuccess && s == TSQ_OK) {
            if (bits_mode) hipLaunchKernelGGL(k_da_build_bits_ovf, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, ia);
            else hipLaunchKernelGGL(k_da_build_ovf, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, ia);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 51, ctx->dscratch + 51, 8, hipMemcpyDeviceToHost, ctx->stream);
        ctx->pinned[52] = 0;
        if (e == hipSuccess && cols_l1) e = hipMemcpyAsync(ctx->pinned + 52, ctl.as<uint32_t>() + g.nregions, 4, hipMemcpyDeviceToHost, ctx->stream);  // rows that missed their region
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        float ms = 0;
        if (e == hipSuccess && e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) j->da_build_ms = ms;
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        j->st.kernel_launches += 2;
    }
    uint32_t f_over = ((const uint32_t*)(ctx->pinned + 51))[0], f_dup = ((const uint32_t*)(ctx->pinned + 51))[1];
    if (cols_l1 && !local_fail && s == TSQ_OK && e == hipSuccess && !f_over && !f_dup && ((const uint32_t*)(ctx->pinned + 52))[0] == 0) {  // a unique build side, every row in its region: this level 1 is dm_prepare_build's
        auto hand = [](DevBuf& to, DevBuf& from) {
            to.release();
            to = from;
            from.p = nullptr;
            from.cap = 0;
        };
        hand(j->dm_l1ent, ent);
        hand(j->dm_l1ctl, ctl);
        hand(j->dm_l1vend, vend);
        hand(j->dm_l1nn, nnm);
        for (int v = 0; v < l1_n; v++) hand(j->dm_l1pay[v], pay[v]);
        j->dm_l1_cap = g.cap;
        j->dm_l1_nulls = l1_nulls;
        j->dm_l1_ready = true;
    }
    release_all();
    if (sc) {  // one more agreement: did every rank get its images, did any rank's cell overflow / hold a duplicate (bit cells)
        int64_t fl[3] = {(local_fail || s != TSQ_OK || e != hipSuccess) ? 1 : 0, (int64_t)(f_over != 0), (int64_t)(bits_mode && f_dup != 0)};
        const tsq_status cs = tsq_comm_allreduce_host_i64(sc, fl, 3, 1);
        if (s == TSQ_OK && e == hipSuccess && cs != TSQ_OK) s = cs;
        if (fl[0] || fl[1] || fl[2]) {
            j->da_img.release();
            if (pre_status != TSQ_OK) return pre_status;
            if (s != TSQ_OK) return s;
            if (e_mm != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("packed-key range: ") + hipGetErrorString(e_mm));
            if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("packed-key images: ") + hipGetErrorString(e));
            // a PEER failed: this rank is healthy, but the plan (shared images or the exchange) must stop on every rank together
            if (fl[0]) return tsq_fail(h, TSQ_ERR_INVALID, "shared build side: another rank failed before its images were ready");
            return TSQ_OK;
        }
        // ---- the images of all ranks, summed: byte cells as bytes, bit cells as 32-bit words (tsq_dapack.h: what the sum can do wrong
        // shows in its population)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        const auto t0 = std::chrono::steady_clock::now();
        tsq_status as = tsq_comm_allreduce_dev_sum(sc, j->da_img.p, bits_mode ? img_bytes / 4 : img_bytes, bits_mode ? 4 : 1);
        j->shared_allreduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (as != TSQ_OK) {
            tsq_fail(h, as, "shared build side: all-reduce of the images failed");
            j->da_img.release();
            return as;
        }
        DaImageCheckArgs ca;
        memset(&ca, 0, sizeof ca);
        ca.img = (const uint4*)j->da_img.p;
        ca.n16 = img_bytes / 16;
        ca.bits = bits_mode ? 1 : 0;
        ca.out = (unsigned long long*)(ctx->dscratch + 52);
        e = hipMemsetAsync(ctx->dscratch + 52, 0, 16, ctx->stream);
        if (e == hipSuccess && e0) e = hipEventRecord(e0, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_da_image_check, dim3(tsq_grid_for(ctx, (int64_t)ca.n16, 256)), dim3(256), 0, ctx->stream, ca);
            e = hipGetLastError();
        }
        if (e == hipSuccess && e1) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 52, ctx->dscratch + 52, 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        float ms = 0;
        if (e == hipSuccess && e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) j->da_build_ms += ms;
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        j->st.kernel_launches++;
        if (e != hipSuccess) { j->da_img.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("shared build side, image check: ") + hipGetErrorString(e)); }
        if (!tsq_da_shared_images_ok(ctx->pinned[52], usable)) {  // a cell passed 255 / a key lives on two ranks (bit cells): identical on every rank
            j->da_img.release();
            return TSQ_OK;
        }
        j->da_unique = ctx->pinned[53] == 0;
        j->shared_image_bytes = (int64_t)img_bytes;
        j->shared_usable_local = (int64_t)usable_local;
        j->da_state = 1;
        return TSQ_OK;
    }
    if (s != TSQ_OK) { j->da_img.release(); return s; }
    if (e != hipSuccess) { j->da_img.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("packed-key images: ") + hipGetErrorString(e)); }
    if (f_over || (bits_mode && f_dup)) {  // a key with more than 255 build rows (bit cells: with more than one): the 64-bit route keeps this join
        j->da_img.release();
        return TSQ_OK;
    }
    j->da_unique = f_dup == 0;
    j->da_state = 1;
    return TSQ_OK;
}
// every column on both sides an 8-byte type, at most TSQ_DA_MAXCOLS per side (what the travelling-columns routes take)
bool dm_cols_shape(const tsq_join* j) {
    if (tsq_knob(j->ctx, TSQ_KNOB_DA_LDS_BUILD, 1) == 0 || j->count_only) return false;
    if (j->cfg.n_probe_cols > TSQ_DA_MAXCOLS || j->cfg.n_build_cols > TSQ_DA_MAXCOLS) return false;
    for (int c = 0; c < j->cfg.n_probe_cols; c++)
        if (j->cfg.probe_types[c] == TSQ_F32 || j->cfg.probe_types[c] == TSQ_BYTES) return false;
    for (int c = 0; c < j->cfg.n_build_cols; c++)
        if (j->cfg.build_types[c] == TSQ_F32 || j->cfg.build_types[c] == TSQ_BYTES) return false;
    return true;
}

tsq_status da_probe(tsq_join* j, const tsq_colset& pcs, int64_t nrows, const uint8_t* sel = nullptr) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nrows, 1024 * 16);
    if (g.nregions * g.cap >= 0xffffffffULL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "radix probe batch too large");
    TSQ_TRY(j->rkeys.reserve(ctx, h, g.ent_bytes));
    TSQ_TRY(j->rctl.reserve(ctx, h, g.ctl_bytes));
    TSQ_TRY(j->rvend.reserve(ctx, h, g.nregions * 4));
    TSQ_TRY(j->rovf.reserve(ctx, h, (size_t)nrows * 4 + 64));
    DaStore st;
    memset(&st, 0, sizeof st);
    st.ent = j->rkeys.p;
    st.cursor = j->rctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.valid_end = j->rvend.as<uint32_t>();
    st.ovf = j->rovf.as<uint32_t>();
    st.ovf_cap = (uint32_t)nrows;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = g.cap;
    TSQ_HIP(h, hipMemsetAsync(j->rctl.p, 0, g.ctl_bytes, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->rvend.p, 0xff, g.nregions * 4, ctx->stream));
    hipEvent_t* re = j->rev[j->st.radix_batches % tsq_join::RING];
    for (int e = 0; e < 3; e++)
        if (!re[e]) TSQ_HIP(h, hipEventCreate(&re[e]));
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(re[0], ctx->stream));
    DaSrc src;
    DaMk mk;
    if (da_mk_usable(j, pcs, st, sel, mk)) {  // several key columns: composed in the partition kernel's registers (no composite column in HBM)
        memset(&src, 0, sizeof src);
        src.nrows = nrows;
        TSQ_TRY(da_launch_partition(j, src, st, false, false, &mk));
    } else {
        TSQ_TRY(da_probe_key(j, pcs, nrows, src, sel));
        TSQ_TRY(da_launch_partition(j, src, st));
    }
    TSQ_HIP(h, hipEventRecord(re[1], ctx->stream));
    DaProbeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.st = st;
    pa.img = j->da_img.as<uint8_t>();
    pa.counters = j->counters.as<unsigned long long>();
    const size_t img_lds = j->da_bits ? ((size_t)1 << j->da_ebits) / 8 : ((size_t)1 << j->da_ebits);
    if (j->da_bits) {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<1024, uint32_t, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds));
        const uint32_t per_cu = img_lds <= (64u << 10) ? 2u : 1u;
        hipLaunchKernelGGL((k_da_probe_count<1024, uint32_t, false, false, true>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus * per_cu)), dim3(1024), img_lds, ctx->stream, pa);
    } else if (j->da_ebits > 16) {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<1024, uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds));
        hipLaunchKernelGGL((k_da_probe_count<1024, uint32_t>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus)), dim3(1024), img_lds, ctx->stream, pa);
    } else {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<512, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds));
        hipLaunchKernelGGL((k_da_probe_count<512, uint16_t>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus * 2)), dim3(512), img_lds, ctx->stream, pa);
    }
    TSQ_HIP(h, hipGetLastError());
    if (j->da_bits) hipLaunchKernelGGL((k_da_probe_ovf<false, true>), dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    else hipLaunchKernelGGL(k_da_probe_ovf<false>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
    j->have_probe_ev = true;
    j->st.kernel_launches += 2;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)st.bits;
    j->st.probe_route = TSQ_ROUTE_PACKED;
    j->st.packed_key_bits = (int32_t)j->da_dm.b;
    return TSQ_OK;
}

// ---- materialising packed route (K4d): which rows join, as (probe row, build row) pairs; the columns follow through the pairs
// Eligible: inner / left outer / right outer join on ONE integer key with a packable build side (da_prepare), no outer filter,
// no OtherConditions, no selected[], not ordered; any number and type of payload columns, NULLs anywhere.
bool da_emit_eligible(const tsq_join* j, int64_t nrows, const uint8_t* selected_dev) {
    if (j->radix_mode == TSQ_RADIX_OFF || j->packing_mode == TSQ_RADIX_OFF || j->count_only || (j->multi && !da_multi_ok(j)) || j->never_match || j->ordered) return false;
    if (!j->conds_h.empty() || (!j->filters_h.empty() && !j->filters_folded)) return false;  // (selected[]: the packed kernels treat a row with selected == 0 like a NULL key)
    if (nrows <= 0 || nrows > 0x7fffffffLL || j->da_state < 0 || j->da_rows_state < 0) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE || j->packing_mode == TSQ_RADIX_FORCE) return true;
    // AUTO: the pairs come out in PARTITION order, so the gather of the probe-side columns is as random as the build side's (the
    // direct route emits in probe order: 9 ms vs 16 ms per 1e8 x 1e8 (k, v) rows) — until the payload columns travel with the
    // entries this route is taken on request only
    const bool env_on = tsq_knob(j->ctx, TSQ_KNOB_PACKED_EMIT_PAIRS, 0) != 0;
    return env_on && nrows >= (4 << 20);
}

// the build rows sorted by word + the coarse ranks (once per build side)
tsq_status da_prepare_rows(tsq_join* j) {
    if (j->da_rows_state) return TSQ_OK;
    j->da_rows_state = -1;
    if (j->da_state != 1 || j->da_ebits > 16 || j->da_ebits < 5) return TSQ_OK;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const int kc = j->ks.bidx[0];
    const int64_t nb = j->bcols[kc].rows;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nb, 1024 * 16);
    if (g.nregions * g.cap >= 0xffffffffULL) return TSQ_OK;
    DevBuf ent, idx, ctl, vend, ovf, ovfi;
    auto release_all = [&]() {
        for (DevBuf* x : {&ent, &idx, &ctl, &vend, &ovf, &ovfi}) x->release();
    };
    tsq_status s = ent.reserve(ctx, h, g.ent_bytes);
    if (s == TSQ_OK) s = idx.reserve(ctx, h, g.nregions * g.cap * 4 + 256);
    if (s == TSQ_OK) s = ctl.reserve(ctx, h, g.ctl_bytes);
    if (s == TSQ_OK) s = vend.reserve(ctx, h, g.nregions * 4);
    if (s == TSQ_OK) s = ovf.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s == TSQ_OK) s = ovfi.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s == TSQ_OK) s = j->da_coarse.reserve(ctx, h, (((size_t)1 << j->da_dm.b) >> 5) * 4 + 64);
    if (s == TSQ_OK) s = j->da_pstart.reserve(ctx, h, ((size_t)g.P + 1) * 4 + 64);
    if (s == TSQ_OK) s = j->da_brows.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s != TSQ_OK) { release_all(); return s; }
    DaStore st;
    memset(&st, 0, sizeof st);
    st.ent = ent.p;
    st.idx = idx.as<uint32_t>();
    st.cursor = ctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.valid_end = vend.as<uint32_t>();
    st.ovf = ovf.as<uint32_t>();
    st.ovf_idx = ovfi.as<uint32_t>();
    st.ovf_cap = (uint32_t)nb;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = g.cap;
    DaSrc src;
    da_build_key(j, src);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctl.p, 0, g.ctl_bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(vend.p, 0xff, g.nregions * 4, ctx->stream);
    if (e == hipSuccess) s = da_launch_partition(j, src, st, true, false);
    if (e == hipSuccess && s == TSQ_OK) {
        hipLaunchKernelGGL(k_da_part_starts, dim3(1), dim3(1024), 0, ctx->stream, st, j->da_pstart.as<uint32_t>());
        e = hipGetLastError();
    }
    DaRowsArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.st = st;
    ra.img = j->da_img.as<uint8_t>();
    ra.coarse = j->da_coarse.as<uint32_t>();
    ra.pstart = j->da_pstart.as<uint32_t>();
    ra.brows = j->da_brows.as<uint32_t>();
    const size_t cells = (size_t)1 << j->da_ebits, lds = 2 * cells + (cells >> 5) * 4;
    if (e == hipSuccess && s == TSQ_OK) e = hipFuncSetAttribute((const void*)k_da_build_rows<1024, uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && s == TSQ_OK) {
        hipLaunchKernelGGL((k_da_build_rows<1024, uint16_t>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus)), dim3(1024), lds, ctx->stream, ra);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 52, st.ovf_count, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    float ms = 0;
    if (e == hipSuccess && e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) j->da_build_ms += ms;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    release_all();
    j->st.kernel_launches += 3;
    if (s != TSQ_OK) return s;
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("packed-key build rows: ") + hipGetErrorString(e));
    if (((const uint32_t*)(ctx->pinned + 52))[0] != 0) {  // skewed build keys: some rows missed their region — keep the other routes
        for (DevBuf* b : {&j->da_coarse, &j->da_pstart, &j->da_brows}) b->release();
        return TSQ_OK;
    }
    j->da_rows_state = 1;
    return TSQ_OK;
}

template <class F>
tsq_status materialise_pairs(tsq_join* j, const tsq_colset& pcs, ProbeArgs& a, int64_t nrows, int64_t out_rows, F&& produce_pairs);

tsq_status da_emit(tsq_join* j, const tsq_colset& pcs, ProbeArgs& a, int64_t nrows, const uint8_t* sel = nullptr) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const bool outer = j->cfg.join_type != TSQ_JOIN_INNER;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nrows, 1024 * 16);
    if (g.nregions * g.cap >= 0xffffffffULL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "radix probe batch too large");
    TSQ_TRY(j->rkeys.reserve(ctx, h, g.ent_bytes));
    TSQ_TRY(j->ridx.reserve(ctx, h, g.nregions * g.cap * 4 + 256));
    TSQ_TRY(j->rctl.reserve(ctx, h, g.ctl_bytes));
    TSQ_TRY(j->rvend.reserve(ctx, h, g.nregions * 4));
    TSQ_TRY(j->rovf.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_TRY(j->rovfidx.reserve(ctx, h, (size_t)nrows * 4 + 64));
    if (outer) TSQ_TRY(j->rmiss.reserve(ctx, h, (size_t)nrows * 4 + 64));
    const size_t n_pc = (size_t)g.P + 1;  // [partition], then one more word: the exclusive scan leaves the total there
    TSQ_TRY(j->tkcnt.reserve(ctx, h, n_pc * 8 + 64));
    DaStore st;
    memset(&st, 0, sizeof st);
    st.ent = j->rkeys.p;
    st.idx = j->ridx.as<uint32_t>();
    st.cursor = j->rctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.miss_count = st.cursor + g.nregions + 1;
    st.miss = j->rmiss.as<uint32_t>();
    st.valid_end = j->rvend.as<uint32_t>();
    st.ovf = j->rovf.as<uint32_t>();
    st.ovf_idx = j->rovfidx.as<uint32_t>();
    st.ovf_cap = (uint32_t)nrows;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = g.cap;
    TSQ_HIP(h, hipMemsetAsync(j->rctl.p, 0, g.ctl_bytes, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->rvend.p, 0xff, g.nregions * 4, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->tkcnt.p, 0, n_pc * 8, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(ctx->dscratch + 52, 0, 16, ctx->stream));  // [52] joined rows of the overflow list, [53] its output cursor
    hipEvent_t* re = j->rev[j->st.radix_batches % tsq_join::RING];
    for (int e = 0; e < 3; e++)
        if (!re[e]) TSQ_HIP(h, hipEventCreate(&re[e]));
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(re[0], ctx->stream));
    DaSrc src;
    TSQ_TRY(da_probe_key(j, pcs, nrows, src, sel));
    TSQ_TRY(da_launch_partition(j, src, st, true, outer));
    TSQ_HIP(h, hipEventRecord(re[1], ctx->stream));
    // ---- sizing pass: output rows per partition, their exclusive scan
    DaProbeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.st = st;
    pa.img = j->da_img.as<uint8_t>();
    pa.counters = (unsigned long long*)(ctx->dscratch + 52);
    pa.pcount = j->tkcnt.as<unsigned long long>();
    const size_t cells = (size_t)1 << j->da_ebits;
    const dim3 pgrid(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus * 2));
    if (outer) {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<512, uint16_t, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells));
        hipLaunchKernelGGL((k_da_probe_count<512, uint16_t, true, true>), pgrid, dim3(512), cells, ctx->stream, pa);
        TSQ_HIP(h, hipGetLastError());
        hipLaunchKernelGGL(k_da_probe_ovf<true>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    } else {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<512, uint16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells));
        hipLaunchKernelGGL((k_da_probe_count<512, uint16_t, true, false>), pgrid, dim3(512), cells, ctx->stream, pa);
        TSQ_HIP(h, hipGetLastError());
        hipLaunchKernelGGL(k_da_probe_ovf<false>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    }
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, pa.pcount, (int)n_pc);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 52, pa.pcount + g.P, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 53, ctx->dscratch + 52, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 54, st.miss_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    const int64_t part_rows = (int64_t)ctx->pinned[52], ovf_rows = (int64_t)ctx->pinned[53];
    const int64_t miss_rows = outer ? (int64_t)((const uint32_t*)(ctx->pinned + 54))[0] : 0;
    const int64_t out_rows = part_rows + ovf_rows + miss_rows;
    j->st.kernel_launches += 3;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)st.bits;
    j->st.probe_route = TSQ_ROUTE_PACKED;
    j->st.packed_key_bits = (int32_t)j->da_dm.b;
    if (out_rows == 0) {
        TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
        TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
        j->have_probe_ev = true;
        return TSQ_OK;
    }
    return materialise_pairs(j, pcs, a, nrows, out_rows, [&]() -> tsq_status {
        DaEmitArgs ea;
        memset(&ea, 0, sizeof ea);
        ea.st = st;
        ea.img = pa.img;
        ea.coarse = j->da_coarse.as<uint32_t>();
        ea.pstart = j->da_pstart.as<uint32_t>();
        ea.brows = j->da_brows.as<uint32_t>();
        ea.pbase = pa.pcount;
        ea.pairs = a.pairs;
        ea.ovf_cursor = (unsigned long long*)(ctx->dscratch + 53);
        ctx->pinned[55] = (uint64_t)part_rows;
        TSQ_HIP(h, hipMemcpyAsync(ctx->dscratch + 53, ctx->pinned + 55, 8, hipMemcpyHostToDevice, ctx->stream));
        const size_t lds = cells + (cells >> 5) * 4;
        if (outer) {
            TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_emit_pairs<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_da_emit_pairs<512, true>), pgrid, dim3(512), lds, ctx->stream, ea);
            TSQ_HIP(h, hipGetLastError());
            hipLaunchKernelGGL(k_da_emit_ovf<true>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, ea);
            TSQ_HIP(h, hipGetLastError());
            if (miss_rows) {
                hipLaunchKernelGGL(k_da_emit_miss, dim3(tsq_grid_for(ctx, miss_rows, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)st.miss, (uint32_t)miss_rows,
                                   a.pairs + part_rows + ovf_rows);
                TSQ_HIP(h, hipGetLastError());
            }
        } else {
            TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_emit_pairs<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_da_emit_pairs<512, false>), pgrid, dim3(512), lds, ctx->stream, ea);
            TSQ_HIP(h, hipGetLastError());
            hipLaunchKernelGGL(k_da_emit_ovf<false>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, ea);
            TSQ_HIP(h, hipGetLastError());
        }
        TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
        j->st.kernel_launches += 3;
        return TSQ_OK;
    });
}

// ---- pairs route on BIT cells (tsq_dajoin.h): a UNIQUE build side whose key range needs 4-byte entries (28..30 bits)
bool da_bits_emit_eligible(const tsq_join* j, int64_t nrows) {
    if (j->radix_mode == TSQ_RADIX_OFF || j->packing_mode == TSQ_RADIX_OFF || j->count_only || j->multi || j->never_match || j->ordered) return false;
    if (!j->conds_h.empty() || (!j->filters_h.empty() && !j->filters_folded)) return false;
    if (nrows <= 0 || nrows > 0x7fffffffLL || j->da_state < 0 || j->da_bitrows_state < 0) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE || j->packing_mode == TSQ_RADIX_FORCE) return true;
    return nrows >= (4 << 20);
}
tsq_status da_prepare_rows_bits(tsq_join* j) {
    if (j->da_bitrows_state) return TSQ_OK;
    j->da_bitrows_state = -1;
    if (j->da_state != 1 || !j->da_unique || j->da_ebits <= 16 || j->da_ebits > 19) return TSQ_OK;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const int kc = j->ks.bidx[0];
    const int64_t nb = j->bcols[kc].rows;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nb, 1024 * 8);
    if (g.nregions * g.cap >= 0xffffffffULL) return TSQ_OK;
    const size_t nwords = ((size_t)1 << j->da_dm.b) >> 5;
    DevBuf ent, idx, ctl, vend, ovf, ovfi;
    auto release_all = [&]() {
        for (DevBuf* x : {&ent, &idx, &ctl, &vend, &ovf, &ovfi}) x->release();
    };
    tsq_status s = ent.reserve(ctx, h, g.ent_bytes);
    if (s == TSQ_OK) s = idx.reserve(ctx, h, g.nregions * g.cap * 4 + 256);
    if (s == TSQ_OK) s = ctl.reserve(ctx, h, g.ctl_bytes);
    if (s == TSQ_OK) s = vend.reserve(ctx, h, g.nregions * 4);
    if (s == TSQ_OK) s = ovf.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s == TSQ_OK) s = ovfi.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s == TSQ_OK && !j->da_bits) s = j->da_bitimg.reserve(ctx, h, nwords * 4 + 64);
    if (s == TSQ_OK) s = j->da_coarse.reserve(ctx, h, nwords * 4 + 64);
    if (s == TSQ_OK) s = j->da_pstart.reserve(ctx, h, ((size_t)g.P + 1) * 4 + 64);
    if (s == TSQ_OK) s = j->da_brows.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s != TSQ_OK) { release_all(); return s; }
    DaStore st;
    memset(&st, 0, sizeof st);
    st.ent = ent.p;
    st.idx = idx.as<uint32_t>();
    st.cursor = ctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.valid_end = vend.as<uint32_t>();
    st.ovf = ovf.as<uint32_t>();
    st.ovf_idx = ovfi.as<uint32_t>();
    st.ovf_cap = (uint32_t)nb;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = g.cap;
    DaSrc src;
    da_build_key(j, src);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess && !j->da_bits) {  // the byte cells of a unique build side are 0 / 1: their bit form
        hipLaunchKernelGGL(k_da_bytes_to_bits, dim3(tsq_grid_for(ctx, (int64_t)nwords, 256)), dim3(256), 0, ctx->stream, j->da_img.as<uint8_t>(), j->da_bitimg.as<uint32_t>(), nwords);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemsetAsync(ctl.p, 0, g.ctl_bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(vend.p, 0xff, g.nregions * 4, ctx->stream);
    if (e == hipSuccess) s = da_launch_partition(j, src, st, true, false);
    if (e == hipSuccess && s == TSQ_OK) {
        hipLaunchKernelGGL(k_da_part_starts, dim3(1), dim3(1024), 0, ctx->stream, st, j->da_pstart.as<uint32_t>());
        e = hipGetLastError();
    }
    DaRowsBitsArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.st = st;
    ra.bits = j->da_bits ? j->da_img.as<uint32_t>() : j->da_bitimg.as<uint32_t>();
    ra.coarse = j->da_coarse.as<uint32_t>();
    ra.pstart = j->da_pstart.as<uint32_t>();
    ra.brows = j->da_brows.as<uint32_t>();
    const size_t lds = ((size_t)1 << j->da_ebits) / 4;  // bits + coarse
    if (e == hipSuccess && s == TSQ_OK) e = hipFuncSetAttribute((const void*)k_da_build_rows_bits<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess && s == TSQ_OK) {
        hipLaunchKernelGGL((k_da_build_rows_bits<1024>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus)), dim3(1024), lds, ctx->stream, ra);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 52, st.ovf_count, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    float ms = 0;
    if (e == hipSuccess && e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) j->da_build_ms += ms;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    release_all();
    j->st.kernel_launches += 4;
    if (s != TSQ_OK) return s;
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("bit-cell build rows: ") + hipGetErrorString(e));
    if (((const uint32_t*)(ctx->pinned + 52))[0] != 0) return TSQ_OK;  // skewed build keys: some rows missed their region — the other routes keep this join
    j->da_bitrows_state = 1;
    return TSQ_OK;
}

tsq_status da_emit_bits(tsq_join* j, const tsq_colset& pcs, ProbeArgs& a, int64_t nrows, const uint8_t* sel) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const bool outer = j->cfg.join_type != TSQ_JOIN_INNER;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nrows, 1024 * 8);
    if (g.nregions * g.cap >= 0xffffffffULL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "radix probe batch too large");
    TSQ_TRY(j->rkeys.reserve(ctx, h, g.ent_bytes));
    TSQ_TRY(j->ridx.reserve(ctx, h, g.nregions * g.cap * 4 + 256));
    TSQ_TRY(j->rctl.reserve(ctx, h, g.ctl_bytes));
    TSQ_TRY(j->rvend.reserve(ctx, h, g.nregions * 4));
    TSQ_TRY(j->rovf.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_TRY(j->rovfidx.reserve(ctx, h, (size_t)nrows * 4 + 64));
    if (outer) TSQ_TRY(j->rmiss.reserve(ctx, h, (size_t)nrows * 4 + 64));
    const size_t n_pc = (size_t)g.P + 1;
    TSQ_TRY(j->tkcnt.reserve(ctx, h, n_pc * 8 + 64));
    DaStore st;
    memset(&st, 0, sizeof st);
    st.ent = j->rkeys.p;
    st.idx = j->ridx.as<uint32_t>();
    st.cursor = j->rctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.miss_count = st.cursor + g.nregions + 1;
    st.miss = j->rmiss.as<uint32_t>();
    st.valid_end = j->rvend.as<uint32_t>();
    st.ovf = j->rovf.as<uint32_t>();
    st.ovf_idx = j->rovfidx.as<uint32_t>();
    st.ovf_cap = (uint32_t)nrows;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = g.cap;
    TSQ_HIP(h, hipMemsetAsync(j->rctl.p, 0, g.ctl_bytes, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->rvend.p, 0xff, g.nregions * 4, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->tkcnt.p, 0, n_pc * 8, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(ctx->dscratch + 52, 0, 16, ctx->stream));
    hipEvent_t* re = j->rev[j->st.radix_batches % tsq_join::RING];
    for (int e = 0; e < 3; e++)
        if (!re[e]) TSQ_HIP(h, hipEventCreate(&re[e]));
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(re[0], ctx->stream));
    DaSrc src;
    TSQ_TRY(da_probe_key(j, pcs, nrows, src, sel));
    TSQ_TRY(da_launch_partition(j, src, st, true, outer));
    TSQ_HIP(h, hipEventRecord(re[1], ctx->stream));
    // ---- sizing pass: output rows per partition (one bit test per entry), their exclusive scan
    const uint32_t* bits = j->da_bits ? j->da_img.as<uint32_t>() : j->da_bitimg.as<uint32_t>();
    DaProbeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.st = st;
    pa.img = (const uint8_t*)bits;
    pa.counters = (unsigned long long*)(ctx->dscratch + 52);
    pa.pcount = j->tkcnt.as<unsigned long long>();
    const size_t img_lds = ((size_t)1 << j->da_ebits) / 8;
    const dim3 pgrid(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus * (img_lds <= (64u << 10) ? 2u : 1u)));
    if (outer) {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<1024, uint32_t, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds));
        hipLaunchKernelGGL((k_da_probe_count<1024, uint32_t, true, true, true>), pgrid, dim3(1024), img_lds, ctx->stream, pa);
        TSQ_HIP(h, hipGetLastError());
        hipLaunchKernelGGL((k_da_probe_ovf<true, true>), dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    } else {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<1024, uint32_t, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_lds));
        hipLaunchKernelGGL((k_da_probe_count<1024, uint32_t, true, false, true>), pgrid, dim3(1024), img_lds, ctx->stream, pa);
        TSQ_HIP(h, hipGetLastError());
        hipLaunchKernelGGL((k_da_probe_ovf<false, true>), dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    }
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, pa.pcount, (int)n_pc);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 52, pa.pcount + g.P, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 53, ctx->dscratch + 52, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 54, st.miss_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    const int64_t part_rows = (int64_t)ctx->pinned[52], ovf_rows = (int64_t)ctx->pinned[53];
    const int64_t miss_rows = outer ? (int64_t)((const uint32_t*)(ctx->pinned + 54))[0] : 0;
    const int64_t out_rows = part_rows + ovf_rows + miss_rows;
    j->st.kernel_launches += 3;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)st.bits;
    j->st.probe_route = TSQ_ROUTE_PACKED;
    j->st.packed_key_bits = (int32_t)j->da_dm.b;
    if (out_rows == 0) {
        TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
        TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
        j->have_probe_ev = true;
        return TSQ_OK;
    }
    return materialise_pairs(j, pcs, a, nrows, out_rows, [&]() -> tsq_status {
        DaEmitBitsArgs ea;
        memset(&ea, 0, sizeof ea);
        ea.st = st;
        ea.bits = bits;
        ea.coarse = j->da_coarse.as<uint32_t>();
        ea.pstart = j->da_pstart.as<uint32_t>();
        ea.brows = j->da_brows.as<uint32_t>();
        ea.pbase = pa.pcount;
        ea.pairs = a.pairs;
        ea.ovf_cursor = (unsigned long long*)(ctx->dscratch + 53);
        ctx->pinned[55] = (uint64_t)part_rows;
        TSQ_HIP(h, hipMemcpyAsync(ctx->dscratch + 53, ctx->pinned + 55, 8, hipMemcpyHostToDevice, ctx->stream));
        const size_t lds = ((size_t)1 << j->da_ebits) / 4;
        const dim3 egrid(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus * (lds <= (64u << 10) ? 2u : 1u)));
        if (outer) {
            TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_emit_pairs_bits<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_da_emit_pairs_bits<1024, true>), egrid, dim3(1024), lds, ctx->stream, ea);
            TSQ_HIP(h, hipGetLastError());
            hipLaunchKernelGGL(k_da_emit_ovf_bits<true>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, ea);
            TSQ_HIP(h, hipGetLastError());
            if (miss_rows) {
                hipLaunchKernelGGL(k_da_emit_miss, dim3(tsq_grid_for(ctx, miss_rows, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)st.miss, (uint32_t)miss_rows,
                                   a.pairs + part_rows + ovf_rows);
                TSQ_HIP(h, hipGetLastError());
            }
        } else {
            TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_emit_pairs_bits<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_da_emit_pairs_bits<1024, false>), egrid, dim3(1024), lds, ctx->stream, ea);
            TSQ_HIP(h, hipGetLastError());
            hipLaunchKernelGGL(k_da_emit_ovf_bits<false>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, ea);
            TSQ_HIP(h, hipGetLastError());
        }
        TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
        j->st.kernel_launches += 3;
        return TSQ_OK;
    });
}

// ---- materialising packed route with travelling columns (K5f + K4e, tsq_dajoin.h): nothing is gathered from all over HBM
// Eligible: what da_emit takes, with every column on both sides an 8-byte type (BIGINT, BIGINT UNSIGNED, DOUBLE) and at most
// TSQ_DA_MAXCOLS columns per side.  NULLs anywhere, inner and outer joins.
bool da_cols_eligible(const tsq_join* j, int64_t nrows, const uint8_t* selected_dev) {
    if (j->radix_mode == TSQ_RADIX_OFF || j->packing_mode == TSQ_RADIX_OFF || j->count_only || (j->multi && !da_multi_ok(j)) || j->never_match || j->ordered) return false;
    // OtherConditions of an INNER join are a filter over the joined rows (joiner.go:351-378: innerJoiner.tryToMatch filters the
    // joined chunk): evaluated on the output batch and compacted (da_post_conditions).  An outer join needs "did ANY match of this
    // outer row pass": with a unique build side an outer row has at most one candidate, and a candidate that fails the conditions
    // turns into the NULL-padded row (onMissMatch, joiner.go:274-281); with duplicate build keys the candidates of an outer row are
    // consecutive rows of the batch and a segmented pass decides (k_outer_segments)
    if (!j->filters_h.empty() && !j->filters_folded) return false;  // (selected[]: the packed kernels treat a row with selected == 0 like a NULL key;
                                                                   //  outer-side filters reach them as such flags: fold_outer_filters)
    if (nrows <= 0 || nrows > 0x7fffffffLL || j->da_state < 0 || j->da_cols_state < 0) return false;
    if (j->cfg.n_probe_cols > TSQ_DA_MAXCOLS || j->cfg.n_build_cols > TSQ_DA_MAXCOLS) return false;
    for (int c = 0; c < j->cfg.n_probe_cols; c++)
        if (j->cfg.probe_types[c] == TSQ_F32 || j->cfg.probe_types[c] == TSQ_BYTES) return false;
    for (int c = 0; c < j->cfg.n_build_cols; c++)
        if (j->cfg.build_types[c] == TSQ_F32 || j->cfg.build_types[c] == TSQ_BYTES) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE || j->packing_mode == TSQ_RADIX_FORCE) return true;
    // AUTO: a batch big enough to pay for a partition pass — and, until the build side is prepared (once per build: ~1.5 ms per 1e8
    // rows and column), big enough next to the build side for that preparation to amortise: a 4 Mi-row probe against 1e8 build rows
    // is better served by the routes that only read the table (the gate used to look at the batch alone)
    if (nrows < (4 << 20)) return false;
    return j->da_cols_state == 1 || j->dm_state == 1 || nrows * 4 >= j->bcols[j->ks.bidx[0]].rows;
}

// Round 4: the build side of the travelling-columns route in one partition pass + one sort pass (tsq_dajoin.h: k_da_coarse,
// k_da_sort_partition) — replaces da_prepare_rows + da_prepare_cols for this route (5.0 -> ~1.5 ms per 1e8 (k, v) rows; the pairs
// route K4d keeps the row-id CSR of da_prepare_rows).
tsq_status da_prepare_cols_direct(tsq_join* j) {
    if (j->da_cols_state) return TSQ_OK;
    j->da_cols_state = -1;
    if (j->da_state != 1 || j->da_bits || j->da_ebits > 16 || j->da_ebits < 5) return TSQ_OK;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const int kb = j->da_multi ? -1 : j->ks.bidx[0];
    const int64_t nb = j->bcols[j->ks.bidx[0]].rows;
    constexpr int T = 1024 * 8;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nb, T);
    if (g.nregions * g.cap >= 0xffffffffULL) return TSQ_OK;
    const size_t slots = g.nregions * g.cap;
    int cols_of[TSQ_DA_MAXCOLS], ncols = 0;  // every build column but the key (the emit kernel recovers the key from the word; a composite key's columns travel)
    bool any_nulls = false;
    for (int c = 0; c < j->cfg.n_build_cols; c++) {
        if (c == kb) continue;
        if (ncols == TSQ_DA_MAXCOLS) return TSQ_OK;
        cols_of[ncols++] = c;
        any_nulls = any_nulls || j->bcols[c].has_nulls;
    }
    DevBuf ent, ctl, vend, ovf, ovfi, nnm, bdup, pay[TSQ_DA_MAXCOLS];
    auto release_all = [&]() {
        for (DevBuf* x : {&ent, &ctl, &vend, &ovf, &ovfi, &nnm, &bdup}) x->release();
        for (auto& b : pay) b.release();
    };
    auto give_up = [&](tsq_status st) {
        release_all();
        for (DevBuf* b : {&j->da_coarse_c, &j->da_pstart_c}) b->release();
        for (int c = 0; c < TSQ_DA_MAXCOLS; c++) {
            j->da_bsorted[c].release();
            j->da_bsorted_nn[c].release();
        }
        return st;
    };
    tsq_status s = ent.reserve(ctx, h, g.ent_bytes);
    if (s == TSQ_OK) s = ctl.reserve(ctx, h, g.ctl_bytes);
    if (s == TSQ_OK) s = vend.reserve(ctx, h, g.nregions * 4);
    if (s == TSQ_OK) s = ovf.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s == TSQ_OK) s = ovfi.reserve(ctx, h, (size_t)nb * 4 + 64);
    for (int v = 0; v < ncols && s == TSQ_OK; v++) s = pay[v].reserve(ctx, h, slots * 8 + 256);
    if (s == TSQ_OK && any_nulls) s = nnm.reserve(ctx, h, slots + 256);
    if (s == TSQ_OK && !j->da_unique) s = bdup.reserve(ctx, h, slots + 256);
    if (s == TSQ_OK) s = j->da_pstart_c.reserve(ctx, h, ((size_t)g.P + 1) * 4 + 64);
    if (s == TSQ_OK) s = j->da_coarse_c.reserve(ctx, h, (((size_t)1 << j->da_dm.b) >> 5) * 4 + 64);
    if (s != TSQ_OK) return give_up(s);
    DaColStore cs;
    memset(&cs, 0, sizeof cs);
    DaStore& st = cs.st;
    st.ent = ent.p;
    st.cursor = ctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.miss_count = st.cursor + g.nregions + 1;
    st.valid_end = vend.as<uint32_t>();
    st.ovf = ovf.as<uint32_t>();
    st.ovf_idx = ovfi.as<uint32_t>();
    st.ovf_cap = (uint32_t)nb;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = g.cap;
    for (int v = 0; v < ncols; v++) cs.pay[v] = pay[v].as<uint64_t>();
    cs.nnmask = any_nulls ? nnm.as<uint8_t>() : nullptr;
    DaColSrc src;
    memset(&src, 0, sizeof src);
    da_build_key(j, src.key);
    src.n_cols = ncols;
    src.any_nulls = any_nulls ? 1 : 0;
    for (int v = 0; v < ncols; v++) {
        src.col[v] = j->bcols[cols_of[v]].data.as<uint64_t>();
        src.nulls[v] = j->bcols[cols_of[v]].has_nulls ? j->bcols[cols_of[v]].nulls.as<uint8_t>() : nullptr;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctl.p, 0, g.ctl_bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(vend.p, 0xff, g.nregions * 4, ctx->stream);
    if (e == hipSuccess) {
        const dim3 grid((unsigned)std::min<int64_t>((nb + T - 1) / T, ctx->num_cus));
        hipLaunchKernelGGL((k_da_partition_cols<1024, 8, false>), grid, dim3(1024), 0, ctx->stream, src, j->da_dm, cs);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_da_part_starts, dim3(1), dim3(1024), 0, ctx->stream, st, j->da_pstart_c.as<uint32_t>());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 52, st.ovf_count, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 56, j->da_pstart_c.as<uint32_t>() + g.P, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    j->st.kernel_launches += 2;
    auto finish_events = [&]() {
        float ms = 0;
        if (e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) j->da_build_ms += ms;
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    if (e != hipSuccess) {
        finish_events();
        return give_up(tsq_fail(h, TSQ_ERR_HIP, std::string("packed build columns: ") + hipGetErrorString(e)));
    }
    const int64_t n = (int64_t)((const uint32_t*)(ctx->pinned + 56))[0];  // the build rows that have a usable key
    if (((const uint32_t*)(ctx->pinned + 52))[0] != 0 || n > nb) {  // skewed build keys: some rows missed their region — the other routes keep this join
        (void)hipEventRecord(e1, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        finish_events();
        return give_up(TSQ_OK);
    }
    DaSortPartArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.cs = cs;
    sa.img = j->da_img.as<uint8_t>();
    sa.coarse = j->da_coarse_c.as<uint32_t>();
    sa.pstart = j->da_pstart_c.as<uint32_t>();
    sa.bdup = j->da_unique ? nullptr : bdup.as<uint8_t>();
    sa.n_cols = ncols;
    for (int v = 0; v < ncols && s == TSQ_OK; v++) {
        const int c = cols_of[v];
        s = j->da_bsorted[c].reserve(ctx, h, (size_t)n * 8 + 64);
        sa.sorted[v] = j->da_bsorted[c].as<uint64_t>();
        if (s == TSQ_OK && j->bcols[c].has_nulls) {
            s = j->da_bsorted_nn[c].reserve(ctx, h, (size_t)n + 64);
            sa.sorted_nn[v] = j->da_bsorted_nn[c].as<uint8_t>();
        }
    }
    if (s != TSQ_OK) {
        finish_events();
        return give_up(s);
    }
    const uint32_t cells = 1u << j->da_ebits;
    // (16 sub-buckets per partition and two workgroups per CU — ~60 KB of LDS each, the entries streamed 16 times from L2 — were
    // measured in round 4: 2.88 ms per column and 1e8 rows against 1.67 ms for 8 sub-buckets and one workgroup per CU with a staging
    // buffer of twice the expected rows: kept)
    sa.sub_bits = j->da_ebits > 13 ? std::min<uint32_t>(3u, j->da_ebits - 13u) : 0u;
    const uint32_t scells = cells >> sa.sub_bits;
    // the staging buffer: twice the expected rows of a sub-bucket (denser sub-buckets take several windows), at most ~100 KB
    const uint64_t expect = (uint64_t)(n / std::max<int64_t>(1, (int64_t)g.P << sa.sub_bits)) * 2 + 1024;
    sa.stage_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(expect, 2048), 12288);
    {
        DaCoarseArgs ca;
        memset(&ca, 0, sizeof ca);
        ca.st = st;
        ca.img = sa.img;
        ca.coarse = j->da_coarse_c.as<uint32_t>();
        e = hipFuncSetAttribute((const void*)k_da_coarse<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((k_da_coarse<1024>), dim3(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus * 2)), dim3(1024), cells, ctx->stream, ca);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess && ncols > 0 && n > 0) {
        const size_t lds = (size_t)2 * scells + (size_t)(scells >> 5) * 4 + (size_t)sa.stage_cap * 9 + 16;
        e = hipFuncSetAttribute((const void*)k_da_sort_partition<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            const uint32_t grid = (uint32_t)std::max(8, (ctx->num_cus / 8) * 8);
            hipLaunchKernelGGL((k_da_sort_partition<1024>), dim3(grid), dim3(1024), lds, ctx->stream, sa);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    finish_events();
    j->st.kernel_launches += 2;
    if (e != hipSuccess) return give_up(tsq_fail(h, TSQ_ERR_HIP, std::string("packed build columns (sort): ") + hipGetErrorString(e)));
    release_all();
    j->da_cols_state = 1;
    return TSQ_OK;
}

// The outer-side filter of an outer join (join.go:328-345: a probe row that fails it goes to onMissMatch — NULL-padded — without
// touching the table) evaluated over the probe batch into one flag byte per row, so that the packed routes take it the way they take
// an externally evaluated selected[] vector: a row with flag 0 behaves like a row with a NULL key.
struct FilterFlagArgs {
    tsq_colset p;
    const tsq_expr_prog* filters;
    int32_t n_filters;
    int64_t n;
    const uint8_t* selected;  // nullptr or the caller's flags: ANDed in
    uint8_t* flags;
    unsigned long long* err;  // err word (preset TSQ_ERRWORD_NONE): an evaluation error leaves the batch to the direct route
    unsigned long long* div0;
};
__global__ void __launch_bounds__(256) k_outer_filter_flags(FilterFlagArgs a) {
    uint64_t errw = TSQ_ERRWORD_NONE;
    uint32_t div0 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        bool sel = !a.selected || a.selected[i] != 0;
        if (sel) {
            tsq_chunk_src src{&a.p, i};
            bool isnull = false;
            int ec = 0, en = 0, d0 = 0;
            sel = false;
            const tsq_status s = tsq_filter_row(a.filters, a.n_filters, src, &sel, &isnull, &ec, &en, &d0);
            div0 += (uint32_t)d0;
            if (s != TSQ_OK) {
                const uint64_t w = tsq_errword(ec, en, (uint64_t)i, s);
                errw = w < errw ? w : errw;
                sel = false;
            }
        }
        a.flags[i] = sel ? 1 : 0;
    }
    if (errw != TSQ_ERRWORD_NONE) atomicMin(a.err, (unsigned long long)errw);
    const uint64_t d = wave_sum_u64(div0);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(a.div0, (unsigned long long)d);
}
// *folded: j->fflags holds the flags and *div0 the warnings the filters raised (added to the statistics by the caller once a packed
// route has taken the batch: the direct route evaluates — and counts — again)
tsq_status fold_outer_filters(tsq_join* j, const tsq_colset& pcs, int64_t nrows, const uint8_t* selected_dev, bool* folded, int64_t* div0) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    *folded = false;
    *div0 = 0;
    TSQ_TRY(j->fflags.reserve(ctx, h, (size_t)nrows + 64));
    FilterFlagArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.p = pcs;
    fa.filters = j->filters_d.as<tsq_expr_prog>();
    fa.n_filters = (int32_t)j->filters_h.size();
    fa.n = nrows;
    fa.selected = selected_dev;
    fa.flags = j->fflags.as<uint8_t>();
    fa.err = (unsigned long long*)(ctx->dscratch + 56);
    fa.div0 = (unsigned long long*)(ctx->dscratch + 57);
    ctx->pinned[56] = TSQ_ERRWORD_NONE;
    ctx->pinned[57] = 0;
    TSQ_HIP(h, hipMemcpyAsync(ctx->dscratch + 56, ctx->pinned + 56, 16, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_outer_filter_flags, dim3(tsq_grid_for(ctx, nrows, 256)), dim3(256), 0, ctx->stream, fa);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 56, ctx->dscratch + 56, 16, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    j->st.kernel_launches++;
    if (ctx->pinned[56] != TSQ_ERRWORD_NONE) return TSQ_OK;  // which error the reference reports depends on the row order: the direct route's business
    *folded = true;
    *div0 = (int64_t)ctx->pinned[57];
    return TSQ_OK;
}

// OtherConditions over a materialised batch of joined rows: row i of the output columns is (left row, right row) at once.
struct PostCondArgs {
    tsq_colset L, R;
    const tsq_expr_prog* conds;
    int32_t n_conds;
    int64_t n;
    uint8_t* keep;
    unsigned long long* err;  // err word (preset TSQ_ERRWORD_NONE): an evaluation error sends the whole batch to the direct route
    unsigned long long* div0; // += division-by-zero warnings of the conditions (NULL result + warning, expression/errors.go:65-77)
    // outer joins: the packed NOT-NULL bitmap of the build side's KEY output column — a joined row holds a build key (NULL keys are
    // never inserted, hash_table.go:161-163), a NULL-padded row of an unmatched outer row does not.  The reference never evaluates
    // the conditions on a padded row (joiner.go onMissMatch): neither does this kernel (nullptr: inner join, every row is a match)
    const uint8_t* matched;
};
__global__ void __launch_bounds__(256) k_post_conds(PostCondArgs a) {
    uint64_t errw = TSQ_ERRWORD_NONE;
    uint32_t div0 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        if (a.matched && tsq_is_null(a.matched, i)) {  // the padded row of an unmatched outer row: it stays what it is
            a.keep[i] = 1;
            continue;
        }
        tsq_joined_src src;
        src.left = &a.L;
        src.right = &a.R;
        src.lrow = src.rrow = i;
        bool sel = false, isnull = false;
        int ec = 0, en = 0, d0 = 0;
        const tsq_status s = tsq_filter_row(a.conds, a.n_conds, src, &sel, &isnull, &ec, &en, &d0);
        div0 += (uint32_t)d0;
        if (s != TSQ_OK) {
            const uint64_t w = tsq_errword(ec, en, (uint64_t)i, s);
            errw = w < errw ? w : errw;
            sel = false;
        }
        a.keep[i] = sel ? 1 : 0;
    }
    if (errw != TSQ_ERRWORD_NONE) atomicMin(a.err, (unsigned long long)errw);
    const uint64_t d = wave_sum_u64(div0);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(a.div0, (unsigned long long)d);
}
// outer join, unique build side: a joined row whose conditions failed becomes the NULL-padded row — the build side's cells go NULL
struct OuterUnmatchArgs {
    const uint8_t* keep;
    int32_t padded_is;  // the value of keep[] that marks a row to pad: 0 (unique build side: the failed candidate itself) or 2 (k_outer_segments)
    int64_t n;
    int32_t n_cols;
    uint8_t* bitmap[TSQ_MAX_COLS];  // the build side's output columns
};
__global__ void __launch_bounds__(256) k_outer_unmatch(OuterUnmatchArgs a) {
    const int64_t nbytes = (a.n + 7) >> 3;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nbytes; b += (int64_t)gridDim.x * 256) {
        uint32_t m = 0;
        for (int k = 0; k < 8 && b * 8 + k < a.n; k++) m |= (int32_t)a.keep[b * 8 + k] != a.padded_is ? (1u << k) : 0u;
        if (m == 0xffu) continue;
        for (int c = 0; c < a.n_cols; c++) a.bitmap[c][b] &= (uint8_t)m;
    }
}
// outer join, build side with duplicate keys: the candidates of an outer row are consecutive rows of the batch, head[] marks the first
// of each.  An outer row none of whose candidates passed the conditions keeps its first row, padded (keep = 2): onMissMatch after
// tryToMatch found nothing (joiner.go:252-281)
// A lane walks the first 32 candidates of its outer row itself; a longer segment (a hot build key: 1e5 duplicates walked by one lane while
// the grid idles — ADVICE r5) is then scanned by the whole wave, 64 candidates per step.
__global__ void __launch_bounds__(256) k_outer_segments(uint8_t* keep, const uint8_t* head, const uint8_t* matched, int64_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t base = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); base < n; base += stride) {  // (wave-uniform: every lane stays in the loop)
        const int64_t i = base + lane;
        const bool mine = i < n && head[i] && !tsq_is_null(matched, i);  // (a padded row is a segment of its own and stays)
        bool any = false, open = false;
        int64_t k = i;
        if (mine) {
            int steps = 0;
            do { any = keep[k] != 0; k++; steps++; } while (!any && k < n && !head[k] && steps < 32);  // (stops at the first candidate that passed)
            open = !any && k < n && !head[k];
        }
        for (uint64_t todo = __ballot(open); todo; todo &= todo - 1) {  // the long segments of this wave's rows, one after the other
            const int l = __ffsll((unsigned long long)todo) - 1;
            int64_t k0 = __shfl(k, l, 64);
            bool found = false;
            for (;;) {
                const int64_t kk = k0 + lane;
                const bool in = kk < n;
                const bool hd = in && head[kk];
                const bool kp = in && keep[kk] != 0;
                const uint64_t mh = __ballot(hd || !in), mk = __ballot(kp);
                // candidates of the segment = the lanes before the first head (or the end of the batch)
                const uint64_t before = mh ? ((mh & (0 - mh)) - 1) : ~0ull;
                if (mk & before) { found = true; break; }
                if (mh) break;
                k0 += 64;
            }
            if ((int)lane == l) any = found;
        }
        if (mine && !any) keep[i] = 2;
    }
}
// filters the batch in place (new, dense column buffers).  *redo: a condition raised an error — which error the reference reports
// depends on the probe row order, so the batch is dropped and the caller runs it through the direct route.
tsq_status da_post_conditions(tsq_join* j, ResultBatch& rb, const std::vector<bool>& may_null_v, bool* redo, const uint8_t* head = nullptr) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const int np = j->cfg.n_probe_cols, nbc = j->cfg.n_build_cols, nout = np + nbc;
    const bool probe_is_left = j->cfg.build_is_right != 0;
    const int nl = probe_is_left ? np : nbc;
    const int64_t n = rb.rows;
    PostCondArgs pa;
    memset(&pa, 0, sizeof pa);
    std::vector<tsq_col> in((size_t)nout), out((size_t)nout);
    for (int oc = 0; oc < nout; oc++) {
        const bool from_probe = probe_is_left ? oc < nl : oc >= nl;
        const int sc = oc < nl ? oc : oc - nl;
        const int32_t type = from_probe ? j->cfg.probe_types[sc] : j->cfg.build_types[sc];
        tsq_colset& cs = oc < nl ? pa.L : pa.R;
        cs.data[sc] = rb.data[oc].p;
        cs.nulls[sc] = may_null_v[oc] ? rb.bitmap[oc].as<uint8_t>() : nullptr;
        cs.type[sc] = type;
        memset(&in[(size_t)oc], 0, sizeof(tsq_col));
        in[(size_t)oc].data = rb.data[oc].p;
        in[(size_t)oc].null_bitmap = may_null_v[oc] ? rb.bitmap[oc].as<uint8_t>() : nullptr;
        in[(size_t)oc].length = n;
        in[(size_t)oc].elem_size = 8;
        in[(size_t)oc].type = type;
        in[(size_t)oc].flags = TSQ_COL_DEVICE;
    }
    pa.L.n = nl;
    pa.R.n = nout - nl;
    pa.conds = j->conds_d.as<tsq_expr_prog>();
    pa.n_conds = (int32_t)j->conds_h.size();
    pa.n = n;
    DevBuf keep;
    TSQ_TRY(keep.reserve(ctx, h, (size_t)n + 64));
    pa.keep = keep.as<uint8_t>();
    pa.err = (unsigned long long*)(ctx->dscratch + 56);
    pa.div0 = (unsigned long long*)(ctx->dscratch + 57);
    if (j->cfg.join_type != TSQ_JOIN_INNER) {  // which output rows are matches: the build side's key column is NOT NULL there
        const int kb = j->ks.bidx[0], okb = probe_is_left ? nl + kb : kb;
        if (!may_null_v[okb]) return tsq_fail(h, TSQ_ERR_HIP, "internal: outer join output column without a bitmap");
        pa.matched = rb.bitmap[okb].as<uint8_t>();
    }
    ctx->pinned[56] = TSQ_ERRWORD_NONE;
    ctx->pinned[57] = 0;
    hipError_t e = hipMemcpyAsync(ctx->dscratch + 56, ctx->pinned + 56, 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_post_conds, dim3(tsq_grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, pa);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 56, ctx->dscratch + 56, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        keep.release();
        return tsq_fail(h, TSQ_ERR_HIP, std::string("conditions over the joined batch: ") + hipGetErrorString(e));
    }
    j->st.kernel_launches++;
    if (ctx->pinned[56] != TSQ_ERRWORD_NONE) {  // (the direct route evaluates the batch again and counts its warnings itself)
        keep.release();
        *redo = true;
        return TSQ_OK;
    }
    j->div0_packed += (int64_t)ctx->pinned[57];
    if (j->cfg.join_type != TSQ_JOIN_INNER) {
        // unique build side: every outer row keeps its one output row, a failed candidate is un-matched.  Duplicates (head != null):
        // failed candidates go, except the first row of an outer row that lost them all, which is padded; then the batch is compacted
        if (head) {
            hipLaunchKernelGGL(k_outer_segments, dim3(tsq_grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, keep.as<uint8_t>(), head, pa.matched, n);
            hipError_t e4 = hipGetLastError();
            if (e4 != hipSuccess) { keep.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("outer join conditions: ") + hipGetErrorString(e4)); }
            j->st.kernel_launches++;
        }
        OuterUnmatchArgs ua;
        memset(&ua, 0, sizeof ua);
        ua.keep = keep.as<uint8_t>();
        ua.padded_is = head ? 2 : 0;
        ua.n = n;
        for (int oc = 0; oc < nout; oc++) {
            const bool from_probe = probe_is_left ? oc < nl : oc >= nl;
            if (from_probe) continue;
            if (!may_null_v[oc]) { keep.release(); return tsq_fail(h, TSQ_ERR_HIP, "internal: outer join output column without a bitmap"); }
            ua.bitmap[ua.n_cols++] = rb.bitmap[oc].as<uint8_t>();
        }
        hipLaunchKernelGGL(k_outer_unmatch, dim3(tsq_grid_for(ctx, (n + 7) / 8, 256)), dim3(256), 0, ctx->stream, ua);
        hipError_t e3 = hipGetLastError();
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(ctx->stream);
        if (e3 != hipSuccess) { keep.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("outer join conditions: ") + hipGetErrorString(e3)); }
        j->st.kernel_launches++;
        if (!head) {
            keep.release();
            return TSQ_OK;
        }
    }
    std::vector<DevBuf> nd((size_t)nout), nbm((size_t)nout);
    tsq_status s = TSQ_OK;
    for (int oc = 0; oc < nout && s == TSQ_OK; oc++) {
        s = nd[(size_t)oc].reserve(ctx, h, ((size_t)n + 8) * 8 + 16);
        if (s == TSQ_OK && may_null_v[oc]) s = nbm[(size_t)oc].reserve(ctx, h, tsq_bitmap_bytes(n) + 16);
        out[(size_t)oc] = in[(size_t)oc];
        out[(size_t)oc].data = nd[(size_t)oc].p;
        out[(size_t)oc].null_bitmap = may_null_v[oc] ? nbm[(size_t)oc].as<uint8_t>() : nullptr;
    }
    int64_t kept = 0;
    if (s == TSQ_OK) {
        s = tsq_chunk_compact(ctx, in.data(), nout, n, keep.as<uint8_t>(), out.data(), &kept);
        if (s != TSQ_OK) tsq_fail(h, s, ctx->hdr.err);
    }
    if (s == TSQ_OK) {
        hipError_t e2 = hipStreamSynchronize(ctx->stream);
        if (e2 != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e2));
    }
    keep.release();
    if (s != TSQ_OK) {
        for (auto& b : nd) b.release();
        for (auto& b : nbm) b.release();
        return s;
    }
    for (int oc = 0; oc < nout; oc++) {
        rb.data[oc].release();
        rb.data[oc] = nd[(size_t)oc];  // shallow move of the buffer handle
        if (may_null_v[oc]) {
            rb.bitmap[oc].release();
            rb.bitmap[oc] = nbm[(size_t)oc];
        }
    }
    rb.rows = kept;
    return TSQ_OK;
}

tsq_status da_emit_cols(tsq_join* j, const tsq_colset& pcs, int64_t nrows, bool* redo, const uint8_t* sel = nullptr) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const bool outer = j->cfg.join_type != TSQ_JOIN_INNER;
    const int np = j->cfg.n_probe_cols, nbc = j->cfg.n_build_cols;
    // TSQ_DA_TRACE=1: host-side time points of one batch on stderr (where the wall time between the kernels goes)
    const bool trace = tsq_knob(ctx, TSQ_KNOB_DA_TRACE, 0) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto tp = [&](const char* what) {
        if (trace) fprintf(stderr, "[da_emit_cols] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    };
    constexpr int T = 1024 * 8;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nrows, T);
    if (g.nregions * g.cap >= 0xffffffffULL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "radix probe batch too large");
    const size_t slots = g.nregions * g.cap;
    TSQ_TRY(j->rkeys.reserve(ctx, h, g.ent_bytes));
    TSQ_TRY(j->rctl.reserve(ctx, h, g.ctl_bytes));
    TSQ_TRY(j->rvend.reserve(ctx, h, g.nregions * 4));
    TSQ_TRY(j->rovf.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_TRY(j->rovfidx.reserve(ctx, h, (size_t)nrows * 4 + 64));
    if (outer) TSQ_TRY(j->rmiss.reserve(ctx, h, (size_t)nrows * 4 + 64));
    const int kc = j->da_multi ? -1 : j->ks.pidx[0], kb = j->da_multi ? -1 : j->ks.bidx[0];  // (-1: a composite key, its columns are payload)
    int trav[TSQ_DA_MAXCOLS], ntrav = 0;  // the probe columns that travel: all but the key
    for (int c = 0; c < np; c++)
        if (c != kc) trav[ntrav++] = c;
    bool any_nulls = false;
    for (int v = 0; v < ntrav; v++) {
        TSQ_TRY(j->rcols[v].reserve(ctx, h, slots * 8 + 256));
        any_nulls = any_nulls || pcs.nulls[trav[v]] != nullptr;
    }
    if (any_nulls) TSQ_TRY(j->rnnmask.reserve(ctx, h, slots + 256));
    const size_t n_pc = (size_t)g.P + 1;
    TSQ_TRY(j->tkcnt.reserve(ctx, h, n_pc * 8 + 64));
    DaColStore cs;
    memset(&cs, 0, sizeof cs);
    DaStore& st = cs.st;
    st.ent = j->rkeys.p;
    st.cursor = j->rctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.miss_count = st.cursor + g.nregions + 1;
    st.miss = j->rmiss.as<uint32_t>();
    st.valid_end = j->rvend.as<uint32_t>();
    st.ovf = j->rovf.as<uint32_t>();
    st.ovf_idx = j->rovfidx.as<uint32_t>();
    st.ovf_cap = (uint32_t)nrows;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = g.cap;
    for (int v = 0; v < ntrav; v++) cs.pay[v] = j->rcols[v].as<uint64_t>();
    cs.nnmask = any_nulls ? j->rnnmask.as<uint8_t>() : nullptr;
    TSQ_HIP(h, hipMemsetAsync(j->rctl.p, 0, g.ctl_bytes, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->rvend.p, 0xff, g.nregions * 4, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->tkcnt.p, 0, n_pc * 8, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(ctx->dscratch + 52, 0, 16, ctx->stream));
    DaColSrc src;
    memset(&src, 0, sizeof src);
    TSQ_TRY(da_probe_key(j, pcs, nrows, src.key, sel));
    src.n_cols = ntrav;
    src.any_nulls = any_nulls ? 1 : 0;
    for (int v = 0; v < ntrav; v++) {
        src.col[v] = (const uint64_t*)pcs.data[trav[v]];
        src.nulls[v] = pcs.nulls[trav[v]];
    }
    hipEvent_t* re = j->rev[j->st.radix_batches % tsq_join::RING];
    for (int e = 0; e < 3; e++)
        if (!re[e]) TSQ_HIP(h, hipEventCreate(&re[e]));
    tp("buffers + memsets queued");
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(re[0], ctx->stream));
    {
        const dim3 grid((unsigned)std::min<int64_t>((nrows + T - 1) / T, ctx->num_cus));
        if (outer) hipLaunchKernelGGL((k_da_partition_cols<1024, 8, true>), grid, dim3(1024), 0, ctx->stream, src, j->da_dm, cs);
        else hipLaunchKernelGGL((k_da_partition_cols<1024, 8, false>), grid, dim3(1024), 0, ctx->stream, src, j->da_dm, cs);
        TSQ_HIP(h, hipGetLastError());
    }
    TSQ_HIP(h, hipEventRecord(re[1], ctx->stream));
    // ---- sizing pass
    DaProbeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.st = st;
    pa.img = j->da_img.as<uint8_t>();
    pa.counters = (unsigned long long*)(ctx->dscratch + 52);
    pa.pcount = j->tkcnt.as<unsigned long long>();
    const size_t cells = (size_t)1 << j->da_ebits;
    const dim3 pgrid(std::min<uint32_t>(g.P, (uint32_t)ctx->num_cus * 2));
    if (outer) {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<512, uint16_t, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells));
        hipLaunchKernelGGL((k_da_probe_count<512, uint16_t, true, true>), pgrid, dim3(512), cells, ctx->stream, pa);
        TSQ_HIP(h, hipGetLastError());
        hipLaunchKernelGGL(k_da_probe_ovf<true>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    } else {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_da_probe_count<512, uint16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cells));
        hipLaunchKernelGGL((k_da_probe_count<512, uint16_t, true, false>), pgrid, dim3(512), cells, ctx->stream, pa);
        TSQ_HIP(h, hipGetLastError());
        hipLaunchKernelGGL(k_da_probe_ovf<false>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    }
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, pa.pcount, (int)n_pc);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 52, pa.pcount + g.P, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 53, ctx->dscratch + 52, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 54, st.miss_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    tp("partition + sizing queued");
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    tp("sizing pass done");
    const int64_t part_rows = (int64_t)ctx->pinned[52], ovf_rows = (int64_t)ctx->pinned[53];
    const int64_t miss_rows = outer ? (int64_t)((const uint32_t*)(ctx->pinned + 54))[0] : 0;
    const int64_t exc_rows = ovf_rows + miss_rows, out_rows = exc_rows + part_rows;
    j->st.kernel_launches += 4;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)st.bits;
    j->st.probe_route = TSQ_ROUTE_PACKED;
    j->st.packed_key_bits = (int32_t)j->da_dm.b;
    j->st.packed_lds_bits = 0;
    if (out_rows == 0) {
        TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
        TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
        j->have_probe_ev = true;
        return TSQ_OK;
    }
    // ---- the output batch: [exception rows | rows of the partitions]; NULLs as one byte per row, packed at the end
    const int nout = np + nbc;
    const bool probe_is_left = j->cfg.build_is_right != 0;
    const int nl = probe_is_left ? np : nbc;
    std::unique_ptr<ResultBatch> rb(new ResultBatch());
    rb->rows = out_rows;
    rb->data.resize(nout);
    rb->notnull.resize(nout);
    rb->bitmap.resize(nout);
    rb->offs.resize(nout);
    rb->nbytes.assign(nout, 0);
    std::vector<bool> may_null_v(nout, false);
    DaEmitColsArgs ea;
    memset(&ea, 0, sizeof ea);
    DaExcArgs xa;
    memset(&xa, 0, sizeof xa);
    for (int oc = 0; oc < nout; oc++) {
        const bool from_probe = probe_is_left ? oc < nl : oc >= nl;
        const int sc = oc < nl ? oc : oc - nl;
        const bool may_null = from_probe ? pcs.nulls[sc] != nullptr : (j->bcols[sc].has_nulls || outer);
        may_null_v[oc] = may_null;
        tsq_status s = rb->data[oc].reserve(ctx, h, ((size_t)out_rows + 8) * 8 + 16);
        if (s == TSQ_OK && may_null) s = rb->notnull[oc].reserve(ctx, h, (size_t)out_rows + 64);
        if (s == TSQ_OK && may_null) s = rb->bitmap[oc].reserve(ctx, h, tsq_bitmap_bytes(out_rows) + 16);
        if (s != TSQ_OK) { rb->release(); return s; }
        if (may_null) TSQ_HIP(h, hipMemsetAsync(rb->notnull[oc].p, 1, (size_t)out_rows, ctx->stream));  // K4e stores the NULL cells' flags only
        uint64_t* od = rb->data[oc].as<uint64_t>();
        uint8_t* of = may_null ? rb->notnull[oc].as<uint8_t>() : nullptr;
        if (from_probe) {
            if (sc == kc) {
                ea.out_pkey = od;
                ea.out_pkey_nn = of;
            } else {
                const int v = (kc < 0 || sc < kc) ? sc : sc - 1;
                ea.out_probe[v] = od;
                ea.out_probe_nn[v] = of;
            }
            xa.pcol[sc] = (const uint64_t*)pcs.data[sc];
            xa.pnull[sc] = pcs.nulls[sc];
            xa.out_probe[sc] = od;
            xa.out_probe_nn[sc] = of;
        } else {
            if (sc == kb) {
                ea.out_bkey = od;
                ea.out_bkey_nn = of;
            } else {
                const int v = (kb < 0 || sc < kb) ? sc : sc - 1;
                ea.out_build[v] = od;
                ea.out_build_nn[v] = of;
                ea.bsorted[v] = j->da_bsorted[sc].as<uint64_t>();
                ea.bsorted_nn[v] = j->bcols[sc].has_nulls ? j->da_bsorted_nn[sc].as<uint8_t>() : nullptr;
            }
            // exception rows read the build side where the emit kernel reads it: the sorted columns (the key cell of a joined row is the
            // probe row's key cell: same flag, same 8 bytes, codec.go:212-240)
            xa.bcol[sc] = sc == kb ? nullptr : j->da_bsorted[sc].as<uint64_t>();
            xa.bnn[sc] = (sc != kb && j->bcols[sc].has_nulls) ? j->da_bsorted_nn[sc].as<uint8_t>() : nullptr;
            xa.out_build[sc] = od;
            xa.out_build_nn[sc] = of;
        }
    }
    // outer join + conditions + duplicate build keys: which output rows start an outer row's candidates
    const bool want_heads = outer && !j->conds_h.empty() && !j->da_unique;
    DevBuf& head = j->heads;
    if (want_heads) {
        tsq_status s = head.reserve(ctx, h, (size_t)out_rows + 64);
        if (s != TSQ_OK) { rb->release(); return s; }
        TSQ_HIP(h, hipMemsetAsync(head.p, 1, (size_t)out_rows, ctx->stream));
        ea.out_head = head.as<uint8_t>();
    }
    tp("output buffers");
    if (exc_rows > 0) {  // overflow-list rows, then the NULL-padded rows of the miss list, as pairs; their cells through the pairs
        TSQ_TRY(j->pairs.reserve(ctx, h, (size_t)exc_rows * 8 + 64));
        DaEmitArgs pe;
        memset(&pe, 0, sizeof pe);
        pe.st = st;
        pe.img = pa.img;
        pe.coarse = j->da_coarse_c.as<uint32_t>();
        pe.pstart = j->da_pstart_c.as<uint32_t>();
        pe.brows = nullptr;  // the pairs name PLACES of the sorted build columns (pstart + rank + k), not build rows
        pe.pairs = j->pairs.as<unsigned long long>();
        pe.ovf_cursor = (unsigned long long*)(ctx->dscratch + 53);  // starts at 0 (cleared above)
        if (outer) hipLaunchKernelGGL(k_da_emit_ovf<true>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pe);
        else hipLaunchKernelGGL(k_da_emit_ovf<false>, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pe);
        TSQ_HIP(h, hipGetLastError());
        if (miss_rows) {
            hipLaunchKernelGGL(k_da_emit_miss, dim3(tsq_grid_for(ctx, miss_rows, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)st.miss, (uint32_t)miss_rows,
                               pe.pairs + ovf_rows);
            TSQ_HIP(h, hipGetLastError());
        }
        xa.pairs = pe.pairs;
        xa.bkey_col = kb;
        xa.pkey_col = kc;
        xa.n = exc_rows;
        xa.n_probe = np;
        xa.n_build = nbc;
        hipLaunchKernelGGL(k_da_gather_exc, dim3(tsq_grid_for(ctx, exc_rows, 256)), dim3(256), 0, ctx->stream, xa);
        TSQ_HIP(h, hipGetLastError());
        if (want_heads) {
            hipLaunchKernelGGL(k_da_exc_heads, dim3(tsq_grid_for(ctx, exc_rows, 256)), dim3(256), 0, ctx->stream, (const unsigned long long*)xa.pairs, exc_rows, head.as<uint8_t>());
            TSQ_HIP(h, hipGetLastError());
        }
        j->st.kernel_launches += 3;
    }
    ea.cs = cs;
    ea.img = pa.img;
    ea.coarse = j->da_coarse_c.as<uint32_t>();
    ea.pstart = j->da_pstart_c.as<uint32_t>();
    ea.pbase = pa.pcount;
    ea.row0 = (unsigned long long)exc_rows;
    ea.dm = j->da_dm;
    ea.n_probe = ntrav;
    ea.n_build = kb < 0 ? nbc : nbc - 1;
    const size_t lds = cells + (cells >> 5) * 4;
    {
        const void* fn = outer ? (j->da_unique ? (const void*)k_da_emit_cols<512, true, true> : (const void*)k_da_emit_cols<512, true, false>)
                               : (j->da_unique ? (const void*)k_da_emit_cols<512, false, true> : (const void*)k_da_emit_cols<512, false, false>);
        TSQ_HIP(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (outer && j->da_unique) hipLaunchKernelGGL((k_da_emit_cols<512, true, true>), pgrid, dim3(512), lds, ctx->stream, ea);
        else if (outer) hipLaunchKernelGGL((k_da_emit_cols<512, true, false>), pgrid, dim3(512), lds, ctx->stream, ea);
        else if (j->da_unique) hipLaunchKernelGGL((k_da_emit_cols<512, false, true>), pgrid, dim3(512), lds, ctx->stream, ea);
        else hipLaunchKernelGGL((k_da_emit_cols<512, false, false>), pgrid, dim3(512), lds, ctx->stream, ea);
    }
    TSQ_HIP(h, hipGetLastError());
    j->st.kernel_launches++;
    for (int oc = 0; oc < nout; oc++)
        if (may_null_v[oc]) TSQ_TRY(tsq_launch_pack_bitmap(ctx, h, rb->notnull[oc].as<uint8_t>(), rb->bitmap[oc].as<uint8_t>(), out_rows));
    if (!j->conds_h.empty()) {
        tsq_status ps = da_post_conditions(j, *rb, may_null_v, redo, want_heads ? head.as<uint8_t>() : nullptr);
        if (ps != TSQ_OK || *redo) {
            rb->release();
            if (*redo) {  // as if this route had not been tried
                j->st.radix_batches--;
                j->st.probe_route = TSQ_ROUTE_DIRECT;
            }
            return ps;
        }
    }
    TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
    j->have_probe_ev = true;
    tp("emit queued");
    if (rb->rows == 0) {
        rb->release();
        return TSQ_OK;
    }
    const tsq_status ds = deliver_batch(j, std::move(rb), may_null_v);
    tp("delivered (stream idle)");
    return ds;
}

// ---------------------------------------------------------------- materialising packed route, build side in LDS (host side; tsq_damat.h)
// Eligible on top of da_cols_eligible: a build side WITHOUT duplicate keys (byte cells 0 / 1), 2-byte entries of 7..16 bits.  S (the
// level-2 fan-out) is the smallest power of two <= 8 that brings the build rows of a final partition into a table of <= 64 KB per
// workgroup (two workgroups per CU); a build side that does not get there (more than ~5e8 rows, or several wide columns) keeps round
// 4's variant.  TSQ_KNOB_DA_LDS_BUILD = 0 switches the route off (A/B measurements, tests of the other variant).
bool dm_eligible(const tsq_join* j) {
    if (tsq_knob(j->ctx, TSQ_KNOB_DA_LDS_BUILD, 1) == 0) return false;
    return j->da_state == 1 && !j->da_bits && j->da_unique && j->da_ebits <= 16 && j->da_ebits >= 7 && j->dm_state >= 0;
}
// (the state the build side must be in before its final partitions are made)
static bool dm_eligible_shape(const tsq_join* j) { return j->da_state == 1 && !j->da_bits && j->da_unique && j->da_ebits <= 16 && j->da_ebits >= 7; }
static size_t dm_split_lds(uint32_t ebits1, bool filter) { return (size_t)512 * 8 * 10 + (filter ? ((size_t)1 << ebits1) / 8 : 0) + 16; }
// level 2 of one side: src (level 1) -> dst; filter: the probe side of an inner join
tsq_status dm_launch_split(tsq_join* j, const DaColStore& src, int n_cols, const DmStore& dst, bool filter) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    DmSplitArgs a;
    memset(&a, 0, sizeof a);
    a.src = src;
    a.n_cols = n_cols;
    a.dst = dst;
    a.bitmap = filter ? j->dm_bitmap.as<uint32_t>() : nullptr;
    const size_t lds = dm_split_lds(src.st.ebits, filter);
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(3, (size_t)(150 * 1024) / lds));  // (measured: 3 workgroups per CU 0.42 ms, 2: 0.46, 1: 0.64 per 1e8 rows)
    const dim3 grid(std::min<uint32_t>(1u << src.st.bits, (uint32_t)ctx->num_cus * per_cu));
    if (filter) {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_dm_split<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_dm_split<512, true>), grid, dim3(512), lds, ctx->stream, a);
    } else {
        TSQ_HIP(h, hipFuncSetAttribute((const void*)k_dm_split<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_dm_split<512, false>), grid, dim3(512), lds, ctx->stream, a);
    }
    TSQ_HIP(h, hipGetLastError());
    j->st.kernel_launches++;
    return TSQ_OK;
}
// the build side, once per build: level 1 with its columns, level 2, the bitmap of the domain; the largest final partition sizes the tables
tsq_status dm_prepare_build(tsq_join* j) {
    if (j->dm_state) return TSQ_OK;
    j->dm_state = -1;
    struct L1Drop {  // whatever of da_prepare's level-1 store this function does not take over goes back to the pool when it returns
        tsq_join* j;
        ~L1Drop() {
            j->dm_l1_ready = false;
            for (DevBuf* b : {&j->dm_l1ent, &j->dm_l1ctl, &j->dm_l1vend, &j->dm_l1nn}) b->release();
            for (auto& b : j->dm_l1pay) b.release();
        }
    } l1_drop{j};
    if (!dm_eligible_shape(j)) return TSQ_OK;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const int kb = j->da_multi ? -1 : j->ks.bidx[0];
    const int64_t nb = j->bcols[j->ks.bidx[0]].rows;
    constexpr int T = 1024 * 8;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nb, T);
    if (g.nregions * g.cap >= 0xffffffffULL) return TSQ_OK;
    const size_t slots = g.nregions * g.cap;
    int ncols = 0;
    bool any_nulls = false;
    for (int c = 0; c < j->cfg.n_build_cols; c++) {
        if (c == kb) continue;
        if (ncols == TSQ_DA_MAXCOLS) return TSQ_OK;
        j->dm_bcol_of[ncols++] = c;
        any_nulls = any_nulls || j->bcols[c].has_nulls;
    }
    if (j->dm_l1_ready && j->dm_l1_nulls != any_nulls) {  // (cannot happen: both look at the same columns)
        j->dm_l1_ready = false;
        for (DevBuf* b : {&j->dm_l1ent, &j->dm_l1ctl, &j->dm_l1vend, &j->dm_l1nn}) b->release();
        for (auto& b : j->dm_l1pay) b.release();
    }
    // S: expected build rows of a final partition (+ 25 % and 6 sigma: the mix spreads any key set evenly) against a table of 64 KB
    const double table_rows = (64.0 * 1024.0) / (8.0 * std::max(1, ncols));
    uint32_t sbits = 0;
    auto expect = [&](uint32_t sb) { const double lam = (double)nb / (double)((size_t)1 << (j->da_pbits + sb)); return lam * 1.25 + 6.0 * sqrt(lam) + 32.0; };
    while (sbits < 3 && j->da_ebits - sbits > 7 && expect(sbits) > table_rows) sbits++;
    const int64_t forced = tsq_knob(ctx, TSQ_KNOB_DA_LDS_BUILD, 1);  // (tests: 2 .. 5 force S = 1 .. 8 on small build sides)
    if (forced >= 2) sbits = std::min<uint32_t>((uint32_t)std::min<int64_t>(forced - 2, 3), j->da_ebits - 5);
    if (expect(sbits) > 2.0 * table_rows) return TSQ_OK;  // (one workgroup per CU would still take 128 KB: beyond that the route is not for this build side)
    const uint32_t S = 1u << sbits, P1 = g.P, Q = P1 * S;
    DevBuf ent, ctl, vend, ovf, ovfi, nnm, pay[TSQ_DA_MAXCOLS];
    const bool have_l1 = j->dm_l1_ready;  // da_prepare partitioned the build side with its columns already
    j->dm_l1_ready = false;
    if (have_l1) {
        auto take = [](DevBuf& to, DevBuf& from) {
            to = from;
            from.p = nullptr;
            from.cap = 0;
        };
        take(ent, j->dm_l1ent);
        take(ctl, j->dm_l1ctl);
        take(vend, j->dm_l1vend);
        take(nnm, j->dm_l1nn);
        for (int v = 0; v < ncols; v++) take(pay[v], j->dm_l1pay[v]);
    }
    auto release_tmp = [&]() {
        for (DevBuf* x : {&ent, &ctl, &vend, &ovf, &ovfi, &nnm}) x->release();
        for (auto& b : pay) b.release();
    };
    auto give_up = [&](tsq_status st) {
        release_tmp();
        for (DevBuf* b : {&j->dm_bent, &j->dm_bnn, &j->dm_boff, &j->dm_bcnt, &j->dm_bitmap}) b->release();
        for (auto& b : j->dm_bpay) b.release();
        return st;
    };
    const uint32_t cap = have_l1 ? j->dm_l1_cap : g.cap;  // (da_prepare sized its regions for 16 Ki-row tiles: a little larger)
    const uint32_t cap1 = 8u * cap + 8u * S;
    const size_t slots2 = (size_t)P1 * cap1;
    if (slots2 >= 0xffffffffULL) return give_up(TSQ_OK);
    tsq_status s = TSQ_OK;
    if (!have_l1) {
        s = ent.reserve(ctx, h, g.ent_bytes);
        if (s == TSQ_OK) s = ctl.reserve(ctx, h, g.ctl_bytes);
        if (s == TSQ_OK) s = vend.reserve(ctx, h, g.nregions * 4);
        if (s == TSQ_OK) s = ovf.reserve(ctx, h, (size_t)nb * 4 + 64);
        if (s == TSQ_OK) s = ovfi.reserve(ctx, h, (size_t)nb * 4 + 64);
        for (int v = 0; v < ncols && s == TSQ_OK; v++) s = pay[v].reserve(ctx, h, slots * 8 + 256);
        if (s == TSQ_OK && any_nulls) s = nnm.reserve(ctx, h, slots + 256);
    }
    if (s == TSQ_OK) s = j->dm_bent.reserve(ctx, h, slots2 * 2 + 256);
    for (int v = 0; v < ncols && s == TSQ_OK; v++) s = j->dm_bpay[v].reserve(ctx, h, slots2 * 8 + 256);
    if (s == TSQ_OK && any_nulls) s = j->dm_bnn.reserve(ctx, h, slots2 + 256);
    if (s == TSQ_OK) s = j->dm_boff.reserve(ctx, h, (size_t)Q * 4 + 64);
    if (s == TSQ_OK) s = j->dm_bcnt.reserve(ctx, h, ((size_t)Q + 1) * 8 + 64);
    const size_t bit_words = ((size_t)1 << j->da_dm.b) / 32;
    if (s == TSQ_OK) s = j->dm_bitmap.reserve(ctx, h, bit_words * 4 + 64);
    if (s != TSQ_OK) {  // no memory for the second copy of the build side: the other variants keep the join
        h->err.clear();
        return give_up(TSQ_OK);
    }
    DaColStore cs;
    memset(&cs, 0, sizeof cs);
    DaStore& st = cs.st;
    st.ent = ent.p;
    st.cursor = ctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.miss_count = st.cursor + g.nregions + 1;
    st.valid_end = vend.as<uint32_t>();
    st.ovf = ovf.as<uint32_t>();
    st.ovf_idx = ovfi.as<uint32_t>();
    st.ovf_cap = (uint32_t)nb;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = cap;
    for (int v = 0; v < ncols; v++) cs.pay[v] = pay[v].as<uint64_t>();
    cs.nnmask = any_nulls ? nnm.as<uint8_t>() : nullptr;
    DaColSrc src;
    memset(&src, 0, sizeof src);
    da_build_key(j, src.key);
    src.n_cols = ncols;
    src.any_nulls = any_nulls ? 1 : 0;
    for (int v = 0; v < ncols; v++) {
        src.col[v] = j->bcols[j->dm_bcol_of[v]].data.as<uint64_t>();
        src.nulls[v] = j->bcols[j->dm_bcol_of[v]].has_nulls ? j->bcols[j->dm_bcol_of[v]].nulls.as<uint8_t>() : nullptr;
    }
    if (have_l1) ctx->pinned[52] = 0;  // (its overflow count was checked by da_prepare)
    DmStore d2;
    memset(&d2, 0, sizeof d2);
    d2.ent = j->dm_bent.as<uint16_t>();
    for (int v = 0; v < ncols; v++) d2.pay[v] = j->dm_bpay[v].as<uint64_t>();
    d2.nnmask = any_nulls ? j->dm_bnn.as<uint8_t>() : nullptr;
    d2.off = j->dm_boff.as<uint32_t>();
    d2.cnt = j->dm_bcnt.as<unsigned long long>();
    d2.cap1 = cap1;
    d2.sbits = sbits;
    d2.ebits2 = j->da_ebits - sbits;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess && !have_l1) e = hipMemsetAsync(ctl.p, 0, g.ctl_bytes, ctx->stream);
    if (e == hipSuccess && !have_l1) e = hipMemsetAsync(vend.p, 0xff, g.nregions * 4, ctx->stream);
    if (e == hipSuccess && !have_l1) {
        const dim3 grid((unsigned)std::min<int64_t>((nb + T - 1) / T, ctx->num_cus));
        hipLaunchKernelGGL((k_da_partition_cols<1024, 8, false>), grid, dim3(1024), 0, ctx->stream, src, j->da_dm, cs);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_da_bytes_to_bits, dim3(tsq_grid_for(ctx, (int64_t)bit_words, 256)), dim3(256), 0, ctx->stream, j->da_img.as<uint8_t>(), j->dm_bitmap.as<uint32_t>(), bit_words);
        e = hipGetLastError();
    }
    if (e == hipSuccess && dm_launch_split(j, cs, ncols, d2, false) != TSQ_OK) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    // the overflow count of level 1 and the rows of every final partition: the host sizes the tables by the largest
    std::vector<unsigned long long> cnt((size_t)Q);
    if (e == hipSuccess && !have_l1) e = hipMemcpyAsync(ctx->pinned + 52, st.ovf_count, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d2.cnt, (size_t)Q * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    float ms = 0;
    if (e == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) j->da_build_ms += ms;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    j->st.kernel_launches += 2;
    if (e != hipSuccess) return give_up(tsq_fail(h, TSQ_ERR_HIP, std::string("packed build side in LDS: ") + hipGetErrorString(e)));
    if (((const uint32_t*)(ctx->pinned + 52))[0] != 0) return give_up(TSQ_OK);  // (rows that missed their level-1 region: cannot happen with unique keys; the other variant copes)
    unsigned long long bmax = 0;
    for (unsigned long long c : cnt) bmax = std::max(bmax, c);
    const uint32_t tab_rows = (uint32_t)((bmax + 31) & ~31ull);
    if (dm_emit_lds(d2.ebits2, tab_rows, ncols, any_nulls) > (size_t)150 * 1024) return give_up(TSQ_OK);
    release_tmp();
    j->dm_sbits = sbits;
    j->dm_tab_rows = std::max<uint32_t>(tab_rows, 32u);
    j->dm_cap1_b = cap1;
    j->dm_nb = ncols;
    j->dm_bnulls = any_nulls;
    j->dm_state = 1;
    return TSQ_OK;
}

// One probe batch: level 1 (+ the miss list of an outer join), level 2 (an inner join drops the rows without a build row), scan, emit.
// *fallback: a run of this batch missed its level-1 region (skewed probe keys) — nothing was delivered, the caller takes round 4's variant.
tsq_status dm_emit_batch(tsq_join* j, const tsq_colset& pcs, int64_t nrows, bool* redo, bool* fallback, const uint8_t* sel = nullptr) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    *fallback = false;
    const bool outer = j->cfg.join_type != TSQ_JOIN_INNER;
    const int np = j->cfg.n_probe_cols, nbc = j->cfg.n_build_cols;
    constexpr int T = 1024 * 8;
    const DaGeom g = da_geometry(j->da_pbits, j->da_ebits, nrows, T);
    if (g.nregions * g.cap >= 0xffffffffULL) { *fallback = true; return TSQ_OK; }
    const size_t slots = g.nregions * g.cap;
    const uint32_t sbits = j->dm_sbits, S = 1u << sbits, P1 = g.P, Q = P1 * S;
    const uint32_t cap1 = 8u * g.cap + 8u * S;
    const size_t slots2 = (size_t)P1 * cap1;
    if (slots2 >= 0xffffffffULL) { *fallback = true; return TSQ_OK; }
    TSQ_TRY(j->rkeys.reserve(ctx, h, g.ent_bytes));
    TSQ_TRY(j->rctl.reserve(ctx, h, g.ctl_bytes));
    TSQ_TRY(j->rvend.reserve(ctx, h, g.nregions * 4));
    TSQ_TRY(j->rovf.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_TRY(j->rovfidx.reserve(ctx, h, (size_t)nrows * 4 + 64));
    if (outer) TSQ_TRY(j->rmiss.reserve(ctx, h, (size_t)nrows * 4 + 64));
    const int kc = j->da_multi ? -1 : j->ks.pidx[0], kb = j->da_multi ? -1 : j->ks.bidx[0];
    int trav[TSQ_DA_MAXCOLS], ntrav = 0;
    for (int c = 0; c < np; c++)
        if (c != kc) trav[ntrav++] = c;
    bool any_nulls = false;
    for (int v = 0; v < ntrav; v++) {
        TSQ_TRY(j->rcols[v].reserve(ctx, h, slots * 8 + 256));
        TSQ_TRY(j->dm_ppay[v].reserve(ctx, h, slots2 * 8 + 256));
        any_nulls = any_nulls || pcs.nulls[trav[v]] != nullptr;
    }
    if (any_nulls) {
        TSQ_TRY(j->rnnmask.reserve(ctx, h, slots + 256));
        TSQ_TRY(j->dm_pnn.reserve(ctx, h, slots2 + 256));
    }
    TSQ_TRY(j->dm_pent.reserve(ctx, h, slots2 * 2 + 256));
    TSQ_TRY(j->dm_poff.reserve(ctx, h, (size_t)Q * 4 + 64));
    TSQ_TRY(j->dm_pcnt.reserve(ctx, h, ((size_t)Q + 1) * 8 + 64));
    DaColStore cs;
    memset(&cs, 0, sizeof cs);
    DaStore& st = cs.st;
    st.ent = j->rkeys.p;
    st.cursor = j->rctl.as<uint32_t>();
    st.ovf_count = st.cursor + g.nregions;
    st.miss_count = st.cursor + g.nregions + 1;
    st.miss = j->rmiss.as<uint32_t>();
    st.valid_end = j->rvend.as<uint32_t>();
    st.ovf = j->rovf.as<uint32_t>();
    st.ovf_idx = j->rovfidx.as<uint32_t>();
    st.ovf_cap = (uint32_t)nrows;
    st.bits = j->da_pbits;
    st.ebits = j->da_ebits;
    st.cap = g.cap;
    for (int v = 0; v < ntrav; v++) cs.pay[v] = j->rcols[v].as<uint64_t>();
    cs.nnmask = any_nulls ? j->rnnmask.as<uint8_t>() : nullptr;
    DmStore p2;
    memset(&p2, 0, sizeof p2);
    p2.ent = j->dm_pent.as<uint16_t>();
    for (int v = 0; v < ntrav; v++) p2.pay[v] = j->dm_ppay[v].as<uint64_t>();
    p2.nnmask = any_nulls ? j->dm_pnn.as<uint8_t>() : nullptr;
    p2.off = j->dm_poff.as<uint32_t>();
    p2.cnt = j->dm_pcnt.as<unsigned long long>();
    p2.cap1 = cap1;
    p2.sbits = sbits;
    p2.ebits2 = j->da_ebits - sbits;
    TSQ_HIP(h, hipMemsetAsync(j->rctl.p, 0, g.ctl_bytes, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->rvend.p, 0xff, g.nregions * 4, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(p2.cnt + Q, 0, 8, ctx->stream));
    DaColSrc src;
    memset(&src, 0, sizeof src);
    TSQ_TRY(da_probe_key(j, pcs, nrows, src.key, sel));
    src.n_cols = ntrav;
    src.any_nulls = any_nulls ? 1 : 0;
    for (int v = 0; v < ntrav; v++) {
        src.col[v] = (const uint64_t*)pcs.data[trav[v]];
        src.nulls[v] = pcs.nulls[trav[v]];
    }
    hipEvent_t* re = j->rev[j->st.radix_batches % tsq_join::RING];
    for (int e = 0; e < 3; e++)
        if (!re[e]) TSQ_HIP(h, hipEventCreate(&re[e]));
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(re[0], ctx->stream));
    {
        const dim3 grid((unsigned)std::min<int64_t>((nrows + T - 1) / T, ctx->num_cus));
        if (outer) hipLaunchKernelGGL((k_da_partition_cols<1024, 8, true>), grid, dim3(1024), 0, ctx->stream, src, j->da_dm, cs);
        else hipLaunchKernelGGL((k_da_partition_cols<1024, 8, false>), grid, dim3(1024), 0, ctx->stream, src, j->da_dm, cs);
        TSQ_HIP(h, hipGetLastError());
    }
    TSQ_HIP(h, hipEventRecord(re[1], ctx->stream));
    TSQ_TRY(dm_launch_split(j, cs, ntrav, p2, !outer));
    TSQ_HIP(h, hipFuncSetAttribute((const void*)k_dm_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)Q + 1) * 8)));
    hipLaunchKernelGGL(k_dm_scan, dim3(1), dim3(1024), ((size_t)Q + 1) * 8, ctx->stream, p2.cnt, Q + 1);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 52, p2.cnt + Q, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 53, st.ovf_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 54, st.miss_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    j->st.kernel_launches += 2;
    const int64_t part_rows = (int64_t)ctx->pinned[52];
    if (((const uint32_t*)(ctx->pinned + 53))[0] != 0) {  // skewed probe keys: a run did not fit its region — round 4's variant has the overflow list
        *fallback = true;
        return TSQ_OK;
    }
    const int64_t miss_rows = outer ? (int64_t)((const uint32_t*)(ctx->pinned + 54))[0] : 0;
    const int64_t exc_rows = miss_rows, out_rows = exc_rows + part_rows;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)st.bits;
    j->st.probe_route = TSQ_ROUTE_PACKED;
    j->st.packed_key_bits = (int32_t)j->da_dm.b;
    j->st.packed_lds_bits = (int32_t)(st.bits + sbits);
    if (out_rows == 0) {
        TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
        TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
        j->have_probe_ev = true;
        return TSQ_OK;
    }
    // ---- the output batch: [NULL-padded rows of the miss list | rows of the final partitions]; NULLs as one byte per row, packed at the end
    const int nout = np + nbc;
    const bool probe_is_left = j->cfg.build_is_right != 0;
    const int nl = probe_is_left ? np : nbc;
    std::unique_ptr<ResultBatch> rb(new ResultBatch());
    rb->rows = out_rows;
    rb->data.resize(nout);
    rb->notnull.resize(nout);
    rb->bitmap.resize(nout);
    rb->offs.resize(nout);
    rb->nbytes.assign(nout, 0);
    std::vector<bool> may_null_v(nout, false);
    DmEmitArgs ea;
    memset(&ea, 0, sizeof ea);
    DaExcArgs xa;
    memset(&xa, 0, sizeof xa);
    auto build_table_col = [&](int sc) {  // build column sc -> its table
        for (int v = 0; v < j->dm_nb; v++)
            if (j->dm_bcol_of[v] == sc) return v;
        return -1;
    };
    for (int oc = 0; oc < nout; oc++) {
        const bool from_probe = probe_is_left ? oc < nl : oc >= nl;
        const int sc = oc < nl ? oc : oc - nl;
        const bool may_null = from_probe ? pcs.nulls[sc] != nullptr : (j->bcols[sc].has_nulls || outer);
        may_null_v[oc] = may_null;
        tsq_status s = rb->data[oc].reserve(ctx, h, ((size_t)out_rows + 8) * 8 + 16);
        if (s == TSQ_OK && may_null) s = rb->notnull[oc].reserve(ctx, h, (size_t)out_rows + 64);
        if (s == TSQ_OK && may_null) s = rb->bitmap[oc].reserve(ctx, h, tsq_bitmap_bytes(out_rows) + 16);
        if (s != TSQ_OK) { rb->release(); return s; }
        // (no preset of the flags: k_dm_emit writes every partition row's flag, k_da_gather_exc every exception row's)
        uint64_t* od = rb->data[oc].as<uint64_t>();
        uint8_t* of = may_null ? rb->notnull[oc].as<uint8_t>() : nullptr;
        if (from_probe) {
            if (sc == kc) {
                ea.out_pkey = od;
                ea.out_pkey_nn = of;  // (a probe key cell of a row that reached a partition is never NULL: the flags are all ones)
            } else {
                const int v = (kc < 0 || sc < kc) ? sc : sc - 1;
                ea.out_probe[v] = od;
                ea.out_probe_nn[v] = of;
            }
            xa.pcol[sc] = (const uint64_t*)pcs.data[sc];
            xa.pnull[sc] = pcs.nulls[sc];
            xa.out_probe[sc] = od;
            xa.out_probe_nn[sc] = of;
        } else {
            if (sc == kb) {
                ea.out_bkey = od;
                ea.out_bkey_nn = of;
            } else {
                const int v = build_table_col(sc);
                ea.out_build[v] = od;
                ea.out_build_nn[v] = of;
            }
            xa.out_build[sc] = od;
            xa.out_build_nn[sc] = of;  // (exception rows of this route are the NULL-padded rows only: no build cell is read)
        }
    }
    if (exc_rows > 0) {  // the NULL-padded rows of the probe rows that never reached a partition (NULL key, key outside the build side's range)
        TSQ_TRY(j->pairs.reserve(ctx, h, (size_t)exc_rows * 8 + 64));
        hipLaunchKernelGGL(k_da_emit_miss, dim3(tsq_grid_for(ctx, miss_rows, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)st.miss, (uint32_t)miss_rows,
                           j->pairs.as<unsigned long long>());
        TSQ_HIP(h, hipGetLastError());
        xa.pairs = j->pairs.as<unsigned long long>();
        xa.bkey_col = kb;
        xa.pkey_col = kc;
        xa.n = exc_rows;
        xa.n_probe = np;
        xa.n_build = nbc;
        hipLaunchKernelGGL(k_da_gather_exc, dim3(tsq_grid_for(ctx, exc_rows, 256)), dim3(256), 0, ctx->stream, xa);
        TSQ_HIP(h, hipGetLastError());
        j->st.kernel_launches += 2;
    }
    if (part_rows > 0) {
        ea.pst = p2;
        ea.bst.ent = j->dm_bent.as<uint16_t>();
        for (int v = 0; v < j->dm_nb; v++) ea.bst.pay[v] = j->dm_bpay[v].as<uint64_t>();
        ea.bst.nnmask = j->dm_bnulls ? j->dm_bnn.as<uint8_t>() : nullptr;
        ea.bst.off = j->dm_boff.as<uint32_t>();
        ea.bst.cnt = j->dm_bcnt.as<unsigned long long>();
        ea.bst.cap1 = j->dm_cap1_b;
        ea.bst.sbits = sbits;
        ea.bst.ebits2 = p2.ebits2;
        ea.dm = j->da_dm;
        ea.pbits = st.bits + sbits;
        ea.row0 = (unsigned long long)exc_rows;
        ea.tab_rows = j->dm_tab_rows;
        ea.n_probe = ntrav;
        ea.n_build = j->dm_nb;
        const size_t lds = dm_emit_lds(p2.ebits2, ea.tab_rows, ea.n_build, j->dm_bnulls);
        const uint32_t per_cu = lds <= (size_t)76 * 1024 ? 2u : 1u;
        const dim3 egrid(std::min<uint32_t>(Q, (uint32_t)ctx->num_cus * per_cu));
        if (per_cu == 2) {  // two 512-thread workgroups per CU: one fills its table while the other streams (measured equal to one of 1024 threads)
            const void* fn = outer ? (const void*)k_dm_emit<512, true> : (const void*)k_dm_emit<512, false>;
            TSQ_HIP(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (outer) hipLaunchKernelGGL((k_dm_emit<512, true>), egrid, dim3(512), lds, ctx->stream, ea);
            else hipLaunchKernelGGL((k_dm_emit<512, false>), egrid, dim3(512), lds, ctx->stream, ea);
        } else {
            const void* fn = outer ? (const void*)k_dm_emit<1024, true> : (const void*)k_dm_emit<1024, false>;
            TSQ_HIP(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (outer) hipLaunchKernelGGL((k_dm_emit<1024, true>), egrid, dim3(1024), lds, ctx->stream, ea);
            else hipLaunchKernelGGL((k_dm_emit<1024, false>), egrid, dim3(1024), lds, ctx->stream, ea);
        }
        TSQ_HIP(h, hipGetLastError());
        j->st.kernel_launches++;
    }
    for (int oc = 0; oc < nout; oc++)
        if (may_null_v[oc]) TSQ_TRY(tsq_launch_pack_bitmap(ctx, h, rb->notnull[oc].as<uint8_t>(), rb->bitmap[oc].as<uint8_t>(), out_rows));
    if (!j->conds_h.empty()) {
        tsq_status ps = da_post_conditions(j, *rb, may_null_v, redo, nullptr);
        if (ps != TSQ_OK || *redo) {
            rb->release();
            if (*redo) {  // as if this route had not been tried
                j->st.radix_batches--;
                j->st.probe_route = TSQ_ROUTE_DIRECT;
                j->st.packed_lds_bits = 0;
            }
            return ps;
        }
    }
    TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
    j->have_probe_ev = true;
    if (rb->rows == 0) {
        rb->release();
        return TSQ_OK;
    }
    return deliver_batch(j, std::move(rb), may_null_v);
}

// ---------------------------------------------------------------- materialising radix path (host side)
// HashJoinExec.Next materialises every joined row (executor/join.go:125-146, joiner.go:351-378, chunk.go:334-356).  The direct
// route sizes the batch with an unpartitioned probe (two random lines per probe row) and then gathers every output cell through
// a row id (one random 8-byte read per cell: 42 G/s).  Here nothing is gathered from all over HBM:
//   probe side : the payload columns travel with the key through the radix partition (tsq_radix.h, <= 2 columns)
//   build side : the payload columns are kept a second time in TABLE-SLOT order (k_table_payload, once per build), so the emit
//                pass reads them next to the slices it is probing
//   keys       : both key columns are recovered from the table word (mix64 is a bijection)
//   rows       : k_lds_probe_count<MODE 1> sizes every ticket, an exclusive scan turns that into output bases, <MODE 2> probes
//                again and writes the joined rows column by column (unspecified row order, as in the reference).
// Eligible: inner join on one BIGINT key of the same signedness, no conditions / filters / selected[] / ordered output,
// <= 3 eight-byte columns per side without NULLs, a sliced table.  Everything else keeps the direct route.
bool radix_emit_eligible(const tsq_join* j, const tsq_colset& pcs, int64_t nrows, const uint8_t* selected_dev) {
    if (j->radix_mode == TSQ_RADIX_OFF || j->count_only || j->multi || j->general || selected_dev || j->never_match || j->ordered || j->chained) return false;
    if (j->tb < TSQ_RADIX_MIN_BITS || nrows <= 0 || nrows > 0x7fffffffLL) return false;
    if (j->cfg.n_probe_cols > 1 + TSQ_LDS_MAXPAY || j->cfg.n_build_cols > 1 + TSQ_LDS_MAXPAY) return false;
    const int32_t kt = j->cfg.build_types[j->ks.bidx[0]];
    if ((kt != TSQ_I64 && kt != TSQ_U64) || j->cfg.probe_types[j->ks.pidx[0]] != kt) return false;
    for (int c = 0; c < j->cfg.n_probe_cols; c++)
        if (j->cfg.probe_types[c] == TSQ_F32 || j->cfg.probe_types[c] == TSQ_BYTES || pcs.nulls[c]) return false;
    for (int c = 0; c < j->cfg.n_build_cols; c++)
        if (j->cfg.build_types[c] == TSQ_F32 || j->cfg.build_types[c] == TSQ_BYTES || j->bcols[c].has_nulls) return false;
    if (!radix_plan_for(j).lds) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE) return true;
    return nrows >= (4 << 20);
}

tsq_status ensure_table_payload(tsq_join* j) {
    if (j->tpay_ready) return TSQ_OK;
    tsq_ctx* ctx = j->ctx;
    const uint64_t nslots = j->nbuckets * TSQ_BUCKET;
    int v = 0;
    for (int c = 0; c < j->cfg.n_build_cols; c++) {
        if (c == j->ks.bidx[0]) continue;
        TSQ_TRY(j->tpay[v].reserve(ctx, &j->hdr, nslots * 8 + 64));
        hipLaunchKernelGGL(k_table_payload, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, j->tkeys.as<uint64_t>(), j->tvals.as<uint32_t>(),
                           j->bcols[c].data.as<uint64_t>(), j->tpay[v].as<uint64_t>(), nslots);
        TSQ_HIP(&j->hdr, hipGetLastError());
        j->st.kernel_launches++;
        v++;
    }
    j->tpay_ready = true;
    return TSQ_OK;
}

tsq_status radix_emit(tsq_join* j, const tsq_colset& pcs, int64_t nrows) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    TSQ_TRY(ensure_table_payload(j));
    const RadixPlan pl = radix_plan_for(j);
    RadixStore st;
    memset(&st, 0, sizeof st);
    st.bits = pl.bits;
    st.R = 8;
    const uint32_t P = 1u << st.bits;
    const int V = j->cfg.n_probe_cols - 1;
    const int K = V == 0 ? 16 : (V == 1 ? 8 : 4), T = 1024 * K;
    const double lam = (double)nrows / ((double)P * 8.0);
    st.cap = (uint32_t)(lam * 1.08 + 8.0 * sqrt(lam) + 2.0 * T / 64.0 + 64.0);
    st.cap = (st.cap + 15u) & ~15u;
    const size_t nregions = (size_t)P * 8;
    if (nregions * st.cap >= 0xffffffffULL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "radix probe batch too large");
    const size_t ctl_words32 = nregions + 16;
    const size_t q_off = (ctl_words32 * 4 + 511) & ~(size_t)511, q_bytes = 8 * TSQ_RADIX_QSTRIDE * 8, ctl_bytes = q_off + q_bytes;
    TSQ_TRY(j->rkeys.reserve(ctx, h, nregions * st.cap * 8 + 256));
    TSQ_TRY(j->rctl.reserve(ctx, h, ctl_bytes));
    TSQ_TRY(j->rvend.reserve(ctx, h, nregions * 4));
    TSQ_TRY(j->rovf.reserve(ctx, h, (size_t)nrows * 8 + 64));
    for (int v = 0; v < V; v++) {
        TSQ_TRY(j->rpay[v].reserve(ctx, h, nregions * st.cap * 8 + 256));
        TSQ_TRY(j->rovfpay[v].reserve(ctx, h, (size_t)nrows * 8 + 64));
        st.pay[v] = j->rpay[v].as<uint64_t>();
        st.ovf_pay[v] = j->rovfpay[v].as<uint64_t>();
    }
    st.keys = j->rkeys.as<uint64_t>();
    st.cursor = j->rctl.as<uint32_t>();
    st.ovf_count = st.cursor + nregions;
    st.queue = (unsigned long long*)((char*)j->rctl.p + q_off);
    st.valid_end = j->rvend.as<uint32_t>();
    st.ovf_keys = j->rovf.as<uint64_t>();
    st.ovf_cap = (uint32_t)nrows;
    const uint32_t ntk = (P >> 3) * pl.S;
    const size_t n_tk = (size_t)8 * ntk + 1;  // [xcd][ticket], then one more word: the exclusive scan leaves the total there
    TSQ_TRY(j->tkcnt.reserve(ctx, h, n_tk * 8 + 64));
    TSQ_HIP(h, hipMemsetAsync(j->rctl.p, 0, ctl_bytes, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->rvend.p, 0xff, nregions * 4, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(j->tkcnt.p, 0, n_tk * 8, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(ctx->dscratch + 4, 0, 16, ctx->stream));  // [4] joined rows of the overflow list, [5] its output cursor

    RadixSrc src;
    memset(&src, 0, sizeof src);
    const int kc = j->ks.pidx[0];
    src.data = pcs.data[kc];
    src.type = pcs.type[kc];
    src.nrows = nrows;
    {
        int v = 0;
        for (int c = 0; c < j->cfg.n_probe_cols; c++) {
            if (c == kc) continue;
            src.vdata[v] = pcs.data[c];
            src.vtype[v] = pcs.type[c];
            v++;
        }
    }
    const int64_t ntiles = (nrows + T - 1) / T;
    const int pgrid = (int)std::min<int64_t>(ntiles, ctx->num_cus);
    hipEvent_t* re = j->rev[j->st.radix_batches % tsq_join::RING];
    for (int e = 0; e < 3; e++)
        if (!re[e]) TSQ_HIP(h, hipEventCreate(&re[e]));
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(re[0], ctx->stream));
    if (V == 0) hipLaunchKernelGGL((k_radix_partition<1024, 16, 4, 0, false, true>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
    else if (V == 1) hipLaunchKernelGGL((k_radix_partition<1024, 8, 4, 1, false, true>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
    else hipLaunchKernelGGL((k_radix_partition<1024, 4, 4, 2, false, true>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipEventRecord(re[1], ctx->stream));
    constexpr int LNT = 1024;
    LdsProbeArgs la;
    memset(&la, 0, sizeof la);
    la.st = st;
    fill_table(j, la.t);
    la.S = pl.S;
    la.nf = pl.nf;
    la.unique = j->unique ? 1u : 0u;
    la.counters = j->counters.as<unsigned long long>();
    la.tk_cnt = j->tkcnt.as<unsigned long long>();
    const int lgrid = std::max(1, ctx->num_cus / 8) * 8;
    const size_t img = (size_t)pl.nf * j->bs * 64;
    const size_t lds1 = img + (size_t)(LNT / 64) * TSQ_LDS_RING_BYTES, lds2 = img + (size_t)(LNT / 64) * TSQ_LDS_RING_BYTES_EMIT;
    if (lds2 + 256 > 160 * 1024) return tsq_fail(h, TSQ_ERR_INVALID, "internal: LDS image + rings exceed 160 KB");
    // ---- sizing pass
    TSQ_HIP(h, hipFuncSetAttribute((const void*)k_lds_probe_count<LNT, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    hipLaunchKernelGGL((k_lds_probe_count<LNT, false, 1>), dim3(lgrid), dim3(LNT), lds1, ctx->stream, la);
    TSQ_HIP(h, hipGetLastError());
    RadixProbeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.st = st;
    pa.t = la.t;
    pa.counters = (unsigned long long*)(ctx->dscratch + 4);
    hipLaunchKernelGGL(k_radix_probe_ovf, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, pa);
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, la.tk_cnt, (int)n_tk);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 32, la.tk_cnt + (n_tk - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 33, ctx->dscratch + 4, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    const int64_t tk_rows = (int64_t)ctx->pinned[32], ovf_rows = (int64_t)ctx->pinned[33], out_rows = tk_rows + ovf_rows;
    j->st.kernel_launches += 4;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)st.bits;
    j->st.probe_route = TSQ_ROUTE_RADIX_LDS;
    if (out_rows == 0) {
        TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
        TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
        j->have_probe_ev = true;
        return TSQ_OK;
    }
    // ---- output batch
    const int nout = j->cfg.n_probe_cols + j->cfg.n_build_cols;
    const bool probe_is_left = j->cfg.build_is_right != 0;
    const int nl = probe_is_left ? j->cfg.n_probe_cols : j->cfg.n_build_cols;
    std::unique_ptr<ResultBatch> rb(new ResultBatch());
    rb->rows = out_rows;
    rb->data.resize(nout);
    rb->notnull.resize(nout);
    rb->bitmap.resize(nout);
    rb->offs.resize(nout);
    rb->nbytes.assign(nout, 0);
    std::vector<bool> may_null_v(nout, false);
    for (int oc = 0; oc < nout; oc++) {
        tsq_status s = rb->data[oc].reserve(ctx, h, ((size_t)out_rows + 8) * 8 + 16);
        if (s != TSQ_OK) { rb->release(); return s; }
    }
    {
        int pv = 0, bv = 0;
        la.n_out_probe = j->cfg.n_probe_cols;
        la.n_out_build = j->cfg.n_build_cols;
        for (int c = 0; c < j->cfg.n_probe_cols; c++) {
            la.out_probe[c] = rb->data[probe_is_left ? c : nl + c].as<uint64_t>();
            la.probe_src[c] = c == kc ? -1 : pv++;
        }
        for (int c = 0; c < j->cfg.n_build_cols; c++) {
            la.out_build[c] = rb->data[probe_is_left ? nl + c : c].as<uint64_t>();
            if (c == j->ks.bidx[0]) {
                la.build_src[c] = -1;
            } else {
                la.bpay[bv] = j->tpay[bv].as<uint64_t>();
                la.bcol[bv] = j->bcols[c].data.as<uint64_t>();
                la.build_src[c] = bv++;
            }
        }
    }
    // ---- emit pass: the same tickets again (queue heads back to zero), then the overflow list behind the tickets' rows
    TSQ_HIP(h, hipMemsetAsync(st.queue, 0, q_bytes, ctx->stream));
    ctx->pinned[34] = (uint64_t)tk_rows;
    TSQ_HIP(h, hipMemcpyAsync(ctx->dscratch + 5, ctx->pinned + 34, 8, hipMemcpyHostToDevice, ctx->stream));
    la.ovf_cursor = (unsigned long long*)(ctx->dscratch + 5);
    TSQ_HIP(h, hipFuncSetAttribute((const void*)k_lds_probe_count<LNT, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    hipLaunchKernelGGL((k_lds_probe_count<LNT, false, 2>), dim3(lgrid), dim3(LNT), lds2, ctx->stream, la);
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_lds_emit_ovf, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, la);
    TSQ_HIP(h, hipGetLastError());
    j->st.kernel_launches += 2;
    TSQ_HIP(h, hipEventRecord(re[2], ctx->stream));
    TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
    j->have_probe_ev = true;
    return deliver_batch(j, std::move(rb), may_null_v);
}

// The joined rows of a batch are known as (probe row, build row) pairs (build row TSQ_PAIR_MISS = the NULL-padded row of an outer
// join): allocate the output batch, let `produce_pairs` fill a.pairs[0, out_rows), copy the columns through the pairs (K4b / K4c)
// and hand the batch to tsq_join_pull.  Shared by the direct route (K3 + K4a) and the packed-key route (tsq_dajoin.h).
template <class F>
tsq_status materialise_pairs(tsq_join* j, const tsq_colset& pcs, ProbeArgs& a, int64_t nrows, int64_t out_rows, F&& produce_pairs) {
    tsq_ctx* ctx = j->ctx;
    const int nout = j->cfg.n_probe_cols + j->cfg.n_build_cols;
    std::unique_ptr<ResultBatch> rb(new ResultBatch());
    rb->rows = out_rows;
    rb->data.resize(nout);
    rb->notnull.resize(nout);
    rb->bitmap.resize(nout);
    rb->offs.resize(nout);
    rb->nbytes.assign(nout, 0);
    const bool outer = j->cfg.join_type != TSQ_JOIN_INNER;
    const int nl = a.probe_is_left ? j->cfg.n_probe_cols : j->cfg.n_build_cols;
    if (nrows > 0xffffffffLL) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "internal: probe slice too large");
    TSQ_TRY(j->pairs.reserve(ctx, &j->hdr, (size_t)out_rows * 8 + 64));
    a.pairs = j->pairs.as<unsigned long long>();
    GatherArgs ga;
    memset(&ga, 0, sizeof ga);
    ga.pairs = a.pairs;
    ga.rows = out_rows;
    std::vector<bool> may_null_v(nout);
    for (int oc = 0; oc < nout; oc++) {
        const bool from_probe = a.probe_is_left ? oc < nl : oc >= nl;
        const int sc = oc < nl ? oc : oc - nl;
        const int32_t type = from_probe ? j->cfg.probe_types[sc] : j->cfg.build_types[sc];
        const bool src_nulls = from_probe ? pcs.nulls[sc] != nullptr : j->bcols[sc].has_nulls;
        const bool may_null = src_nulls || (outer && !from_probe);
        may_null_v[oc] = may_null;
        GatherCol& gc0 = ga.col[oc];
        if (!j->used_out.empty() && !j->used_out[(size_t)oc]) {  // the parent never reads this column: it is not gathered at all
            gc0.es = 0;
            gc0.from_probe = from_probe ? 1 : 0;
            may_null_v[oc] = false;
            continue;
        }
        tsq_status s = type == TSQ_BYTES ? rb->offs[oc].reserve(ctx, &j->hdr, ((size_t)out_rows + 2) * 8 + 16)
                                         : rb->data[oc].reserve(ctx, &j->hdr, ((size_t)out_rows + 8) * tsq_elem_size(type) + 16);
        if (s == TSQ_OK && may_null) s = rb->bitmap[oc].reserve(ctx, &j->hdr, tsq_bitmap_bytes(out_rows) + 16);
        if (s != TSQ_OK) { rb->release(); return s; }
        GatherCol& gc = ga.col[oc];
        if (type == TSQ_BYTES) {  // es == 0: skipped by k_gather_cols, gathered below
            gc.es = 0;
            gc.from_probe = from_probe ? 1 : 0;
            continue;
        }
        gc.src = from_probe ? a.p.data[sc] : a.b.data[sc];
        gc.src_nulls = src_nulls ? (from_probe ? a.p.nulls[sc] : a.b.nulls[sc]) : nullptr;
        gc.dst = rb->data[oc].p;
        gc.dst_bitmap = may_null ? rb->bitmap[oc].as<uint8_t>() : nullptr;
        gc.es = tsq_elem_size(type);
        gc.from_probe = from_probe ? 1 : 0;
        // The build-side key of an inner join on one integer column of the same type IS the probe key (same flag, same 8
        // bytes: codec.go:212-240): copy it from the probe column — consecutive rows — instead of gathering it through the
        // build row id (a random 8-byte read per joined row, 42 G/s).
        if (!from_probe && !outer && !j->multi && sc == j->ks.bidx[0]) {
            const int pkc = j->ks.pidx[0];
            if (j->cfg.probe_types[pkc] == type && (type == TSQ_I64 || type == TSQ_U64)) {
                gc.src = a.p.data[pkc];
                gc.src_nulls = a.p.nulls[pkc];
                gc.from_probe = 1;
            }
        }
    }
    TSQ_TRY(produce_pairs());
    {
        const int64_t groups = (out_rows + 7) / 8;
        const int gx = (int)std::min<int64_t>((groups + 255) / 256, (int64_t)ctx->num_cus * 8);
        hipLaunchKernelGGL(k_gather_cols, dim3(gx, nout), dim3(256), 0, ctx->stream, ga);
        TSQ_HIP(&j->hdr, hipGetLastError());
        j->st.kernel_launches++;
    }
    // var-len output columns: lengths -> offsets (exclusive scan) -> bytes
    for (int oc = 0; oc < nout; oc++) {
        const bool from_probe = a.probe_is_left ? oc < nl : oc >= nl;
        const int sc = oc < nl ? oc : oc - nl;
        const int32_t type = from_probe ? j->cfg.probe_types[sc] : j->cfg.build_types[sc];
        if (type != TSQ_BYTES || (!j->used_out.empty() && !j->used_out[(size_t)oc])) continue;
        VarGatherArgs va;
        memset(&va, 0, sizeof va);
        va.pairs = a.pairs;
        va.rows = out_rows;
        va.from_probe = from_probe ? 1 : 0;
        va.src_offs = from_probe ? pcs.offs[sc] : j->bcols[sc].offs.as<int64_t>();
        va.src_data = (const uint8_t*)(from_probe ? pcs.data[sc] : j->bcols[sc].data.p);
        va.src_nulls = from_probe ? pcs.nulls[sc] : (j->bcols[sc].has_nulls ? j->bcols[sc].nulls.as<uint8_t>() : nullptr);
        va.out_offs = rb->offs[oc].as<int64_t>();
        va.out_bitmap = may_null_v[oc] ? rb->bitmap[oc].as<uint8_t>() : nullptr;
        const int gl = tsq_grid_for(ctx, (out_rows + 7) / 8, 256);
        hipLaunchKernelGGL(k_varlen_len, dim3(gl), dim3(256), 0, ctx->stream, va);
        TSQ_HIP(&j->hdr, hipGetLastError());
        DevBuf scratch;
        tsq_status s = tsq_launch_scan64(ctx, &j->hdr, va.out_offs, out_rows, scratch);
        if (s == TSQ_OK) {
            hipError_t e = hipMemcpyAsync(ctx->pinned + 42, va.out_offs + out_rows, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) s = tsq_fail(&j->hdr, TSQ_ERR_HIP, std::string("var-len gather: ") + hipGetErrorString(e));
        }
        scratch.release();
        if (s != TSQ_OK) { rb->release(); return s; }
        const int64_t nbytes = (int64_t)ctx->pinned[42];
        rb->nbytes[oc] = nbytes;
        s = rb->data[oc].reserve(ctx, &j->hdr, (size_t)nbytes + 64);
        if (s != TSQ_OK) { rb->release(); return s; }
        va.out_data = rb->data[oc].as<uint8_t>();
        if (nbytes > 0) {
            if (nbytes / out_rows > 32) hipLaunchKernelGGL(k_varlen_copy_wave, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, va);
            else hipLaunchKernelGGL(k_varlen_copy_rows, dim3(tsq_grid_for(ctx, out_rows, 256)), dim3(256), 0, ctx->stream, va);
            TSQ_HIP(&j->hdr, hipGetLastError());
        }
        j->st.kernel_launches += 5;
    }
    TSQ_HIP(&j->hdr, hipEventRecord(j->ev[3], ctx->stream));
    j->have_probe_ev = true;
    return deliver_batch(j, std::move(rb), may_null_v);
}

// ---- several integer key columns whose fields need 29..63 bits: COUNT(*) through a single-key CHILD join on the composite (round 4)
// The packed composite (k_da_compose) is exact for any total width up to 63 bits: equal composites <=> equal cells in every column
// (codec.go:243-338).  Beyond 28 bits the packed images do not take it, but every SINGLE-key route does: the build side's composite
// column becomes the one BIGINT UNSIGNED key column of a child join, a probe batch is composed and pushed to the child, and the child
// picks its route (bit cells, the 64-bit LDS route, ...) as for any 64-bit key.  A row that cannot match (a NULL key cell, a probe cell
// outside its field) composes to ~0: the child compares BIGINT UNSIGNED with BIGINT, where cells >= 2^63 never match (codec.go:219-224).
tsq_status probe_batch(tsq_join* j, const tsq_colset& pcs, int64_t nrows, const uint8_t* selected_dev);
// ---------------------------------------------------------------- key-record route (host side; tsq_keyrec.h)
bool kr_count_eligible(const tsq_join* j, int64_t nrows, const uint8_t* selected_dev) {
    if (j->radix_mode == TSQ_RADIX_OFF || !j->multi || j->kr_state < 0 || tsq_knob(j->ctx, TSQ_KNOB_KEYREC, 1) == 0) return false;
    if (!j->count_only || j->checksum || j->general_cfg || selected_dev || j->never_match || j->ordered) return false;
    if (nrows <= 0 || nrows > 0x7fffffffLL) return false;
    const int64_t nb = j->bcols[j->ks.bidx[0]].rows;
    if (nb <= 0 || nb > (int64_t)TSQ_KR_MAXP * TSQ_KR_FILL) return false;  // (larger build sides: the partitions would not fit the LDS tables)
    if (j->radix_mode == TSQ_RADIX_FORCE) return true;
    return nrows >= (1 << 16) && nb >= (1 << 16);
}
// hist -> offsets -> scan [-> check] -> scatter of one side.  `check`: read the flags after the scan (a synchronisation): *ok = every
// record fits and no partition holds more than TSQ_KR_CAP of them
// the digests of the string key columns of one side (digest mode): dig[k] <- k_kr_digest of column key_cols[k]
tsq_status kr_digests(tsq_join* j, const tsq_colset& cs, const int32_t* key_cols, int64_t nrows, DevBuf* dig) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    for (int k = 0; k < j->ks.n_keys; k++) {
        const int c = key_cols[k];
        if (cs.type[c] != TSQ_BYTES) continue;
        TSQ_TRY(dig[k].reserve(ctx, h, (size_t)nrows * 8 + 64));
        KrDigestArgs da;
        memset(&da, 0, sizeof da);
        da.data = (const uint8_t*)cs.data[c];
        da.offs = cs.offs[c];
        da.nulls = cs.nulls[c];
        da.nrows = nrows;
        da.out = dig[k].as<uint64_t>();
        da.weak = tsq_knob(ctx, TSQ_KNOB_KEYREC, 1) == 3 ? 1 : 0;
        // long cells: a wave per row (the host knows the average length from the offsets' ends)
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 43, cs.offs[c], 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 44, cs.offs[c] + nrows, 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        const int64_t bytes = (int64_t)ctx->pinned[44] - (int64_t)ctx->pinned[43];
        if (nrows > 0 && bytes / nrows >= 64) hipLaunchKernelGGL(k_kr_digest<true>, dim3(tsq_grid_for(ctx, nrows * 64, 256)), dim3(256), 0, ctx->stream, da);
        else hipLaunchKernelGGL(k_kr_digest<false>, dim3(tsq_grid_for(ctx, nrows, 256)), dim3(256), 0, ctx->stream, da);
        TSQ_HIP(h, hipGetLastError());
        j->st.kernel_launches++;
    }
    return TSQ_OK;
}
tsq_status kr_pass(tsq_join* j, const tsq_colset& cs, const int32_t* key_cols, int64_t nrows, uint32_t pbits, DevBuf& counts, DevBuf& pstart, DevBuf& rec,
                   bool check, bool* ok, DevBuf* ids = nullptr, const uint8_t* selected = nullptr, bool list_rows_without_key = false, const DevBuf* dig = nullptr,
                   bool* toolong_out = nullptr) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const uint32_t P = 1u << pbits;
    KrArgs a;
    memset(&a, 0, sizeof a);
    a.src.cs = cs;
    a.src.n_keys = j->ks.n_keys;
    for (int k = 0; k < j->ks.n_keys; k++) a.src.col[k] = key_cols[k];
    a.src.nrows = nrows;
    a.src.selected = selected;
    {
        int32_t kt[TSQ_MAX_KEYS];
        for (int k = 0; k < j->ks.n_keys; k++) kt[k] = cs.type[key_cols[k]];
        a.src.layout = kr_layout_of(kt, j->ks.n_keys);
    }
    if (dig) {  // digest mode: the string cells as (length, digest); run-time cell positions
        a.src.layout = 0;
        for (int k = 0; k < j->ks.n_keys; k++) a.src.digest[k] = cs.type[key_cols[k]] == TSQ_BYTES ? dig[k].as<uint64_t>() : nullptr;
    }
    a.pbits = pbits;
    const int64_t chunks = (nrows + TSQ_KR_NT - 1) / TSQ_KR_NT;
    a.n_wg = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(1024, std::max<int64_t>(8, tsq_knob(ctx, TSQ_KNOB_KR_WG, TSQ_KR_MAXWG))), chunks));
    a.rows_per_wg = ((chunks + a.n_wg - 1) / a.n_wg) * TSQ_KR_NT;
    if (list_rows_without_key) {  // the outer side of an outer join: rows with a NULL key cell, cells beyond a record, selected == 0
        TSQ_TRY(j->kr_norec.reserve(ctx, h, (size_t)nrows * 4 + 64));
        a.norec = j->kr_norec.as<uint32_t>();
        a.norec_all = 1;
        a.norec_count = (unsigned long long*)(ctx->dscratch + 58);
        TSQ_HIP(h, hipMemsetAsync(ctx->dscratch + 58, 0, 8, ctx->stream));
    }
    TSQ_TRY(counts.reserve(ctx, h, (size_t)a.n_wg * P * 4 + 64));
    TSQ_TRY(pstart.reserve(ctx, h, ((size_t)P + 1) * 4 + 64));
    TSQ_TRY(rec.reserve(ctx, h, (size_t)nrows * TSQ_KR_BYTES + 64));
    TSQ_TRY(j->kr_flags.reserve(ctx, h, 64));
    if (ids) {
        TSQ_TRY(ids->reserve(ctx, h, (size_t)nrows * 4 + 64));
        a.ids = ids->as<uint32_t>();
    }
    a.counts = counts.as<uint32_t>();
    a.pstart = pstart.as<uint32_t>();
    a.rec = rec.as<unsigned long long>();
    a.flags = j->kr_flags.as<uint32_t>();
    if (check) TSQ_HIP(h, hipMemsetAsync(a.flags, 0, 16, ctx->stream));
    const size_t lds = (size_t)P * 4;
    TSQ_HIP(h, hipFuncSetAttribute((const void*)k_kr_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TSQ_HIP(h, hipFuncSetAttribute((const void*)k_kr_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_kr_hist, dim3(a.n_wg), dim3(TSQ_KR_NT), lds, ctx->stream, a);
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_kr_offsets, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, a);
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_kr_scan, dim3(1), dim3(1024), 0, ctx->stream, a.pstart, P, a.flags);
    TSQ_HIP(h, hipGetLastError());
    if (check) {
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 40, a.flags, 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        const uint32_t* f = (const uint32_t*)(ctx->pinned + 40);
        *ok = (f[0] & 1u) == 0 && f[1] <= TSQ_KR_CAP;
        if (toolong_out) *toolong_out = (f[0] & 1u) != 0;
        if (!*ok) return TSQ_OK;
    }
    hipLaunchKernelGGL(k_kr_scatter, dim3(a.n_wg), dim3(TSQ_KR_NT), lds, ctx->stream, a);
    TSQ_HIP(h, hipGetLastError());
    j->st.kernel_launches += 4;
    return TSQ_OK;
}
tsq_status kr_prepare(tsq_join* j) {
    if (j->kr_state) return TSQ_OK;
    j->kr_state = -1;
    const int64_t nb = j->bcols[j->ks.bidx[0]].rows;
    uint32_t pbits = 0;
    while (((int64_t)1 << pbits) * TSQ_KR_FILL < nb && (1u << pbits) < TSQ_KR_MAXP) pbits++;  // ~8192 build records per partition (half of them at the power of two above)
    while (pbits < 9 && ((int64_t)256 << pbits) <= nb) pbits++;  // ... and two workgroups for every CU while a partition keeps 128 records or more (one partition = one workgroup)
    tsq_colset bcs;
    tsq_fill_colset(bcs, j->bcols);
    bool ok = false, toolong = false;
    tsq_status s = kr_pass(j, bcs, j->ks.bidx, nb, pbits, j->kr_counts, j->kr_bstart, j->kr_brec, true, &ok, &j->kr_bids, nullptr, false, nullptr, &toolong);
    if (s == TSQ_OK && !ok && toolong) {
        // a key that does not fit a record: its string cells as (length, digest of the bytes) — 13 bytes each — and the matches verified
        // byte for byte (the reference's own benchmark joins on a 5 KiB varstring, executor/benchmark_test.go:328-360)
        bool any_str = false;
        uint32_t need = 0;
        for (int k = 0; k < j->ks.n_keys; k++) {
            const bool str = j->cfg.build_types[j->ks.bidx[k]] == TSQ_BYTES;
            any_str = any_str || str;
            need += str ? TSQ_KR_DIGEST_CELL : 9u;
        }
        if (any_str && need <= TSQ_KR_BYTES) {
            s = kr_digests(j, bcs, j->ks.bidx, nb, j->kr_bdig);
            if (s == TSQ_OK) s = kr_pass(j, bcs, j->ks.bidx, nb, pbits, j->kr_counts, j->kr_bstart, j->kr_brec, true, &ok, &j->kr_bids, nullptr, false, j->kr_bdig);
            j->kr_digest = s == TSQ_OK && ok;
        }
    }
    if (s != TSQ_OK || !ok) {
        for (auto& b : j->kr_bdig) b.release();
        j->kr_bids.release();  // a key that does not fit a record, or a partition too large for LDS (one key with thousands of rows): the other routes keep this join
        for (DevBuf* b : {&j->kr_counts, &j->kr_bstart, &j->kr_brec}) b->release();
        return s;
    }
    j->kr_pbits = pbits;
    j->kr_state = 1;
    return TSQ_OK;
}
static void kr_launch_probe(tsq_ctx* ctx, uint32_t grid, const KrProbeArgs& pa) {
    if (pa.n_verify) hipLaunchKernelGGL(k_kr_probe<true>, dim3(grid), dim3(TSQ_KR_PNT), 0, ctx->stream, pa);
    else hipLaunchKernelGGL(k_kr_probe<false>, dim3(grid), dim3(TSQ_KR_PNT), 0, ctx->stream, pa);
}
// digest mode: what the probe kernel compares byte for byte — the string key columns of both sides
void kr_verify_args(tsq_join* j, const tsq_colset& pcs, KrProbeArgs& pa) {
    if (!j->kr_digest) return;
    for (int k = 0; k < j->ks.n_keys; k++) {
        const int bc = j->ks.bidx[k], pc = j->ks.pidx[k];
        if (j->cfg.build_types[bc] != TSQ_BYTES) continue;
        const int v = pa.n_verify++;
        pa.vb_data[v] = j->bcols[bc].data.as<uint8_t>();
        pa.vb_offs[v] = j->bcols[bc].offs.as<int64_t>();
        pa.vp_data[v] = (const uint8_t*)pcs.data[pc];
        pa.vp_offs[v] = pcs.offs[pc];
    }
    pa.bids = j->kr_bids.as<uint32_t>();
    pa.pids = j->kr_pids.as<uint32_t>();
}
tsq_status kr_count_batch(tsq_join* j, const tsq_colset& pcs, int64_t nrows) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    bool ok = true;
    if (j->kr_digest) {
        TSQ_TRY(kr_digests(j, pcs, j->ks.pidx, nrows, j->kr_pdig));
        TSQ_TRY(kr_pass(j, pcs, j->ks.pidx, nrows, j->kr_pbits, j->kr_counts, j->kr_pstart, j->kr_prec, false, &ok, &j->kr_pids, nullptr, false, j->kr_pdig));
    } else
    TSQ_TRY(kr_pass(j, pcs, j->ks.pidx, nrows, j->kr_pbits, j->kr_counts, j->kr_pstart, j->kr_prec, false, &ok));
    KrProbeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.brec = j->kr_brec.as<unsigned long long>();
    pa.bstart = j->kr_bstart.as<uint32_t>();
    pa.prec = j->kr_prec.as<unsigned long long>();
    pa.pstart = j->kr_pstart.as<uint32_t>();
    pa.P = 1u << j->kr_pbits;
    pa.counters = j->counters.as<unsigned long long>();
    pa.flags = j->kr_flags.as<uint32_t>();
    kr_verify_args(j, pcs, pa);
    const int grid = (int)std::min<uint32_t>(pa.P, (uint32_t)ctx->num_cus * 2);
    kr_launch_probe(ctx, grid, pa);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
    j->have_probe_ev = true;
    j->st.kernel_launches++;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)j->kr_pbits;
    j->st.keyrec_digests = j->kr_digest ? 1 : 0;
    j->st.probe_route = TSQ_ROUTE_KEYREC;
    return TSQ_OK;
}

// the MATERIALISING form: the joined (probe row, build row) pairs of the records, then the usual column gather (k_gather_cols, var-len
// columns included) — inner joins without conditions; the reference's BenchmarkHashJoinExec shape (benchmark_test.go:352-360) with its rows
bool kr_emit_eligible(const tsq_join* j, int64_t nrows, const uint8_t* selected_dev) {
    if (j->radix_mode == TSQ_RADIX_OFF || !j->multi || j->kr_state < 0 || tsq_knob(j->ctx, TSQ_KNOB_KEYREC, 1) == 0) return false;
    // inner and outer joins; outer-side filters arrive as flags (fold_outer_filters), OtherConditions keep the direct route
    (void)selected_dev;
    if (j->count_only || !j->conds_h.empty() || (!j->filters_h.empty() && !j->filters_folded) || j->never_match || j->ordered) return false;
    if (nrows <= 0 || nrows > 0x7fffffffLL) return false;
    const int64_t nb = j->bcols[j->ks.bidx[0]].rows;
    if (nb <= 0 || nb > (int64_t)TSQ_KR_MAXP * TSQ_KR_FILL || nb > 0xffffffffLL) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE) return true;
    return nrows >= (1 << 16) && nb >= (1 << 16);
}
tsq_status kr_emit_batch(tsq_join* j, const tsq_colset& pcs, ProbeArgs& a, int64_t nrows, const uint8_t* selected_dev) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    TSQ_HIP(h, hipEventRecord(j->ev[2], ctx->stream));
    bool ok = true;
    const bool outer = j->cfg.join_type != TSQ_JOIN_INNER;
    if (j->kr_digest) TSQ_TRY(kr_digests(j, pcs, j->ks.pidx, nrows, j->kr_pdig));
    TSQ_TRY(kr_pass(j, pcs, j->ks.pidx, nrows, j->kr_pbits, j->kr_counts, j->kr_pstart, j->kr_prec, false, &ok, &j->kr_pids, selected_dev, outer, j->kr_digest ? j->kr_pdig : nullptr));
    KrProbeArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.brec = j->kr_brec.as<unsigned long long>();
    pa.bstart = j->kr_bstart.as<uint32_t>();
    pa.prec = j->kr_prec.as<unsigned long long>();
    pa.pstart = j->kr_pstart.as<uint32_t>();
    pa.P = 1u << j->kr_pbits;
    pa.counters = j->counters.as<unsigned long long>();
    pa.flags = j->kr_flags.as<uint32_t>();
    pa.bids = j->kr_bids.as<uint32_t>();
    pa.pids = j->kr_pids.as<uint32_t>();
    pa.outer = outer ? 1 : 0;
    kr_verify_args(j, pcs, pa);
    if (j->kr_digest) {  // the emit launch reuses the sizing launch's byte comparisons
        TSQ_TRY(j->kr_vmask.reserve(ctx, h, (size_t)nrows * 4 + 64));
        pa.vmask = j->kr_vmask.as<uint32_t>();
        pa.vmode = 1;
    }
    // sizing launch: joined rows per partition; their exclusive scan = every partition's first output row
    TSQ_TRY(j->kr_pcnt.reserve(ctx, h, ((size_t)pa.P + 1) * 8 + 64));
    pa.part_cnt = j->kr_pcnt.as<unsigned long long>();
    TSQ_HIP(h, hipMemsetAsync(pa.part_cnt, 0, ((size_t)pa.P + 1) * 8, ctx->stream));  // (partitions without rows on one side are skipped by the kernel)
    const int grid = (int)std::min<uint32_t>(pa.P, (uint32_t)ctx->num_cus * 2);
    kr_launch_probe(ctx, grid, pa);  // pairs == nullptr
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_kr_scan64, dim3(1), dim3(1024), 0, ctx->stream, pa.part_cnt, pa.P);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 41, pa.part_cnt + pa.P, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (outer) TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 42, ctx->dscratch + 58, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    const int64_t part_rows = (int64_t)ctx->pinned[41];
    const int64_t keyless = outer ? (int64_t)ctx->pinned[42] : 0;  // outer rows without a key: one NULL-padded row each, after the partitions' rows
    const int64_t out_rows = part_rows + keyless;
    j->st.kernel_launches += 2;
    j->st.radix_batches++;
    j->st.radix_bits = (int32_t)j->kr_pbits;
    j->st.keyrec_digests = j->kr_digest ? 1 : 0;
    j->st.probe_route = TSQ_ROUTE_KEYREC;
    if (out_rows == 0) {
        TSQ_HIP(h, hipEventRecord(j->ev[3], ctx->stream));
        j->have_probe_ev = true;
        return TSQ_OK;
    }
    return materialise_pairs(j, pcs, a, nrows, out_rows, [&]() -> tsq_status {
        pa.pairs = a.pairs;
        if (pa.vmode == 1) pa.vmode = 2;
        kr_launch_probe(ctx, grid, pa);
        TSQ_HIP(h, hipGetLastError());
        j->st.kernel_launches++;
        if (keyless > 0) {
            hipLaunchKernelGGL(k_kr_miss_pairs, dim3(tsq_grid_for(ctx, keyless, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)j->kr_norec.p, keyless, a.pairs + part_rows);
            TSQ_HIP(h, hipGetLastError());
            j->st.kernel_launches++;
        }
        return TSQ_OK;
    });
}

bool wide_count_eligible(const tsq_join* j, int64_t nrows, const uint8_t* selected_dev) {
    if (j->radix_mode == TSQ_RADIX_OFF || j->packing_mode == TSQ_RADIX_OFF || !j->multi || !da_multi_ok(j) || j->wide_state < 0) return false;
    if (!j->count_only || j->checksum || j->general_cfg || selected_dev || j->never_match) return false;
    if (nrows <= 0 || nrows > 0x7fffffffLL) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE || j->packing_mode == TSQ_RADIX_FORCE) return true;
    return nrows >= (4 << 20) && j->bcols[j->ks.bidx[0]].rows >= (1 << 20);
}
tsq_status wide_prepare(tsq_join* j) {
    if (j->wide_state) return TSQ_OK;
    j->wide_state = -1;
    bool ok = false;
    TSQ_TRY(da_compose_build(j, &ok, 63));
    if (!ok) { j->da_ckey.release(); return TSQ_OK; }
    tsq_join_cfg cc;
    memset(&cc, 0, sizeof cc);
    cc.join_type = TSQ_JOIN_INNER;
    cc.build_is_right = 1;
    cc.n_keys = 1;
    cc.n_build_cols = cc.n_probe_cols = 1;
    cc.build_types[0] = TSQ_U64;
    cc.probe_types[0] = TSQ_I64;  // (mixed signedness on purpose: a composite of ~0 — "cannot match" — is dropped on both sides)
    cc.max_chunk_size = j->cfg.max_chunk_size;
    cc.concurrency = j->cfg.concurrency;
    cc.probe_batch_rows = j->cfg.probe_batch_rows;
    tsq_join* c = nullptr;
    tsq_status s = tsq_join_create(j->ctx, &cc, &c);
    if (s != TSQ_OK) return TSQ_OK;
    c->radix_mode = j->radix_mode;
    c->packing_mode = j->packing_mode;
    tsq_col bc;
    memset(&bc, 0, sizeof bc);
    bc.data = j->da_ckey.p;
    bc.length = j->bcols[j->ks.bidx[0]].rows;
    bc.elem_size = 8;
    bc.type = TSQ_U64;
    bc.flags = TSQ_COL_DEVICE;
    s = tsq_join_build_push(c, &bc, 1, bc.length);
    if (s == TSQ_OK) s = tsq_join_build_finish(c);
    if (s == TSQ_OK) s = tsq_join_set_count_only(c, 1);
    if (s != TSQ_OK) {
        tsq_fail(&j->hdr, s, c->hdr.err);
        tsq_join_destroy(c);
        return s == TSQ_ERR_UNSUPPORTED ? TSQ_OK : s;
    }
    j->wide = c;
    j->wide_state = 1;
    return TSQ_OK;
}
tsq_status wide_count_batch(tsq_join* j, const tsq_colset& pcs, int64_t nrows) {
    DaSrc src;
    const bool was = j->da_multi;
    j->da_multi = true;  // (da_probe_key composes the batch with j->da_fields)
    const tsq_status ks = da_probe_key(j, pcs, nrows, src);
    j->da_multi = was;
    TSQ_TRY(ks);
    tsq_colset cs;
    memset(&cs, 0, sizeof cs);
    cs.n = 1;
    cs.data[0] = src.data;
    cs.type[0] = TSQ_I64;
    tsq_join* c = j->wide;
    const tsq_status s = probe_batch(c, cs, nrows, nullptr);
    if (s != TSQ_OK) return tsq_fail(&j->hdr, s, c->hdr.err);
    j->st.probe_route = c->st.probe_route;
    j->st.packed_key_bits = c->st.packed_key_bits;
    j->st.radix_bits = c->st.radix_bits;
    j->st.radix_batches++;
    return TSQ_OK;
}

// hit ratio of a probe batch against the packed images, from a strided sample of its keys (k_da_sample): one small kernel + one sync
tsq_status da_sample_hit_ratio(tsq_join* j, const tsq_colset& pcs, int64_t nrows, const uint8_t* sel, double* rho) {
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    DaSampleArgs sa;
    memset(&sa, 0, sizeof sa);
    const int kc = j->ks.pidx[0];
    sa.src.data = (const uint64_t*)pcs.data[kc];
    sa.src.nulls = pcs.nulls[kc];
    sa.src.nrows = nrows;
    sa.src.sel = sel;
    sa.dm = j->da_dm;
    sa.img = j->da_img.as<uint8_t>();
    sa.n_samples = std::min<int64_t>(nrows, 1 << 16);
    sa.stride = std::max<int64_t>(1, nrows / sa.n_samples);
    sa.out = (unsigned long long*)(ctx->dscratch + 58);
    TSQ_HIP(h, hipMemsetAsync(ctx->dscratch + 58, 0, 16, ctx->stream));
    hipLaunchKernelGGL(k_da_sample, dim3((unsigned)((sa.n_samples + 255) / 256)), dim3(256), 0, ctx->stream, sa);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 58, ctx->dscratch + 58, 16, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    j->st.kernel_launches++;
    *rho = ctx->pinned[59] ? (double)ctx->pinned[58] / (double)ctx->pinned[59] : 0.0;
    return TSQ_OK;
}

// run the probe kernels over one device-resident batch described by pcs / selected
tsq_status probe_batch_routes(tsq_join* j, const tsq_colset& pcs, int64_t nrows, const uint8_t* selected_dev, const uint8_t* caller_selected);
tsq_status probe_batch(tsq_join* j, const tsq_colset& pcs, int64_t nrows, const uint8_t* selected_dev) {
    if (nrows == 0) return TSQ_OK;
    j->filters_folded = false;
    if (!j->filters_h.empty() && !j->shared && !j->count_only) {
        // outer-side filters: worth a pass over the batch when a materialising packed route would take the batch without them
        j->filters_folded = true;
        const bool packed = da_cols_eligible(j, nrows, selected_dev) || da_emit_eligible(j, nrows, selected_dev) || da_bits_emit_eligible(j, nrows) ||
                            kr_emit_eligible(j, nrows, selected_dev);
        j->filters_folded = false;
        if (packed) {
            bool folded = false;
            int64_t div0 = 0;
            TSQ_TRY(fold_outer_filters(j, pcs, nrows, selected_dev, &folded, &div0));
            if (folded) {
                j->filters_folded = true;
                const int64_t direct_before = j->direct_batches;
                const tsq_status s = probe_batch_routes(j, pcs, nrows, j->fflags.as<uint8_t>(), selected_dev);
                j->filters_folded = false;
                if (s == TSQ_OK && j->direct_batches == direct_before) j->div0_packed += div0;  // (the direct route counted its own)
                return s;
            }
        }
    }
    return probe_batch_routes(j, pcs, nrows, selected_dev, selected_dev);
}
// selected_dev: what the packed routes see (the caller's flags, or those with the outer-side filters folded in); caller_selected: what
// the direct route sees next to the filters themselves
tsq_status probe_batch_routes(tsq_join* j, const tsq_colset& pcs, int64_t nrows, const uint8_t* selected_dev, const uint8_t* caller_selected) {
    tsq_ctx* ctx = j->ctx;
    ProbeArgs a;
    memset(&a, 0, sizeof a);
    a.p = pcs;
    tsq_fill_colset(a.b, j->bcols);
    a.ks = j->ks;
    fill_table(j, a.t);
    a.nrows = nrows;
    a.selected = caller_selected;
    a.filters = j->filters_d.as<tsq_expr_prog>();
    a.n_filters = (int32_t)j->filters_h.size();
    a.conds = j->conds_d.as<tsq_expr_prog>();
    a.n_conds = (int32_t)j->conds_h.size();
    a.join_type = j->cfg.join_type;
    a.probe_is_left = j->cfg.build_is_right ? 1 : 0;
    a.counters = j->counters.as<unsigned long long>();
    j->st.probe_rows += nrows;

    if (j->shared) {  // the images of a build side sharded over several GPUs: there is no other table to fall back to
        if (!j->count_only || j->checksum || j->general_cfg || nrows > 0x7fffffffLL)
            return tsq_fail(&j->hdr, TSQ_ERR_UNSUPPORTED, "a shared build side (tsq_join_build_finish_shared) answers COUNT(*) of an inner join without conditions");
        return da_probe(j, pcs, nrows, selected_dev);
    }
    // the 64-bit table may have been left for the first batch that needs it (tsq_join_build_finish: table_can_wait)
    auto need_table = [&]() -> tsq_status {
        if (j->table_ready) return TSQ_OK;
        TSQ_TRY(build_table(j));
        fill_table(j, a.t);
        return TSQ_OK;
    };
    if (radix_eligible(j, nrows, selected_dev)) {
        TSQ_TRY(da_prepare(j));
        if (j->da_state == 1) return da_probe(j, pcs, nrows, selected_dev);
        if (!selected_dev) {  // (the 64-bit radix route knows no selected[])
            TSQ_TRY(need_table());
            if (radix_eligible(j, nrows, selected_dev)) return radix_probe(j, pcs, nrows);  // (the real table may be chained / too small)
        }
    }
    if (da_multi_count_eligible(j, nrows, selected_dev)) {  // several integer key columns: the packed route ...
        TSQ_TRY(da_prepare(j));
        if (j->da_state == 1) return da_probe(j, pcs, nrows, selected_dev);
    }
    if (wide_count_eligible(j, nrows, selected_dev)) {  // ... or, with fields of 29..63 bits, a single-key child join on the composite
        TSQ_TRY(wide_prepare(j));
        if (j->wide_state == 1) return wide_count_batch(j, pcs, nrows);
    }
    if (kr_count_eligible(j, nrows, selected_dev)) {  // ... or key records: any key columns (strings included) whose cells fit 32 bytes
        TSQ_TRY(kr_prepare(j));
        if (j->kr_state == 1) return kr_count_batch(j, pcs, nrows);
    }
    // ---- materialising packed routes.  Which one: when most probe rows join, the probe columns travel with the entries (K5f + K4e);
    // a SELECTIVE batch (few rows join: a sample of its keys against the images tells, k_da_sample) is better served by (probe row,
    // build row) pairs + a gather of the few joined rows (K4d) — moving every column of every row through the partition is waste.
    const bool forced = j->radix_mode == TSQ_RADIX_FORCE || j->packing_mode == TSQ_RADIX_FORCE;
    bool prefer_pairs = false;
    const bool pairs_knob_set = ctx->knob[TSQ_KNOB_DA_PAIRS_BELOW_PERMILLE] != TSQ_KNOB_DEFAULT;  // (tests force either variant with it)
    if ((!forced || pairs_knob_set) && !j->count_only && !j->da_multi && da_cols_eligible(j, nrows, selected_dev) && j->conds_h.empty()) {
        TSQ_TRY(da_prepare(j, nullptr, TSQ_OK, true));
        if (j->da_state == 1 && !j->da_bits) {
            double rho = 1.0;
            TSQ_TRY(da_sample_hit_ratio(j, pcs, nrows, selected_dev, &rho));
            prefer_pairs = rho * 1000.0 < (double)tsq_knob(ctx, TSQ_KNOB_DA_PAIRS_BELOW_PERMILLE, 350);
            j->last_sampled_hit_ratio = rho;
        }
    }
    if (prefer_pairs) {
        if (j->dm_l1_ready) {  // (the level-1 store da_prepare kept for the LDS variant: a selective join does not take it)
            j->dm_l1_ready = false;
            for (DevBuf* b : {&j->dm_l1ent, &j->dm_l1ctl, &j->dm_l1vend, &j->dm_l1nn}) b->release();
            for (auto& b : j->dm_l1pay) b.release();
        }
        TSQ_TRY(da_prepare_rows(j));
        if (j->da_rows_state == 1) return da_emit(j, pcs, a, nrows, selected_dev);
    }
    if (da_cols_eligible(j, nrows, selected_dev)) {
        TSQ_TRY(da_prepare(j, nullptr, TSQ_OK, true));
        // a build side without duplicate keys: its final partitions sit in LDS (tsq_damat.h) — unless this batch's keys are skewed
        bool lds_redo = false;
        if (dm_eligible(j)) {
            TSQ_TRY(dm_prepare_build(j));
            if (j->dm_state == 1) {
                bool fallback = false;
                TSQ_TRY(dm_emit_batch(j, pcs, nrows, &lds_redo, &fallback, selected_dev));
                if (!fallback && !lds_redo) return TSQ_OK;
            }
        }
        // conditions of an OUTER join over a build side with duplicate keys: "did ANY candidate of this outer row pass" is a segmented
        // reduction over the batch (the candidates of an outer row are consecutive output rows: k_outer_segments)
        if (!lds_redo) {  // (a condition that raised an error: the direct route reports it in the reference's order)
            TSQ_TRY(da_prepare_cols_direct(j));
            if (j->da_cols_state == 1) {
                bool redo = false;
                TSQ_TRY(da_emit_cols(j, pcs, nrows, &redo, selected_dev));
                if (!redo) return TSQ_OK;
            }
        }
    }
    if (!selected_dev && radix_emit_eligible(j, pcs, nrows, selected_dev)) {
        TSQ_TRY(need_table());
        if (radix_emit_eligible(j, pcs, nrows, selected_dev)) return radix_emit(j, pcs, nrows);
    }
    if (da_emit_eligible(j, nrows, selected_dev)) {
        TSQ_TRY(da_prepare(j));
        TSQ_TRY(da_prepare_rows(j));
        if (j->da_rows_state == 1) return da_emit(j, pcs, a, nrows, selected_dev);
    }
    if (da_bits_emit_eligible(j, nrows)) {  // a unique build side whose key range needs 4-byte entries: pairs through bit cells
        TSQ_TRY(da_prepare(j));
        TSQ_TRY(da_prepare_rows_bits(j));
        if (j->da_bitrows_state == 1) return da_emit_bits(j, pcs, a, nrows, selected_dev);
    }
    if (kr_emit_eligible(j, nrows, selected_dev)) {  // several key columns / string keys, materialising: pairs out of the key records
        TSQ_TRY(kr_prepare(j));
        if (j->kr_state == 1) return kr_emit_batch(j, pcs, a, nrows, selected_dev);
    }
    TSQ_TRY(need_table());
    TSQ_HIP(&j->hdr, hipEventRecord(j->ev[2], ctx->stream));
    j->st.probe_route = TSQ_ROUTE_DIRECT;
    j->direct_batches++;
    if (j->count_only) {
        TSQ_TRY(dispatch_count(j, a, j->checksum));
        TSQ_HIP(&j->hdr, hipEventRecord(j->ev[3], ctx->stream));
        j->have_probe_ev = true;
        return TSQ_OK;
    }
    // emit mode: size the batch first (K3), then materialise (K4); both walk contiguous rows per workgroup
    const int egrid = tsq_grid_for(ctx, nrows, 256);
    TSQ_TRY(j->bbase.reserve(ctx, &j->hdr, (size_t)egrid * 8 + 64));
    TSQ_TRY(j->firstcnt.reserve(ctx, &j->hdr, (size_t)nrows * 8 + 64));
    a.block_base = j->bbase.as<unsigned long long>();
    a.first_cnt = j->firstcnt.as<unsigned long long>();
    a.ordered = j->ordered ? 1 : 0;
    a.rows_per_block = (((nrows + egrid - 1) / egrid) + 63) & ~(int64_t)63;
    unsigned long long before[8], after[8];
    TSQ_TRY(read_counters(j, before));
    TSQ_TRY(dispatch_count(j, a, false));
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, ctx->stream, a.block_base, egrid);
    TSQ_HIP(&j->hdr, hipGetLastError());
    TSQ_TRY(read_counters(j, after));
    TSQ_TRY(status_from_errword(j, after[3]));
    const int64_t out_rows = (int64_t)(after[0] - before[0]);
    if (out_rows == 0) {
        TSQ_HIP(&j->hdr, hipEventRecord(j->ev[3], ctx->stream));
        j->have_probe_ev = true;
        return TSQ_OK;
    }
    return materialise_pairs(j, pcs, a, nrows, out_rows, [&]() -> tsq_status {
        TSQ_TRY(reset_counters(j, true));
        return dispatch_emit(j, a);
    });
}

// Host chunks, fixed-width columns: the H2D copies of the rows staged so far are queued while the caller is still pushing (every 256 Ki
// rows), so that a flush has only the last slice — not the whole batch, 64 MB for 4 Mi (k, v) rows — between it and its kernels.  The
// device columns of the batch are reserved for a whole batch first (a reserve that grows would move the rows already there); the copies
// are stream-ordered behind the previous batch's kernels, which read the same columns.
#define TSQ_STAGE_EARLY_ROWS (256 << 10)
tsq_status probe_stage_early(tsq_join* j) {
    HostStage& sg = j->stage;
    tsq_ctx* ctx = j->ctx;
    for (size_t c = 0; c < j->pcols.size(); c++)
        if (j->pcols[c].type == TSQ_BYTES) return TSQ_OK;
    for (size_t c = 0; c < j->pcols.size(); c++) {
        ColStore& cs = j->pcols[c];
        const int es = cs.elem();
        TSQ_TRY(cs.data.reserve(ctx, &j->hdr, (size_t)sg.cap * es + 64));
        TSQ_HIP(&j->hdr, hipMemcpyAsync((char*)cs.data.p + (size_t)j->stage_sent * es, (const char*)sg.data[c].p + (size_t)j->stage_sent * es,
                                         (size_t)(sg.staged - j->stage_sent) * es, hipMemcpyHostToDevice, ctx->stream));
    }
    j->stage_sent = sg.staged;
    return TSQ_OK;
}

tsq_status probe_flush(tsq_join* j) {
    HostStage& sg = j->stage;
    if (sg.staged == 0) return TSQ_OK;
    tsq_ctx* ctx = j->ctx;
    DevBuf tmp, tmp2;
    if (j->stage_sent > 0) {  // (fixed-width columns only: the rest of the rows, then the bitmaps of the whole batch)
        tsq_status s = probe_stage_early(j);
        for (size_t c = 0; c < j->pcols.size() && s == TSQ_OK; c++) {
            ColStore& cs = j->pcols[c];
            cs.clear();
            s = tsq_col_append_bitmap(ctx, &j->hdr, cs, sg.bitmap((int)c), sg.staged, false, tmp);
            cs.rows = sg.staged;
            j->st.h2d_bytes += sg.staged * cs.elem();
        }
        if (s != TSQ_OK) { tmp.release(); j->stage_sent = 0; return s; }
    } else
    for (size_t c = 0; c < j->pcols.size(); c++) {
        j->pcols[c].clear();
        tsq_status s = sg.append_to(ctx, &j->hdr, (int)c, j->pcols[c], tmp, tmp2);
        if (s != TSQ_OK) { tmp.release(); tmp2.release(); return s; }
        j->st.h2d_bytes += j->pcols[c].type == TSQ_BYTES ? sg.nbytes[c] + sg.staged * 8 : sg.staged * j->pcols[c].elem();
    }
    const uint8_t* sel_dev = nullptr;
    if (sg.sel_any) {
        tsq_status s = j->psel.reserve(ctx, &j->hdr, (size_t)sg.staged + 16);
        if (s != TSQ_OK) { tmp.release(); tmp2.release(); return s; }
        hipError_t e = hipMemcpyAsync(j->psel.p, sg.sel.p, (size_t)sg.staged, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { tmp.release(); tmp2.release(); return tsq_fail(&j->hdr, TSQ_ERR_HIP, std::string("hipMemcpyAsync(sel): ") + hipGetErrorString(e)); }
        sel_dev = j->psel.as<uint8_t>();
    }
    // staging (pinned) memory is reused by the next pushes: the call waits for the H2D copies — not for the kernels behind them, which run
    // beside the staging of the next batch (round 6; a materialising batch has read its counters back by then anyway)
    const bool overlap = (tsq_knob(ctx, TSQ_KNOB_HOST_OVERLAP, 3) & 1) != 0;
    if (overlap) {
        hipError_t ee = j->ev_h2d ? hipSuccess : hipEventCreateWithFlags(&j->ev_h2d, hipEventDisableTiming);
        if (ee == hipSuccess) ee = hipEventRecord(j->ev_h2d, ctx->stream);
        if (ee != hipSuccess) { tmp.release(); tmp2.release(); return tsq_fail(&j->hdr, TSQ_ERR_HIP, std::string("hipEventRecord(h2d): ") + hipGetErrorString(ee)); }
    }
    tsq_colset pcs;
    tsq_fill_colset(pcs, j->pcols);
    tsq_status s = probe_batch(j, pcs, sg.staged, sel_dev);
    hipError_t e = overlap ? hipEventSynchronize(j->ev_h2d) : hipStreamSynchronize(ctx->stream);
    tmp.release();
    tmp2.release();
    sg.reset();
    j->stage_sent = 0;
    if (s != TSQ_OK) return s;
    if (e != hipSuccess) return tsq_fail(&j->hdr, TSQ_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return TSQ_OK;
}

// ---------------------------------------------------------------- table geometry + partitioned build (host side)
// Sliced tables (tsq_jointable.h): load factor ~0.75 over 8-slot buckets, 2^tb slices of bs buckets with at most
// TSQ_BP_MAX_SLICE_ROWS rows each on average (a slice is one LDS image of the partitioned build; four of them are one
// image of the LDS probe).  Small or skew-rebuilt tables are one slice at load factor <= 0.5 (est_build_rows is only a
// hint in the reference too, hash_table.go:84-96).
void table_geometry(tsq_join* j, int64_t nb, bool sliced) {
    j->tb = 0;
    j->bs = (uint32_t)std::max<uint64_t>(16, (uint64_t)((nb + 3) / 4));
    if (sliced) {
        double lf = (double)tsq_knob(j->ctx, TSQ_KNOB_TABLE_LF_PERMILLE, 750) / 1000.0;
        if (!(lf >= 0.3 && lf <= 0.9)) lf = 0.75;
        const uint64_t nbk = std::max<uint64_t>(64, (uint64_t)ceil((double)nb / (8.0 * lf)));
        const uint64_t bs_max = std::min<uint64_t>(TSQ_BP_MAX_SLICE, (uint64_t)(TSQ_BP_MAX_SLICE_ROWS / (8.0 * lf)));
        uint32_t bits = 0;
        while (((nbk + (1ull << bits) - 1) >> bits) > bs_max) bits++;
        if (bits >= TSQ_RADIX_MIN_BITS && bits <= 19) {
            j->tb = bits;
            j->bs = (uint32_t)((nbk + (1ull << bits) - 1) >> bits);
        }
    }
    j->nbuckets = (uint64_t)j->bs << j->tb;
}

// Single key column, enough rows to pay for three passes, sliced geometry chosen by the caller (Q = 2^tb slices of m = bs buckets).
bool build_partitioned_eligible(const tsq_join* j, int64_t nb) {
    if (j->radix_mode == TSQ_RADIX_OFF || j->multi || j->never_match || j->tb < 2) return false;
    if (nb >= 0xffffffffLL) return false;
    if (j->radix_mode == TSQ_RADIX_FORCE) return nb >= (1 << 16);
    return nb >= (4 << 20);
}

tsq_status build_partitioned(tsq_join* j, int64_t nb, uint32_t sent_cap, bool* done) {
    *done = false;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const uint32_t bits = j->tb;
    uint32_t b1, b2;
    if (bits >= 16) { b2 = 8; b1 = bits - 8; }
    else if (bits >= 9) { b1 = 8; b2 = bits - 8; }
    else { b1 = bits - 1; b2 = 1; }
    const uint32_t Q = 1u << bits, P1 = 1u << b1;
    const uint32_t m = j->bs;
    const uint64_t nbuckets = j->nbuckets;
    // pass-1 store (8 XCC regions per partition), as in radix_probe
    RadixStore st;
    memset(&st, 0, sizeof st);
    st.bits = b1;
    st.R = 8;
    constexpr int NT = 1024, K1 = 8, T1 = NT * K1;
    const double lam = (double)nb / ((double)P1 * 8.0);
    st.cap = (uint32_t)(lam * 1.08 + 8.0 * sqrt(lam) + 2.0 * T1 / 64.0 + 64.0);
    st.cap = (st.cap + 15u) & ~15u;
    const size_t nregions = (size_t)P1 * 8, slots1 = nregions * st.cap;
    const double lam2 = (double)nb / (double)Q;
    uint32_t cap2 = (uint32_t)(lam2 * 1.08 + 8.0 * sqrt(lam2) + 64.0);
    cap2 = (cap2 + 15u) & ~15u;
    const size_t slots2 = (size_t)Q * cap2;
    if (slots1 >= 0xffffffffULL || slots2 >= 0xffffffffULL || cap2 > 4096) return TSQ_OK;  // k_build_images<512, 8> holds a slice's rows in registers
    DevBuf k1, i1, ctl, vend, ok1, oi1, k2, i2, cnt2, orows;
    auto release_all = [&]() {
        for (DevBuf* b : {&k1, &i1, &ctl, &vend, &ok1, &oi1, &k2, &i2, &cnt2, &orows}) b->release();
    };
    tsq_status s = TSQ_OK;
    const size_t ctl_bytes = nregions * 4 + 64;  // cursors | [nregions] pass-1 overflow count | [nregions+1] row-list count
    if (s == TSQ_OK) s = k1.reserve(ctx, h, slots1 * 8 + 256);
    if (s == TSQ_OK) s = i1.reserve(ctx, h, slots1 * 4 + 256);
    if (s == TSQ_OK) s = ctl.reserve(ctx, h, ctl_bytes);
    if (s == TSQ_OK) s = vend.reserve(ctx, h, nregions * 4);
    if (s == TSQ_OK) s = ok1.reserve(ctx, h, (size_t)nb * 8 + 64);
    if (s == TSQ_OK) s = oi1.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s == TSQ_OK) s = k2.reserve(ctx, h, slots2 * 8 + 256);
    if (s == TSQ_OK) s = i2.reserve(ctx, h, slots2 * 4 + 256);
    if (s == TSQ_OK) s = cnt2.reserve(ctx, h, (size_t)Q * 4);
    if (s == TSQ_OK) s = orows.reserve(ctx, h, (size_t)nb * 4 + 64);
    if (s == TSQ_OK) s = j->tkeys.reserve(ctx, h, nbuckets * TSQ_BUCKET * 8);
    if (s == TSQ_OK) s = j->tvals.reserve(ctx, h, nbuckets * TSQ_BUCKET * 4);
    if (s != TSQ_OK) { release_all(); return s; }
    st.keys = k1.as<uint64_t>();
    st.idx = i1.as<uint32_t>();
    st.cursor = ctl.as<uint32_t>();
    st.ovf_count = st.cursor + nregions;
    st.valid_end = vend.as<uint32_t>();
    st.ovf_keys = ok1.as<uint64_t>();
    st.ovf_idx = oi1.as<uint32_t>();
    st.ovf_cap = (uint32_t)nb;
    SubStore sub;
    memset(&sub, 0, sizeof sub);
    sub.keys = k2.as<uint64_t>();
    sub.idx = i2.as<uint32_t>();
    sub.count = cnt2.as<uint32_t>();
    sub.ovf_rows = orows.as<uint32_t>();
    sub.ovf_count = st.cursor + nregions + 1;
    sub.ovf_cap = (uint32_t)nb;
    sub.b1 = b1;
    sub.b2 = b2;
    sub.cap2 = cap2;
    hipError_t e = hipEventRecord(j->ev[0], ctx->stream);  // after the allocations: build_kernel_ms is kernel time
    if (e == hipSuccess) e = hipMemsetAsync(ctl.p, 0, ctl_bytes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(vend.p, 0xff, nregions * 4, ctx->stream);
    RadixSrc src;
    memset(&src, 0, sizeof src);
    const int kc = j->ks.bidx[0];
    src.data = j->bcols[kc].data.p;
    src.nulls = j->bcols[kc].has_nulls ? j->bcols[kc].nulls.as<uint8_t>() : nullptr;
    src.type = j->cfg.build_types[kc];
    src.skip_high = j->ks.skip_high;
    src.nrows = nb;
    ImageArgs ia;
    memset(&ia, 0, sizeof ia);
    ia.in = sub;
    fill_table(j, ia.t);
    ia.m = m;
    ia.sent_rows = j->sent.as<uint32_t>();
    ia.sent_cap = sent_cap;
    ia.sent_total = (uint32_t*)(ctx->dscratch + 1);
    ia.inserted = (unsigned long long*)ctx->dscratch;
    ia.fail = (uint32_t*)(ctx->dscratch + 2);
    const bool cas_images = tsq_knob(ctx, TSQ_KNOB_BUILD_IMAGES_CAS, 0) != 0;
    const size_t img_bytes = (size_t)m * TSQ_BUCKET * 12 + (cas_images ? 0 : (size_t)m * 4);
    if (e == hipSuccess && img_bytes > 48 * 1024)
        e = hipFuncSetAttribute(cas_images ? (const void*)k_build_images<512, 8> : (const void*)k_build_images_cnt<512, 8>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)img_bytes);
    if (e == hipSuccess) {
        const int64_t ntiles = (nb + T1 - 1) / T1;
        hipLaunchKernelGGL((k_radix_partition<NT, K1, 4, 0, true, true>), dim3((unsigned)std::min<int64_t>(ntiles, ctx->num_cus)), dim3(NT), 0, ctx->stream, src, st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL((k_radix_subpartition<1024, 8>), dim3(P1), dim3(1024), 0, ctx->stream, st, sub);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        const dim3 igrid(std::min<uint32_t>(Q, (uint32_t)ctx->num_cus * 2));
        if (cas_images) hipLaunchKernelGGL((k_build_images<512, 8>), igrid, dim3(512), img_bytes, ctx->stream, ia);
        else hipLaunchKernelGGL((k_build_images_cnt<512, 8>), igrid, dim3(512), img_bytes, ctx->stream, ia);
        e = hipGetLastError();
    }
    j->st.kernel_launches += 3;
    // rows handed back: pass-1 overflow (st.ovf_idx) and the row list of passes 2 and 3
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 16, st.ovf_count, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { release_all(); return tsq_fail(h, TSQ_ERR_HIP, std::string("partitioned build: ") + hipGetErrorString(e)); }
    const uint32_t n_ovf1 = ((const uint32_t*)(ctx->pinned + 16))[0], n_rows = ((const uint32_t*)(ctx->pinned + 16))[1];
    BuildArgs a;
    memset(&a, 0, sizeof a);
    tsq_fill_colset(a.b, j->bcols);
    a.ks = j->ks;
    fill_table(j, a.t);
    a.sent_rows = j->sent.as<uint32_t>();
    a.sent_cap = sent_cap;
    a.sent_total = (uint32_t*)(ctx->dscratch + 1);
    a.inserted = (unsigned long long*)ctx->dscratch;
    a.fail = (uint32_t*)(ctx->dscratch + 2);
    for (int pass = 0; pass < 2; pass++) {
        a.nrows = pass == 0 ? n_ovf1 : n_rows;
        a.row_list = pass == 0 ? st.ovf_idx : sub.ovf_rows;
        if (a.nrows == 0) continue;
        s = launch_build<false>(j, a);
        if (s != TSQ_OK) { release_all(); return s; }
    }
    j->build_handed_back = (int64_t)n_ovf1 + n_rows;
    e = hipStreamSynchronize(ctx->stream);  // the scratch buffers go back to the pool
    release_all();
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("partitioned build: ") + hipGetErrorString(e));
    *done = true;
    return TSQ_OK;
}

}  // namespace

// ====================================================================== C-ABI
TSQ_API tsq_status tsq_join_create(tsq_ctx* ctx, const tsq_join_cfg* cfg, tsq_join** out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !cfg || !out) return tsq_fail(nullptr, TSQ_ERR_INVALID, "tsq_join_create: NULL argument");
    *out = nullptr;
    tsq_handle_hdr* ch = &ctx->hdr;
    if (cfg->join_type < TSQ_JOIN_INNER || cfg->join_type > TSQ_JOIN_RIGHT_OUTER)
        return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "only inner/left outer/right outer joins exist (joiner.go:105-116)");
    // outer joins: the probe side is the outer side (join.go:31-60): left outer => build is right
    if (cfg->join_type == TSQ_JOIN_LEFT_OUTER && !cfg->build_is_right)
        return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "left outer join must build on the right child");
    if (cfg->join_type == TSQ_JOIN_RIGHT_OUTER && cfg->build_is_right)
        return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "right outer join must build on the left child");
    if (cfg->n_keys < 1 || cfg->n_keys > TSQ_MAX_KEYS) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "1..4 join key columns supported");
    if (cfg->n_build_cols < 1 || cfg->n_build_cols > TSQ_MAX_COLS || cfg->n_probe_cols < 1 || cfg->n_probe_cols > TSQ_MAX_COLS)
        return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "1..16 columns per side supported");
    for (int c = 0; c < cfg->n_build_cols; c++)
        if (cfg->build_types[c] < TSQ_I64 || cfg->build_types[c] > TSQ_BYTES) return tsq_fail(ch, TSQ_ERR_INVALID, "unknown build column type");
    for (int c = 0; c < cfg->n_probe_cols; c++)
        if (cfg->probe_types[c] < TSQ_I64 || cfg->probe_types[c] > TSQ_BYTES) return tsq_fail(ch, TSQ_ERR_INVALID, "unknown probe column type");
    for (int k = 0; k < cfg->n_keys; k++) {
        if (cfg->build_key_idx[k] < 0 || cfg->build_key_idx[k] >= cfg->n_build_cols || cfg->probe_key_idx[k] < 0 ||
            cfg->probe_key_idx[k] >= cfg->n_probe_cols)
            return tsq_fail(ch, TSQ_ERR_INVALID, "join key index out of range");
    }
    if ((cfg->n_other_conds > 0 && !cfg->other_conds) || (cfg->n_outer_filters > 0 && !cfg->outer_filters) ||
        cfg->n_other_conds < 0 || cfg->n_outer_filters < 0 || cfg->n_other_conds > 16 || cfg->n_outer_filters > 16)
        return tsq_fail(ch, TSQ_ERR_INVALID, "bad condition/filter list");
    const int nleft = cfg->build_is_right ? cfg->n_probe_cols : cfg->n_build_cols;
    (void)nleft;
    for (int e = 0; e < cfg->n_other_conds; e++) {
        const char* why = "";
        int32_t jt[2 * TSQ_MAX_COLS];  // the joined row lhs || rhs (joiner.go:145-150)
        for (int c = 0; c < cfg->n_probe_cols + cfg->n_build_cols; c++) {
            const bool from_probe = cfg->build_is_right ? c < cfg->n_probe_cols : c >= cfg->n_build_cols;
            jt[c] = from_probe ? cfg->probe_types[cfg->build_is_right ? c : c - cfg->n_build_cols] : cfg->build_types[cfg->build_is_right ? c - cfg->n_probe_cols : c];
        }
        tsq_status s = tsq_validate_prog(cfg->other_conds[e], cfg->n_probe_cols + cfg->n_build_cols, &why, jt);
        if (s != TSQ_OK) return tsq_fail(ch, s, std::string("other condition: ") + why);
        if (cfg->other_conds[e].result_type == TSQ_BYTES) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "other condition: a string-valued condition keeps the Go evaluator");
    }
    for (int e = 0; e < cfg->n_outer_filters; e++) {
        const char* why = "";
        tsq_status s = tsq_validate_prog(cfg->outer_filters[e], cfg->n_probe_cols, &why, cfg->probe_types);
        if (s != TSQ_OK) return tsq_fail(ch, s, std::string("outer filter: ") + why);
        if (cfg->outer_filters[e].result_type == TSQ_BYTES) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "outer filter: a string-valued filter keeps the Go evaluator");
    }
    TSQ_HIP(ch, hipSetDevice(ctx->device));
    std::unique_ptr<tsq_join> j(new tsq_join());
    j->hdr.magic = TSQ_MAGIC_JOIN;
    j->ctx = ctx;
    j->cfg = *cfg;
    j->conds_h.assign(cfg->other_conds, cfg->other_conds + cfg->n_other_conds);
    j->filters_h.assign(cfg->outer_filters, cfg->outer_filters + cfg->n_outer_filters);
    j->cfg.other_conds = nullptr;
    j->cfg.outer_filters = nullptr;
    if (j->cfg.max_chunk_size <= 0) j->cfg.max_chunk_size = 1024;
    if (j->cfg.probe_batch_rows <= 0) j->cfg.probe_batch_rows = 4 << 20;
    j->cfg.probe_batch_rows = (j->cfg.probe_batch_rows + 63) & ~(int64_t)63;

    j->bcols.resize(cfg->n_build_cols);
    for (int c = 0; c < cfg->n_build_cols; c++) j->bcols[c].type = cfg->build_types[c];
    j->pcols.resize(cfg->n_probe_cols);
    for (int c = 0; c < cfg->n_probe_cols; c++) j->pcols[c].type = cfg->probe_types[c];

    // key plan
    KeySpec& ks = j->ks;
    ks.n_keys = cfg->n_keys;
    j->multi = cfg->n_keys > 1;
    for (int k = 0; k < cfg->n_keys; k++) {
        ks.bidx[k] = cfg->build_key_idx[k];
        ks.pidx[k] = cfg->probe_key_idx[k];
        const int32_t bt = cfg->build_types[ks.bidx[k]], pt = cfg->probe_types[ks.pidx[k]];
        // a string key cell is (compactBytesFlag, bytes) (codec.go:233-235): the table stores a hash of the bytes and every match is
        // verified against the build row's bytes — the multi-column route, also for a single string key
        if (bt == TSQ_BYTES || pt == TSQ_BYTES) j->multi = true;
        if ((bt == TSQ_BYTES) != (pt == TSQ_BYTES)) j->never_match = true;  // flag 2 vs 5 / 8 / 9
        else if (bt == TSQ_BYTES) continue;
        if (is_int_class(bt) != is_int_class(pt)) j->never_match = true;  // flag 8/9 vs 5 (codec.go:217-235)
        if (!j->multi && is_int_class(bt) && bt != pt) ks.skip_high = 1;  // flag 8 vs 9 for cells >= 2^63
    }
    j->general = cfg->join_type != TSQ_JOIN_INNER || cfg->n_other_conds > 0 || cfg->n_outer_filters > 0;
    j->general_cfg = j->general;

    tsq_handle_hdr* h = &j->hdr;
    TSQ_TRY(j->counters.reserve(ctx, h, 64));
    if (!j->conds_h.empty()) {
        TSQ_TRY(j->conds_d.reserve(ctx, h, j->conds_h.size() * sizeof(tsq_expr_prog)));
        TSQ_HIP(h, hipMemcpy(j->conds_d.p, j->conds_h.data(), j->conds_h.size() * sizeof(tsq_expr_prog), hipMemcpyHostToDevice));
    }
    if (!j->filters_h.empty()) {
        TSQ_TRY(j->filters_d.reserve(ctx, h, j->filters_h.size() * sizeof(tsq_expr_prog)));
        TSQ_HIP(h, hipMemcpy(j->filters_d.p, j->filters_h.data(), j->filters_h.size() * sizeof(tsq_expr_prog), hipMemcpyHostToDevice));
    }
    for (int i = 0; i < 4; i++) TSQ_HIP(h, hipEventCreate(&j->ev[i]));
    {
        tsq_status s = reset_counters(j.get(), false);
        if (s != TSQ_OK) { tsq_fail(ch, s, j->hdr.err); return s; }
    }
    *out = j.release();
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_build_push(tsq_join* j, const tsq_col* cols, int32_t n_cols, int64_t nrows) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    TSQ_TRY(check_cancel(j));
    if (j->build_done) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "build_push after build_finish");
    if (nrows < 0 || (!cols && nrows > 0)) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "bad arguments");
    if (nrows == 0) return TSQ_OK;
    bool dev = false;
    TSQ_TRY(tsq_validate_cols(&j->hdr, cols, n_cols, j->cfg.n_build_cols, j->cfg.build_types, nrows, &dev));
    TSQ_HIP(&j->hdr, hipSetDevice(j->ctx->device));
    if (j->bcols[0].rows + j->stage.staged + nrows >= 0xfffffff0LL)
        return tsq_fail(&j->hdr, TSQ_ERR_UNSUPPORTED, "build side exceeds 2^32 rows per GPU: partition across GPUs first");
    if (dev) {
        TSQ_TRY(build_flush(j));
        // the first chunk of a build side handed over with TSQ_COL_RETAIN: kept where it is (hash_table.go:146-169: PutChunk keeps the chunk)
        bool retain = j->bcols[0].rows == 0;
        for (int c = 0; c < n_cols && retain; c++) {
            const size_t need = (size_t)nrows * (size_t)tsq_elem_size(cols[c].type);
            retain = (cols[c].flags & TSQ_COL_RETAIN) && cols[c].type != TSQ_BYTES && j->bcols[c].rows == 0 && tsq_user_alloc_bytes(j->ctx, cols[c].data) >= need + 64 &&
                     (!cols[c].null_bitmap || tsq_user_alloc_bytes(j->ctx, cols[c].null_bitmap) >= (size_t)tsq_bitmap_bytes(nrows) + 8);
        }
        if (retain) {
            for (int c = 0; c < n_cols; c++) {
                ColStore& cs = j->bcols[c];
                cs.data.adopt(cols[c].data, (size_t)nrows * (size_t)cs.elem() + 64);
                if (cols[c].null_bitmap) {
                    cs.nulls.adopt(cols[c].null_bitmap, (size_t)tsq_bitmap_bytes(nrows) + 8);
                    cs.has_nulls = true;
                }
                cs.rows = nrows;
            }
            return TSQ_OK;
        }
        DevBuf tmp, tmp2;
        for (int c = 0; c < n_cols; c++) {
            tsq_status s = cols[c].type == TSQ_BYTES
                               ? tsq_col_append_varlen(j->ctx, &j->hdr, j->bcols[c], cols[c].data, cols[c].offsets, cols[c].null_bitmap, nrows, true, tmp, tmp2)
                               : tsq_col_append(j->ctx, &j->hdr, j->bcols[c], cols[c].data, cols[c].null_bitmap, nrows, true, tmp);
            if (s != TSQ_OK) { tmp.release(); tmp2.release(); return s; }
        }
        tmp.release();
        tmp2.release();
        return TSQ_OK;
    }
    if (j->stage.cap == 0) TSQ_TRY(j->stage.init(&j->hdr, n_cols, j->cfg.build_types, 1 << 20));
    int64_t off = 0;
    while (off < nrows) {
        int64_t n = std::min<int64_t>(nrows - off, j->stage.room());
        j->stage.add(cols, off, n, nullptr);
        off += n;
        if (j->stage.room() == 0) TSQ_TRY(build_flush(j));
    }
    return TSQ_OK;
}

// the 64-bit hash table of the build side (tsq_jointable.h).  Built by tsq_join_build_finish — or, when the build side looks packable
// (below), by the first probe batch that needs it: the packed routes never read it, and 2.6 ms per 1e8 build rows
// is more than the packed images cost.
static tsq_status build_table(tsq_join* j) {
    if (j->table_ready) return TSQ_OK;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    const int64_t nb = j->bcols[0].rows;
    uint32_t sent_cap = 1024;
    TSQ_TRY(j->sent.reserve(ctx, h, sent_cap * 4));
    j->sent_count = 0;
    j->build_inserted = 0;
    bool sliced = nb >= 32768 && !j->never_match && j->radix_mode != TSQ_RADIX_OFF;
    bool part_done = false;
    for (int attempt = 0;; attempt++) {
        table_geometry(j, nb, sliced);
        const uint64_t nbuckets = j->nbuckets;
        if (nbuckets >> 32) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "build side too large for one table: partition across GPUs first");
        // dscratch[0] = inserted (u64), dscratch[1] low = sent_total (u32), dscratch[2] low = fail flag (u32)
        TSQ_HIP(h, hipMemsetAsync(ctx->dscratch, 0, 24, ctx->stream));
        part_done = false;
        if (nb > 0 && build_partitioned_eligible(j, nb)) {
            TSQ_TRY(build_partitioned(j, nb, sent_cap, &part_done));  // records ev[0]
            if (part_done) {
                TSQ_HIP(h, hipEventRecord(j->ev[1], ctx->stream));
                j->have_build_ev = true;
            }
        }
        if (!part_done) {
            TSQ_TRY(j->tkeys.reserve(ctx, h, nbuckets * TSQ_BUCKET * 8));
            TSQ_TRY(j->tvals.reserve(ctx, h, nbuckets * TSQ_BUCKET * 4));
            TSQ_HIP(h, hipMemsetAsync(j->tkeys.p, 0x80, nbuckets * TSQ_BUCKET * 8, ctx->stream));
            if (j->chained) {
                TSQ_TRY(j->tnext.reserve(ctx, h, (size_t)nb * 4 + 64));
                TSQ_HIP(h, hipMemsetAsync(j->tvals.p, 0xff, nbuckets * TSQ_BUCKET * 4, ctx->stream));
            }
        }
        if (nb == 0 || j->never_match) {
            TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
            break;
        }
        BuildArgs a;
        memset(&a, 0, sizeof a);
        tsq_fill_colset(a.b, j->bcols);
        a.ks = j->ks;
        fill_table(j, a.t);
        a.row0 = 0;
        a.nrows = nb;
        a.sent_rows = j->sent.as<uint32_t>();
        a.sent_cap = sent_cap;
        a.sent_total = (uint32_t*)(ctx->dscratch + 1);
        a.inserted = (unsigned long long*)ctx->dscratch;
        a.fail = (uint32_t*)(ctx->dscratch + 2);
        if (!part_done) {
            TSQ_HIP(h, hipEventRecord(j->ev[0], ctx->stream));
            if (j->chained) {
                ChainArgs ca;
                ca.b = a;
                ca.next = j->tnext.as<uint32_t>();
                const int grid = tsq_grid_for(ctx, a.nrows, 256);
                if (j->multi) hipLaunchKernelGGL(k_build_insert_chained<true>, dim3(grid), dim3(256), 0, ctx->stream, ca);
                else hipLaunchKernelGGL(k_build_insert_chained<false>, dim3(grid), dim3(256), 0, ctx->stream, ca);
                TSQ_HIP(h, hipGetLastError());
                j->st.kernel_launches++;
            } else if (j->multi) TSQ_TRY(launch_build<true>(j, a));
            else TSQ_TRY(launch_build<false>(j, a));
            TSQ_HIP(h, hipEventRecord(j->ev[1], ctx->stream));
            j->have_build_ev = true;
        }
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned, ctx->dscratch, 24, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        j->unique = !j->chained && (ctx->pinned[2] >> 32) == 0;
        if ((uint32_t)ctx->pinned[2]) {
            // A slice could not take all of its rows (skewed / heavily duplicated keys): once more as one slice at load
            // factor 0.5.  If even that walks more than TSQ_MAX_WALK buckets for one row, a key has tens of thousands of
            // duplicates and every insert would scan its whole run: the Go operator (O(1) per row, hash_table.go:247-256)
            // is the better executor for that input.
            if (sliced && attempt == 0) {
                sliced = false;
                j->st.build_slice_retries++;
                continue;
            }
            if (!j->chained) {  // a key with tens of thousands of build rows: one slot per distinct word + row chains (rowHashMap's layout)
                j->chained = true;
                sliced = false;
                j->st.build_slice_retries++;
                continue;
            }
            return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "the build side does not fit one open-addressing table: partition across GPUs first");
        }
        j->build_inserted = (int64_t)ctx->pinned[0];
        uint32_t sent_total = (uint32_t)ctx->pinned[1];
        if (sent_total > sent_cap) {  // rare: many rows carry the sentinel table word — collect them all
            TSQ_TRY(j->sent.reserve(ctx, h, (size_t)sent_total * 4));
            TSQ_HIP(h, hipMemsetAsync(ctx->dscratch + 1, 0, 8, ctx->stream));
            a.sent_rows = j->sent.as<uint32_t>();
            a.sent_cap = sent_total;
            int grid = tsq_grid_for(ctx, nb, 256);
            if (j->multi) hipLaunchKernelGGL(k_collect_sentinel<true>, dim3(grid), dim3(256), 0, ctx->stream, a);
            else hipLaunchKernelGGL(k_collect_sentinel<false>, dim3(grid), dim3(256), 0, ctx->stream, a);
            TSQ_HIP(h, hipGetLastError());
            TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        }
        j->sent_count = sent_total;
        break;
    }
    j->st.table_bytes = (int64_t)(j->nbuckets * TSQ_BUCKET * 12);
    j->st.table_buckets = (int64_t)j->nbuckets;
    j->st.table_slice_bits = (int32_t)j->tb;
    j->st.build_partitioned = part_done ? 1 : 0;
    j->st.build_rows_inserted = j->build_inserted;
    j->table_ready = true;
    return TSQ_OK;
}

// May the 64-bit table wait for a probe batch that needs it?  Yes when the packed routes are likely to serve every batch: one integer
// key column (or several that compose), nothing switched off, and a build side of the size AUTO packs (or packing FORCEd).  The
// decision only moves WORK: a batch that takes another route builds the table first (probe_batch: need_table).
static bool table_can_wait(const tsq_join* j, int64_t nb) {
    if (j->never_match || j->radix_mode == TSQ_RADIX_OFF || nb <= 0 || nb >= 0xffffffffLL) return false;
    if (tsq_knob(j->ctx, TSQ_KNOB_LAZY_TABLE, 1) == 0 || !j->filters_h.empty() || j->ordered) return false;
    const bool packable = j->multi ? da_multi_ok(j) : (is_int_class(j->cfg.build_types[j->ks.bidx[0]]) && is_int_class(j->cfg.probe_types[j->ks.pidx[0]]));
    if (packable && j->packing_mode != TSQ_RADIX_OFF && tsq_knob(j->ctx, TSQ_KNOB_PACKED_KEYS, 1) != 0)
        return j->packing_mode == TSQ_RADIX_FORCE || nb >= tsq_knob(j->ctx, TSQ_KNOB_DA_MIN_BUILD_ROWS, (int64_t)(1 << 20));
    // string keys / key columns that do not compose: the key-record route serves the batches it takes (kr_count_eligible, kr_emit_eligible)
    // without the table — inserting 1e5 build rows keyed by a 5 KiB string hashed every byte of them for nothing (round 6)
    if (j->multi && !packable && tsq_knob(j->ctx, TSQ_KNOB_KEYREC, 1) != 0 && j->conds_h.empty() && nb <= (int64_t)TSQ_KR_MAXP * TSQ_KR_FILL)
        return j->radix_mode == TSQ_RADIX_FORCE || nb >= (1 << 16);
    return false;
}

TSQ_API tsq_status tsq_join_build_finish(tsq_join* j) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    TSQ_TRY(check_cancel(j));
    if (j->build_done) return TSQ_OK;
    tsq_ctx* ctx = j->ctx;
    tsq_handle_hdr* h = &j->hdr;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    TSQ_TRY(build_flush(j));
    const int64_t nb = j->bcols[0].rows;
    j->st.build_rows = nb;
    if (table_can_wait(j, nb)) {
        // the geometry is host arithmetic: the route choices that ask for it (radix_eligible: table bytes, slices) get the planned one
        table_geometry(j, nb, nb >= 32768);
    } else {
        TSQ_TRY(build_table(j));
    }
    j->build_done = true;
    j->stage.release();  // staging is re-initialised for the probe schema
    return TSQ_OK;
}

// The build side of a COUNT(*) join SHARDED over the ranks of a communicator, without moving a probe row: every rank pushes ITS
// build rows, then all ranks call this instead of tsq_join_build_finish.  See da_prepare(j, comm) and DESIGN.md §6.
TSQ_API tsq_status tsq_join_build_finish_shared(tsq_join* j, tsq_comm* c, int32_t* shared_out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    const tsq_status cancelled = check_cancel(j);  // (rank-local: carried into the collective below)
    tsq_handle_hdr* h = &j->hdr;
    if (!shared_out) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_join_build_finish_shared: NULL argument");
    *shared_out = 0;
    if (!tsq_comm_usable(c, j->ctx)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_join_build_finish_shared: the communicator does not belong to the join's context");
    if (j->build_done) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_join_build_finish_shared after build_finish");
    // rank-local failures from here on are carried INTO the collective (da_prepare's pre_status) instead of returned before it
    tsq_status pre = cancelled;
    if (pre == TSQ_OK && hipSetDevice(j->ctx->device) != hipSuccess) pre = tsq_fail(h, TSQ_ERR_HIP, "tsq_join_build_finish_shared: hipSetDevice failed");
    if (pre == TSQ_OK) pre = build_flush(j);
    // what the images can answer (the same on every rank: it is the plan): COUNT(*) of an inner equi-join on one integer column
    if (j->general || j->multi || j->never_match || j->ordered || j->radix_mode == TSQ_RADIX_OFF) return pre;
    const bool was_count_only = j->count_only;
    j->count_only = true;
    j->da_state = 0;
    const tsq_status s = da_prepare(j, c, pre);
    if (s != TSQ_OK || j->da_state != 1) {  // not packable (or this rank failed): the handle still holds its rows, nothing else
        j->count_only = was_count_only;
        j->da_state = 0;
        return s != TSQ_OK ? s : pre;
    }
    j->shared = true;
    j->st.build_rows = j->bcols[0].rows;
    j->st.build_rows_inserted = j->shared_usable_local;
    j->build_done = true;
    j->stage.release();
    *shared_out = 1;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_set_count_only(tsq_join* j, int32_t on) {
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    if (j->st.probe_rows > 0 || j->stage.staged > 0) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "count-only must be chosen before the first probe row");
    j->count_only = on != 0;
    if (!j->count_only) j->checksum = false;
    return TSQ_OK;
}
// Inline projection of the join's output (the planner's column pruning: a parent that reads 5 of 10 joined columns): the routes that
// gather the output through (probe row, build row) pairs skip the unused columns entirely; tsq_join_pull leaves them untouched.
TSQ_API tsq_status tsq_join_set_used_columns(tsq_join* j, const uint8_t* used, int32_t n_cols) {
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    if (j->st.probe_rows > 0 || j->stage.staged > 0) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "the used columns must be chosen before the first probe row");
    if (!used) { j->used_out.clear(); return TSQ_OK; }
    if (n_cols != j->cfg.n_probe_cols + j->cfg.n_build_cols) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "tsq_join_set_used_columns: one flag per output column (left child's, then right child's)");
    j->used_out.assign(used, used + n_cols);
    return TSQ_OK;
}
TSQ_API tsq_status tsq_join_set_ordered(tsq_join* j, int32_t on) {
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    if (j->st.probe_rows > 0 || j->stage.staged > 0) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "ordered output must be chosen before the first probe row");
    j->ordered = on != 0;
    return TSQ_OK;
}
TSQ_API tsq_status tsq_join_set_radix(tsq_join* j, int32_t mode) {
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    if (mode < TSQ_RADIX_AUTO || mode > TSQ_RADIX_FORCE) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "radix mode must be -1 (auto), 0 (off) or 1 (force)");
    j->radix_mode = mode;
    return TSQ_OK;
}
TSQ_API tsq_status tsq_join_set_key_packing(tsq_join* j, int32_t mode) {
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    if (mode < TSQ_RADIX_AUTO || mode > TSQ_RADIX_FORCE) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "key packing mode must be -1 (auto), 0 (off) or 1 (force)");
    if (j->da_state != 0) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "key packing must be chosen before the first probe batch");
    j->packing_mode = mode;
    return TSQ_OK;
}
TSQ_API tsq_status tsq_join_set_checksum(tsq_join* j, int32_t on) {
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    if (!j->count_only) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "checksum needs count-only mode");
    for (int c = 0; on && c < j->cfg.n_build_cols + j->cfg.n_probe_cols; c++)
        if ((c < j->cfg.n_build_cols ? j->cfg.build_types[c] : j->cfg.probe_types[c - j->cfg.n_build_cols]) == TSQ_BYTES)
            return tsq_fail(&j->hdr, TSQ_ERR_UNSUPPORTED, "the row checksum does not cover var-len columns");
    j->checksum = on != 0;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_probe_push(tsq_join* j, const tsq_col* cols, int32_t n_cols, int64_t nrows, const uint8_t* selected) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    TSQ_TRY(check_cancel(j));
    if (!j->build_done) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "probe_push before build_finish");
    if (j->probe_done) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "probe_push after probe_finish");
    if (nrows < 0 || (!cols && nrows > 0)) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "bad arguments");
    if (nrows == 0) return TSQ_OK;
    bool dev = false;
    TSQ_TRY(tsq_validate_cols(&j->hdr, cols, n_cols, j->cfg.n_probe_cols, j->cfg.probe_types, nrows, &dev));
    TSQ_HIP(&j->hdr, hipSetDevice(j->ctx->device));
    if (j->st.probe_rows == 0 && j->stage.staged == 0) j->host_mode = !dev;
    if (selected) j->general = true;
    if (dev) {
        TSQ_TRY(probe_flush(j));
        tsq_colset pcs;
        tsq_colset_from_cols(pcs, cols, n_cols);
        // process in slices so an emit batch stays bounded
        // device-resident input: larger emit batches (every batch costs a count pass, two host syncs and its output buffers)
        // (the materialising radix path partitions the whole push at once: its passes are the better the longer the partitions)
        // (the packed pairs routes, once an earlier slice has prepared them: every slice costs a count pass, two host syncs and its
        // output buffers — Q3's lineitem join took 15 slices of a 600 M-row side; 128 Mi rows keep the partition store at 1.5 GB)
        const bool whole = j->count_only || radix_emit_eligible(j, pcs, nrows, selected) || (da_cols_eligible(j, nrows, selected) && j->da_cols_state >= 0);
        int64_t slice = 0;
        for (int64_t off = 0; off < nrows; off += slice) {
            const bool pairs_ready = j->da_state == 1 && j->da_unique && (j->da_rows_state == 1 || j->da_bitrows_state == 1) && j->conds_h.empty() && j->filters_h.empty();  // (unique build side: at most one output row per probe row)
            slice = whole ? nrows : std::max<int64_t>(j->cfg.probe_batch_rows, pairs_ready ? (128 << 20) : (32 << 20));
            const int64_t n = std::min<int64_t>(slice, nrows - off);
            tsq_colset s;
            tsq_colset_slice(s, pcs, off);  // slices are multiples of 64 rows
            TSQ_TRY(check_cancel(j));
            TSQ_TRY(probe_batch(j, s, n, selected ? selected + off : nullptr));
        }
        return TSQ_OK;
    }
    if (j->stage.cap == 0) TSQ_TRY(j->stage.init(&j->hdr, n_cols, j->cfg.probe_types, j->cfg.probe_batch_rows));
    int64_t off = 0;
    while (off < nrows) {
        int64_t n = std::min<int64_t>(nrows - off, j->stage.room());
        j->stage.add(cols, off, n, selected);
        off += n;
        if (j->stage.room() == 0) {
            TSQ_TRY(check_cancel(j));
            TSQ_TRY(probe_flush(j));
        } else if (j->stage.staged - j->stage_sent >= TSQ_STAGE_EARLY_ROWS && (tsq_knob(j->ctx, TSQ_KNOB_HOST_OVERLAP, 3) & 2) != 0) {
            TSQ_TRY(probe_stage_early(j));
        }
    }
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_probe_finish(tsq_join* j) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    TSQ_TRY(check_cancel(j));
    if (!j->build_done) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "probe_finish before build_finish");
    TSQ_HIP(&j->hdr, hipSetDevice(j->ctx->device));
    TSQ_TRY(probe_flush(j));
    j->probe_done = true;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_pull(tsq_join* j, tsq_col* out_cols, int32_t n_cols, int64_t cap_rows, int64_t* nrows_out, int32_t* eos) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    if (!nrows_out || !eos) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "NULL out pointer");
    *nrows_out = 0;
    *eos = 0;
    TSQ_TRY(check_cancel(j));
    if (j->count_only) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "pull in count-only mode");
    const int nout = j->cfg.n_probe_cols + j->cfg.n_build_cols;
    if (n_cols != nout) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "pull: column count must be n_probe_cols + n_build_cols");
    TSQ_HIP(&j->hdr, hipSetDevice(j->ctx->device));
    while (!j->results.empty() && j->results.front()->cursor >= j->results.front()->rows) {
        j->results.front()->release();
        j->results.pop_front();
    }
    if (j->results.empty()) {
        if (j->probe_done) *eos = 1;
        return TSQ_OK;
    }
    ResultBatch* rb = j->results.front().get();
    if (rb->pending) {
        // its copies are still on the way: while the probe side has more to push the caller gets "no rows yet" (it pushes its next chunk, as
        // after any pull that found nothing) instead of waiting beside an idle staging buffer; after probe_finish the call waits
        if (!j->probe_done && hipEventQuery(rb->ready) == hipErrorNotReady) return TSQ_OK;
        TSQ_TRY(settle_batch(j, rb));
    }
    const int64_t n = std::min<int64_t>(cap_rows, rb->rows - rb->cursor);
    if (n <= 0) return TSQ_OK;
    const bool probe_is_left = j->cfg.build_is_right != 0;
    const int nl = probe_is_left ? j->cfg.n_probe_cols : j->cfg.n_build_cols;
    for (int oc = 0; oc < nout; oc++) {
        const bool from_probe = probe_is_left ? oc < nl : oc >= nl;
        const int sc = oc < nl ? oc : oc - nl;
        const int32_t type = from_probe ? j->cfg.probe_types[sc] : j->cfg.build_types[sc];
        const int es = tsq_elem_size(type);
        tsq_col& o = out_cols[oc];
        const bool odev = o.flags & TSQ_COL_DEVICE;
        const bool has_bm = rb->on_host ? rb->hbitmap[oc].p != nullptr : rb->bitmap[oc].p != nullptr;
        if ((o.flags & TSQ_COL_BORROW) && odev && !rb->on_host && type != TSQ_BYTES) {  // pointers into the result batch: no copy
            if ((rb->cursor & 7) != 0) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "device pull: cap_rows must keep the cursor a multiple of 8");
            o.data = rb->data[oc].p ? (char*)rb->data[oc].p + (size_t)rb->cursor * es : nullptr;  // (nullptr: a column the parent does not use)
            o.null_bitmap = has_bm ? rb->bitmap[oc].as<uint8_t>() + (rb->cursor >> 3) : nullptr;
            o.length = n;
            o.type = type;
            o.elem_size = es;
            continue;
        }
        if ((o.flags & TSQ_COL_BORROW) && !odev && rb->on_host && type != TSQ_BYTES) {  // round 6: the same for HOST pulls — pointers into the pinned result batch
            if ((rb->cursor & 7) != 0) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "borrowed pull: cap_rows must keep the cursor a multiple of 8");
            o.data = rb->hdata[oc].p ? (char*)rb->hdata[oc].p + (size_t)rb->cursor * es : nullptr;
            o.null_bitmap = has_bm ? (uint8_t*)rb->hbitmap[oc].p + (rb->cursor >> 3) : nullptr;
            o.length = n;
            o.type = type;
            o.elem_size = es;
            continue;
        }
        if (!j->used_out.empty() && !j->used_out[(size_t)oc]) {  // not materialised (tsq_join_set_used_columns): nothing to copy
            o.length = n;
            continue;
        }
        if (!o.data) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "pull: out column data == NULL");
        if (type == TSQ_BYTES && !o.offsets) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "pull: var-len out column needs an offsets buffer (tsq_join_peek tells the data bytes)");
        if (rb->on_host && !odev) {
            if (type == TSQ_BYTES) {  // cells [cursor, cursor + n): their bytes, and the offsets moved to start at 0
                const int64_t* so = (const int64_t*)rb->hoffs[oc].p + rb->cursor;
                memcpy(o.data, (const char*)rb->hdata[oc].p + so[0], (size_t)(so[n] - so[0]));
                for (int64_t i = 0; i <= n; i++) o.offsets[i] = so[i] - so[0];
            } else
            tsq_host_copy(o.data, (const char*)rb->hdata[oc].p + (size_t)rb->cursor * es, (size_t)n * es);
            if (o.null_bitmap) {
                if (!has_bm) memset(o.null_bitmap, 0xff, tsq_bitmap_bytes(n));
                else if ((rb->cursor & 7) == 0) {
                    memcpy(o.null_bitmap, (const uint8_t*)rb->hbitmap[oc].p + (rb->cursor >> 3), tsq_bitmap_bytes(n));
                } else {
                    const uint8_t* src = (const uint8_t*)rb->hbitmap[oc].p;
                    memset(o.null_bitmap, 0, tsq_bitmap_bytes(n));
                    for (int64_t i = 0; i < n; i++) {
                        const int64_t s = rb->cursor + i;
                        if ((src[s >> 3] >> (s & 7)) & 1) o.null_bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
                    }
                }
            } else if (has_bm) {
                return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "pull: column may contain NULLs but no null_bitmap buffer was given");
            }
        } else if (!rb->on_host && odev) {
            if (type == TSQ_BYTES) {
                const int64_t* so = rb->offs[oc].as<int64_t>() + rb->cursor;
                TSQ_HIP(&j->hdr, hipMemcpyAsync(j->ctx->pinned + 44, so, 8, hipMemcpyDeviceToHost, j->ctx->stream));
                TSQ_HIP(&j->hdr, hipMemcpyAsync(j->ctx->pinned + 45, so + n, 8, hipMemcpyDeviceToHost, j->ctx->stream));
                TSQ_HIP(&j->hdr, hipStreamSynchronize(j->ctx->stream));
                const int64_t b0 = (int64_t)j->ctx->pinned[44], b1 = (int64_t)j->ctx->pinned[45];
                if (b1 > b0) TSQ_HIP(&j->hdr, hipMemcpyAsync(o.data, (const char*)rb->data[oc].p + b0, (size_t)(b1 - b0), hipMemcpyDeviceToDevice, j->ctx->stream));
                TSQ_TRY(tsq_launch_offsets_rebase(j->ctx, &j->hdr, o.offsets, so, n + 1, -b0));
            } else
            TSQ_HIP(&j->hdr, hipMemcpyAsync(o.data, (const char*)rb->data[oc].p + (size_t)rb->cursor * es, (size_t)n * es, hipMemcpyDeviceToDevice, j->ctx->stream));
            if (o.null_bitmap) {
                if ((rb->cursor & 7) != 0) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "device pull: cap_rows must keep the cursor a multiple of 8");
                if (!has_bm) TSQ_HIP(&j->hdr, hipMemsetAsync(o.null_bitmap, 0xff, tsq_bitmap_bytes(n), j->ctx->stream));
                else TSQ_HIP(&j->hdr, hipMemcpyAsync(o.null_bitmap, rb->bitmap[oc].as<uint8_t>() + (rb->cursor >> 3), tsq_bitmap_bytes(n), hipMemcpyDeviceToDevice, j->ctx->stream));
            } else if (has_bm) {
                return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "pull: column may contain NULLs but no null_bitmap buffer was given");
            }
        } else {
            return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "pull: output placement (host/device) must match the probe pushes");
        }
        o.length = n;
        o.type = type;
        o.elem_size = type == TSQ_BYTES ? -1 : es;
    }
    if (!rb->on_host) TSQ_HIP(&j->hdr, hipStreamSynchronize(j->ctx->stream));
    rb->cursor += n;
    *nrows_out = n;
    return TSQ_OK;
}

// What the next tsq_join_pull of up to cap_rows rows will deliver: the row count and, per output column, the data bytes of a
// var-len column (0 for fixed-width columns) — a Go caller sizes its chunk.Column.data with it (column.go:207-211 grows by append).
TSQ_API tsq_status tsq_join_peek(tsq_join* j, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_out, int32_t n_cols) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    if (!nrows_out || !bytes_out) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "NULL out pointer");
    const int nout = j->cfg.n_probe_cols + j->cfg.n_build_cols;
    if (n_cols != nout) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "peek: column count must be n_probe_cols + n_build_cols");
    *nrows_out = 0;
    for (int oc = 0; oc < nout; oc++) bytes_out[oc] = 0;
    TSQ_TRY(check_cancel(j));
    TSQ_HIP(&j->hdr, hipSetDevice(j->ctx->device));
    while (!j->results.empty() && j->results.front()->cursor >= j->results.front()->rows) {
        j->results.front()->release();
        j->results.pop_front();
    }
    if (j->results.empty()) return TSQ_OK;
    ResultBatch* rb = j->results.front().get();
    TSQ_TRY(settle_batch(j, rb));  // (waits: what peek says and what the next pull delivers must be the same batch)
    const int64_t n = std::min<int64_t>(cap_rows, rb->rows - rb->cursor);
    if (n <= 0) return TSQ_OK;
    *nrows_out = n;
    for (int oc = 0; oc < nout; oc++) {
        if (rb->on_host ? !(oc < (int)rb->hoffs.size() && rb->hoffs[oc].p) : !rb->offs[oc].p) continue;
        if (rb->on_host) {
            const int64_t* so = (const int64_t*)rb->hoffs[oc].p + rb->cursor;
            bytes_out[oc] = so[n] - so[0];
        } else {
            const int64_t* so = rb->offs[oc].as<int64_t>() + rb->cursor;
            TSQ_HIP(&j->hdr, hipMemcpyAsync(j->ctx->pinned + 44, so, 8, hipMemcpyDeviceToHost, j->ctx->stream));
            TSQ_HIP(&j->hdr, hipMemcpyAsync(j->ctx->pinned + 45, so + n, 8, hipMemcpyDeviceToHost, j->ctx->stream));
            TSQ_HIP(&j->hdr, hipStreamSynchronize(j->ctx->stream));
            bytes_out[oc] = (int64_t)j->ctx->pinned[45] - (int64_t)j->ctx->pinned[44];
        }
    }
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_count(tsq_join* j, int64_t* rows_out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN || !rows_out) return TSQ_ERR_INVALID;
    TSQ_TRY(check_cancel(j));
    TSQ_HIP(&j->hdr, hipSetDevice(j->ctx->device));
    if (!j->count_only) {
        *rows_out = j->total_out;
        return TSQ_OK;
    }
    unsigned long long c[8];
    TSQ_TRY(read_counters(j, c));
    TSQ_TRY(status_from_errword(j, c[3]));
    int64_t child = 0;
    if (j->wide) {  // the batches that went through the composite-key child
        const tsq_status ws = tsq_join_count(j->wide, &child);
        if (ws != TSQ_OK) return tsq_fail(&j->hdr, ws, j->wide->hdr.err);
    }
    *rows_out = (int64_t)c[0] + child;
    j->st.out_rows = *rows_out;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_checksum(tsq_join* j, uint64_t* sum_out, uint64_t* xor_out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN || !sum_out || !xor_out) return TSQ_ERR_INVALID;
    if (!j->count_only || !j->checksum) return tsq_fail(&j->hdr, TSQ_ERR_INVALID, "checksum mode is not enabled");
    TSQ_HIP(&j->hdr, hipSetDevice(j->ctx->device));
    unsigned long long c[8];
    TSQ_TRY(read_counters(j, c));
    TSQ_TRY(status_from_errword(j, c[3]));
    *sum_out = c[1];
    *xor_out = c[2];
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_cancel(tsq_join* j) {
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return TSQ_ERR_INVALID;
    j->cancelled.store(1);
    return TSQ_OK;
}

TSQ_API tsq_status tsq_join_stats(tsq_join* j, tsq_stats* out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN || !out) return TSQ_ERR_INVALID;
    TSQ_HIP(&j->hdr, hipSetDevice(j->ctx->device));
    TSQ_HIP(&j->hdr, hipStreamSynchronize(j->ctx->stream));
    float ms = 0;
    if (j->have_build_ev && hipEventElapsedTime(&ms, j->ev[0], j->ev[1]) == hipSuccess) j->st.build_kernel_ms = ms;
    if (j->have_probe_ev && hipEventElapsedTime(&ms, j->ev[2], j->ev[3]) == hipSuccess) j->st.probe_kernel_ms = ms;
    j->st.partition_kernel_ms = 0;
    j->st.radix_overflow_rows = 0;
    j->st.build_handed_back_rows = j->build_handed_back;
    j->st.packed_build_ms = j->da_build_ms;
    {   // warnings of OtherConditions / outer filters so far: counters[4] of the direct kernels + the packed route's own count
        unsigned long long d0 = 0;
        if (hipMemcpy(&d0, j->counters.as<unsigned long long>() + 4, 8, hipMemcpyDeviceToHost) == hipSuccess) j->st.div_by_zero_warnings = (int64_t)d0 + j->div0_packed;
    }
    j->st.shared_build = j->shared ? 1 : 0;
    j->st.shared_image_bytes = j->shared_image_bytes;
    j->st.shared_allreduce_ms = j->shared_allreduce_ms;
    j->st.radix_probe_kernel_ms = j->st.partition_kernel_ms_sum = j->st.radix_probe_kernel_ms_sum = 0;
    j->st.radix_timed_batches = 0;
    if (j->st.radix_batches > 0 && j->rctl.p) {
        const int64_t nt = std::min<int64_t>(j->st.radix_batches, tsq_join::RING);
        for (int64_t b = j->st.radix_batches - nt; b < j->st.radix_batches; b++) {
            hipEvent_t* re = j->rev[b % tsq_join::RING];
            float pm = 0, qm = 0;
            if (!re[0] || !re[1] || !re[2]) continue;
            if (hipEventElapsedTime(&pm, re[0], re[1]) != hipSuccess || hipEventElapsedTime(&qm, re[1], re[2]) != hipSuccess) continue;
            j->st.partition_kernel_ms_sum += pm;
            j->st.radix_probe_kernel_ms_sum += qm;
            j->st.radix_timed_batches++;
            j->st.partition_kernel_ms = pm;
            j->st.radix_probe_kernel_ms = qm;
        }
        uint32_t ovf = 0;  // overflow count of the last radix batch sits behind the cursors
        const size_t nregions = ((size_t)1 << j->st.radix_bits) * 8;
        if (hipMemcpy(&ovf, j->rctl.as<uint32_t>() + nregions, 4, hipMemcpyDeviceToHost) == hipSuccess) j->st.radix_overflow_rows = ovf;
    }
    *out = j->st;
    return TSQ_OK;
}

TSQ_API void tsq_join_destroy(tsq_join* j) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(j, TSQ_MAGIC_JOIN));
    if (!j || j->hdr.magic != TSQ_MAGIC_JOIN) return;
    (void)hipSetDevice(j->ctx->device);
    (void)hipStreamSynchronize(j->ctx->stream);  // Close() drains in-flight work (join.go:81-107)
    if (j->wide) {
        tsq_join_destroy(j->wide);
        j->wide = nullptr;
    }
    for (auto& c : j->bcols) c.release();
    for (auto& c : j->pcols) c.release();
    if (j->copy_stream) (void)hipStreamSynchronize(j->copy_stream);
    for (auto& r : j->results) r->release();
    j->results.clear();
    if (j->copy_stream) (void)hipStreamDestroy(j->copy_stream);
    if (j->ev_emit) (void)hipEventDestroy(j->ev_emit);
    if (j->ev_h2d) (void)hipEventDestroy(j->ev_h2d);
    j->tkeys.release();
    j->tvals.release();
    j->tnext.release();
    j->sent.release();
    j->psel.release();
    j->counters.release();
    j->conds_d.release();
    j->filters_d.release();
    j->fflags.release();
    j->heads.release();
    j->stage.release();
    for (int i = 0; i < 4; i++)
        if (j->ev[i]) (void)hipEventDestroy(j->ev[i]);
    for (int i = 0; i < tsq_join::RING; i++)
        for (int e = 0; e < 3; e++)
            if (j->rev[i][e]) (void)hipEventDestroy(j->rev[i][e]);
    j->bbase.release();
    j->pairs.release();
    j->firstcnt.release();
    j->rkeys.release();
    j->rctl.release();
    j->rvend.release();
    j->rovf.release();
    j->tkcnt.release();
    j->da_img.release();
    j->da_ckey.release();
    j->rckey.release();
    for (DevBuf* b : {&j->da_coarse, &j->da_pstart, &j->da_coarse_c, &j->da_pstart_c, &j->da_brows, &j->ridx, &j->rovfidx, &j->rmiss, &j->rnnmask}) b->release();
    for (DevBuf* b : {&j->kr_brec, &j->kr_bstart, &j->kr_counts, &j->kr_prec, &j->kr_pstart, &j->kr_flags, &j->kr_bids, &j->kr_pids, &j->kr_pcnt, &j->kr_norec}) b->release();
    for (int k = 0; k < TSQ_MAX_KEYS; k++) {
        j->kr_bdig[k].release();
        j->kr_pdig[k].release();
        j->kr_vmask.release();
    }
    for (DevBuf* b : {&j->dm_bent, &j->dm_bnn, &j->dm_boff, &j->dm_bcnt, &j->dm_bitmap, &j->dm_pent, &j->dm_pnn, &j->dm_poff, &j->dm_pcnt}) b->release();
    for (int c = 0; c < TSQ_DA_MAXCOLS; c++) {
        j->da_bsorted[c].release();
        j->da_bsorted_nn[c].release();
        j->rcols[c].release();
        j->dm_bpay[c].release();
        j->dm_ppay[c].release();
        j->dm_l1pay[c].release();
    }
    for (DevBuf* b : {&j->dm_l1ent, &j->dm_l1ctl, &j->dm_l1vend, &j->dm_l1nn}) b->release();
    for (int v = 0; v < TSQ_LDS_MAXPAY; v++) {
        j->rpay[v].release();
        j->rovfpay[v].release();
        j->tpay[v].release();
    }
    j->hdr.magic = 0;
    delete j;
}
