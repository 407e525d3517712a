// tsq_agg.hip — hash aggregation for gfx950 (MI355X).  Replaces executor/aggregate.go's partial
// workers / shuffle / final workers and executor/aggfuncs/* (citations at each piece).
//
// Data layout in HBM: one open-addressed group table, structure of arrays, capacity `cap` slots
// (+2 special slots: cap = the group whose key word equals the EMPTY sentinel, cap+1 = the NULL
// group; NULL is a regular group in GROUP BY, util/codec/codec.go:718-719):
//     tag[cap+2]            uint64   single key: the key word itself (exact);
//                                    multi key : 64-bit mix of all (null,word) cells
//     gkey[k][cap+2]        uint64   key words of the group (output of firstrow(key) / verification)
//     gknull[cap+2]         uint8    bitmask of NULL key cells (multi key only)
//     per aggregate i: acc[i], aux[i], cnt[i] (uint64) and seen[i] (uint8)
// Group key words follow HashGroupKey (codec.go:713-746): ints by value (UNSIGNED ignored), reals by
// their memcomparable image (float.go:22-30) so that -0.0 and +0.0 share a group.
//
// Update = one pass per pushed batch: find-or-claim the slot with a 64-bit CAS, then one or two
// device-scope atomics per aggregate.  int64 SUM/AVG accumulate in 128 bits (lo += v with carry into
// hi); overflow is reported iff the exact group sum leaves the BIGINT range (func_sum.go:133-137
// reports it as soon as a running sum overflows, which depends on worker interleaving there).
//
// Var-len cells (string group keys: codec.go:738-744; firstRow4String, func_first_row.go:193-230; maxMin4String,
// func_max_min.go:312-378).  The reference deep-copies the string a partial result keeps (stringutil.Copy); here the
// operator keeps the bytes of every var-len input column it was pushed — one growing heap per column in HBM — and a cell
// is a 64-bit REFERENCE into that heap, byte offset << 24 | length.  A reference fits the 8-byte words of gkey[] / acc[],
// so claiming a group, FIRST_ROW and table growth move strings as words; only the equality check of a string key and the
// MAX/MIN comparison (a CAS loop on the reference) touch the bytes.  A string key makes the aggregate take the multi-key
// path (tag = hash of the bytes, the bytes themselves verified in phase 1).
#include "tsq_stage.h"
#include "tsq_aggfast.h"
#include "tsq_daagg.h"
#include "tsq_keydict.h"

#include <memory>

#define TSQ_EMPTY_TAG 0x8080808080808080ULL
#define TSQ_BUSY_TAG 0x8080808080808082ULL  /* multi key, phase 1: slot claimed, its key cells not yet published */

struct AggState {  // device pointers of one aggregate's state arrays
    unsigned long long* acc;
    unsigned long long* aux;
    unsigned long long* cnt;
    uint8_t* seen;
};
struct AggPlan {
    int32_t n_keys;
    int32_t key_col[TSQ_MAX_GROUP_KEYS];
    int32_t n_aggs;
    tsq_agg_func f[TSQ_MAX_AGGS];
};
struct AggTable {
    unsigned long long* tag;
    unsigned long long* gkey[TSQ_MAX_GROUP_KEYS];
    uint8_t* gknull;
    AggState st[TSQ_MAX_AGGS];
    uint64_t cap;
};
struct AggArgs {
    tsq_colset in;
    AggPlan plan;
    AggTable t;
    int64_t nrows;
    int64_t row_base;               // global row number of row 0 (diagnostics only)
    const uint32_t* retry_in;       // optional: process only these rows
    uint32_t* retry_out;            // rows that found no slot within the probe limit
    unsigned long long* counters;   // [0]=new groups [1]=retry count [2]=rows whose tag was another key's (resolved) [5]=a string cell too long for a reference
    int32_t phase;                  // multi key: 0 = claim slots, 1 = verify + update
    uint32_t* slot_of;              // multi key: slot found by phase 0 for item r (0xffffffff = handed back), read by phase 1
    uint32_t tag_bits;              // 0, or (tests) keep only this many bits of the multi-key tag so that distinct keys collide
    uint64_t bail_after;            // once this many items were handed back the table is too small: the rest skip the walk
};

// group key word of one cell (codec.go:713-746 semantics, see header)
__device__ __forceinline__ uint64_t group_key_word(const tsq_colset& cs, int c, int64_t row) {
    if (cs.type[c] == TSQ_F32 || cs.type[c] == TSQ_F64) {
        double f = cs.type[c] == TSQ_F32 ? (double)((const float*)cs.data[c])[row] : ((const double*)cs.data[c])[row];
        uint64_t u = tsq_f64_bits(f);
        return f >= 0 ? (u | 0x8000000000000000ULL) : ~u;  // float.go:22-30 (−0.0 >= 0 is true)
    }
    return ((const uint64_t*)cs.data[c])[row];
}
// inverse of the float image, for firstrow(key)/output
__device__ __forceinline__ uint64_t group_key_word_decode(uint64_t w, int32_t type) {
    if (type == TSQ_F32 || type == TSQ_F64) {
        uint64_t u = (w & 0x8000000000000000ULL) ? (w & ~0x8000000000000000ULL) : ~w;  // float.go:32-40
        if (type == TSQ_F32) {
            float f = (float)tsq_bits_f64(u);
            uint32_t b;
            memcpy(&b, &f, 4);
            return b;
        }
        return u;
    }
    return w;
}

// order preserving images for atomicMax/atomicMin on uint64
__device__ __forceinline__ uint64_t ord_image(const tsq_colset& cs, int c, int32_t type, int64_t row) {
    switch (type) {
        case TSQ_I64: return ((const uint64_t*)cs.data[c])[row] ^ 0x8000000000000000ULL;
        case TSQ_U64: return ((const uint64_t*)cs.data[c])[row];
        default: {
            double f = cs.type[c] == TSQ_F32 ? (double)((const float*)cs.data[c])[row] : ((const double*)cs.data[c])[row];
            uint64_t u = tsq_f64_bits(f);
            return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
        }
    }
}
__device__ __forceinline__ uint64_t ord_image_decode(uint64_t w, int32_t type) {
    switch (type) {
        case TSQ_I64: return w ^ 0x8000000000000000ULL;
        case TSQ_U64: return w;
        default: {
            uint64_t u = (w >> 63) ? (w & ~0x8000000000000000ULL) : ~w;
            if (type == TSQ_F32) {
                float f = (float)tsq_bits_f64(u);
                uint32_t b;
                memcpy(&b, &f, 4);
                return b;
            }
            return u;
        }
    }
}

__device__ __forceinline__ bool is_real_type(int32_t t) { return t == TSQ_F32 || t == TSQ_F64; }

// ---- references to var-len cells (see the file header)
#define TSQ_REF_LEN_BITS 24
#define TSQ_REF_MAXLEN ((1ull << TSQ_REF_LEN_BITS) - 2)  /* a longer cell raises counters[5]: UNSUPPORTED */
#define TSQ_REF_NONE (~0ull)                             /* MAX/MIN of strings: no value yet */
__device__ __forceinline__ uint64_t ref_len(uint64_t ref) { return ref & ((1ull << TSQ_REF_LEN_BITS) - 1); }
__device__ __forceinline__ uint64_t ref_off(uint64_t ref) { return ref >> TSQ_REF_LEN_BITS; }
// the cell (c, row) of a batch whose var-len columns ARE the heap (data = heap base, offsets absolute)
__device__ __forceinline__ uint64_t str_ref(const tsq_colset& cs, int c, int64_t row, unsigned long long* too_long) {
    const int64_t o = cs.offs[c][row], n = cs.offs[c][row + 1] - o;
    if ((uint64_t)n > TSQ_REF_MAXLEN) {
        __hip_atomic_store(too_long, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (uint64_t)o << TSQ_REF_LEN_BITS;
    }
    return ((uint64_t)o << TSQ_REF_LEN_BITS) | (uint64_t)n;
}
__device__ __forceinline__ int ref_cmp(const void* heap, uint64_t x, uint64_t y) {  // types.CompareString: binary collation
    const uint8_t* h = (const uint8_t*)heap;
    return tsq_cmp_bytes(h + ref_off(x), (uint32_t)ref_len(x), h + ref_off(y), (uint32_t)ref_len(y));
}
__device__ __forceinline__ bool ref_equal(const void* heap, uint64_t x, uint64_t y) {
    if (x == y) return true;
    if (ref_len(x) != ref_len(y)) return false;
    return ref_cmp(heap, x, y) == 0;
}
// the 8-byte image of a cell as FIRST_ROW keeps it
__device__ __forceinline__ uint64_t agg_cell(const tsq_colset& cs, int c, int64_t row, unsigned long long* too_long) {
    return cs.type[c] == TSQ_BYTES ? str_ref(cs, c, row, too_long) : tsq_cell_raw(cs, c, row);
}

// 128-bit accumulate of a signed 64-bit addend: lo += v (returns carry), hi += sign(v) + carry
__device__ __forceinline__ void add128(unsigned long long* lo, unsigned long long* hi, int64_t v) {
    const unsigned long long uv = (unsigned long long)v;
    const unsigned long long old = atomicAdd(lo, uv);
    const long long carry = (old + uv < old) ? 1 : 0;
    const long long d = carry + (v < 0 ? -1 : 0);
    if (d) atomicAdd(hi, (unsigned long long)d);
}

// applies every aggregate of `plan` for input row `row` to slot `s`.
// `winner` = this thread created the group (writes FIRST_ROW values, func_first_row.go:67-81:
// any row of the group is a legitimate "first" row under parallel workers).
__device__ __forceinline__ void agg_update_slot(const AggArgs& a, uint64_t s, int64_t row, bool winner) {
    for (int i = 0; i < a.plan.n_aggs; i++) {
        const tsq_agg_func f = a.plan.f[i];
        const AggState st = a.t.st[i];
        const bool merge = f.mode == TSQ_MODE_FINAL || f.mode == TSQ_MODE_PARTIAL2;
        const bool arg_null = f.arg_col >= 0 ? tsq_is_null(a.in.nulls[f.arg_col], row) : false;
        switch (f.func) {
            case TSQ_AGG_COUNT:  // func_count.go:33-119
                if (arg_null) break;
                atomicAdd(&st.acc[s], merge ? ((const unsigned long long*)a.in.data[f.arg_col])[row] : 1ull);
                break;
            case TSQ_AGG_SUM:  // func_sum.go:60-154
                if (arg_null) break;
                if (is_real_type(f.arg_type)) {
                    double v = a.in.type[f.arg_col] == TSQ_F32 ? (double)((const float*)a.in.data[f.arg_col])[row]
                                                                : ((const double*)a.in.data[f.arg_col])[row];
                    atomicAdd((double*)&st.acc[s], v);
                } else {
                    add128(&st.acc[s], &st.aux[s], ((const int64_t*)a.in.data[f.arg_col])[row]);
                }
                st.seen[s] = 1;
                break;
            case TSQ_AGG_AVG: {  // func_avg.go:62-128,164-216
                int vc = merge ? f.arg_col2 : f.arg_col;
                if (arg_null) break;
                if (merge && tsq_is_null(a.in.nulls[vc], row)) break;
                if (is_real_type(f.arg_type)) {
                    double v = a.in.type[vc] == TSQ_F32 ? (double)((const float*)a.in.data[vc])[row] : ((const double*)a.in.data[vc])[row];
                    atomicAdd((double*)&st.acc[s], v);
                } else {
                    add128(&st.acc[s], &st.aux[s], ((const int64_t*)a.in.data[vc])[row]);
                }
                atomicAdd(&st.cnt[s], merge ? ((const unsigned long long*)a.in.data[f.arg_col])[row] : 1ull);
                break;
            }
            case TSQ_AGG_MAX:  // func_max_min.go:81-117 (+uint/float variants)
            case TSQ_AGG_MIN:
                if (arg_null) break;
                if (f.arg_type == TSQ_BYTES) {  // maxMin4String (func_max_min.go:337-362): the reference of the best string so far
                    const unsigned long long mine = str_ref(a.in, f.arg_col, row, &a.counters[5]);
                    unsigned long long cur = __hip_atomic_load(&st.acc[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (;;) {
                        if (cur != TSQ_REF_NONE) {
                            const int c = ref_cmp(a.in.data[f.arg_col], mine, cur);
                            if (f.func == TSQ_AGG_MAX ? c <= 0 : c >= 0) break;
                        }
                        const unsigned long long prev = atomicCAS(&st.acc[s], cur, mine);
                        if (prev == cur) break;
                        cur = prev;
                    }
                } else if (f.func == TSQ_AGG_MAX) {
                    atomicMax(&st.acc[s], (unsigned long long)ord_image(a.in, f.arg_col, f.arg_type, row));
                } else {
                    atomicMin(&st.acc[s], (unsigned long long)ord_image(a.in, f.arg_col, f.arg_type, row));
                }
                st.seen[s] = 1;
                break;
            case TSQ_AGG_FIRSTROW:
                if (!winner) break;
                st.acc[s] = arg_null ? 0ull : agg_cell(a.in, f.arg_col, row, &a.counters[5]);
                st.seen[s] = arg_null ? 0 : 1;
                break;
        }
    }
}

// the 64-bit tag of a several-column group key: a chain over the key words (shared by the row upsert and the merge of packed
// partial groups — the same key must reach the same tag whichever way its rows came)
#define TSQ_AGG_TAG_SEED 0x6A09E667F3BCC908ULL
__device__ __forceinline__ uint64_t agg_tag_step(uint64_t h, uint64_t hw, bool isnull) {
    return tsq_splitmix64(h ^ hw) + (isnull ? 0x9E3779B97F4A7C15ULL : 0);
}
__device__ __forceinline__ uint64_t agg_tag_finish(uint64_t h, uint32_t tag_bits) {
    if (tag_bits) h &= (1ull << tag_bits) - 1;
    return (h == TSQ_EMPTY_TAG || h == TSQ_BUSY_TAG) ? h ^ 1 : h;
}

// A walk longer than this means the table is over-full for this batch (the host keeps load <= 0.5 for the groups it
// knows, where a 256-slot run is astronomically unlikely): the item is handed back and the host grows the table.
#define TSQ_AGG_PROBE_LIMIT 256

// adds the per-thread values of a 256-thread workgroup to *dst with one device atomic (all threads must call it)
__device__ __forceinline__ void block_add_u32(unsigned long long* dst, uint32_t v) {
    __shared__ unsigned int s_sum;
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_sum, v);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum) atomicAdd(dst, (unsigned long long)s_sum);
}

// K7 — group-table upsert.  Replaces HashAggPartialWorker.updatePartialResult
// (executor/aggregate.go:332-350): getGroupKey (:359-394) + getPartialResult (:396-410) + the per
// row UpdatePartialResult calls.  SINGLE key: one fused pass.  MULTI key: phase 0 claims slots by
// 64-bit tag and the claimer stores the key cells; phase 1 (a later launch, so the cells are
// visible) verifies the cells and applies the aggregates.  Phase 0 leaves the slot of every item in slot_of[], so phase 1
// does not walk the table again — unless the slot belongs to ANOTHER key with the same 64-bit tag: then the row walks on
// from there, comparing cells, and finds or claims its own group (see the BUSY protocol below).  Two keys never merge
// and the stream never has to be abandoned for a collision.
template <bool MULTI>
__global__ void __launch_bounds__(256) k_agg_update(AggArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t new_groups = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.nrows; r += stride) {
        const int64_t row = a.retry_in ? (int64_t)a.retry_in[r] : r;
        uint64_t tag, slot;
        uint64_t kw[TSQ_MAX_GROUP_KEYS];
        uint32_t nullmask = 0;
        bool special = false;
        if (a.plan.n_keys == 0) {  // no GROUP BY: one global group, kept in the NULL-group slot
            slot = a.t.cap + 1;
            special = true;
            tag = 0;
        } else if (!MULTI) {
            const int c = a.plan.key_col[0];
            if (tsq_is_null(a.in.nulls[c], row)) { slot = a.t.cap + 1; special = true; tag = 0; kw[0] = 0; nullmask = 1; }
            else {
                kw[0] = group_key_word(a.in, c, row);
                tag = kw[0];
                if (tag == TSQ_EMPTY_TAG) { slot = a.t.cap; special = true; }
            }
        } else {
            uint64_t h = TSQ_AGG_TAG_SEED;
            for (int k = 0; k < a.plan.n_keys; k++) {
                const int c = a.plan.key_col[k];
                const bool isn = tsq_is_null(a.in.nulls[c], row);
                uint64_t hw;
                if (a.in.type[c] == TSQ_BYTES) {  // the group keeps a reference; the tag (phase 0 only) hashes the bytes
                    kw[k] = isn ? 0 : str_ref(a.in, c, row, &a.counters[5]);
                    hw = (isn || a.phase != 0) ? 0 : tsq_hash_bytes((const uint8_t*)a.in.data[c] + ref_off(kw[k]), (int64_t)ref_len(kw[k]));
                } else {
                    kw[k] = isn ? 0 : group_key_word(a.in, c, row);
                    hw = kw[k];
                }
                nullmask |= isn ? (1u << k) : 0u;
                h = agg_tag_step(h, hw, isn);
            }
            tag = agg_tag_finish(h, a.tag_bits);
        }
        bool winner = false;
        if (MULTI && a.phase == 1) {
            const uint32_t s32 = a.slot_of[r];
            if (s32 == 0xffffffffu) continue;  // handed back by phase 0
            slot = s32;
        } else if (special) {
            // the two special slots are claimed through their tag word as well (EMPTY -> 1)
            if (__hip_atomic_load(&a.t.tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == TSQ_EMPTY_TAG)
                winner = atomicCAS(&a.t.tag[slot], (unsigned long long)TSQ_EMPTY_TAG, 1ull) == TSQ_EMPTY_TAG;
        } else {
            slot = tsq_mulhi64(tsq_mix64(tag), a.t.cap);
            bool found = false;
            // a batch with far more new groups than free slots: after bail_after failed walks nobody walks any more
            const bool hopeless = __hip_atomic_load(&a.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > a.bail_after;
            for (int probe = 0; probe < TSQ_AGG_PROBE_LIMIT && !hopeless; probe++) {
                unsigned long long cur = __hip_atomic_load(&a.t.tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == TSQ_EMPTY_TAG) {
                    cur = atomicCAS(&a.t.tag[slot], (unsigned long long)TSQ_EMPTY_TAG, (unsigned long long)tag);
                    if (cur == TSQ_EMPTY_TAG) { winner = true; found = true; break; }
                }
                if (cur == tag) { found = true; break; }
                slot = slot + 1 == a.t.cap ? 0 : slot + 1;
            }
            if (!found) {  // table (nearly) full: hand the row back, the host grows the table
                uint32_t i = (uint32_t)atomicAdd(&a.counters[1], 1ull);
                a.retry_out[i] = (uint32_t)row;
                if (MULTI) a.slot_of[r] = 0xffffffffu;
                continue;
            }
        }
        if (MULTI && a.phase == 0) a.slot_of[r] = (uint32_t)slot;
        if (winner) {
            new_groups++;
            for (int k = 0; k < a.plan.n_keys; k++) a.t.gkey[k][slot] = kw[k];
            if (a.t.gknull) a.t.gknull[slot] = (uint8_t)nullmask;
        }
        if (!MULTI) {
            agg_update_slot(a, slot, row, winner);
        } else if (a.phase == 0) {
            if (winner) {  // FIRST_ROW values come from the claimer
                for (int i = 0; i < a.plan.n_aggs; i++) {
                    const tsq_agg_func f = a.plan.f[i];
                    if (f.func != TSQ_AGG_FIRSTROW) continue;
                    const bool arg_null = tsq_is_null(a.in.nulls[f.arg_col], row);
                    a.t.st[i].acc[slot] = arg_null ? 0ull : agg_cell(a.in, f.arg_col, row, &a.counters[5]);
                    a.t.st[i].seen[slot] = arg_null ? 0 : 1;
                }
            }
        } else {
            auto keys_at = [&](uint64_t sl) -> bool {
                bool same = a.t.gknull[sl] == (uint8_t)nullmask;
                for (int k = 0; k < a.plan.n_keys && same; k++) {
                    const int c = a.plan.key_col[k];
                    if (a.in.type[c] == TSQ_BYTES) same = ((nullmask >> k) & 1u) || ref_equal(a.in.data[c], a.t.gkey[k][sl], kw[k]);
                    else same = a.t.gkey[k][sl] == kw[k];
                }
                return same;
            };
            bool claimed = false;
            if (!keys_at(slot)) {
                // Two different keys share a 64-bit tag: this row's group lives further along the probe sequence, or nowhere yet.
                // Everything phase 0 wrote is visible now, so tags AND cells are compared; a new group is claimed as BUSY, its
                // cells written, then its tag published — a lane that meets BUSY looks at the slot again (the claimer never
                // waits for anybody, so the wave makes progress whatever its lanes do).
                atomicAdd(&a.counters[2], 1ull);  // statistic: rows that went the long way
                bool found = false;
                int probe = 0;
                slot = slot + 1 == a.t.cap ? 0 : slot + 1;
                while (probe < TSQ_AGG_PROBE_LIMIT && !found) {
                    unsigned long long cur = __hip_atomic_load(&a.t.tag[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (cur == TSQ_EMPTY_TAG) {
                        cur = atomicCAS(&a.t.tag[slot], (unsigned long long)TSQ_EMPTY_TAG, (unsigned long long)TSQ_BUSY_TAG);
                        if (cur == TSQ_EMPTY_TAG) {
                            for (int k = 0; k < a.plan.n_keys; k++) a.t.gkey[k][slot] = kw[k];
                            a.t.gknull[slot] = (uint8_t)nullmask;
                            __hip_atomic_store(&a.t.tag[slot], (unsigned long long)tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                            claimed = found = true;
                            new_groups++;
                        }
                        continue;  // lost the race: the slot is BUSY or published now, look again
                    }
                    if (cur == TSQ_BUSY_TAG) continue;
                    if (cur == tag && keys_at(slot)) {
                        found = true;
                        break;
                    }
                    slot = slot + 1 == a.t.cap ? 0 : slot + 1;
                    probe++;
                }
                if (!found) {  // the run is too long for this table: the host grows it and runs the row again
                    const uint32_t i = (uint32_t)atomicAdd(&a.counters[1], 1ull);
                    a.retry_out[i] = (uint32_t)row;
                    continue;
                }
            }
            agg_update_slot(a, slot, row, claimed);
        }
    }
    // one device atomic per workgroup: a same-address atomic per THREAD costs ~11 ns each, chip-wide (3e5 of them were
    // most of a 0.25 ms launch over 8e5 rows)
    block_add_u32(&a.counters[0], new_groups);
}


#include "tsq_streamagg.h"

// K7b — merge of LDS partial groups into the HBM group table.  Replaces HashAggFinalWorker.consumeIntermData
// (executor/aggregate.go:424-427, STUB; intended per courses/proj5-part3) + AggFunc.MergePartialResult
// (aggfuncs/func_count.go:51-55, func_sum.go:96-111, func_avg.go:86-113, func_max_min.go:60-79): one
// find-or-claim per partial group, then one or two device atomics per aggregate.
struct MergeArgs {
    AfPlan plan;
    AfPartials in;
    AggTable t;
    int64_t n;
    const uint32_t* retry_in;
    uint32_t* retry_out;
    unsigned long long* counters;  // [0]=new groups [1]=retry count
    uint64_t bail_after;           // as in AggArgs
};
// the words of partial group `rec` into the aggregates of slot `slot` (FIRSTROW is the caller's: it needs the key)
__device__ __forceinline__ void merge_apply(const AfPlan& plan, const AfPartials& in, const AggTable& t, uint64_t slot, uint32_t rec) {
    for (int i = 0; i < plan.n_aggs; i++) {
        const AfAgg f = plan.f[i];
        const AggState st = t.st[i];
        if (f.func == TSQ_AGG_FIRSTROW) continue;
        const unsigned long long w0 = in.w[f.w][rec];
        switch (f.func) {
            case TSQ_AGG_COUNT: atomicAdd(&st.acc[slot], w0); break;
            case TSQ_AGG_SUM:
            case TSQ_AGG_AVG:
                if (af_is_real(f.type)) {
                    atomicAdd((double*)&st.acc[slot], tsq_bits_f64(w0));
                    if (f.func == TSQ_AGG_AVG) atomicAdd(&st.cnt[slot], in.w[f.w + 1][rec]);
                } else {  // 128-bit add of (lo, hi)
                    const unsigned long long old = atomicAdd(&st.acc[slot], w0);
                    const unsigned long long hi = in.w[f.w + 1][rec] + ((old + w0 < old) ? 1ull : 0ull);
                    if (hi) atomicAdd(&st.aux[slot], hi);
                    if (f.func == TSQ_AGG_AVG) atomicAdd(&st.cnt[slot], in.w[f.w + 2][rec]);
                }
                if (f.func == TSQ_AGG_SUM) st.seen[slot] = 1;
                break;
            case TSQ_AGG_MAX:
                atomicMax(&st.acc[slot], w0);
                st.seen[slot] = 1;
                break;
            case TSQ_AGG_MIN:
                atomicMin(&st.acc[slot], w0);
                st.seen[slot] = 1;
                break;
        }
    }
}
__global__ void __launch_bounds__(256) k_agg_merge(MergeArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t new_groups = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.n; r += stride) {
        const uint32_t rec = a.retry_in ? a.retry_in[r] : (uint32_t)r;
        const unsigned long long tag = a.in.key[rec];
        uint64_t slot;
        bool winner = false;
        if (tag == TSQ_EMPTY_TAG) {  // the sentinel key word has its own slot (see file header)
            slot = a.t.cap;
            if (__hip_atomic_load(&a.t.tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == TSQ_EMPTY_TAG)
                winner = atomicCAS(&a.t.tag[slot], (unsigned long long)TSQ_EMPTY_TAG, 1ull) == TSQ_EMPTY_TAG;
        } else {
            slot = tsq_mulhi64(tsq_mix64(tag), a.t.cap);
            bool found = false;
            const bool hopeless = __hip_atomic_load(&a.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > a.bail_after;
            for (int probe = 0; probe < TSQ_AGG_PROBE_LIMIT && !hopeless; probe++) {
                unsigned long long cur = __hip_atomic_load(&a.t.tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == TSQ_EMPTY_TAG) {
                    cur = atomicCAS(&a.t.tag[slot], (unsigned long long)TSQ_EMPTY_TAG, tag);
                    if (cur == TSQ_EMPTY_TAG) { winner = true; found = true; break; }
                }
                if (cur == tag) { found = true; break; }
                slot = slot + 1 == a.t.cap ? 0 : slot + 1;
            }
            if (!found) {
                const uint32_t i = (uint32_t)atomicAdd(&a.counters[1], 1ull);
                a.retry_out[i] = rec;
                continue;
            }
        }
        if (winner) {
            new_groups++;
            a.t.gkey[0][slot] = tag;
        }
        if (winner) {  // firstrow(group key): any row of the group (func_first_row.go:67-81)
            for (int i = 0; i < a.plan.n_aggs; i++)
                if (a.plan.f[i].func == TSQ_AGG_FIRSTROW) {
                    a.t.st[i].acc[slot] = group_key_word_decode(tag, a.plan.key_type);
                    a.t.st[i].seen[slot] = 1;
                }
        }
        merge_apply(a.plan, a.in, a.t, slot, rec);
    }
    // one device atomic per workgroup: a same-address atomic per THREAD costs ~11 ns each, chip-wide (3e5 of them were
    // most of a 0.25 ms launch over 8e5 rows)
    block_add_u32(&a.counters[0], new_groups);
}

// K7c — merge of packed partial groups whose key is SEVERAL columns (tsq_daagg.h: key word = the fields d).  Same protocol as the
// several-column row upsert (k_agg_update<true>): phase 0 claims a slot by the 64-bit tag of the decoded cells and the claimer
// stores them, phase 1 (a later launch) compares the cells and merges the words — or walks on when another key owns the tag.
struct MergeMultiArgs {
    MergeArgs m;
    DaAggKeys ks;
    int32_t phase;
    uint32_t* slot_of;
    uint32_t tag_bits;
};
__global__ void __launch_bounds__(256) k_agg_merge_multi(MergeMultiArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t new_groups = 0;
    const int nk = a.ks.n;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.m.n; r += stride) {
        const uint32_t rec = a.m.retry_in ? a.m.retry_in[r] : (uint32_t)r;
        const uint32_t d = (uint32_t)a.m.in.key[rec];
        uint64_t kw[TSQ_DAAGG_MAXK];
        uint32_t nullmask = 0;
        uint64_t h = TSQ_AGG_TAG_SEED;
        for (int k = 0; k < nk; k++) {
            bool isn;
            kw[k] = daagg_field_cell(a.ks, d, k, &isn);
            nullmask |= isn ? (1u << k) : 0u;
            h = agg_tag_step(h, kw[k], isn);
        }
        const unsigned long long tag = agg_tag_finish(h, a.tag_bits);
        auto first_rows = [&](uint64_t sl) {  // firstrow(key column): the cell itself
            for (int i = 0; i < a.m.plan.n_aggs; i++) {
                const int fk = a.ks.fr_key[i];
                if (fk < 0) continue;
                a.m.t.st[i].acc[sl] = kw[fk];
                a.m.t.st[i].seen[sl] = ((nullmask >> fk) & 1u) ? 0 : 1;
            }
        };
        uint64_t slot;
        if (a.phase == 0) {
            slot = tsq_mulhi64(tsq_mix64(tag), a.m.t.cap);
            bool found = false, winner = false;
            const bool hopeless = __hip_atomic_load(&a.m.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > a.m.bail_after;
            for (int probe = 0; probe < TSQ_AGG_PROBE_LIMIT && !hopeless; probe++) {
                unsigned long long cur = __hip_atomic_load(&a.m.t.tag[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == TSQ_EMPTY_TAG) {
                    cur = atomicCAS(&a.m.t.tag[slot], (unsigned long long)TSQ_EMPTY_TAG, tag);
                    if (cur == TSQ_EMPTY_TAG) { winner = true; found = true; break; }
                }
                if (cur == tag) { found = true; break; }
                slot = slot + 1 == a.m.t.cap ? 0 : slot + 1;
            }
            if (!found) {
                const uint32_t i = (uint32_t)atomicAdd(&a.m.counters[1], 1ull);
                a.m.retry_out[i] = rec;
                a.slot_of[r] = 0xffffffffu;
                continue;
            }
            a.slot_of[r] = (uint32_t)slot;
            if (winner) {
                new_groups++;
                for (int k = 0; k < nk; k++) a.m.t.gkey[k][slot] = kw[k];
                a.m.t.gknull[slot] = (uint8_t)nullmask;
                first_rows(slot);
            }
            continue;
        }
        const uint32_t s32 = a.slot_of[r];
        if (s32 == 0xffffffffu) continue;  // handed back by phase 0
        slot = s32;
        auto keys_at = [&](uint64_t sl) -> bool {
            bool same = a.m.t.gknull[sl] == (uint8_t)nullmask;
            for (int k = 0; k < nk && same; k++) same = a.m.t.gkey[k][sl] == kw[k];
            return same;
        };
        if (!keys_at(slot)) {  // another key owns this tag: the BUSY protocol of k_agg_update<true>, phase 1
            atomicAdd(&a.m.counters[2], 1ull);
            bool found = false;
            int probe = 0;
            slot = slot + 1 == a.m.t.cap ? 0 : slot + 1;
            while (probe < TSQ_AGG_PROBE_LIMIT && !found) {
                unsigned long long cur = __hip_atomic_load(&a.m.t.tag[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == TSQ_EMPTY_TAG) {
                    cur = atomicCAS(&a.m.t.tag[slot], (unsigned long long)TSQ_EMPTY_TAG, (unsigned long long)TSQ_BUSY_TAG);
                    if (cur == TSQ_EMPTY_TAG) {
                        for (int k = 0; k < nk; k++) a.m.t.gkey[k][slot] = kw[k];
                        a.m.t.gknull[slot] = (uint8_t)nullmask;
                        first_rows(slot);
                        __hip_atomic_store(&a.m.t.tag[slot], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        found = true;
                        new_groups++;
                    }
                    continue;  // lost the race: the slot is BUSY or published now, look again
                }
                if (cur == TSQ_BUSY_TAG) continue;
                if (cur == tag && keys_at(slot)) {
                    found = true;
                    break;
                }
                slot = slot + 1 == a.m.t.cap ? 0 : slot + 1;
                probe++;
            }
            if (!found) {
                const uint32_t i = (uint32_t)atomicAdd(&a.m.counters[1], 1ull);
                a.m.retry_out[i] = rec;
                continue;
            }
        }
        merge_apply(a.m.plan, a.m.in, a.m.t, slot, rec);
    }
    block_add_u32(&a.m.counters[0], new_groups);
}

// re-insert every occupied slot of an old table into a bigger one (table growth)
struct RehashArgs {
    AggTable from, to;
    AggPlan plan;
    int32_t multi;
};
__global__ void __launch_bounds__(256) k_agg_rehash(RehashArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < a.from.cap + 2; s += stride) {
        const unsigned long long tag = a.from.tag[s];
        if (tag == TSQ_EMPTY_TAG) continue;
        uint64_t d;
        if (s >= a.from.cap) d = a.to.cap + (s - a.from.cap);
        else {
            d = tsq_mulhi64(tsq_mix64(tag), a.to.cap);
            for (;;) {  // distinct groups only: every tag is unique unless two keys collide (then both keep a slot)
                if (atomicCAS(&a.to.tag[d], (unsigned long long)TSQ_EMPTY_TAG, tag) == TSQ_EMPTY_TAG) break;
                d = d + 1 == a.to.cap ? 0 : d + 1;
            }
        }
        if (s >= a.from.cap) a.to.tag[d] = tag;
        for (int k = 0; k < a.plan.n_keys; k++) a.to.gkey[k][d] = a.from.gkey[k][s];
        if (a.from.gknull) a.to.gknull[d] = a.from.gknull[s];
        for (int i = 0; i < a.plan.n_aggs; i++) {
            a.to.st[i].acc[d] = a.from.st[i].acc[s];
            a.to.st[i].aux[d] = a.from.st[i].aux[s];
            a.to.st[i].cnt[d] = a.from.st[i].cnt[s];
            a.to.st[i].seen[d] = a.from.st[i].seen[s];
        }
    }
}

// initial values of the state arrays (MIN starts at all ones; everything else at zero)
__global__ void __launch_bounds__(256) k_fill_u64(unsigned long long* p, unsigned long long v, uint64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// K8 — finalise: compacts occupied slots into output columns.  Replaces consumeIntermData /
// getFinalResult (aggregate.go:424-457) and AggFunc.AppendFinalResult2Chunk of every function.
struct FinalArgs {
    AggTable t;
    AggPlan plan;
    int32_t key_type[TSQ_MAX_GROUP_KEYS];
    void* out_data[2 * TSQ_MAX_AGGS];
    uint8_t* out_notnull[2 * TSQ_MAX_AGGS];
    unsigned long long* counters;  // [3] = output cursor, [4] = overflow flag (BIGINT)
    uint64_t ordered_groups;       // StreamAggExec: slots [0, ordered_groups) are the groups in input order, group s -> output row s
    int32_t ordered;
};
__device__ __forceinline__ bool sum128_fits(unsigned long long lo, unsigned long long hi) {
    return hi == ((lo >> 63) ? ~0ull : 0ull);  // hi must be the sign extension of lo
}
// Output positions: a workgroup takes TSQ_FINAL_CHUNK consecutive slots, counts the occupied ones, reserves its output
// range with ONE device atomic and hands positions out from an LDS cursor.  (One returning device atomic per wave on the
// single cursor cost ~11 ns each, chip-wide: 1.3 ms for an 8 M-slot table.)
#define TSQ_FINAL_CHUNK 4096
__global__ void __launch_bounds__(256) k_agg_finalize(FinalArgs a) {
    __shared__ unsigned long long s_cnt, s_cur;
    const uint64_t nslots = a.ordered ? a.ordered_groups : a.t.cap + 2;
    const uint64_t nchunks = (nslots + TSQ_FINAL_CHUNK - 1) / TSQ_FINAL_CHUNK;
    const int lane = threadIdx.x & 63;
    for (uint64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const uint64_t lo = ch * TSQ_FINAL_CHUNK;
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < TSQ_FINAL_CHUNK; i += 256) mine += (lo + i < nslots && a.t.tag[lo + i] != TSQ_EMPTY_TAG) ? 1u : 0u;
        for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if (lane == 0 && mine) atomicAdd(&s_cnt, (unsigned long long)mine);
        __syncthreads();
        if (threadIdx.x == 0) s_cur = s_cnt ? atomicAdd(&a.counters[3], s_cnt) : 0ull;
        __syncthreads();
        if (s_cnt == 0) continue;  // block-uniform
      for (uint32_t i0 = 0; i0 < TSQ_FINAL_CHUNK; i0 += 256) {
        const uint64_t s = lo + i0 + threadIdx.x;
        const bool occ = s < nslots && a.t.tag[s] != TSQ_EMPTY_TAG;
        const unsigned long long m = __ballot(occ);
        if (!m) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&s_cur, (unsigned long long)__popcll(m));
        base = __shfl(base, 0, 64);
        if (!occ) continue;
        const uint64_t pos = a.ordered ? s : base + __popcll(m & ((1ull << lane) - 1));
        int oc = 0;
        for (int i = 0; i < a.plan.n_aggs; i++) {
            const tsq_agg_func f = a.plan.f[i];
            const AggState st = a.t.st[i];
            const bool partial_out = f.mode == TSQ_MODE_PARTIAL1 || f.mode == TSQ_MODE_PARTIAL2;
            const bool real = is_real_type(f.arg_type);
            switch (f.func) {
                case TSQ_AGG_COUNT:
                    ((uint64_t*)a.out_data[oc])[pos] = st.acc[s];
                    if (a.out_notnull[oc]) a.out_notnull[oc][pos] = 1;
                    oc++;
                    break;
                case TSQ_AGG_SUM: {
                    const bool nn = st.seen[s] != 0;
                    if (!real && nn && !sum128_fits(st.acc[s], st.aux[s])) atomicOr(&a.counters[4], 1ull);
                    ((uint64_t*)a.out_data[oc])[pos] = nn ? st.acc[s] : 0;
                    a.out_notnull[oc][pos] = nn ? 1 : 0;
                    oc++;
                    break;
                }
                case TSQ_AGG_AVG: {
                    const unsigned long long cnt = st.cnt[s];
                    if (!real && cnt && !sum128_fits(st.acc[s], st.aux[s])) atomicOr(&a.counters[4], 1ull);
                    if (partial_out) {  // (count, sum) columns (descriptor.go:70-81)
                        ((uint64_t*)a.out_data[oc])[pos] = cnt;
                        if (a.out_notnull[oc]) a.out_notnull[oc][pos] = 1;
                        oc++;
                        ((uint64_t*)a.out_data[oc])[pos] = st.acc[s];
                        if (a.out_notnull[oc]) a.out_notnull[oc][pos] = 1;
                        oc++;
                    } else {
                        uint64_t v = 0;
                        if (cnt) {
                            if (real) v = tsq_f64_bits(tsq_bits_f64(st.acc[s]) / (double)(long long)cnt);  // func_avg.go:154-162
                            else v = (uint64_t)tsq_godiv((int64_t)st.acc[s], (int64_t)cnt);                 // func_avg.go:47-55
                        }
                        ((uint64_t*)a.out_data[oc])[pos] = v;
                        a.out_notnull[oc][pos] = cnt ? 1 : 0;
                        oc++;
                    }
                    break;
                }
                case TSQ_AGG_MAX:
                case TSQ_AGG_MIN: {
                    const bool nn = st.seen[s] != 0;
                    const uint64_t v = !nn ? 0 : (f.arg_type == TSQ_BYTES ? st.acc[s] : ord_image_decode(st.acc[s], f.arg_type));
                    if (f.arg_type == TSQ_F32) ((uint32_t*)a.out_data[oc])[pos] = (uint32_t)v;
                    else ((uint64_t*)a.out_data[oc])[pos] = v;
                    a.out_notnull[oc][pos] = nn ? 1 : 0;
                    oc++;
                    break;
                }
                case TSQ_AGG_FIRSTROW: {
                    const bool nn = st.seen[s] != 0;
                    if (f.arg_type == TSQ_F32) ((uint32_t*)a.out_data[oc])[pos] = nn ? (uint32_t)st.acc[s] : 0u;
                    else ((uint64_t*)a.out_data[oc])[pos] = nn ? st.acc[s] : 0ull;
                    a.out_notnull[oc][pos] = nn ? 1 : 0;
                    oc++;
                    break;
                }
            }
        }
      }
    }
}

// K8a' — the rows of an aggregate whose every group lives in the DENSE partial state of the packed route (tsq_daagg.h: one cell per packed
// key word; no row took the exception path, nothing was merged into the hash table): straight from the cells.  Replaces k_daagg_dense_emit
// -> k_agg_merge -> k_agg_finalize (every group through a partial-group list and a random upsert before it is read back from the table:
// 1.2 of 3.3 ms for 4.3e6 groups).  A cell's words are the partial-group words of tsq_aggfast.h: COUNT = the count; SUM / AVG of integers =
// (sum of the low halves, sum of the high halves[, count]); of reals = (sum[, count]); MAX / MIN = the ordered image.  Every aggregate of
// a touched cell has seen a value: rows with a NULL argument cell are exception rows, and this kernel only runs when there were none.
struct DenseFinalArgs {
    AfPlan fplan;
    AggPlan plan;
    DaDomain dm;
    int32_t key_type;
    const unsigned long long* dense_w[TSQ_AF_MAXW];
    const uint32_t* dense_touch;
    // pg form (partitioned groups, tsq_aggfast.h K7p): the cells are the slots of the sub-tables, a slot is a group when its table word is
    // not TSQ_AF_EMPTY, and the word gives the key back (tsq_unmix64: mix64 is a bijection)
    const unsigned long long* pg_key;
    uint64_t ncells;
    void* out_data[2 * TSQ_MAX_AGGS];
    uint8_t* out_notnull[2 * TSQ_MAX_AGGS];
    unsigned long long* counters;  // [3] = output cursor, [4] = overflow flag (BIGINT)
};
__global__ void __launch_bounds__(256) k_dense_count(const uint32_t* touch, uint64_t nwords, unsigned long long* out) {
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * 256) c += (unsigned long long)__popc(touch[i]);
    c = wave_sum_u64(c);
    __shared__ unsigned long long s_c;  // one device atomic per workgroup (same-address device atomics cost ~11 ns each, chip-wide)
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0 && s_c) atomicAdd(out, s_c);
}
__global__ void __launch_bounds__(256) k_dense_finalize(DenseFinalArgs a) {
    __shared__ uint32_t s_cnt;
    __shared__ unsigned long long s_cur;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t nchunks = (a.ncells + TSQ_FINAL_CHUNK - 1) / TSQ_FINAL_CHUNK;
    for (uint64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const uint64_t lo = ch * TSQ_FINAL_CHUNK;
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        if (a.pg_key) {
            uint32_t c = 0;
            for (uint32_t i = threadIdx.x; i < TSQ_FINAL_CHUNK; i += 256) c += (lo + i < a.ncells && a.pg_key[lo + i] != TSQ_AF_EMPTY) ? 1u : 0u;
            c = (uint32_t)wave_sum_u64(c);
            if (lane == 0 && c) atomicAdd(&s_cnt, c);
        } else if (threadIdx.x < TSQ_FINAL_CHUNK / 32 && lo + 32ull * threadIdx.x < a.ncells) {
            const uint32_t c = (uint32_t)__popc(a.dense_touch[(lo >> 5) + threadIdx.x]);
            if (c) atomicAdd(&s_cnt, c);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cur = s_cnt ? atomicAdd(&a.counters[3], (unsigned long long)s_cnt) : 0ull;
        __syncthreads();
        if (s_cnt == 0) continue;  // (block-uniform)
        for (uint32_t i0 = 0; i0 < TSQ_FINAL_CHUNK; i0 += 256) {
            const uint64_t u = lo + i0 + threadIdx.x;
            const bool occ = u < a.ncells && (a.pg_key ? a.pg_key[u] != TSQ_AF_EMPTY : (bool)((a.dense_touch[u >> 5] >> (u & 31u)) & 1u));
            const unsigned long long m = __ballot(occ);
            if (!m) continue;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&s_cur, (unsigned long long)__popcll(m));
            base = __shfl(base, 0, 64);
            if (!occ) continue;
            const uint64_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
            int oc = 0;
            for (int i = 0; i < a.plan.n_aggs; i++) {
                const tsq_agg_func f = a.plan.f[i];
                const AfAgg g = a.fplan.f[i];
                const bool partial_out = f.mode == TSQ_MODE_PARTIAL1 || f.mode == TSQ_MODE_PARTIAL2;
                const bool real = is_real_type(f.arg_type);
                switch (f.func) {
                    case TSQ_AGG_COUNT:
                        ((uint64_t*)a.out_data[oc])[pos] = a.dense_w[g.w][u];
                        if (a.out_notnull[oc]) a.out_notnull[oc][pos] = 1;
                        oc++;
                        break;
                    case TSQ_AGG_SUM:
                    case TSQ_AGG_AVG: {
                        unsigned long long sum, cnt = 0;
                        if (real) {
                            sum = a.dense_w[g.w][u];
                            if (f.func == TSQ_AGG_AVG) cnt = a.dense_w[g.w + 1][u];
                        } else {  // (sum of low halves, sum of high halves) -> the 128-bit sum, as k_daagg_dense_emit splits it
                            const unsigned long long lo32 = a.dense_w[g.w][u], hi32 = a.dense_w[g.w + 1][u];
                            sum = (hi32 << 32) + lo32;
                            const unsigned long long hi = (unsigned long long)((long long)hi32 >> 32) + (sum < lo32 ? 1ull : 0ull);
                            if (!sum128_fits(sum, hi)) atomicOr(&a.counters[4], 1ull);
                            if (f.func == TSQ_AGG_AVG) cnt = a.dense_w[g.w + 2][u];
                        }
                        if (f.func == TSQ_AGG_SUM) {
                            ((uint64_t*)a.out_data[oc])[pos] = sum;
                            a.out_notnull[oc][pos] = 1;
                            oc++;
                        } else if (partial_out) {  // (count, sum) columns (descriptor.go:70-81)
                            ((uint64_t*)a.out_data[oc])[pos] = cnt;
                            if (a.out_notnull[oc]) a.out_notnull[oc][pos] = 1;
                            oc++;
                            ((uint64_t*)a.out_data[oc])[pos] = sum;
                            if (a.out_notnull[oc]) a.out_notnull[oc][pos] = 1;
                            oc++;
                        } else {
                            uint64_t v = 0;
                            if (cnt) {
                                if (real) v = tsq_f64_bits(tsq_bits_f64(sum) / (double)(long long)cnt);  // func_avg.go:154-162
                                else v = (uint64_t)tsq_godiv((int64_t)sum, (int64_t)cnt);                 // func_avg.go:47-55
                            }
                            ((uint64_t*)a.out_data[oc])[pos] = v;
                            a.out_notnull[oc][pos] = cnt ? 1 : 0;
                            oc++;
                        }
                        break;
                    }
                    case TSQ_AGG_MAX:
                    case TSQ_AGG_MIN: {
                        const uint64_t v = ord_image_decode(a.dense_w[g.w][u], f.arg_type);
                        if (f.arg_type == TSQ_F32) ((uint32_t*)a.out_data[oc])[pos] = (uint32_t)v;
                        else ((uint64_t*)a.out_data[oc])[pos] = v;
                        a.out_notnull[oc][pos] = 1;
                        oc++;
                        break;
                    }
                    case TSQ_AGG_FIRSTROW: {  // of the group key: the cell's word back to the key (tsq_da_unmix is the inverse of the packing mix)
                        const uint64_t key = a.pg_key ? tsq_unmix64(a.pg_key[u]) : a.dm.kmin + (uint64_t)tsq_da_unmix((uint32_t)u, a.dm.s, a.dm.mask);
                        ((uint64_t*)a.out_data[oc])[pos] = group_key_word_decode(key, a.key_type);
                        a.out_notnull[oc][pos] = 1;
                        oc++;
                        break;
                    }
                }
            }
        }
    }
}

// K8b — var-len output columns (AppendFinalResult2Chunk of firstRow4String / maxMin4String: chk.AppendString): the
// finalize pass left references; their lengths (k_ref_len), the exclusive scan of the lengths = the column's offsets
// (tsq_launch_scan64), then the bytes out of the heap (k_ref_copy: one row per lane, or per wave for long cells).
struct RefOutArgs {
    const unsigned long long* refs;
    const uint8_t* notnull;  // one byte per row
    int64_t rows;
    const uint8_t* heap;
    int64_t* out_offs;       // [rows + 1]
    uint8_t* out_data;
    const uint8_t* heap_child;  // rows [child_from, rows): references into this buffer (the dictionary of group keys, tsq_keydict.h)
    int64_t child_from;         // (rows: no such rows)
};
__global__ void __launch_bounds__(256) k_ref_len(RefOutArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.rows; r += (int64_t)gridDim.x * blockDim.x)
        a.out_offs[r] = a.notnull[r] ? (int64_t)ref_len(a.refs[r]) : 0;  // a NULL cell has no bytes
}
template <bool WAVE>
__global__ void __launch_bounds__(256) k_ref_copy(RefOutArgs a) {
    const int lane = WAVE ? (threadIdx.x & 63) : 0, step = WAVE ? 64 : 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = WAVE ? (t >> 6) : t; r < a.rows; r += WAVE ? (nt >> 6) : nt) {
        if (!a.notnull[r]) continue;
        const uint8_t* s = (r >= a.child_from ? a.heap_child : a.heap) + ref_off(a.refs[r]);
        uint8_t* d = a.out_data + a.out_offs[r];
        const int64_t n = a.out_offs[r + 1] - a.out_offs[r];
        if (WAVE) {
            for (int64_t i = lane; i < n; i += step) d[i] = s[i];
            continue;
        }
        // one row per lane: the cell's bytes are FETCHED first, 32 at a time as four words (a byte loop is a chain of dependent
        // round trips — the stores between the loads may alias them: 0.86 ms for 5e6 cells of ~9 bytes, round 6), then stored
        for (int64_t i0 = 0; i0 < n; i0 += 32) {
            const uint32_t m = n - i0 < 32 ? (uint32_t)(n - i0) : 32u;
            uint64_t w[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) w[q] = q * 8u < m ? kr_load8(s + i0 + q * 8u, m - q * 8u < 8u ? m - q * 8u : 8u) : 0ull;
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
#pragma unroll
                for (uint32_t b = 0; b < 8; b++)
                    if (q * 8u + b < m) d[i0 + q * 8u + b] = (uint8_t)(w[q] >> (8u * b));
            }
        }
    }
}

// ---- compaction of a string heap: the operator appends every pushed var-len cell to the column's heap, but only the
// cells the group table REFERS to are needed afterwards — group keys, FIRST_ROW / MAX / MIN values.  Between two batches the
// referenced strings are copied into a fresh heap and the references rewritten, so the heap grows with the groups, not with the
// input (the reference keeps one copy per group: stringutil.Copy in its partial results).
#define TSQ_HEAP_GC_MAXF 20
struct HeapGcArgs {
    const unsigned long long* tag;
    uint64_t nslots;                          // cap + 2
    int32_t n_fields;
    unsigned long long* ref[TSQ_HEAP_GC_MAXF];  // the field's reference array (gkey[k] or acc[i]), rewritten in place
    const uint8_t* seen[TSQ_HEAP_GC_MAXF];      // aggregate value: its seen[] flags; group key: nullptr
    int32_t nullbit[TSQ_HEAP_GC_MAXF];          // group key k: bit k of gknull[s] says NULL; aggregate value: -1
    const uint8_t* gknull;
    const uint8_t* heap;
    uint8_t* new_heap;
    int64_t* offs;                            // [n_fields * nslots + 1]: lengths, then (scanned) new offsets
};
__device__ __forceinline__ bool heap_gc_live(const HeapGcArgs& a, int f, uint64_t s) {
    if (a.tag[s] == TSQ_EMPTY_TAG) return false;
    if (a.nullbit[f] >= 0) return !a.gknull || !((a.gknull[s] >> a.nullbit[f]) & 1u);
    return a.seen[f][s] != 0 && a.ref[f][s] != TSQ_REF_NONE;
}
__global__ void __launch_bounds__(256) k_heap_gc_len(HeapGcArgs a) {
    const uint64_t n = a.nslots * (uint64_t)a.n_fields;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const int f = (int)(i / a.nslots);
        const uint64_t s = i % a.nslots;
        a.offs[i] = heap_gc_live(a, f, s) ? (int64_t)ref_len(a.ref[f][s]) : 0;
    }
}
__global__ void __launch_bounds__(256) k_heap_gc_move(HeapGcArgs a) {
    const uint64_t n = a.nslots * (uint64_t)a.n_fields;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const int f = (int)(i / a.nslots);
        const uint64_t s = i % a.nslots;
        if (!heap_gc_live(a, f, s)) continue;
        const unsigned long long r = a.ref[f][s];
        const uint64_t len = ref_len(r), to = (uint64_t)a.offs[i];
        const uint8_t* src = a.heap + ref_off(r);
        uint8_t* dst = a.new_heap + to;
        for (uint64_t b = 0; b < len; b++) dst[b] = src[b];
        a.ref[f][s] = (to << TSQ_REF_LEN_BITS) | len;
    }
}

// ====================================================================== host side
// ---- several integer key columns wider than the packed route's 23 bits: ONE 64-bit composite key (round 4)
// GROUP BY l_orderkey, o_orderdate, o_shippriority (Q3: 28 + 12 + 2 bits) used to take the several-column row upsert: a 64-bit TAG of
// the cells in phase 0, the cells compared in a second launch, four FIRST_ROW state arrays — ~9 random lines per row.  When the
// fields of the key columns ([kmin_k, kmin_k + 2^w_k), one more code for NULL, found by the first batch and widened while the sum
// stays <= 63 bits) hold a row's cells, the row's group IS the number d = sum field_k << shift_k — equal d <=> equal cells in every
// column, NULL = NULL (aggregate.go:359-394 / codec.go:713-746: the group key is the concatenation of the encoded cells) — and the
// operator hands (d, the argument columns) to a CHILD aggregate with ONE BIGINT UNSIGNED key: the single-key upsert (one launch, the
// tag is the key), its LDS pre-aggregation when groups repeat, no FIRST_ROW state for the key columns at all (their values are
// decoded from d when the groups are emitted).  A row with a cell outside its field is an EXCEPTION: it takes the several-column
// upsert into this operator's own table; a group lives in exactly one of the two tables (its cells decide), so the result is the
// child's groups followed by the own ones.
#define TSQ_WIDE_NO_NULL (~0ull)
struct WideFields {
    int32_t n;
    uint64_t kmin[TSQ_MAX_GROUP_KEYS], maxd[TSQ_MAX_GROUP_KEYS], nullcode[TSQ_MAX_GROUP_KEYS];
    uint32_t shift[TSQ_MAX_GROUP_KEYS], width[TSQ_MAX_GROUP_KEYS];
};
struct WideComposeArgs {
    WideFields f;
    const uint64_t* col[TSQ_MAX_GROUP_KEYS];
    const uint8_t* nulls[TSQ_MAX_GROUP_KEYS];
    int64_t nrows;
    uint64_t* d;                    // composite per row (exception rows: ~0, never read by the child)
    unsigned long long* counts;     // [0] += exception rows
};
__device__ __forceinline__ uint64_t wide_compose_row(const WideComposeArgs& a, int64_t row, bool* ok) {
    uint64_t d = 0;
    bool good = true;
#pragma unroll
    for (int k = 0; k < TSQ_MAX_GROUP_KEYS; k++) {
        if (k < a.f.n) {
            uint64_t fld;
            if (tsq_is_null(a.nulls[k], row)) {
                fld = a.f.nullcode[k];
                good = good && fld != TSQ_WIDE_NO_NULL;
            } else {
                fld = a.col[k][row] - a.f.kmin[k];
                good = good && fld <= a.f.maxd[k];
            }
            d |= fld << a.f.shift[k];
        }
    }
    *ok = good;
    return good ? d : ~0ull;
}
__global__ void __launch_bounds__(256) k_agg_wide_compose(WideComposeArgs a) {
    uint32_t exc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.nrows; i += (int64_t)gridDim.x * 256) {
        bool ok;
        a.d[i] = wide_compose_row(a, i, &ok);
        exc += ok ? 0u : 1u;
    }
    block_add_u32(&a.counts[0], exc);
}
// the two row lists of a batch that holds exception rows (any order): one device atomic per workgroup and list
struct WideListArgs {
    const uint64_t* d;
    int64_t nrows;
    uint32_t* ok_rows;
    uint32_t* exc_rows;
    unsigned long long* cursors;  // [0] ok, [1] exceptions
};
__global__ void __launch_bounds__(256) k_agg_wide_lists(WideListArgs a) {
    __shared__ uint32_t s_base[2], s_cnt[2];
    for (int64_t base = (int64_t)blockIdx.x * 256; base < a.nrows; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        const bool in = i < a.nrows, exc = in && a.d[i] == ~0ull, ok = in && !exc;
        if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t mo = __ballot(ok), me = __ballot(exc);
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t wo = 0, we = 0;
        if (lane == 0) {
            wo = atomicAdd(&s_cnt[0], (uint32_t)__popcll(mo));
            we = atomicAdd(&s_cnt[1], (uint32_t)__popcll(me));
        }
        wo = __shfl(wo, 0, 64);
        we = __shfl(we, 0, 64);
        __syncthreads();
        if (threadIdx.x < 2 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = (uint32_t)atomicAdd(&a.cursors[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
        __syncthreads();
        const uint64_t below = (1ull << lane) - 1ull;
        if (ok) a.ok_rows[s_base[0] + wo + (uint32_t)__popcll(mo & below)] = (uint32_t)i;
        if (exc) a.exc_rows[s_base[1] + we + (uint32_t)__popcll(me & below)] = (uint32_t)i;
        __syncthreads();
    }
}
// the key columns of the child's groups, decoded from their composite
struct WideDecodeArgs {
    WideFields f;
    const uint64_t* d;
    int64_t n;
    int32_t n_out;
    int32_t key_of[TSQ_MAX_AGGS];        // output column -> key column it is FIRST_ROW of
    uint64_t* out[TSQ_MAX_AGGS];
    uint8_t* out_nn[TSQ_MAX_AGGS];
};
__global__ void __launch_bounds__(256) k_agg_wide_decode(WideDecodeArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const uint64_t d = a.d[i];
        for (int o = 0; o < a.n_out; o++) {
            const int k = a.key_of[o];
            const uint64_t fld = a.f.width[k] >= 64 ? d : ((d >> a.f.shift[k]) & ((1ull << a.f.width[k]) - 1ull));
            const bool isnull = a.f.nullcode[k] != TSQ_WIDE_NO_NULL && fld == a.f.nullcode[k];
            a.out[o][i] = isnull ? 0ull : a.f.kmin[k] + fld;
            a.out_nn[o][i] = isnull ? 0 : 1;
        }
    }
}

struct AggTableBufs {
    DevBuf tag, gknull;
    DevBuf gkey[TSQ_MAX_GROUP_KEYS];
    DevBuf acc[TSQ_MAX_AGGS], aux[TSQ_MAX_AGGS], cnt[TSQ_MAX_AGGS], seen[TSQ_MAX_AGGS];
    uint64_t cap = 0;
    void release() {
        tag.release();
        gknull.release();
        for (auto& b : gkey) b.release();
        for (int i = 0; i < TSQ_MAX_AGGS; i++) { acc[i].release(); aux[i].release(); cnt[i].release(); seen[i].release(); }
        cap = 0;
    }
};

struct tsq_agg {
    tsq_handle_hdr hdr;
    tsq_ctx* ctx = nullptr;
    tsq_agg_cfg cfg;
    AggPlan plan{};
    bool multi = false;
    std::atomic<int> cancelled{0};
    AggTableBufs tb;
    int64_t groups = 0;
    DevBuf counters, retry[2];
    HostStage stage;
    std::vector<ColStore> icols;  // device batch for host pushes
    // var-len input columns: the bytes of every pushed cell stay here until the operator is destroyed (file header); a batch
    // starts at a heap row that is a multiple of 8, so its null bitmap starts on a byte
    std::vector<ColStore> heap;
    bool has_str = false;
    std::vector<int64_t> heap_gc_at;  // per var-len input column: compact its heap when it holds more bytes than this
    int64_t heap_gcs = 0;
    uint32_t test_tag_bits = 0;  // TSQ_AGG_TAG_BITS (tests): truncated multi-key tags, so that distinct keys share a tag
    bool finished = false;
    int64_t in_rows = 0;
    // output
    int n_out = 0;
    std::vector<int32_t> out_types;
    std::vector<DevBuf> odata, onn, obitmap;
    std::vector<PinnedBuf> hdata, hbitmap;
    std::vector<int32_t> out_arg_col;   // var-len output column -> the input column whose heap its references point into
    std::vector<DevBuf> ooffs, obytes;  // var-len output columns: offsets[out_rows + 1] and the bytes (odata holds the references)
    std::vector<int64_t> onbytes;
    std::vector<PinnedBuf> hoffs;
    int64_t out_rows = 0, out_cursor = 0;
    bool out_on_host = false, host_mode = true;
    tsq_stats st{};
    // LDS pre-aggregation path (tsq_aggfast.h)
    int32_t fast_mode = TSQ_AGGFAST_AUTO;
    bool fast_ok = false;       // the plan is expressible in LDS words
    AfPlan fplan{};
    DevBuf slot_of;             // multi key: phase 0 -> phase 1 slot numbers
    DevBuf fkey, fw[TSQ_AF_MAXW], fctl, fexc;      // partial groups | counters (partials, exceptions) | exception row ids
    DevBuf rkeys, rpay[TSQ_RADIX_MAXV], rctl, rvend, rokeys, ropay[TSQ_RADIX_MAXV];  // partitioned rows (H mode)
    // the dense packed route on two streams (round 6): the partitioned store exists twice — k_agg_da / k_daagg_ovf of batch i read one
    // on `side` while the partition pass of batch i + 1 fills the other on the context's stream.  side_ev[s]: store s has been read
    DevBuf r2keys, r2pay[TSQ_RADIX_MAXV], r2ctl, r2vend, r2okeys, r2opay[TSQ_RADIX_MAXV];
    hipStream_t side = nullptr;
    hipEvent_t side_part = nullptr, side_ev[2] = {nullptr, nullptr};
    bool side_busy[2] = {false, false};
    int side_state = 0;         // 0: not tried, 1: stream and events exist, -1: not usable
    int side_turn = 0;
    int64_t side_batches = 0;
    int64_t fast_batches = 0, fast_fallbacks = 0;
    // partitioned groups (tsq_aggfast.h K7p): about as many groups as rows — the group table is a set of LDS-sized sub-tables in HBM
    int pg_state = 0;            // 0: not tried, 1: in use, -1: not usable
    uint32_t pg_pbits = 0, pg_sbits = 0;
    int64_t pg_rows = 0, pg_batches = 0;
    DevBuf pg_key, pg_w[TSQ_AF_MAXW], pg_used;
    // every batch rewrites all the sub-tables (load, apply, store): batches much smaller than the state wait in these columns (the key
    // column and the argument columns, with their null bitmaps) until they are worth a pass — or until the end
    std::vector<ColStore> pg_pend;
    int64_t pg_pend_rows = 0;
    bool pg_in_flush = false;
    // packed-key pre-aggregation (tsq_daagg.h): the key range the first large batch showed
    int da_state = 0;  // 0: not tried, 1: in use, -1: not usable (range too wide, float key, too many rows outside the range)
    DaDomain da_dm{};
    uint32_t da_pbits = 0, da_ebits = 0;
    int64_t packed_batches = 0;
    uint32_t da_paybytes = 8;   // width of the one travelling argument column in the partitioned store (8: as it is)
    // dense partial state of the one-key packed route (tsq_daagg.h): words [k][u] + touch bits, folded into the table at finish
    int dense_state = 0;        // 0: not tried, 1: in use, -1: not usable
    DevBuf dense_w[TSQ_AF_MAXW], dense_touch;
    int64_t dense_rows = 0;     // rows that went into the state since it was last emptied
    int64_t dense_flushes = 0;
    // several integer key columns as the fields of one packed word (tsq_daagg.h): mk_n > 1.  da_low: the word fits one LDS table
    int mk_n = 0;
    int32_t mk_col[TSQ_DAAGG_MAXK] = {0, 0, 0, 0};
    DaAggKeys da_keys{};
    bool da_low = false;
    // several integer key columns as ONE 64-bit composite key handed to a child aggregate (see k_agg_wide_compose)
    int wide_state = 0;        // 0: not tried, 1: in use, -1: not usable (this plan, or the fields of the first batch exceed 63 bits)
    bool wide_ok = false;      // the plan allows it (integer key columns, fixed-width inputs)
    bool is_wide_child = false;
    bool da_hint = false;      // packed route: the key range is given (unsigned keys in [da_hint_min, da_hint_max]) instead of sampled alone
    uint64_t da_hint_min = 0, da_hint_max = 0;
    WideFields wide_f{};
    tsq_agg* wide = nullptr;   // the child: GROUP BY d
    DevBuf wide_d, wide_okrows, wide_excrows;
    int32_t wide_child_out[TSQ_MAX_AGGS * 2];  // own output column -> the child's output column, or -1 - k: decoded from d (key column k)
    int64_t wide_batches = 0, wide_exception_rows = 0;
    // string keys / several key columns through a DICTIONARY of group keys (tsq_keydict.h): the child (`wide`, wide_state, wide_child_out and the
    // row lists above are shared with the composite-key route) groups by the dense id of a row's key record
    bool kd_ok = false;        // the plan allows it
    bool has_str_key = false;  // some group key column is a string
    bool kd_mode = false;      // the child's key is a dictionary id (wide_state == 1)
    uint32_t kd_pbits = 0;
    int32_t kd_npay = 0, kd_paycol[TSQ_KR_MAXPAY] = {0, 0, 0, 0};
    DevBuf kd_counts, kd_pstart, kd_rec, kd_ids, kd_flags, kd_pay[TSQ_KR_MAXPAY], kd_paynn, kd_paybm[TSQ_KR_MAXPAY], kd_norec, kd_ridx, kd_gid;
    DevBuf kd_drec, kd_dids, kd_dcount, kd_dloc, kd_ctl;  // the dictionary; kd_ctl: [0] ids handed out, [1] exception rows of the batch, [2] rows without a record, [3..4] list cursors
    int64_t kd_next_host = 0;  // ids handed out, as of the last batch
    // StreamAggExec (tsq_streamagg.h): the input arrives ordered by the group keys, the table is an ARRAY of groups in input order
    bool stream = false;
    DevBuf sa_cnt;             // heads per 2048-row chunk of the current batch
    DevBuf hotkeys;            // packed aggregate: the sampled hot keys of the current batch (tsq_daagg.h DaAggHot)
    int64_t hot_age = 0;       // batches since the operator started: the hot keys are sampled every fourth
    int64_t stream_batches = 0;
};

namespace {

tsq_status agg_cancelled(tsq_agg* a) {
    if (a->cancelled.load()) return tsq_fail(&a->hdr, TSQ_ERR_CANCELLED, "aggregate cancelled");
    return TSQ_OK;
}

void fill_agg_table(const tsq_agg* a, const AggTableBufs& b, AggTable& t) {
    memset(&t, 0, sizeof t);
    t.tag = b.tag.as<unsigned long long>();
    for (int k = 0; k < a->plan.n_keys; k++) t.gkey[k] = b.gkey[k].as<unsigned long long>();
    t.gknull = (a->multi || a->stream) ? b.gknull.as<uint8_t>() : nullptr;
    for (int i = 0; i < a->plan.n_aggs; i++) {
        t.st[i].acc = b.acc[i].as<unsigned long long>();
        t.st[i].aux = b.aux[i].as<unsigned long long>();
        t.st[i].cnt = b.cnt[i].as<unsigned long long>();
        t.st[i].seen = b.seen[i].as<uint8_t>();
    }
    t.cap = b.cap;
}

tsq_status alloc_table(tsq_agg* a, AggTableBufs& b, uint64_t cap) {
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const uint64_t n = cap + 2;
    b.cap = cap;
    TSQ_TRY(b.tag.reserve(ctx, h, n * 8));
    TSQ_HIP(h, hipMemsetAsync(b.tag.p, 0x80, n * 8, ctx->stream));
    for (int k = 0; k < a->plan.n_keys; k++) TSQ_TRY(b.gkey[k].reserve(ctx, h, n * 8));
    if (a->multi || a->stream) {
        TSQ_TRY(b.gknull.reserve(ctx, h, n));
        TSQ_HIP(h, hipMemsetAsync(b.gknull.p, 0, n, ctx->stream));
    }
    for (int i = 0; i < a->plan.n_aggs; i++) {
        TSQ_TRY(b.acc[i].reserve(ctx, h, n * 8));
        TSQ_TRY(b.aux[i].reserve(ctx, h, n * 8));
        TSQ_TRY(b.cnt[i].reserve(ctx, h, n * 8));
        TSQ_TRY(b.seen[i].reserve(ctx, h, n));
        if (a->plan.f[i].func == TSQ_AGG_MIN || (a->plan.f[i].func == TSQ_AGG_MAX && a->plan.f[i].arg_type == TSQ_BYTES)) {
            int grid = tsq_grid_for(ctx, (int64_t)n, 256);
            hipLaunchKernelGGL(k_fill_u64, dim3(grid), dim3(256), 0, ctx->stream, b.acc[i].as<unsigned long long>(), ~0ull, n);
            TSQ_HIP(h, hipGetLastError());
        } else {
            TSQ_HIP(h, hipMemsetAsync(b.acc[i].p, 0, n * 8, ctx->stream));
        }
        TSQ_HIP(h, hipMemsetAsync(b.aux[i].p, 0, n * 8, ctx->stream));
        TSQ_HIP(h, hipMemsetAsync(b.cnt[i].p, 0, n * 8, ctx->stream));
        TSQ_HIP(h, hipMemsetAsync(b.seen[i].p, 0, n, ctx->stream));
    }
    return TSQ_OK;
}

tsq_status grow_table(tsq_agg* a, uint64_t new_cap) {
    tsq_ctx* ctx = a->ctx;
    AggTableBufs nb;
    tsq_status s = alloc_table(a, nb, new_cap);
    if (s != TSQ_OK) { nb.release(); return s; }
    RehashArgs ra;
    memset(&ra, 0, sizeof ra);
    fill_agg_table(a, a->tb, ra.from);
    fill_agg_table(a, nb, ra.to);
    ra.plan = a->plan;
    ra.multi = a->multi;
    int grid = tsq_grid_for(ctx, (int64_t)a->tb.cap + 2, 256);
    hipLaunchKernelGGL(k_agg_rehash, dim3(grid), dim3(256), 0, ctx->stream, ra);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { nb.release(); return tsq_fail(&a->hdr, TSQ_ERR_HIP, std::string("rehash: ") + hipGetErrorString(e)); }
    a->tb.release();
    a->tb = nb;  // shallow move of the buffer handles
    a->st.kernel_launches++;
    return TSQ_OK;
}

tsq_status launch_update(tsq_agg* a, AggArgs& args) {
    int grid = tsq_grid_for(a->ctx, args.nrows, 256);
    if (a->multi) hipLaunchKernelGGL(k_agg_update<true>, dim3(grid), dim3(256), 0, a->ctx->stream, args);
    else hipLaunchKernelGGL(k_agg_update<false>, dim3(grid), dim3(256), 0, a->ctx->stream, args);
    TSQ_HIP(&a->hdr, hipGetLastError());
    a->st.kernel_launches++;
    return TSQ_OK;
}

// find-or-claim loop shared by the row upsert and the partial-group merge: run `launch` over n items,
// grow the table and re-run it on the items it handed back until none is left.
template <class Launch>
tsq_status upsert_loop(tsq_agg* a, int64_t n0, const uint32_t* retry_in0, Launch&& launch) {
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    if (n0 == 0) return TSQ_OK;
    // keep the load factor <= 0.5 for the groups known so far; items that still find no slot are retried
    if ((uint64_t)a->groups * 2 > a->tb.cap) TSQ_TRY(grow_table(a, a->tb.cap * 4));
    // first batch into an empty table of unknown cardinality: room for every item of the batch being a new group, up to
    // 4 Mi slots — instead of one launch that hands nearly everything back and a growth step
    if (a->groups == 0 && (uint64_t)n0 * 2 > a->tb.cap && a->tb.cap < (1u << 22)) {
        uint64_t want = a->tb.cap;
        while (want < (uint64_t)n0 * 2 && want < (1u << 22)) want <<= 1;
        TSQ_TRY(grow_table(a, want));
    }
    TSQ_TRY(a->retry[0].reserve(ctx, h, (size_t)n0 * 4 + 16));
    TSQ_TRY(a->retry[1].reserve(ctx, h, (size_t)n0 * 4 + 16));
    unsigned long long* counters = a->counters.as<unsigned long long>();
    const uint32_t* retry_in = retry_in0;
    int64_t n = n0;
    int which = 0;
    for (int round = 0; round < 40; round++) {
        TSQ_TRY(agg_cancelled(a));
        AggTable t;
        fill_agg_table(a, a->tb, t);
        TSQ_HIP(h, hipMemsetAsync(counters, 0, 3 * 8, ctx->stream));
        TSQ_TRY(launch(t, n, retry_in, a->retry[which].as<uint32_t>()));
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned, counters, 3 * 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        a->groups += (int64_t)ctx->pinned[0];
        const uint64_t n_retry = ctx->pinned[1];
        a->st.build_handed_back_rows += (int64_t)ctx->pinned[2];  // rows whose 64-bit tag was shared by another key (resolved in phase 1)
        if (n_retry == 0) return TSQ_OK;
        // every item handed back may be a new group: size for that, within x4 .. x64 of the current table
        // (rounded to a power of two, so that a query that runs again finds its table arrays in the context pool)
        uint64_t want = 1;
        while (want < ((uint64_t)a->groups + n_retry) * 2) want <<= 1;
        TSQ_TRY(grow_table(a, std::max<uint64_t>(a->tb.cap * 4, std::min<uint64_t>(a->tb.cap * 64, want))));
        retry_in = a->retry[which].as<uint32_t>();
        n = (int64_t)n_retry;
        which ^= 1;
    }
    return tsq_fail(h, TSQ_ERR_HIP, "aggregate: table growth did not converge");
}

// rows [0, nrows) of `in` (or only the rows listed in rows_in) through the row-at-a-time upsert
tsq_status agg_rows(tsq_agg* a, const tsq_colset& in, int64_t nrows, const uint32_t* rows_in) {
    if (nrows == 0) return TSQ_OK;
    if (nrows >= 0xffffffffLL) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "internal: batch too large");
    AggArgs args;
    memset(&args, 0, sizeof args);
    args.in = in;
    args.plan = a->plan;
    args.row_base = a->in_rows;
    args.counters = a->counters.as<unsigned long long>();
    args.tag_bits = a->test_tag_bits;
    if (a->multi) {
        TSQ_TRY(a->slot_of.reserve(a->ctx, &a->hdr, (size_t)nrows * 4 + 16));
        args.slot_of = a->slot_of.as<uint32_t>();
    }
    return upsert_loop(a, nrows, rows_in, [&](const AggTable& t, int64_t n, const uint32_t* retry_in, uint32_t* retry_out) -> tsq_status {
        if (a->multi && t.cap + 2 >= 0xffffffffULL) return tsq_fail(&a->hdr, TSQ_ERR_UNSUPPORTED, "multi-key aggregate: more than 2^32 group slots");
        args.t = t;
        args.bail_after = std::max<uint64_t>(1024, t.cap / 16);
        args.nrows = n;
        args.retry_in = retry_in;
        args.retry_out = retry_out;
        args.phase = 0;
        TSQ_TRY(launch_update(a, args));
        if (a->multi) {  // phase 1 only for rows that found a slot; rows handed back are retried as a whole
            args.phase = 1;
            TSQ_TRY(launch_update(a, args));
        }
        return TSQ_OK;
    });
}

// ---------------------------------------------------------------- LDS pre-aggregation (host side)
uint32_t af_slots(const AfPlan& p) { return p.W <= 3 ? 4096u : 2048u; }

template <int MODE>
tsq_status launch_lds(tsq_agg* a, AfLdsArgs& la, int grid) {
    hipStream_t st = a->ctx->stream;
    switch (la.plan.W) {
        case 1: hipLaunchKernelGGL((k_agg_lds<MODE, 1>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        case 2: hipLaunchKernelGGL((k_agg_lds<MODE, 2>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        case 3: hipLaunchKernelGGL((k_agg_lds<MODE, 3>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        case 4: hipLaunchKernelGGL((k_agg_lds<MODE, 4>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        default: hipLaunchKernelGGL((k_agg_lds<MODE, 5>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
    }
    TSQ_HIP(&a->hdr, hipGetLastError());
    a->st.kernel_launches++;
    return TSQ_OK;
}

// ---- packed-key H mode (tsq_daagg.h)
tsq_status launch_agg_da(tsq_agg* a, DaAggLdsArgs& la, int grid, hipStream_t st) {
    if (la.plan.W <= 3 && la.st.ebits <= 11) {  // half-size tables: two workgroups per CU
        switch (la.plan.W) {
            case 1: hipLaunchKernelGGL((k_agg_da<1, 2048>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
            case 2: hipLaunchKernelGGL((k_agg_da<2, 2048>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
            default: hipLaunchKernelGGL((k_agg_da<3, 2048>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        }
        TSQ_HIP(&a->hdr, hipGetLastError());
        a->st.kernel_launches++;
        return TSQ_OK;
    }
    {   // the two commonest plans: update descriptors as compile-time constants (tsq_daagg.h, SIG)
        const uint32_t* wd = la.plan.wdesc;
        // (default 2 since round 6: k_agg_da is bound by the instructions it issues — one LDS atomic per row instead of two: C3 5.98 -> 5.64 ms,
        // the Zipf variant 15.0 -> 12.6 ms; round 4 measured 2 % and left it off)
        const int64_t sig_knob = tsq_knob(a->ctx, TSQ_KNOB_DAAGG_SIG, 2);
        const bool sig_on = sig_knob != 0;
        int sig = 0;
        if (sig_on && la.plan.W == 3 && wd[0] == af_wdesc(AF_W_ADD_LO32, 0, TSQ_I64) && wd[1] == af_wdesc(AF_W_ADD_HI32, 0, TSQ_I64) && wd[2] == af_wdesc(AF_W_ADD1, 0, 0)) sig = 1;
        if (sig_on && la.plan.W == 2 && wd[0] == af_wdesc(AF_W_ADD_REAL, 0, TSQ_F64) && wd[1] == af_wdesc(AF_W_ADD1, 0, 0)) sig = 2;
        // SUM(BIGINT) + COUNT(*) over 2-byte argument cells into the dense state, fewer than 2^24 rows per partition: count and sum share one LDS word
        if (sig == 1 && sig_knob >= 2 && la.dense_touch != nullptr && la.st.paybytes == 2 && (uint64_t)la.st.cap * 8 < ((uint64_t)1 << (64 - TSQ_DAAGG_PACK_SHIFT))) sig = 3;
        if (sig) {
            if (sig == 3) hipLaunchKernelGGL((k_agg_da<3, 4096, 3>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la);
            else if (sig == 1) hipLaunchKernelGGL((k_agg_da<3, 4096, 1>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la);
            else hipLaunchKernelGGL((k_agg_da<2, 4096, 2>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la);
            TSQ_HIP(&a->hdr, hipGetLastError());
            a->st.kernel_launches++;
            return TSQ_OK;
        }
    }
    switch (la.plan.W) {  // W <= 3: 4096 cells per partition (96 KB of LDS words), else 2048
        case 1: hipLaunchKernelGGL((k_agg_da<1, 4096>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        case 2: hipLaunchKernelGGL((k_agg_da<2, 4096>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        case 3: hipLaunchKernelGGL((k_agg_da<3, 4096>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        case 4: hipLaunchKernelGGL((k_agg_da<4, 2048>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
        default: hipLaunchKernelGGL((k_agg_da<5, 2048>), dim3(grid), dim3(TSQ_AF_NT), 0, st, la); break;
    }
    TSQ_HIP(&a->hdr, hipGetLastError());
    a->st.kernel_launches++;
    return TSQ_OK;
}
// several key columns: the range of every column in the first large batch gives its field (a nullable column gets one more code
// for NULL); the fields must fit TSQ_DAAGG_MAX_BITS together
tsq_status da_agg_setup_multi(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const AfPlan& pl = a->fplan;
    DaAggKeys& ks = a->da_keys;
    ks.n = a->mk_n;
    uint32_t total = 0;
    for (int k = 0; k < a->mk_n; k++) {
        const int c = a->mk_col[k];
        DaMinMaxArgs ma;
        memset(&ma, 0, sizeof ma);
        ma.src.data = (const uint64_t*)in.data[c];
        ma.src.nulls = in.nulls[c];
        ma.src.nrows = nrows;
        ma.flip = a->cfg.group_key_type[k] == TSQ_I64 ? 0x8000000000000000ULL : 0ULL;
        ma.out = (unsigned long long*)(ctx->dscratch + 48);
        ctx->pinned[48] = ~0ULL;
        ctx->pinned[49] = 0;
        ctx->pinned[50] = 0;
        TSQ_HIP(h, hipMemcpyAsync(ctx->dscratch + 48, ctx->pinned + 48, 24, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_da_minmax, dim3(tsq_grid_for(ctx, nrows, 256)), dim3(256), 0, ctx->stream, ma);
        TSQ_HIP(h, hipGetLastError());
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 48, ctx->dscratch + 48, 24, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        a->st.kernel_launches++;
        const bool nullable = in.nulls[c] != nullptr;
        uint64_t kmin = 0, range = 0;
        if (ctx->pinned[50] != 0) {
            kmin = ctx->pinned[48] ^ ma.flip;
            range = (ctx->pinned[49] ^ ma.flip) - kmin;
        }
        if (range >> TSQ_DAAGG_MAX_BITS) return TSQ_OK;
        const uint64_t codes = range + 1 + (nullable ? 1 : 0);
        uint32_t w = 0;
        while (((uint64_t)1 << w) < codes) w++;
        ks.kmin[k] = kmin;
        ks.width[k] = w;
        ks.shift[k] = total;
        ks.nullcode[k] = nullable ? (uint32_t)(((uint64_t)1 << w) - 1) : TSQ_DAAGG_NO_NULL;
        ks.maxd[k] = (uint32_t)(((uint64_t)1 << w) - 1) - (nullable ? 1u : 0u);
        if (w == 0) ks.maxd[k] = 0;
        total += w;
        if (total > TSQ_DAAGG_MAX_BITS) return TSQ_OK;
    }
    const int log2c_env = (int)tsq_knob(a->ctx, TSQ_KNOB_DAAGG_LOG2C, 0);  // (experiment knob: 11 = half-size tables)
    const uint32_t log2c = (log2c_env >= 9 && log2c_env <= 12 && pl.W <= 3) ? (uint32_t)log2c_env : (pl.W <= 3 ? 12u : 11u);
    a->da_low = total <= log2c;  // the word fits one LDS table: no partition pass (k_agg_da_low)
    uint32_t b = std::max(total, log2c + TSQ_RADIX_MIN_BITS);
    if (b - log2c > TSQ_RADIX_MAX_BITS) return TSQ_OK;
    a->da_pbits = std::max<uint32_t>(b - log2c, 5);  // >= 32 partitions x 8 region shares: a workgroup for every CU (k_agg_da)
    a->da_ebits = b - a->da_pbits;
    a->da_dm.kmin = 0;  // the partial group's key word is the field word d itself
    a->da_dm.range = ((uint64_t)1 << b) - 1;
    a->da_dm.b = b;
    a->da_dm.s = (b + 1) / 2;
    a->da_dm.mask = (uint32_t)(((uint64_t)1 << b) - 1);
    a->da_dm.skip_high = 0;
    a->da_state = 1;
    return TSQ_OK;
}

// the key range of the first large batch decides: [kmin, kmin + 2^b) with b <= TSQ_DAAGG_MAX_BITS, or the 64-bit H mode
tsq_status da_agg_setup(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    if (a->da_state) return TSQ_OK;
    a->da_state = -1;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const bool env_off = tsq_knob(a->ctx, TSQ_KNOB_PACKED_KEYS, 1) == 0;
    const AfPlan& pl = a->fplan;
    if (env_off || (pl.key_type != TSQ_I64 && pl.key_type != TSQ_U64)) return TSQ_OK;
    if (a->mk_n > 1) return da_agg_setup_multi(a, in, nrows);
    // the key range — and, for one integer argument column, its value range — over a sample of the batch (one 256-row block of
    // every 16 once the batch has 4 Mi rows: 2 GB of keys would take 0.4 ms to read, the sample 0.03): a key or an argument the
    // sample did not show is an exception row later (tsq_daagg.h), never a wrong result
    DaAggRangeArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.data[0] = (const uint64_t*)in.data[pl.key_col];
    ra.nulls[0] = in.nulls[pl.key_col];
    ra.flip[0] = pl.key_type == TSQ_I64 ? 0x8000000000000000ULL : 0ULL;
    ra.ncols = 1;
    const bool narrow_try = pl.V == 1 && (pl.vtype[0] == TSQ_I64 || pl.vtype[0] == TSQ_U64) && tsq_knob(ctx, TSQ_KNOB_AGG_NARROW_CELLS, 1) != 0;
    if (narrow_try) {
        ra.data[1] = (const uint64_t*)in.data[pl.vcol[0]];
        ra.nulls[1] = in.nulls[pl.vcol[0]];
        ra.flip[1] = 0;  // as unsigned numbers: a negative BIGINT is a huge value and keeps the full cell
        ra.ncols = 2;
    }
    ra.nrows = nrows;
    ra.every = nrows >= (4 << 20) ? 16 : 1;
    ra.out = (unsigned long long*)(ctx->dscratch + 48);
    for (int c = 0; c < 2; c++) {
        ctx->pinned[48 + 3 * c] = ~0ULL;
        ctx->pinned[49 + 3 * c] = 0;
        ctx->pinned[50 + 3 * c] = 0;
    }
    TSQ_HIP(h, hipMemcpyAsync(ctx->dscratch + 48, ctx->pinned + 48, 48, hipMemcpyHostToDevice, ctx->stream));
    {
        const int64_t sblocks = ((nrows + 255) / 256 + ra.every - 1) / ra.every;
        hipLaunchKernelGGL(k_daagg_sample_range, dim3((unsigned)std::min<int64_t>(sblocks, (int64_t)ctx->num_cus * 8)), dim3(256), 0, ctx->stream, ra);
    }
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 48, ctx->dscratch + 48, 48, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    a->st.kernel_launches++;
    if (ctx->pinned[50] == 0) return TSQ_OK;
    uint64_t kmin = ctx->pinned[48] ^ ra.flip[0], kmax = ctx->pinned[49] ^ ra.flip[0];
    if (a->da_hint) {  // the caller knows the keys' range (the dictionary's group ids, kd_batch): nothing the sample missed becomes an exception row
        kmin = a->da_hint_min;
        kmax = std::max<uint64_t>(kmax, a->da_hint_max);
    }
    const uint64_t range = kmax - kmin;
    a->da_paybytes = 8;
    if (narrow_try && ctx->pinned[53] != 0) {
        const uint64_t vmax = ctx->pinned[52];
        a->da_paybytes = vmax < (1ull << 16) ? 2u : (vmax < (1ull << 32) ? 4u : 8u);
    }
    const int log2c_env = (int)tsq_knob(a->ctx, TSQ_KNOB_DAAGG_LOG2C, 0);  // (experiment knob: 11 = half-size tables)
    const uint32_t log2c = (log2c_env >= 9 && log2c_env <= 12 && pl.W <= 3) ? (uint32_t)log2c_env : (pl.W <= 3 ? 12u : 11u);
    if (range >> TSQ_DAAGG_MAX_BITS) return TSQ_OK;
    uint32_t b = log2c + TSQ_RADIX_MIN_BITS;
    while ((range >> b) != 0) b++;
    if (b - log2c > TSQ_RADIX_MAX_BITS) return TSQ_OK;
    a->da_pbits = std::max<uint32_t>(b - log2c, 5);  // >= 32 partitions x 8 region shares: a workgroup for every CU (k_agg_da)
    a->da_ebits = b - a->da_pbits;
    a->da_dm.kmin = kmin;
    a->da_dm.range = ((uint64_t)1 << b) - 1;  // the whole 2^b window above kmin: later batches may bring somewhat larger keys
    a->da_dm.b = b;
    a->da_dm.s = (b + 1) / 2;
    a->da_dm.mask = (uint32_t)(((uint64_t)1 << b) - 1);
    a->da_dm.skip_high = 0;
    a->da_state = 1;
    return TSQ_OK;
}

// n_part partial groups (key word + W words each) into the group table: consumeIntermData (aggregate.go:424-427)
tsq_status merge_partials(tsq_agg* a, const AfPartials& parts, uint32_t n_part) {
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const bool mk = a->mk_n > 1;
    MergeArgs ma;
    memset(&ma, 0, sizeof ma);
    ma.plan = a->fplan;
    ma.in = parts;
    ma.counters = a->counters.as<unsigned long long>();
    if (mk) TSQ_TRY(a->slot_of.reserve(ctx, h, (size_t)n_part * 4 + 16));
    return upsert_loop(a, (int64_t)n_part, nullptr, [&](const AggTable& t, int64_t n, const uint32_t* retry_in, uint32_t* retry_out) -> tsq_status {
        ma.t = t;
        ma.bail_after = std::max<uint64_t>(1024, t.cap / 16);
        ma.n = n;
        ma.retry_in = retry_in;
        ma.retry_out = retry_out;
        if (mk) {  // several key columns: claim by tag, then compare the cells (two launches, as the row upsert does)
            if (t.cap + 2 >= 0xffffffffULL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "multi-key aggregate: more than 2^32 group slots");
            MergeMultiArgs mm;
            memset(&mm, 0, sizeof mm);
            mm.m = ma;
            mm.ks = a->da_keys;
            mm.slot_of = a->slot_of.as<uint32_t>();
            mm.tag_bits = a->test_tag_bits;
            for (int phase = 0; phase < 2; phase++) {
                mm.phase = phase;
                hipLaunchKernelGGL(k_agg_merge_multi, dim3(tsq_grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, mm);
                TSQ_HIP(h, hipGetLastError());
                a->st.kernel_launches++;
            }
            return TSQ_OK;
        }
        hipLaunchKernelGGL(k_agg_merge, dim3(tsq_grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, ma);
        TSQ_HIP(h, hipGetLastError());
        a->st.kernel_launches++;
        return TSQ_OK;
    });
}

// ---- the side stream of the dense packed route: everything that reads the dense state or reuses the partitioned stores on the
// context's stream waits for the launches still running there
tsq_status side_join(tsq_agg* a) {
    for (int s = 0; s < 2; s++) {
        if (!a->side_busy[s]) continue;
        TSQ_HIP(&a->hdr, hipStreamWaitEvent(a->ctx->stream, a->side_ev[s], 0));
        a->side_busy[s] = false;
    }
    return TSQ_OK;
}
bool side_setup(tsq_agg* a) {
    if (a->side_state) return a->side_state == 1;
    a->side_state = -1;
    if (hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking) != hipSuccess) { a->side = nullptr; return false; }
    bool ok = hipEventCreateWithFlags(&a->side_part, hipEventDisableTiming) == hipSuccess;
    for (int s = 0; s < 2 && ok; s++) ok = hipEventCreateWithFlags(&a->side_ev[s], hipEventDisableTiming) == hipSuccess;
    if (!ok) return false;
    a->side_state = 1;
    return true;
}

// ---- dense partial state of the one-key packed route (tsq_daagg.h, K7f)
void da_dense_args(tsq_agg* a, DaAggDenseArgs& da) {
    memset(&da, 0, sizeof da);
    da.plan = a->fplan;
    da.dm = a->da_dm;
    for (int k = 0; k < a->fplan.W; k++) da.dense_w[k] = a->dense_w[k].as<unsigned long long>();
    da.dense_touch = a->dense_touch.as<uint32_t>();
    da.ncells = (uint64_t)1 << a->da_dm.b;
}
// first use: the accumulators of all 2^b packed words, set to the words' initial values.  Not usable (dense_state = -1) when the
// memory is not there: the batch then appends partial groups as before.
tsq_status da_dense_setup(tsq_agg* a) {
    if (a->dense_state) return TSQ_OK;
    a->dense_state = -1;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    if (tsq_knob(ctx, TSQ_KNOB_AGG_DENSE, 1) == 0 || a->mk_n > 1 || a->da_ebits < 5) return TSQ_OK;
    const size_t ncells = (size_t)1 << a->da_dm.b;
    for (int k = 0; k < a->fplan.W; k++)
        if (a->dense_w[k].reserve(ctx, h, ncells * 8) != TSQ_OK) return TSQ_OK;
    if (a->dense_touch.reserve(ctx, h, ncells / 8 + 64) != TSQ_OK) return TSQ_OK;
    DaAggDenseArgs da;
    da_dense_args(a, da);
    da.init_only = 1;
    hipLaunchKernelGGL(k_daagg_dense_emit, dim3(tsq_grid_for(ctx, (int64_t)ncells, 256)), dim3(256), 0, ctx->stream, da);
    TSQ_HIP(h, hipGetLastError());
    a->st.kernel_launches++;
    a->dense_state = 1;
    a->dense_rows = 0;
    return TSQ_OK;
}
// the touched cells become partial groups and are merged into the table; the state is empty (initial words) afterwards
tsq_status da_dense_flush(tsq_agg* a) {
    if (a->dense_state != 1 || a->dense_rows == 0) return TSQ_OK;
    TSQ_TRY(side_join(a));
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const AfPlan& pl = a->fplan;
    const size_t ncells = (size_t)1 << a->da_dm.b;
    TSQ_TRY(a->fkey.reserve(ctx, h, ncells * 8));
    for (int k = 0; k < pl.W; k++) TSQ_TRY(a->fw[k].reserve(ctx, h, ncells * 8));
    TSQ_TRY(a->fctl.reserve(ctx, h, 64));
    TSQ_HIP(h, hipMemsetAsync(a->fctl.p, 0, 64, ctx->stream));
    DaAggDenseArgs da;
    da_dense_args(a, da);
    da.out.key = a->fkey.as<unsigned long long>();
    for (int k = 0; k < pl.W; k++) da.out.w[k] = a->fw[k].as<unsigned long long>();
    da.out.count = a->fctl.as<uint32_t>();
    da.out.cap = (uint32_t)ncells;
    hipLaunchKernelGGL(k_daagg_dense_emit, dim3(tsq_grid_for(ctx, (int64_t)ncells, 256)), dim3(256), 0, ctx->stream, da);
    TSQ_HIP(h, hipGetLastError());
    a->st.kernel_launches++;
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 8, a->fctl.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    const uint32_t n_part = ((const uint32_t*)(ctx->pinned + 8))[0];
    a->dense_rows = 0;
    a->dense_flushes++;
    return merge_partials(a, da.out, n_part);
}

// one batch through LDS pre-aggregation.  *done = false: nothing was merged, the caller runs the row path.
// ---- partitioned groups (tsq_aggfast.h K7p)
tsq_status pg_pend_batch(tsq_agg* a, const tsq_colset& in, int64_t nrows);
// what H mode takes: partitions of 2^10 at most, each half-filling one LDS table
static bool pg_h_fits(const AfPlan& pl, int64_t groups_est) { return ((double)groups_est * 1.3 / (double)af_slots(pl)) <= 1024.0; }
// the sub-tables for `est` groups at half load; refused beyond 2^11 partitions x 16 sub-tables (6.7e7 groups of <= 3 words)
tsq_status pg_setup(tsq_agg* a, int64_t est) {
    if (a->pg_state) return TSQ_OK;
    a->pg_state = -1;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const AfPlan& pl = a->fplan;
    const bool trace = tsq_knob(ctx, TSQ_KNOB_DA_TRACE, 0) != 0;
    if (a->mk_n > 1 || a->plan.n_keys != 1 || a->groups != 0 || a->in_rows != 0 || a->stream) {
        if (trace) fprintf(stderr, "[pg_setup] refused: mk_n %d n_keys %d groups %lld in_rows %lld stream %d\n", a->mk_n, a->plan.n_keys, (long long)a->groups, (long long)a->in_rows, (int)a->stream);
        return TSQ_OK;
    }
    for (int k = 0; k < pl.W; k++)
        if (pl.init[k] != 0 && pl.init[k] != ~0ull) {  // (the state is initialised with memsets)
            if (trace) fprintf(stderr, "[pg_setup] refused: init[%d] = %llx\n", k, pl.init[k]);
            return TSQ_OK;
        }
    const uint64_t S = af_slots(pl);
    const uint64_t want = (uint64_t)((double)est / 0.6) + S;  // (an LDS table takes groups up to 7/8 of its slots; the estimate may be low)
    uint32_t pbits = 8, sbits = 0;
    while (pbits < 11 && (S << pbits) < want) pbits++;
    while (sbits < 4 && (S << (pbits + sbits)) < want) sbits++;
    const int64_t forced = tsq_knob(ctx, TSQ_KNOB_AGG_PG, 1);  // (tests: v >= 2 -> 2^(v - 2) sub-tables whatever the estimate: small states that fill up)
    if (forced >= 2) {
        const uint32_t tot = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(forced - 2, 15));  // (two sub-tables at least: the partition pass needs a bit)
        pbits = std::min<uint32_t>(tot, 11u);
        sbits = tot - pbits;
    } else if ((S << (pbits + sbits)) < want) return TSQ_OK;
    const size_t slots = (size_t)S << (pbits + sbits), subs = (size_t)1 << (pbits + sbits);
    tsq_status s = a->pg_key.reserve(ctx, h, slots * 8 + 64);
    for (int k = 0; k < pl.W && s == TSQ_OK; k++) s = a->pg_w[k].reserve(ctx, h, slots * 8 + 64);
    if (s == TSQ_OK) s = a->pg_used.reserve(ctx, h, subs * 4 + 64);
    if (s != TSQ_OK) {  // no memory for the state: the other modes keep the aggregate
        h->err.clear();
        a->pg_key.release();
        for (auto& b : a->pg_w) b.release();
        a->pg_used.release();
        return TSQ_OK;
    }
    TSQ_HIP(h, hipMemsetAsync(a->pg_key.p, 0x80, slots * 8, ctx->stream));  // TSQ_AF_EMPTY
    for (int k = 0; k < pl.W; k++) TSQ_HIP(h, hipMemsetAsync(a->pg_w[k].p, pl.init[k] ? 0xff : 0x00, slots * 8, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(a->pg_used.p, 0, subs * 4, ctx->stream));
    a->pg_pbits = pbits;
    a->pg_sbits = sbits;
    a->pg_rows = 0;
    a->pg_state = 1;
    return TSQ_OK;
}
// the merging way out: every group of the sub-tables as a partial group into the table in HBM (slot ranges of 2^24: the partial list
// of a range cannot overflow), then the state is empty again
tsq_status pg_flush(tsq_agg* a) {
    if (a->pg_state != 1 || a->pg_rows == 0) return TSQ_OK;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const AfPlan& pl = a->fplan;
    const uint64_t S = af_slots(pl), slots = S << (a->pg_pbits + a->pg_sbits), step = (uint64_t)1 << 24;
    const size_t pcap = (size_t)std::min<uint64_t>(slots, step);
    TSQ_TRY(a->fkey.reserve(ctx, h, pcap * 8));
    for (int k = 0; k < pl.W; k++) TSQ_TRY(a->fw[k].reserve(ctx, h, pcap * 8));
    TSQ_TRY(a->fctl.reserve(ctx, h, 64));
    for (uint64_t lo = 0; lo < slots; lo += step) {
        AfPgEmitArgs ea;
        memset(&ea, 0, sizeof ea);
        ea.plan = pl;
        ea.out.key = a->fkey.as<unsigned long long>();
        for (int k = 0; k < pl.W; k++) {
            ea.out.w[k] = a->fw[k].as<unsigned long long>();
            ea.pg_w[k] = a->pg_w[k].as<unsigned long long>();
        }
        ea.out.count = a->fctl.as<uint32_t>();
        ea.out.cap = (uint32_t)pcap;
        ea.pg_key = a->pg_key.as<unsigned long long>();
        ea.lo = lo;
        ea.n = std::min<uint64_t>(step, slots - lo);
        TSQ_HIP(h, hipMemsetAsync(a->fctl.p, 0, 64, ctx->stream));
        hipLaunchKernelGGL(k_pg_emit, dim3(tsq_grid_for(ctx, (int64_t)ea.n, 256)), dim3(256), 0, ctx->stream, ea);
        TSQ_HIP(h, hipGetLastError());
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 8, a->fctl.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        a->st.kernel_launches++;
        const uint32_t n_part = ((const uint32_t*)(ctx->pinned + 8))[0];
        if (n_part) TSQ_TRY(merge_partials(a, ea.out, n_part));
    }
    const size_t subs = (size_t)1 << (a->pg_pbits + a->pg_sbits);
    TSQ_HIP(h, hipMemsetAsync(a->pg_key.p, 0x80, slots * 8, ctx->stream));
    for (int k = 0; k < pl.W; k++) TSQ_HIP(h, hipMemsetAsync(a->pg_w[k].p, pl.init[k] ? 0xff : 0x00, slots * 8, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(a->pg_used.p, 0, subs * 4, ctx->stream));
    a->pg_rows = 0;
    return TSQ_OK;
}

tsq_status agg_batch_fast(tsq_agg* a, const tsq_colset& in, int64_t nrows, int64_t groups_est, bool* done) {
    *done = false;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const AfPlan& pl = a->fplan;
    const uint32_t S = af_slots(pl);
    const bool mk = a->mk_n > 1;  // several key columns: the packed route or nothing
    // partitioned groups: once the sub-tables hold groups every batch goes to them (a group has ONE home)
    bool pg = a->pg_state == 1;
    const int64_t pg_knob = tsq_knob(ctx, TSQ_KNOB_AGG_PG, 1);  // 0: never; >= 2 (tests): always, with 2^(v - 2) sub-tables
    if (!pg && pg_knob >= 2 && !mk && a->pg_state == 0) {
        TSQ_TRY(pg_setup(a, groups_est));
        pg = a->pg_state == 1;
    }
    // (rows in the dense state of the packed route are groups the table has not seen yet: `groups_est` says nothing then)
    const bool low = !pg && !mk && groups_est <= (int64_t)(S / 2) && !(a->da_state == 1 && a->dense_state == 1 && a->dense_rows > 0);
    uint32_t bits = 0;
    if (!low && !pg) TSQ_TRY(da_agg_setup(a, in, nrows));
    const bool packed = !pg && !low && a->da_state == 1;
    if (!low && !packed && !pg && !mk && pg_knob != 0 && !pg_h_fits(pl, groups_est)) {  // more groups than LDS tables hold per batch: the sub-tables
        TSQ_TRY(pg_setup(a, groups_est));
        pg = a->pg_state == 1;
    }
    if (pg && !a->pg_in_flush) {  // (the batch that made the state: it waits like the later ones unless it is worth a pass by itself)
        *done = true;
        return pg_pend_batch(a, in, nrows);
    }
    if (pg && a->pg_rows + nrows > ((int64_t)1 << 31)) TSQ_TRY(pg_flush(a));  // (the 32-bit halves of its int64 sums must not wrap)
    if (mk && !packed) return TSQ_OK;
    const bool packed_low = packed && mk && a->da_low;
    if (pg) bits = a->pg_pbits;
    else if (!low && !packed) {
        // H: partitions small enough that their groups half-fill one LDS table, at least 256 of them (parallelism)
        // (2^11 partitions — tables 1/8 full, shorter walks — were measured: k_agg_lds 0.66 -> 0.64 ms per 1e8 rows, but the partition
        // kernel with a payload column writes 32-byte runs then and goes from 0.76 to 0.83 ms: kept at 2^10)
        bits = 8;
        while (bits < 10 && ((double)groups_est * 2.2 / (double)S) > (double)(1u << bits)) bits++;
        if (((double)groups_est * 1.3 / (double)S) > (double)(1u << bits)) return TSQ_OK;  // too many groups for LDS tables
    }
    // partial-group buffer: every workgroup may emit a table, plus spilled rows; beyond cap the batch is redone row by row
    // a packed batch with few partitions splits each of them over up to 8 workgroups (k_agg_da: one share of the XCC regions each)
    uint32_t da_nsplit = 1;
    if (packed && !packed_low)
        while (da_nsplit < 8 && ((uint32_t)1 << a->da_pbits) * da_nsplit < (uint32_t)ctx->num_cus) da_nsplit *= 2;
    // one key column on the packed route: the LDS tables are folded into the dense state, no partial groups leave the batch
    bool dense = false;
    if (packed && !packed_low && !mk) {
        TSQ_TRY(da_dense_setup(a));
        dense = a->dense_state == 1;
        const int64_t dk = tsq_knob(ctx, TSQ_KNOB_AGG_DENSE, 1);
        if (dense && a->dense_rows + nrows > (dk > 1 ? dk : ((int64_t)1 << 31))) TSQ_TRY(da_dense_flush(a));  // (its lo32 sums must not wrap)
    }
    const size_t nblocks = (low || packed_low) ? (size_t)ctx->num_cus : (((size_t)1 << (packed ? a->da_pbits : bits)) * da_nsplit);
    // (partitioned groups: every row may be spilled, and the batch cannot be redone — the sub-tables have taken its rows)
    const size_t pcap = pg ? (size_t)nrows + 4096 : (dense ? 4096 : std::min<size_t>(nblocks * S + (size_t)nrows / 8 + 4096, 0x7fffffffULL));
    TSQ_TRY(a->fkey.reserve(ctx, h, pcap * 8));
    for (int k = 0; k < pl.W; k++) TSQ_TRY(a->fw[k].reserve(ctx, h, pcap * 8));
    TSQ_TRY(a->fctl.reserve(ctx, h, 64));
    TSQ_TRY(a->fexc.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_HIP(h, hipMemsetAsync(a->fctl.p, 0, 64, ctx->stream));
    AfLdsArgs la;
    memset(&la, 0, sizeof la);
    la.plan = pl;
    la.out.key = a->fkey.as<unsigned long long>();
    for (int k = 0; k < pl.W; k++) la.out.w[k] = a->fw[k].as<unsigned long long>();
    la.out.count = a->fctl.as<uint32_t>();
    la.out.cap = (uint32_t)pcap;
    la.in = in;
    la.nrows = nrows;
    la.exc_rows = a->fexc.as<uint32_t>();
    la.exc_count = a->fctl.as<uint32_t>() + 1;
    if (!packed || packed_low) TSQ_TRY(side_join(a));  // (the other modes use the first partitioned store and the partial-group buffers)
    if (low) {
        TSQ_TRY(launch_lds<0>(a, la, (int)std::min<int64_t>(ctx->num_cus, (nrows + TSQ_AF_NT - 1) / TSQ_AF_NT)));
    } else if (packed_low) {
        DaAggLowArgs lo;
        memset(&lo, 0, sizeof lo);
        lo.plan = pl;
        lo.out = la.out;
        lo.ks = a->da_keys;
        lo.src.nrows = nrows;
        for (int k = 0; k < a->mk_n; k++) {
            lo.src.mkdata[k] = in.data[a->mk_col[k]];
            lo.src.mknulls[k] = in.nulls[a->mk_col[k]];
        }
        for (int v = 0; v < pl.V; v++) {
            lo.src.vdata[v] = in.data[pl.vcol[v]];
            lo.src.vnulls[v] = in.nulls[pl.vcol[v]];
            lo.src.vtype[v] = in.type[pl.vcol[v]];
        }
        lo.src.exc_rows = la.exc_rows;
        lo.src.exc_count = la.exc_count;
        const dim3 lgrid((unsigned)std::min<int64_t>(ctx->num_cus, (nrows + TSQ_AF_NT - 1) / TSQ_AF_NT));
        switch (pl.W) {
            case 1: hipLaunchKernelGGL((k_agg_da_low<1, 4096>), lgrid, dim3(TSQ_AF_NT), 0, ctx->stream, lo); break;
            case 2: hipLaunchKernelGGL((k_agg_da_low<2, 4096>), lgrid, dim3(TSQ_AF_NT), 0, ctx->stream, lo); break;
            case 3: hipLaunchKernelGGL((k_agg_da_low<3, 4096>), lgrid, dim3(TSQ_AF_NT), 0, ctx->stream, lo); break;
            case 4: hipLaunchKernelGGL((k_agg_da_low<4, 2048>), lgrid, dim3(TSQ_AF_NT), 0, ctx->stream, lo); break;
            default: hipLaunchKernelGGL((k_agg_da_low<5, 2048>), lgrid, dim3(TSQ_AF_NT), 0, ctx->stream, lo); break;
        }
        TSQ_HIP(h, hipGetLastError());
        a->st.kernel_launches++;
        a->packed_batches++;
    } else if (packed) {
        DaAggStore st;
        memset(&st, 0, sizeof st);
        st.bits = a->da_pbits;
        st.ebits = a->da_ebits;
        st.paybytes = (pl.V == 1 && !mk) ? a->da_paybytes : 8u;
        const uint32_t P = 1u << st.bits;
        // two 512-thread workgroups per CU (tsq_daagg.h, WITH_ROW = false): one argument column and the overflow store of the dense
        // state; tiles of 4096 rows; knob 0: the 1024-thread kernel, 1 (default): narrow cells only, 2: 8-byte cells too
        const int64_t part2_knob = tsq_knob(ctx, TSQ_KNOB_DAAGG_PART2, 1);  // (measured: 7.43 -> 7.11 ms per 1e9 rows with 2-byte cells, 8.58 -> 8.83 ms with 8-byte cells)
        const bool part2 = dense && pl.V == 1 && part2_knob != 0 && (st.paybytes != 8 || part2_knob >= 2);
        const int K = part2 ? 8 : (pl.V == 0 ? 16 : (pl.V == 1 ? 8 : 4)), T = (part2 ? 512 : 1024) * K;  // (part2 with 16 rows per lane: 110 spilled VGPRs at the 128 it may use)
        const double tiles = ceil((double)nrows / T);
        const double lam = std::max((double)nrows / ((double)P * 8.0), ceil(tiles / 8.0) * std::min<double>((double)T, (double)nrows) / (double)P);
        st.cap = (uint32_t)(lam * 1.08 + 8.0 * sqrt(lam) + 2.0 * T / 64.0 + 64.0);
        st.cap = (st.cap + 63u) & ~63u;
        const size_t nregions = (size_t)P * 8, slots = nregions * st.cap;
        if (slots >= 0xffffffffULL) return TSQ_OK;
        // two streams (round 6, knob AGG_OVERLAP; OFF unless the knob is set): the dense state takes the batch's rows from the store on
        // a side stream while the next batch is being partitioned into the other store.  Measured on C3 (1e9 rows / 1e6 groups, four
        // batches): 7.0 ms on one stream, 10.1 ms on two — both kernels fill the chip with persistent workgroups and both live on
        // the LDS pipeline (staged scatter / LDS atomics), so beside each other they run at less than half speed each
        // (profiles/r06_ab_measurements.txt).  Kept as the A/B VERDICT r5 item 3 asked for, with its parity tests.
        const int64_t ovk = tsq_knob(ctx, TSQ_KNOB_AGG_OVERLAP, 0);  // (1: batches of 2^24 rows or more; v >= 2, tests: of v rows or more)
        const bool overlap = dense && !mk && ovk != 0 && nrows >= (ovk > 1 ? ovk : ((int64_t)1 << 24)) && side_setup(a);
        const int sx = overlap ? a->side_turn : 0;
        if (!overlap) TSQ_TRY(side_join(a));
        else if (a->side_busy[sx]) {  // (the batch before the last one has been read out of this store)
            TSQ_HIP(h, hipStreamWaitEvent(ctx->stream, a->side_ev[sx], 0));
            a->side_busy[sx] = false;
        }
        DevBuf& rkeys = sx ? a->r2keys : a->rkeys;
        DevBuf* rpay = sx ? a->r2pay : a->rpay;
        DevBuf& rctl = sx ? a->r2ctl : a->rctl;
        DevBuf& rvend = sx ? a->r2vend : a->rvend;
        DevBuf& rokeys = sx ? a->r2okeys : a->rokeys;
        DevBuf* ropay = sx ? a->r2opay : a->ropay;
        TSQ_TRY(rkeys.reserve(ctx, h, slots * 2 + 256));
        for (int v = 0; v < pl.V; v++) TSQ_TRY(rpay[v].reserve(ctx, h, slots * (v == 0 ? st.paybytes : 8u) + 256));
        TSQ_TRY(rctl.reserve(ctx, h, nregions * 4 + 64));
        TSQ_TRY(rvend.reserve(ctx, h, nregions * 4));
        st.ent = rkeys.as<uint16_t>();
        st.cursor = rctl.as<uint32_t>();
        st.valid_end = rvend.as<uint32_t>();
        for (int v = 0; v < pl.V; v++) st.pay[v] = rpay[v].as<uint64_t>();
        if (dense) {  // skewed keys: the runs that do not fit their regions are aggregated from an overflow store (k_daagg_ovf) — it
                      // holds a whole batch, so it cannot fill
            TSQ_TRY(rokeys.reserve(ctx, h, (size_t)nrows * 4 + 64));
            for (int v = 0; v < pl.V; v++) TSQ_TRY(ropay[v].reserve(ctx, h, (size_t)nrows * 8 + 64));
            st.ovf_u = rokeys.as<uint32_t>();
            for (int v = 0; v < pl.V; v++) st.ovf_pay[v] = ropay[v].as<uint64_t>();
            st.ovf_count = st.cursor + nregions;
            st.ovf_cap = (uint32_t)nrows;
        }
        TSQ_HIP(h, hipMemsetAsync(rctl.p, 0, nregions * 4 + 64, ctx->stream));
        TSQ_HIP(h, hipMemsetAsync(rvend.p, 0xff, nregions * 4, ctx->stream));
        DaAggSrc src;
        memset(&src, 0, sizeof src);
        src.kdata = in.data[pl.key_col];
        src.knulls = in.nulls[pl.key_col];
        src.nrows = nrows;
        for (int k = 0; k < a->mk_n; k++) {
            src.mkdata[k] = in.data[a->mk_col[k]];
            src.mknulls[k] = in.nulls[a->mk_col[k]];
        }
        for (int v = 0; v < pl.V; v++) {
            src.vdata[v] = in.data[pl.vcol[v]];
            src.vnulls[v] = in.nulls[pl.vcol[v]];
            src.vtype[v] = in.type[pl.vcol[v]];
        }
        src.exc_rows = la.exc_rows;
        src.exc_count = la.exc_count;
        DaAggKeys ks = a->da_keys;
        if (!mk) ks.n = 1;
        // hot keys of the batch (tsq_daagg.h, round 5): sampled, then aggregated inside the partition kernel — with a dense state
        // (where their partial groups go) and one integer key column; knob DAAGG_HOT = 0 switches it off (A/B measurements)
        DaAggHot hot;
        memset(&hot, 0, sizeof hot);
        if (dense && !mk && part2 && pl.W <= TSQ_DAAGG_HOT_MAXW && tsq_knob(ctx, TSQ_KNOB_DAAGG_HOT, 1) != 0 && (in.type[pl.key_col] == TSQ_I64 || in.type[pl.key_col] == TSQ_U64)) {
            TSQ_TRY(a->hotkeys.reserve(ctx, h, (TSQ_DAAGG_HOT_MAX + 16 + 2 * TSQ_DAAGG_HOT_TABLE) * 4));
            DaAggHotSampleArgs ha;
            memset(&ha, 0, sizeof ha);
            ha.kdata = (const uint64_t*)in.data[pl.key_col];
            ha.knulls = in.nulls[pl.key_col];
            ha.nrows = nrows;
            ha.dm = a->da_dm;
            ha.keys = a->hotkeys.as<uint32_t>();
            ha.n = ha.keys + TSQ_DAAGG_HOT_MAX;
            ha.table = ha.keys + TSQ_DAAGG_HOT_MAX + 16;
            // the hot set of a batch serves the next three as well (a stale set costs speed, never a result: a key that is no longer
            // hot is aggregated in the partition kernel all the same, one that became hot travels like any other key)
            if (a->hot_age % 4 == 0) {
                TSQ_HIP(h, hipMemsetAsync(ha.table, 0xff, (size_t)TSQ_DAAGG_HOT_TABLE * 4, ctx->stream));  // words: TSQ_DA_NONE
                TSQ_HIP(h, hipMemsetAsync(ha.table + TSQ_DAAGG_HOT_TABLE, 0, (size_t)TSQ_DAAGG_HOT_TABLE * 4, ctx->stream));
                hipLaunchKernelGGL(k_daagg_hot_sample, dim3(TSQ_DAAGG_HOT_SAMPLE / 1024), dim3(1024), 0, ctx->stream, ha);
                TSQ_HIP(h, hipGetLastError());
                hipLaunchKernelGGL(k_daagg_hot_list, dim3(1), dim3(1024), 0, ctx->stream, ha);
                TSQ_HIP(h, hipGetLastError());
                a->st.kernel_launches += 2;
            }
            a->hot_age++;
            hot.keys = ha.keys;
            hot.n = ha.n;
            hot.W = pl.W;
            for (int k = 0; k < pl.W; k++) {
                hot.dense_w[k] = a->dense_w[k].as<unsigned long long>();
                hot.init[k] = pl.init[k];
                hot.wdesc[k] = pl.wdesc[k];
            }
            hot.dense_touch = a->dense_touch.as<uint32_t>();
        }
        const int pgrid = (int)std::min<int64_t>((nrows + T - 1) / T, (int64_t)ctx->num_cus * (part2 ? 2 : 1));
        if (part2 && st.paybytes == 2) hipLaunchKernelGGL((k_daagg_partition<512, 8, 1, 2, false>), dim3(pgrid), dim3(512), 0, ctx->stream, src, a->da_dm, st, ks, hot);
        else if (part2 && st.paybytes == 4) hipLaunchKernelGGL((k_daagg_partition<512, 8, 1, 4, false>), dim3(pgrid), dim3(512), 0, ctx->stream, src, a->da_dm, st, ks, hot);
        else if (part2) hipLaunchKernelGGL((k_daagg_partition<512, 8, 1, 8, false>), dim3(pgrid), dim3(512), 0, ctx->stream, src, a->da_dm, st, ks, hot);
        else if (pl.V == 0) hipLaunchKernelGGL((k_daagg_partition<1024, 16, 0>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, a->da_dm, st, ks, hot);
        else if (pl.V == 1 && st.paybytes == 4) hipLaunchKernelGGL((k_daagg_partition<1024, 8, 1, 4>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, a->da_dm, st, ks, hot);
        else if (pl.V == 1 && st.paybytes == 2) hipLaunchKernelGGL((k_daagg_partition<1024, 8, 1, 2>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, a->da_dm, st, ks, hot);
        else if (pl.V == 1) hipLaunchKernelGGL((k_daagg_partition<1024, 8, 1>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, a->da_dm, st, ks, hot);
        else hipLaunchKernelGGL((k_daagg_partition<1024, 4, 2>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, a->da_dm, st, ks, hot);
        TSQ_HIP(h, hipGetLastError());
        a->st.kernel_launches++;
        DaAggLdsArgs da;
        memset(&da, 0, sizeof da);
        da.plan = pl;
        da.out = la.out;
        da.st = st;
        da.dm = a->da_dm;
        da.nsplit = da_nsplit;
        if (dense) {
            for (int k = 0; k < pl.W; k++) da.dense_w[k] = a->dense_w[k].as<unsigned long long>();
            da.dense_touch = a->dense_touch.as<uint32_t>();
            a->dense_rows += nrows;
        }
        const uint32_t wg_per_cu = (pl.W <= 3 && st.ebits <= 11) ? 2u : 1u;
        const int agrid = (int)std::min<uint32_t>(P * da.nsplit, (uint32_t)ctx->num_cus * wg_per_cu);
        hipStream_t astream = ctx->stream;
        if (overlap) {  // the rest of the batch on the side stream, behind the partition pass
            TSQ_HIP(h, hipEventRecord(a->side_part, ctx->stream));
            TSQ_HIP(h, hipStreamWaitEvent(a->side, a->side_part, 0));
            astream = a->side;
            da.concurrent = 1;
        }
        TSQ_TRY(launch_agg_da(a, da, agrid, astream));
        if (dense) {
            DaAggOvfArgs oa;
            memset(&oa, 0, sizeof oa);
            oa.plan = pl;
            oa.st = st;
            for (int k = 0; k < pl.W; k++) oa.dense_w[k] = da.dense_w[k];
            oa.dense_touch = da.dense_touch;
            const dim3 ogrid((unsigned)ctx->num_cus);
            switch (pl.W) {
                case 1: hipLaunchKernelGGL((k_daagg_ovf<1>), ogrid, dim3(TSQ_AF_NT), 0, astream, oa); break;
                case 2: hipLaunchKernelGGL((k_daagg_ovf<2>), ogrid, dim3(TSQ_AF_NT), 0, astream, oa); break;
                case 3: hipLaunchKernelGGL((k_daagg_ovf<3>), ogrid, dim3(TSQ_AF_NT), 0, astream, oa); break;
                case 4: hipLaunchKernelGGL((k_daagg_ovf<4>), ogrid, dim3(TSQ_AF_NT), 0, astream, oa); break;
                default: hipLaunchKernelGGL((k_daagg_ovf<5>), ogrid, dim3(TSQ_AF_NT), 0, astream, oa); break;
            }
            TSQ_HIP(h, hipGetLastError());
            a->st.kernel_launches++;
        }
        if (overlap) {
            TSQ_HIP(h, hipEventRecord(a->side_ev[sx], a->side));
            a->side_busy[sx] = true;
            a->side_turn ^= 1;
            a->side_batches++;
        }
        a->packed_batches++;
    } else {
        RadixStore st;
        memset(&st, 0, sizeof st);
        st.bits = bits;
        st.R = 8;
        const uint32_t P = 1u << bits;
        const int K = pl.V <= 1 ? 8 : 4, T = 1024 * K;
        const double lam = (double)nrows / ((double)P * 8.0);
        st.cap = (uint32_t)(lam * 1.08 + 8.0 * sqrt(lam) + 2.0 * T / 64.0 + 64.0);
        st.cap = (st.cap + 15u) & ~15u;
        const size_t nregions = (size_t)P * 8, slots = nregions * st.cap;
        if (slots >= 0xffffffffULL) return TSQ_OK;
        TSQ_TRY(a->rkeys.reserve(ctx, h, slots * 8 + 256));
        for (int v = 0; v < pl.V; v++) TSQ_TRY(a->rpay[v].reserve(ctx, h, slots * 8 + 256));
        TSQ_TRY(a->rctl.reserve(ctx, h, nregions * 4 + 64));
        TSQ_TRY(a->rvend.reserve(ctx, h, nregions * 4));
        TSQ_TRY(a->rokeys.reserve(ctx, h, (size_t)nrows * 8 + 64));
        for (int v = 0; v < pl.V; v++) TSQ_TRY(a->ropay[v].reserve(ctx, h, (size_t)nrows * 8 + 64));
        st.keys = a->rkeys.as<uint64_t>();
        st.cursor = a->rctl.as<uint32_t>();
        st.ovf_count = st.cursor + nregions;
        st.valid_end = a->rvend.as<uint32_t>();
        st.ovf_keys = a->rokeys.as<uint64_t>();
        st.ovf_cap = (uint32_t)nrows;
        for (int v = 0; v < pl.V; v++) {
            st.pay[v] = a->rpay[v].as<uint64_t>();
            st.ovf_pay[v] = a->ropay[v].as<uint64_t>();
        }
        TSQ_HIP(h, hipMemsetAsync(a->rctl.p, 0, nregions * 4 + 64, ctx->stream));
        TSQ_HIP(h, hipMemsetAsync(a->rvend.p, 0xff, nregions * 4, ctx->stream));
        RadixSrc src;
        memset(&src, 0, sizeof src);
        src.data = in.data[pl.key_col];
        src.nulls = in.nulls[pl.key_col];
        src.type = in.type[pl.key_col];
        src.nrows = nrows;
        src.key_kind = 1;
        for (int v = 0; v < pl.V; v++) {
            src.vdata[v] = in.data[pl.vcol[v]];
            src.vnulls[v] = in.nulls[pl.vcol[v]];
            src.vtype[v] = in.type[pl.vcol[v]];
        }
        src.exc_rows = la.exc_rows;
        src.exc_count = la.exc_count;
        const int pgrid = (int)std::min<int64_t>((nrows + T - 1) / T, ctx->num_cus);
        // HASHED: the store receives table words mix64(group key word) — see tsq_aggfast.h
        if (pl.V == 0) hipLaunchKernelGGL((k_radix_partition<1024, 16, 4, 0, false, true>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
        else if (pl.V == 1) hipLaunchKernelGGL((k_radix_partition<1024, 8, 4, 1, false, true>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
        else hipLaunchKernelGGL((k_radix_partition<1024, 4, 4, 2, false, true>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
        TSQ_HIP(h, hipGetLastError());
        a->st.kernel_launches++;
        la.st = st;
        if (pg) {
            AfPgArgs ga;
            memset(&ga, 0, sizeof ga);
            ga.plan = pl;
            ga.out = la.out;
            ga.st = st;
            ga.pg_key = a->pg_key.as<unsigned long long>();
            for (int k = 0; k < pl.W; k++) ga.pg_w[k] = a->pg_w[k].as<unsigned long long>();
            ga.pg_used = a->pg_used.as<uint32_t>();
            ga.sbits = a->pg_sbits;
            const dim3 ggrid(std::min<uint32_t>(P << a->pg_sbits, (uint32_t)ctx->num_cus * (pl.W <= 1 ? 2u : 1u)));
            switch (pl.W) {
                case 1: hipLaunchKernelGGL((k_agg_pg<1>), ggrid, dim3(TSQ_AF_NT), 0, ctx->stream, ga); break;
                case 2: hipLaunchKernelGGL((k_agg_pg<2>), ggrid, dim3(TSQ_AF_NT), 0, ctx->stream, ga); break;
                case 3: hipLaunchKernelGGL((k_agg_pg<3>), ggrid, dim3(TSQ_AF_NT), 0, ctx->stream, ga); break;
                case 4: hipLaunchKernelGGL((k_agg_pg<4>), ggrid, dim3(TSQ_AF_NT), 0, ctx->stream, ga); break;
                default: hipLaunchKernelGGL((k_agg_pg<5>), ggrid, dim3(TSQ_AF_NT), 0, ctx->stream, ga); break;
            }
            TSQ_HIP(h, hipGetLastError());
            AfPgOvfArgs oa;
            memset(&oa, 0, sizeof oa);
            oa.plan = pl;
            oa.out = la.out;
            oa.st = st;
            hipLaunchKernelGGL(k_pg_ovf, dim3(ctx->num_cus), dim3(256), 0, ctx->stream, oa);  // the overflow list of skewed partitions (usually empty)
            TSQ_HIP(h, hipGetLastError());
            a->st.kernel_launches += 2;
            a->pg_rows += nrows;
            a->pg_batches++;
        } else {
            TSQ_TRY(launch_lds<1>(a, la, (int)P));
            TSQ_TRY(launch_lds<2>(a, la, 8));  // the overflow list of skewed partitions (usually empty)
        }
    }
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 8, a->fctl.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    const uint32_t n_part = ((const uint32_t*)(ctx->pinned + 8))[0], n_exc = ((const uint32_t*)(ctx->pinned + 8))[1];
    const uint32_t n_misfit = ((const uint32_t*)(ctx->pinned + 8))[2];  // exception rows whose argument did not fit the narrow cells
    if (((const uint32_t*)(ctx->pinned + 8))[3]) return tsq_fail(h, TSQ_ERR_HIP, "internal: the overflow store of the packed aggregate was full");
    if (n_part > la.out.cap) {  // more partial groups than the buffer holds: nothing was merged yet, redo the batch row by row
        if (pg) return tsq_fail(h, TSQ_ERR_HIP, "internal: the spill list of the partitioned groups was full");  // (sized for every row)
        a->fast_fallbacks++;
        return TSQ_OK;
    }
    TSQ_TRY(merge_partials(a, la.out, n_part));
    if (n_exc) TSQ_TRY(agg_rows(a, in, (int64_t)n_exc, a->fexc.as<uint32_t>()));
    if (packed && a->da_paybytes != 8 && (int64_t)n_misfit > nrows / 64) a->da_paybytes = 8;  // the sample did not describe the argument column: full cells from here on
    if (packed && (int64_t)(n_exc - n_misfit) > nrows / 4) a->da_state = -1;  // the range of the first batch does not describe the input: 64-bit H mode from here on
    a->fast_batches++;
    *done = true;
    return TSQ_OK;
}

// one device-resident batch: LDS pre-aggregation when the plan and the batch allow it, the row upsert otherwise
// ---- several integer key columns as one 64-bit composite key (k_agg_wide_compose): the fields from the first batch, the child
tsq_status agg_batch(tsq_agg* a, const tsq_colset& in, int64_t nrows);

// ---------------------------------------------------------------- StreamAggExec (host side; kernels in tsq_streamagg.h)
// the group array grows by copy: a group's number never changes (no rehash)
tsq_status stream_grow(tsq_agg* a, uint64_t new_cap) {
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    AggTableBufs nb;
    tsq_status s = alloc_table(a, nb, new_cap);
    if (s != TSQ_OK) { nb.release(); return s; }
    const size_t g = (size_t)a->groups;
    hipError_t e = hipSuccess;
    auto cp = [&](DevBuf& to, DevBuf& from, size_t bytes) {
        if (e == hipSuccess && bytes && to.p && from.p) e = hipMemcpyAsync(to.p, from.p, bytes, hipMemcpyDeviceToDevice, ctx->stream);
    };
    cp(nb.tag, a->tb.tag, g * 8);
    cp(nb.gknull, a->tb.gknull, g);
    for (int k = 0; k < a->plan.n_keys; k++) cp(nb.gkey[k], a->tb.gkey[k], g * 8);
    for (int i = 0; i < a->plan.n_aggs; i++) {
        cp(nb.acc[i], a->tb.acc[i], g * 8);
        cp(nb.aux[i], a->tb.aux[i], g * 8);
        cp(nb.cnt[i], a->tb.cnt[i], g * 8);
        cp(nb.seen[i], a->tb.seen[i], g);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { nb.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("stream aggregate, growing the group array: ") + hipGetErrorString(e)); }
    a->tb.release();
    a->tb = nb;  // shallow move of the buffer handles
    return TSQ_OK;
}

tsq_status stream_batch(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    if (nrows >= 0x7fffffffLL) return tsq_fail(h, TSQ_ERR_INVALID, "internal: batch too large");
    const uint32_t nchunks = (uint32_t)((nrows + TSQ_SA_CHUNK - 1) / TSQ_SA_CHUNK);
    TSQ_TRY(a->sa_cnt.reserve(ctx, h, ((size_t)nchunks + 1) * 4 + 64));
    StreamAggArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.u.in = in;
    sa.u.plan = a->plan;
    sa.u.nrows = nrows;
    sa.u.row_base = a->in_rows;
    sa.u.counters = a->counters.as<unsigned long long>();
    fill_agg_table(a, a->tb, sa.u.t);
    sa.groups_before = (uint64_t)a->groups;
    sa.chunk_cnt = a->sa_cnt.as<uint32_t>();
    sa.nchunks = nchunks;
    const int grid = (int)std::min<int64_t>(nchunks, (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(k_sa_count, dim3(grid), dim3(TSQ_SA_NT), 0, ctx->stream, sa);
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_sa_scan, dim3(1), dim3(1024), 0, ctx->stream, sa.chunk_cnt, nchunks);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned, sa.chunk_cnt + nchunks, 4, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    const uint64_t heads = ((const uint32_t*)ctx->pinned)[0];
    const uint64_t need = (uint64_t)a->groups + heads;
    if (need > a->tb.cap) {
        TSQ_TRY(stream_grow(a, std::max<uint64_t>(need, a->tb.cap * 2)));
        fill_agg_table(a, a->tb, sa.u.t);
    }
    // plans whose reducing aggregates (COUNT / SUM / AVG / MAX / MIN of fixed-width arguments) number <= 4 keep the open run's partial
    // results in registers, one per lane (k_sa_update_lanes); the others — MAX / MIN of strings, more aggregates — reduce every step
    SaRed red;
    memset(&red, 0, sizeof red);
    bool lanes_ok = true;
    for (int i = 0; i < a->plan.n_aggs; i++) {
        const tsq_agg_func& f = a->plan.f[i];
        if (f.func == TSQ_AGG_FIRSTROW) continue;
        if ((f.func == TSQ_AGG_MAX || f.func == TSQ_AGG_MIN) && f.arg_type == TSQ_BYTES) { lanes_ok = false; break; }
        if (red.n == 4) { lanes_ok = false; break; }
        red.agg[red.n++] = i;
    }
    if (lanes_ok && red.n > 0 && tsq_knob(ctx, TSQ_KNOB_STREAMAGG_LANES, 1) != 0) {
        uint32_t kinds = 0;  // SA_K_* of the reducing aggregates, 4 bits each: the commonest plans have their own instantiation
        for (int q = 0; q < red.n; q++) {
            const tsq_agg_func& f = a->plan.f[red.agg[q]];
            const bool real = f.arg_type == TSQ_F32 || f.arg_type == TSQ_F64;
            const uint32_t k = f.func == TSQ_AGG_COUNT ? SA_K_COUNT : (f.func == TSQ_AGG_SUM || f.func == TSQ_AGG_AVG) ? (real ? SA_K_SUMR : SA_K_SUMI)
                               : f.func == TSQ_AGG_MAX ? SA_K_MAX : SA_K_MIN;
            kinds |= k << (4 * q);
        }
#define TSQ_SA_LAUNCH(NA, K) hipLaunchKernelGGL((k_sa_update_lanes<NA, K>), dim3(grid), dim3(TSQ_SA_NT), 0, ctx->stream, sa, red)
        switch (red.n * 0x10000u + kinds) {
            case 0x10001u: TSQ_SA_LAUNCH(1, 0x1u); break;    // COUNT
            case 0x10002u: TSQ_SA_LAUNCH(1, 0x2u); break;    // SUM / AVG (BIGINT)
            case 0x10003u: TSQ_SA_LAUNCH(1, 0x3u); break;    // SUM / AVG (DOUBLE)
            case 0x20012u: TSQ_SA_LAUNCH(2, 0x12u); break;   // SUM(BIGINT), COUNT — the C3 plan
            case 0x20013u: TSQ_SA_LAUNCH(2, 0x13u); break;   // SUM(DOUBLE), COUNT
            case 0x30412u: TSQ_SA_LAUNCH(3, 0x412u); break;  // SUM(BIGINT), COUNT, MAX
            case 0x30413u: TSQ_SA_LAUNCH(3, 0x413u); break;
            default:
                switch (red.n) {
                    case 1: TSQ_SA_LAUNCH(1, 0u); break;
                    case 2: TSQ_SA_LAUNCH(2, 0u); break;
                    case 3: TSQ_SA_LAUNCH(3, 0u); break;
                    default: TSQ_SA_LAUNCH(4, 0u); break;
                }
        }
#undef TSQ_SA_LAUNCH
    } else {
        hipLaunchKernelGGL(k_sa_update, dim3(grid), dim3(TSQ_SA_NT), 0, ctx->stream, sa);
    }
    TSQ_HIP(h, hipGetLastError());
    a->groups = (int64_t)need;
    a->st.kernel_launches += 3;
    a->stream_batches++;
    return TSQ_OK;
}

tsq_status wide_setup(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    if (a->wide_state) return TSQ_OK;
    a->wide_state = -1;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    WideFields& f = a->wide_f;
    memset(&f, 0, sizeof f);
    f.n = a->plan.n_keys;
    uint32_t need[TSQ_MAX_GROUP_KEYS], total = 0;
    uint64_t lo[TSQ_MAX_GROUP_KEYS], range[TSQ_MAX_GROUP_KEYS];
    bool nullable[TSQ_MAX_GROUP_KEYS];
    for (int k = 0; k < f.n; k++) {
        const int c = a->plan.key_col[k];
        DaMinMaxArgs ma;
        memset(&ma, 0, sizeof ma);
        ma.src.data = (const uint64_t*)in.data[c];
        ma.src.nulls = in.nulls[c];
        ma.src.nrows = nrows;
        ma.flip = a->cfg.group_key_type[k] == TSQ_I64 ? 0x8000000000000000ULL : 0ULL;
        ma.out = (unsigned long long*)(ctx->dscratch + 48);
        ctx->pinned[48] = ~0ULL;
        ctx->pinned[49] = 0;
        ctx->pinned[50] = 0;
        TSQ_HIP(h, hipMemcpyAsync(ctx->dscratch + 48, ctx->pinned + 48, 24, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_da_minmax, dim3(tsq_grid_for(ctx, nrows, 256)), dim3(256), 0, ctx->stream, ma);
        TSQ_HIP(h, hipGetLastError());
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 48, ctx->dscratch + 48, 24, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        a->st.kernel_launches++;
        lo[k] = range[k] = 0;
        if (ctx->pinned[50] != 0) {
            lo[k] = ctx->pinned[48] ^ ma.flip;
            range[k] = (ctx->pinned[49] ^ ma.flip) - lo[k];
        }
        // NULL gets a code whether or not this batch holds one (a later batch may: GROUP BY makes NULL a group, codec.go:718-719)
        nullable[k] = true;
        uint32_t w = 0;
        while (w < 64 && (range[k] >> w) != 0) w++;
        need[k] = w + 1;  // the observed range, doubled: room for the NULL code and for somewhat larger keys of later batches
        total += need[k];
    }
    if (total > 63) return TSQ_OK;
    // widen the fields while the word has room (later batches: order numbers keep growing, dates move on), at most 8 bits each
    uint32_t spare = 63 - total;
    for (int round = 0; round < 8 && spare; round++)
        for (int k = 0; k < f.n && spare; k++) {
            need[k]++;
            spare--;
        }
    uint32_t at = 0;
    for (int k = 0; k < f.n; k++) {
        const uint64_t cells = need[k] >= 64 ? ~0ull : ((1ull << need[k]) - 1ull);  // the last code is NULL's
        // the window starts a quarter of its slack below the smallest key seen (unless that would wrap around zero of the key's order)
        const uint64_t slack = (cells - 1 > range[k]) ? (cells - 1 - range[k]) / 4 : 0;
        const bool is_signed = a->cfg.group_key_type[k] == TSQ_I64;
        const uint64_t floor_key = is_signed ? 0x8000000000000000ULL : 0ULL;  // the smallest cell of the order, as a 64-bit word
        const uint64_t room = lo[k] - floor_key;                               // (wrapping subtraction: distance above the floor)
        f.kmin[k] = lo[k] - (slack < room ? slack : room);
        f.width[k] = need[k];
        f.shift[k] = at;
        f.nullcode[k] = nullable[k] ? cells : TSQ_WIDE_NO_NULL;
        f.maxd[k] = cells - 1;
        at += need[k];
    }
    // ---- the child: GROUP BY d (one BIGINT UNSIGNED key = input column n_input_cols), the same aggregates minus FIRST_ROW(key column)
    tsq_agg_cfg cc = a->cfg;
    if (cc.n_input_cols >= TSQ_MAX_COLS) return TSQ_OK;
    const int dcol = cc.n_input_cols;
    cc.input_types[dcol] = TSQ_U64;
    cc.n_input_cols = dcol + 1;
    cc.n_group_keys = 1;
    cc.group_key_col[0] = dcol;
    cc.group_key_type[0] = TSQ_U64;
    cc.n_aggs = 0;
    memset(&cc.aggs[0], 0, sizeof(tsq_agg_func));
    cc.aggs[0].func = TSQ_AGG_FIRSTROW;
    cc.aggs[0].mode = TSQ_MODE_COMPLETE;
    cc.aggs[0].arg_col = dcol;
    cc.aggs[0].arg_col2 = -1;
    cc.aggs[0].arg_type = TSQ_U64;
    cc.n_aggs = 1;
    int child_oc = 1, own_oc = 0;
    for (int i = 0; i < a->cfg.n_aggs; i++) {
        const tsq_agg_func& fn = a->cfg.aggs[i];
        const bool partial_out = fn.mode == TSQ_MODE_PARTIAL1 || fn.mode == TSQ_MODE_PARTIAL2;
        const int outs = (fn.func == TSQ_AGG_AVG && partial_out) ? 2 : 1;
        int key = -1;
        if (fn.func == TSQ_AGG_FIRSTROW)
            for (int k = 0; k < f.n; k++)
                if (fn.arg_col == a->plan.key_col[k]) key = k;
        if (key >= 0) {
            a->wide_child_out[own_oc++] = -1 - key;
            continue;
        }
        if (cc.n_aggs >= TSQ_MAX_AGGS) return TSQ_OK;
        cc.aggs[cc.n_aggs++] = fn;
        for (int o = 0; o < outs; o++) a->wide_child_out[own_oc++] = child_oc++;
    }
    tsq_agg* child = nullptr;
    const tsq_status cs = tsq_agg_create(ctx, &cc, &child);
    if (cs != TSQ_OK) return TSQ_OK;  // (a plan the single-key operator refuses: the several-column upsert keeps this aggregate)
    child->is_wide_child = true;
    child->fast_mode = a->fast_mode;
    child->host_mode = false;
    a->wide = child;
    a->wide_state = 1;
    return TSQ_OK;
}

tsq_status wide_batch(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    TSQ_TRY(a->wide_d.reserve(ctx, h, (size_t)nrows * 8 + 64));
    WideComposeArgs ca;
    memset(&ca, 0, sizeof ca);
    ca.f = a->wide_f;
    for (int k = 0; k < ca.f.n; k++) {
        ca.col[k] = (const uint64_t*)in.data[a->plan.key_col[k]];
        ca.nulls[k] = in.nulls[a->plan.key_col[k]];
    }
    ca.nrows = nrows;
    ca.d = a->wide_d.as<uint64_t>();
    ca.counts = (unsigned long long*)(ctx->dscratch + 60);
    TSQ_HIP(h, hipMemsetAsync(ctx->dscratch + 60, 0, 24, ctx->stream));
    hipLaunchKernelGGL(k_agg_wide_compose, dim3(tsq_grid_for(ctx, nrows, 256)), dim3(256), 0, ctx->stream, ca);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 60, ctx->dscratch + 60, 8, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    a->st.kernel_launches++;
    const int64_t n_exc = (int64_t)ctx->pinned[60];
    // the child's batch: the caller's columns + d
    tsq_agg* c = a->wide;
    tsq_colset cin = in;
    const int dcol = a->cfg.n_input_cols;
    cin.n = dcol + 1;
    cin.data[dcol] = a->wide_d.p;
    cin.nulls[dcol] = nullptr;
    cin.type[dcol] = TSQ_U64;
    cin.offs[dcol] = nullptr;
    a->wide_batches++;
    if (n_exc == 0) {
        TSQ_TRY(agg_batch(c, cin, nrows));
        if (c->hdr.err.size()) h->err = c->hdr.err;
        c->in_rows += nrows;
        return TSQ_OK;
    }
    // exception rows (a cell outside its field): they take the several-column upsert into this operator's own table, the others go
    // to the child as a row list
    a->wide_exception_rows += n_exc;
    TSQ_TRY(a->wide_okrows.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_TRY(a->wide_excrows.reserve(ctx, h, (size_t)n_exc * 4 + 64));
    WideListArgs la;
    memset(&la, 0, sizeof la);
    la.d = ca.d;
    la.nrows = nrows;
    la.ok_rows = a->wide_okrows.as<uint32_t>();
    la.exc_rows = a->wide_excrows.as<uint32_t>();
    la.cursors = (unsigned long long*)(ctx->dscratch + 61);
    hipLaunchKernelGGL(k_agg_wide_lists, dim3(tsq_grid_for(ctx, nrows, 256)), dim3(256), 0, ctx->stream, la);
    TSQ_HIP(h, hipGetLastError());
    a->st.kernel_launches++;
    if (nrows - n_exc > 0) {
        const tsq_status s = agg_rows(c, cin, nrows - n_exc, la.ok_rows);
        if (s != TSQ_OK) { h->err = c->hdr.err; return s; }
    }
    c->in_rows += nrows - n_exc;
    return agg_rows(a, in, n_exc, la.exc_rows);
}

// ---------------------------------------------------------------- GROUP BY through the dictionary of group keys (host side; tsq_keydict.h)
tsq_status kd_setup(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    (void)in;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    // partitions: at most ~8192 groups each on average (two thirds of a partition's places: + 45 sigma stays below TSQ_KR_CAP); the planner's
    // estimate, else as many groups as the first batch has rows
    const int64_t est = a->cfg.est_groups > 0 ? a->cfg.est_groups : nrows;
    uint32_t pbits = 0;
    while (((int64_t)TSQ_KR_FILL << pbits) < est && (1u << pbits) < TSQ_KR_MAXP) pbits++;
    if (nrows >= (1 << 16) && pbits < 8) pbits = 8;  // (one workgroup works on one partition: later batches may be much larger than the first)
    const size_t P = (size_t)1 << pbits;
    // ---- the child: GROUP BY id (input column 0) over the travelled argument columns (input columns 1 ..)
    tsq_agg_cfg cc = a->cfg;
    cc.n_input_cols = 1 + a->kd_npay;
    cc.input_types[0] = TSQ_U64;
    for (int v = 0; v < a->kd_npay; v++) cc.input_types[1 + v] = a->cfg.input_types[a->kd_paycol[v]];
    cc.n_group_keys = 1;
    cc.group_key_col[0] = 0;
    cc.group_key_type[0] = TSQ_U64;
    cc.est_groups = 0;
    memset(&cc.aggs[0], 0, sizeof(tsq_agg_func));
    cc.aggs[0].func = TSQ_AGG_FIRSTROW;
    cc.aggs[0].mode = TSQ_MODE_COMPLETE;
    cc.aggs[0].arg_col = 0;
    cc.aggs[0].arg_col2 = -1;
    cc.aggs[0].arg_type = TSQ_U64;
    cc.n_aggs = 1;
    auto child_col = [&](int c) {
        for (int v = 0; v < a->kd_npay; v++)
            if (a->kd_paycol[v] == c) return 1 + v;
        return -1;
    };
    int child_oc = 1, own_oc = 0;
    for (int i = 0; i < a->cfg.n_aggs; i++) {
        const tsq_agg_func& fn = a->cfg.aggs[i];
        const bool partial_out = fn.mode == TSQ_MODE_PARTIAL1 || fn.mode == TSQ_MODE_PARTIAL2;
        const int outs = (fn.func == TSQ_AGG_AVG && partial_out) ? 2 : 1;
        if (fn.func == TSQ_AGG_FIRSTROW) {
            int key = -1;
            for (int k = 0; k < a->plan.n_keys; k++)
                if (fn.arg_col == a->plan.key_col[k]) key = k;
            a->wide_child_out[own_oc++] = -1 - key;
            continue;
        }
        if (cc.n_aggs >= TSQ_MAX_AGGS) return TSQ_OK;
        tsq_agg_func g = fn;
        g.arg_col = fn.arg_col >= 0 ? child_col(fn.arg_col) : -1;
        g.arg_col2 = fn.arg_col2 >= 0 ? child_col(fn.arg_col2) : -1;
        cc.aggs[cc.n_aggs++] = g;
        for (int o = 0; o < outs; o++) a->wide_child_out[own_oc++] = child_oc++;
    }
    // the dictionary: P * 12288 places of 32 + 4 bytes.  No memory for it (a first batch of 1e8 rows without an estimate asks for ~7 GB)
    // means "route not usable", not a failed aggregate: the several-column upsert keeps the rows
    auto drop_dictionary = [&]() {
        for (DevBuf* b : {&a->kd_drec, &a->kd_dids, &a->kd_dcount, &a->kd_ctl}) b->release();
        h->err.clear();
        return TSQ_OK;
    };
    tsq_status rs = a->kd_drec.reserve(ctx, h, P * TSQ_KR_CAP * TSQ_KR_BYTES + 64);
    if (rs == TSQ_OK) rs = a->kd_dids.reserve(ctx, h, P * TSQ_KR_CAP * 4 + 64);
    if (rs == TSQ_OK) rs = a->kd_dcount.reserve(ctx, h, P * 4 + 64);
    if (rs == TSQ_OK) rs = a->kd_ctl.reserve(ctx, h, 64);
    if (rs != TSQ_OK) return drop_dictionary();
    TSQ_HIP(h, hipMemsetAsync(a->kd_dcount.p, 0, P * 4, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(a->kd_ctl.p, 0, 64, ctx->stream));
    tsq_agg* child = nullptr;
    const tsq_status cs = tsq_agg_create(ctx, &cc, &child);
    if (cs != TSQ_OK) return drop_dictionary();  // (a plan the single-key operator refuses: the several-column upsert keeps this aggregate)
    child->is_wide_child = true;
    child->fast_mode = a->fast_mode;
    child->host_mode = false;
    a->wide = child;
    a->wide_state = 1;
    a->kd_mode = true;
    a->kd_pbits = pbits;
    a->kd_next_host = 0;
    return TSQ_OK;
}

tsq_status kd_batch(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const uint32_t P = 1u << a->kd_pbits;
    // ---- the rows' key records, partitioned by their hash; the argument cells travel
    KrArgs ka;
    memset(&ka, 0, sizeof ka);
    ka.src.cs = in;
    ka.src.n_keys = a->plan.n_keys;
    for (int k = 0; k < a->plan.n_keys; k++) ka.src.col[k] = a->plan.key_col[k];
    ka.src.keep_nulls = 1;
    ka.src.layout = kr_layout_of(a->cfg.group_key_type, a->plan.n_keys);
    ka.src.nrows = nrows;
    ka.pbits = a->kd_pbits;
    const int64_t chunks = (nrows + TSQ_KR_NT - 1) / TSQ_KR_NT;
    ka.n_wg = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(1024, std::max<int64_t>(8, tsq_knob(ctx, TSQ_KNOB_KR_WG, TSQ_KR_MAXWG))), chunks));
    ka.rows_per_wg = ((chunks + ka.n_wg - 1) / ka.n_wg) * TSQ_KR_NT;
    TSQ_TRY(a->kd_counts.reserve(ctx, h, (size_t)ka.n_wg * P * 4 + 64));
    TSQ_TRY(a->kd_pstart.reserve(ctx, h, ((size_t)P + 1) * 4 + 64));
    const bool slots = a->kd_npay <= 3 && tsq_knob(ctx, TSQ_KNOB_KEYREC, 1) != 2;  // (knob value 2: separate arrays, for A/B and the tests)
    TSQ_TRY(a->kd_rec.reserve(ctx, h, (size_t)nrows * (slots ? 64 : TSQ_KR_BYTES) + 64));
    TSQ_TRY(a->kd_ids.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_TRY(a->kd_norec.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_TRY(a->kd_flags.reserve(ctx, h, 64));
    bool any_nulls = false;
    ka.n_pay = a->kd_npay;
    for (int v = 0; v < a->kd_npay; v++) {
        TSQ_TRY(a->kd_pay[v].reserve(ctx, h, (size_t)nrows * 8 + 64));
        ka.pay_src[v] = (const uint64_t*)in.data[a->kd_paycol[v]];
        ka.pay_nulls[v] = in.nulls[a->kd_paycol[v]];
        ka.pay_dst[v] = a->kd_pay[v].as<uint64_t>();
        any_nulls = any_nulls || in.nulls[a->kd_paycol[v]] != nullptr;
    }
    if (any_nulls) {
        TSQ_TRY(a->kd_paynn.reserve(ctx, h, (size_t)nrows + 64));
        ka.pay_nn = a->kd_paynn.as<uint8_t>();
    }
    unsigned long long* ctl = a->kd_ctl.as<unsigned long long>();
    ka.counts = a->kd_counts.as<uint32_t>();
    ka.pstart = a->kd_pstart.as<uint32_t>();
    ka.rec = a->kd_rec.as<unsigned long long>();
    ka.ids = a->kd_ids.as<uint32_t>();
    if (slots) ka.slot = ka.rec;
    ka.flags = a->kd_flags.as<uint32_t>();
    ka.norec = a->kd_norec.as<uint32_t>();
    ka.norec_count = ctl + 2;
    TSQ_HIP(h, hipMemsetAsync(ka.flags, 0, 16, ctx->stream));
    TSQ_HIP(h, hipMemsetAsync(ctl + 1, 0, 32, ctx->stream));  // [1] exception rows, [2] rows without a record, [3..4] list cursors
    const size_t lds = (size_t)P * 4;
    TSQ_HIP(h, hipFuncSetAttribute((const void*)k_kr_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TSQ_HIP(h, hipFuncSetAttribute((const void*)k_kr_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_kr_hist, dim3(ka.n_wg), dim3(TSQ_KR_NT), lds, ctx->stream, ka);
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_kr_offsets, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, ka);
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_kr_scan, dim3(1), dim3(1024), 0, ctx->stream, ka.pstart, P, ka.flags);
    TSQ_HIP(h, hipGetLastError());
    hipLaunchKernelGGL(k_kr_scatter, dim3(ka.n_wg), dim3(TSQ_KR_NT), lds, ctx->stream, ka);
    TSQ_HIP(h, hipGetLastError());
    // ---- records -> group ids (new keys enter the dictionary)
    TSQ_TRY(a->kd_dloc.reserve(ctx, h, (size_t)(a->kd_next_host + nrows) * 4 + 64, true, (size_t)a->kd_next_host * 4));
    TSQ_TRY(a->kd_ridx.reserve(ctx, h, (size_t)nrows * 4 + 64));
    TSQ_TRY(a->kd_gid.reserve(ctx, h, (size_t)nrows * 8 + 64));
    KdArgs da;
    memset(&da, 0, sizeof da);
    da.prec = ka.rec;
    da.stride = slots ? 8u : 4u;
    if (slots) {
        da.n_pay = a->kd_npay;
        for (int v = 0; v < a->kd_npay; v++) da.pay_dst[v] = a->kd_pay[v].as<uint64_t>();
        da.pay_nn = any_nulls ? a->kd_paynn.as<uint8_t>() : nullptr;
        da.ids = ka.ids;
    }
    da.pstart = ka.pstart;
    da.P = P;
    da.drec = a->kd_drec.as<unsigned long long>();
    da.dids = a->kd_dids.as<uint32_t>();
    da.dcount = a->kd_dcount.as<uint32_t>();
    da.dloc = a->kd_dloc.as<uint32_t>();
    da.next_id = ctl;
    da.ridx = a->kd_ridx.as<uint32_t>();
    da.gid = a->kd_gid.as<uint64_t>();
    da.counters = ctl + 1;
    hipLaunchKernelGGL(k_kd_assign, dim3(std::min<uint32_t>(P, (uint32_t)ctx->num_cus * 2)), dim3(TSQ_KD_NT), 0, ctx->stream, da);
    TSQ_HIP(h, hipGetLastError());
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 60, ctl, 24, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    a->st.kernel_launches += 5;
    a->kd_next_host = (int64_t)ctx->pinned[60];
    const int64_t n_exc = (int64_t)ctx->pinned[61], n_norec = (int64_t)ctx->pinned[62];
    const int64_t n_rec = nrows - n_norec;
    // ---- the child's batch: ids + the travelled columns, in partition order
    tsq_agg* c = a->wide;
    tsq_colset cin;
    memset(&cin, 0, sizeof cin);
    cin.n = 1 + a->kd_npay;
    cin.data[0] = a->kd_gid.p;
    cin.type[0] = TSQ_U64;
    for (int v = 0; v < a->kd_npay; v++) {
        cin.data[1 + v] = a->kd_pay[v].p;
        cin.type[1 + v] = a->cfg.input_types[a->kd_paycol[v]];
        if (in.nulls[a->kd_paycol[v]] && n_rec > 0) {  // the NOT-NULL bits of the column, in partition order
            TSQ_TRY(a->kd_paybm[v].reserve(ctx, h, tsq_bitmap_bytes(n_rec) + 64));
            hipLaunchKernelGGL(k_kd_nn_bitmap, dim3(tsq_grid_for(ctx, (n_rec + 7) / 8, 256)), dim3(256), 0, ctx->stream, (const uint8_t*)a->kd_paynn.p, n_rec, (uint32_t)v,
                               a->kd_paybm[v].as<uint8_t>());
            TSQ_HIP(h, hipGetLastError());
            cin.nulls[1 + v] = a->kd_paybm[v].as<uint8_t>();
            a->st.kernel_launches++;
        }
    }
    a->wide_batches++;
    c->cfg.est_groups = std::max<int64_t>(1, a->kd_next_host);  // (the child need not learn its cardinality from a prefix of the batch)
    c->da_hint = true;  // ... and its keys are the ids 0 .. next - 1 (the packed window is the power of two above)
    c->da_hint_min = 0;
    c->da_hint_max = (uint64_t)std::max<int64_t>(1, a->kd_next_host) - 1;
    // ... nor find its table too small in the middle of a merge (a failed merge is a second merge): every id is a group of the child
    if (!c->multi && c->tb.cap < (uint64_t)a->kd_next_host * 2) {
        const tsq_status gs = grow_table(c, (uint64_t)a->kd_next_host * 3 + 16);
        if (gs != TSQ_OK) { h->err = c->hdr.err; return gs; }
    }
    if (n_exc == 0) {
        if (n_rec > 0) {
            const tsq_status s = agg_batch(c, cin, n_rec);
            if (s != TSQ_OK) { h->err = c->hdr.err; return s; }
            c->in_rows += n_rec;
        }
    } else {
        // rows whose partition of the dictionary is full: the several-column upsert into this operator's own table (by source row); the others
        // go to the child as a list of positions
        TSQ_TRY(a->wide_okrows.reserve(ctx, h, (size_t)n_rec * 4 + 64));
        TSQ_TRY(a->wide_excrows.reserve(ctx, h, (size_t)n_exc * 4 + 64));
        KdListArgs la;
        memset(&la, 0, sizeof la);
        la.gid = da.gid;
        la.ids = ka.ids;
        la.n = n_rec;
        la.ok_pos = a->wide_okrows.as<uint32_t>();
        la.exc_rows = a->wide_excrows.as<uint32_t>();
        la.cursors = ctl + 3;
        hipLaunchKernelGGL(k_kd_lists, dim3(tsq_grid_for(ctx, n_rec, 256)), dim3(256), 0, ctx->stream, la);
        TSQ_HIP(h, hipGetLastError());
        a->st.kernel_launches++;
        if (n_rec - n_exc > 0) {
            const tsq_status s = agg_rows(c, cin, n_rec - n_exc, la.ok_pos);
            if (s != TSQ_OK) { h->err = c->hdr.err; return s; }
        }
        c->in_rows += n_rec - n_exc;
        TSQ_TRY(agg_rows(a, in, n_exc, la.exc_rows));
    }
    a->wide_exception_rows += n_exc + n_norec;
    if (n_norec > 0) TSQ_TRY(agg_rows(a, in, n_norec, ka.norec));  // key cells that do not fit a record
    return TSQ_OK;
}

// ---- partitioned groups: small batches wait (every pass rewrites the whole state)
tsq_status agg_batch(tsq_agg* a, const tsq_colset& in, int64_t nrows);
tsq_status pg_pend_flush(tsq_agg* a) {
    if (a->pg_pend_rows == 0) return TSQ_OK;
    tsq_colset in;
    memset(&in, 0, sizeof in);
    in.n = a->cfg.n_input_cols;
    for (int c = 0; c < in.n; c++) {
        const ColStore& cs = a->pg_pend[c];
        in.type[c] = a->cfg.input_types[c];
        in.data[c] = cs.rows ? cs.data.p : nullptr;  // (columns the plan does not read were not kept)
        in.nulls[c] = (cs.rows && cs.has_nulls) ? cs.nulls.as<uint8_t>() : nullptr;
    }
    const int64_t n = a->pg_pend_rows;
    a->pg_in_flush = true;
    const tsq_status s = agg_batch(a, in, n);
    a->pg_in_flush = false;
    for (auto& cs : a->pg_pend) cs.clear();
    a->pg_pend_rows = 0;
    return s;
}
tsq_status pg_pend_batch(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    tsq_ctx* ctx = a->ctx;
    const AfPlan& pl = a->fplan;
    const int64_t slots = (int64_t)af_slots(pl) << (a->pg_pbits + a->pg_sbits);
    // rows that make a pass over the state worth it (tests that force the mode, knob >= 2, send every batch straight through: the keys of a
    // batch must meet their groups of the earlier ones)
    const int64_t worth = tsq_knob(ctx, TSQ_KNOB_AGG_PG, 1) >= 2 ? 0 : std::max<int64_t>(4 << 20, slots / 2);
    if (a->pg_pend_rows == 0 && nrows >= worth) {  // a big batch: straight through
        a->pg_in_flush = true;
        const tsq_status s = agg_batch(a, in, nrows);
        a->pg_in_flush = false;
        return s;
    }
    if (a->pg_pend_rows + nrows >= ((int64_t)1 << 31)) TSQ_TRY(pg_pend_flush(a));
    if (a->pg_pend.empty()) {
        a->pg_pend.resize(a->cfg.n_input_cols);
        for (int c = 0; c < a->cfg.n_input_cols; c++) a->pg_pend[c].type = a->cfg.input_types[c];
    }
    // the columns the plan reads: the key column and the argument columns
    bool used[TSQ_MAX_COLS] = {};
    used[pl.key_col] = true;
    for (int v = 0; v < pl.V; v++) used[pl.vcol[v]] = true;
    DevBuf tmp;
    for (int c = 0; c < a->cfg.n_input_cols; c++) {
        if (!used[c]) continue;
        ColStore& cs = a->pg_pend[c];
        if (cs.rows != a->pg_pend_rows) { tmp.release(); return tsq_fail(&a->hdr, TSQ_ERR_HIP, "internal: pending columns out of step"); }
        const tsq_status s = tsq_col_append(ctx, &a->hdr, cs, in.data[c], in.nulls[c], nrows, true, tmp);
        if (s != TSQ_OK) { tmp.release(); return s; }
    }
    tmp.release();
    a->pg_pend_rows += nrows;
    if (a->pg_pend_rows >= worth) return pg_pend_flush(a);
    return TSQ_OK;
}

tsq_status agg_batch(tsq_agg* a, const tsq_colset& in, int64_t nrows) {
    if (nrows == 0) return TSQ_OK;
    if (a->stream) return stream_batch(a, in, nrows);
    // several integer key columns wider than the packed route takes: one composite key for a single-key child (wide_setup).  The
    // first batch decides; small first batches (below 64 Ki rows) wait for a bigger one unless the fast paths are FORCEd (tests)
    if (a->wide_ok && a->wide_state >= 0 && a->fast_mode != TSQ_AGGFAST_OFF && nrows < 0x7fffffffLL) {
        if (a->wide_state == 0 && a->groups == 0 && a->in_rows == 0 && (nrows >= (1 << 16) || a->fast_mode == TSQ_AGGFAST_FORCE)) {
            // the packed several-column route (<= 23 bits of fields) is tried first by agg_batch_fast: here only what it cannot take
            bool packed_fits = false;
            if (a->fast_ok && a->mk_n > 1 && (nrows >= (1 << 20) || a->fast_mode == TSQ_AGGFAST_FORCE)) {
                TSQ_TRY(da_agg_setup(a, in, nrows));  // (what agg_batch_fast would do with this batch: idempotent)
                packed_fits = a->da_state == 1;
            }
            if (!packed_fits) TSQ_TRY(wide_setup(a, in, nrows));
            else a->wide_state = -1;
        } else if (a->wide_state == 0 && (a->groups > 0 || a->in_rows > 0)) {
            a->wide_state = -1;  // rows already live in the several-column table: keep one table
        }
        if (a->wide_state == 1) return wide_batch(a, in, nrows);
    }
    // string keys / key columns the composite word cannot hold: the dictionary of group keys + a child GROUP BY id (kd_setup)
    // (integer key columns: only what neither the packed several-column route nor the composite word can hold)
    bool kd_turn = a->kd_ok && a->wide_state <= 0;
    if (kd_turn && !a->has_str_key) kd_turn = a->wide_ok && a->wide_state == -1 && a->da_state != 1;
    if (kd_turn && a->fast_mode != TSQ_AGGFAST_OFF && nrows < 0x7fffffffLL) {
        if (a->groups == 0 && a->in_rows == 0 && (nrows >= (1 << 16) || a->fast_mode == TSQ_AGGFAST_FORCE)) TSQ_TRY(kd_setup(a, in, nrows));
        a->kd_ok = a->kd_mode;  // (rows already live in the several-column table, or the setup refused: keep one table)
    }
    if (a->kd_mode) return kd_batch(a, in, nrows);
    if (a->pg_state == 1 && nrows < 0x7fffffffLL) {  // the groups live in the partitioned sub-tables: every batch goes there, whatever its size
        if (!a->pg_in_flush) return pg_pend_batch(a, in, nrows);
        bool done = false;
        TSQ_TRY(agg_batch_fast(a, in, nrows, 1, &done));
        if (!done) return tsq_fail(&a->hdr, TSQ_ERR_HIP, "internal: a batch of the partitioned groups was not taken");
        return TSQ_OK;
    }
    const bool want_fast = a->fast_ok && a->fast_mode != TSQ_AGGFAST_OFF && nrows < 0x7fffffffLL &&
                           (a->fast_mode == TSQ_AGGFAST_FORCE || nrows >= (1 << 20));
    if (!want_fast) return agg_rows(a, in, nrows, nullptr);
    // cardinality: what the table has seen so far, else the planner's estimate, else learn it from a prefix
    int64_t done_rows = 0;
    int64_t est = a->cfg.est_groups > 0 ? a->cfg.est_groups : 0;
    const int64_t seen_rows = a->in_rows;
    if (seen_rows + done_rows >= (1 << 20) || (a->fast_mode == TSQ_AGGFAST_FORCE && a->groups > 0)) est = std::max<int64_t>(a->groups, 1);
    if (est == 0 && a->pg_state == 0 && a->mk_n <= 1 && a->plan.n_keys == 1 && a->groups == 0 && a->in_rows == 0 && nrows >= (1 << 20) && tsq_knob(a->ctx, TSQ_KNOB_AGG_PG, 1) != 0) {
        // no estimate: 8192 keys of the batch tell "about as many groups as rows" from what the LDS modes take (k_pg_sample) — without a
        // prefix in the table, which would then hold groups that the sub-tables hold too
        tsq_ctx* ctx = a->ctx;
        AfPgSampleArgs sa;
        memset(&sa, 0, sizeof sa);
        sa.src.data = in.data[a->fplan.key_col];
        sa.src.nulls = in.nulls[a->fplan.key_col];
        sa.src.type = in.type[a->fplan.key_col];
        sa.src.nrows = nrows;
        sa.src.key_kind = 1;
        sa.out = (uint32_t*)(ctx->dscratch + 40);
        hipLaunchKernelGGL(k_pg_sample, dim3(1), dim3(1024), 0, ctx->stream, sa);
        TSQ_HIP(&a->hdr, hipGetLastError());
        TSQ_HIP(&a->hdr, hipMemcpyAsync(ctx->pinned + 40, ctx->dscratch + 40, 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(&a->hdr, hipStreamSynchronize(ctx->stream));
        a->st.kernel_launches++;
        const double n = (double)((const uint32_t*)(ctx->pinned + 40))[0], d = (double)((const uint32_t*)(ctx->pinned + 40))[1];
        if (n >= 4096.0) {
            const double ndv = d > 0 ? n * n / (2.0 * d) : 4.0 * (double)nrows;
            if (!pg_h_fits(a->fplan, (int64_t)std::min<double>(ndv, 9e18)))  // (a planner would pass this on as est_groups)
                est = (int64_t)std::min<double>(std::min<double>(ndv * 1.5, (double)nrows), 9e18);
            else if (8.0 * (n - d) <= n)
                // at most an eighth of the sampled keys were new (so at most about an eighth of the rows belong to groups the sample did not
                // see: Good-Turing): the sample has seen most of the groups (uniform keys: N ~ n - d) — twice
                // the distinct keys of the sample is the estimate, and the 1 Mi-row prefix through the row upsert that used to learn it
                // (121 us of BenchmarkAggRows' 0.41 ms: a thousand groups take a million same-address atomics) is not needed.  An
                // estimate that is too low costs speed only: rows that find their LDS table full leave as one-row partial groups
                est = std::max<int64_t>(1, (int64_t)(2.0 * (n - d)));
        }
    }
    if (est == 0 && a->fast_mode != TSQ_AGGFAST_FORCE) {
        const int64_t prefix = std::min<int64_t>(nrows, 1 << 20);  // multiple of 8 rows: bitmap slices stay byte aligned
        TSQ_TRY(agg_rows(a, in, prefix, nullptr));
        done_rows = prefix;
        est = std::max<int64_t>(a->groups, 1);
        if (done_rows == nrows) return TSQ_OK;
    }
    if (est == 0) est = 1;
    tsq_colset rest;
    tsq_colset_slice(rest, in, done_rows);
    bool done = false;
    TSQ_TRY(agg_batch_fast(a, rest, nrows - done_rows, est, &done));
    if (!done) TSQ_TRY(agg_rows(a, rest, nrows - done_rows, nullptr));
    return TSQ_OK;
}

// appends n rows of var-len input column c (pinned staging or the caller's device column) to its heap and pads the heap to
// a multiple of 8 rows; *row0 = heap row of the first appended row
tsq_status heap_append(tsq_agg* a, int c, const void* data, const int64_t* offsets, const uint8_t* bitmap, int64_t n, bool src_dev, DevBuf& t1,
                       DevBuf& t2, int64_t* row0) {
    ColStore& hs = a->heap[c];
    *row0 = hs.rows;
    TSQ_TRY(tsq_col_append_varlen(a->ctx, &a->hdr, hs, data, offsets, bitmap, n, src_dev, t1, t2));
    if (hs.nbytes >= (1ll << 40)) return tsq_fail(&a->hdr, TSQ_ERR_UNSUPPORTED, "aggregate: more than 1 TiB of string cells in one column");
    const int64_t pad = (8 - (hs.rows & 7)) & 7;
    if (pad) {
        static const int64_t zeros[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        TSQ_TRY(tsq_col_append_varlen(a->ctx, &a->hdr, hs, zeros, zeros, nullptr, pad, false, t1, t2));
    }
    return TSQ_OK;
}
// points the var-len columns of a batch at their heaps (data = heap base, offsets absolute, bitmap from the batch's first row)
void colset_use_heap(const tsq_agg* a, tsq_colset& in, int c, int64_t row0) {
    const ColStore& hs = a->heap[c];
    in.data[c] = hs.data.p;
    in.offs[c] = hs.offs.as<int64_t>() + row0;
    in.nulls[c] = hs.has_nulls ? hs.nulls.as<uint8_t>() + (row0 >> 3) : nullptr;
    in.type[c] = TSQ_BYTES;
}

// between two batches: a heap that has outgrown its mark keeps only the strings the group table refers to
tsq_status agg_heap_gc(tsq_agg* a) {
    if (!a->has_str) return TSQ_OK;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    const bool gc_env = ctx->knob[TSQ_KNOB_AGG_HEAP_GC_BYTES] != TSQ_KNOB_DEFAULT;  // (test knob, read per call: a compaction after every batch)
    const int64_t gc_min = tsq_knob(ctx, TSQ_KNOB_AGG_HEAP_GC_BYTES, (int64_t)(256 << 20));
    if (a->heap_gc_at.empty()) a->heap_gc_at.assign(a->heap.size(), gc_min);
    if (gc_env)
        for (auto& m : a->heap_gc_at) m = std::min<int64_t>(m, gc_min);
    for (size_t c = 0; c < a->heap.size(); c++) {
        ColStore& hs = a->heap[c];
        if (hs.type != TSQ_BYTES || hs.nbytes <= a->heap_gc_at[c]) continue;
        HeapGcArgs ga;
        memset(&ga, 0, sizeof ga);
        AggTable t;
        fill_agg_table(a, a->tb, t);
        ga.tag = t.tag;
        ga.nslots = t.cap + 2;
        ga.gknull = t.gknull;
        ga.heap = (const uint8_t*)hs.data.p;
        for (int k = 0; k < a->plan.n_keys; k++)
            if (a->cfg.group_key_type[k] == TSQ_BYTES && a->plan.key_col[k] == (int)c) {
                ga.ref[ga.n_fields] = t.gkey[k];
                ga.nullbit[ga.n_fields] = k;
                ga.n_fields++;
            }
        for (int i = 0; i < a->plan.n_aggs; i++) {
            const tsq_agg_func& f = a->plan.f[i];
            if (f.arg_type != TSQ_BYTES || f.arg_col != (int)c) continue;
            if (f.func != TSQ_AGG_FIRSTROW && f.func != TSQ_AGG_MAX && f.func != TSQ_AGG_MIN) continue;
            ga.ref[ga.n_fields] = t.st[i].acc;
            ga.seen[ga.n_fields] = t.st[i].seen;
            ga.nullbit[ga.n_fields] = -1;
            ga.n_fields++;
        }
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        int64_t live = 0;
        DevBuf nd, offs, scratch;
        if (ga.n_fields > 0) {
            const uint64_t n = ga.nslots * (uint64_t)ga.n_fields;
            tsq_status s = offs.reserve(ctx, h, (n + 1) * 8 + 64);
            if (s != TSQ_OK) return s;
            ga.offs = offs.as<int64_t>();
            hipLaunchKernelGGL(k_heap_gc_len, dim3(tsq_grid_for(ctx, (int64_t)n, 256)), dim3(256), 0, ctx->stream, ga);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { offs.release(); return tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e)); }
            s = tsq_launch_scan64(ctx, h, ga.offs, (int64_t)n, scratch);  // exclusive, and offs[n] = the live bytes
            if (s == TSQ_OK) {
                e = hipMemcpyAsync(ctx->pinned + 43, ga.offs + n, 8, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
            }
            if (s == TSQ_OK) {
                live = (int64_t)ctx->pinned[43];
                s = nd.reserve(ctx, h, (size_t)live + 64);
            }
            if (s == TSQ_OK) {
                ga.new_heap = nd.as<uint8_t>();
                hipLaunchKernelGGL(k_heap_gc_move, dim3(tsq_grid_for(ctx, (int64_t)n, 256)), dim3(256), 0, ctx->stream, ga);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
            }
            offs.release();
            scratch.release();
            if (s != TSQ_OK) { nd.release(); return s; }
            a->st.kernel_launches += 3;
        }
        // the compacted heap: the live strings, no rows (the next batch appends behind them, tsq_col_append_varlen)
        if (ga.n_fields > 0) {
            hs.data.release();
            hs.data = nd;  // shallow move of the buffer handle
        }
        hs.nbytes = live;
        hs.rows = 0;
        a->heap_gc_at[c] = std::max<int64_t>(gc_min, 2 * live);
        a->heap_gcs++;
    }
    return TSQ_OK;
}

tsq_status agg_flush(tsq_agg* a) {
    HostStage& sg = a->stage;
    if (sg.staged == 0) return TSQ_OK;
    if (sg.failed) return tsq_fail(&a->hdr, TSQ_ERR_OOM_DEVICE, "pinned staging for a var-len column could not grow");
    DevBuf tmp, tmp2;
    std::vector<int64_t> row0(a->icols.size(), 0);
    for (size_t c = 0; c < a->icols.size(); c++) {
        tsq_status s;
        if (a->icols[c].type == TSQ_BYTES) {
            s = heap_append(a, (int)c, sg.data[c].p, (const int64_t*)sg.offs[c].p, sg.bitmap((int)c), sg.staged, false, tmp, tmp2, &row0[c]);
            a->st.h2d_bytes += sg.nbytes[c] + sg.staged * 8;
        } else {
            a->icols[c].rows = 0;
            a->icols[c].has_nulls = false;
            s = tsq_col_append(a->ctx, &a->hdr, a->icols[c], sg.data[c].p, sg.bitmap((int)c), sg.staged, false, tmp);
            a->st.h2d_bytes += sg.staged * a->icols[c].elem();
        }
        if (s != TSQ_OK) { tmp.release(); tmp2.release(); return s; }
    }
    tsq_colset in;
    tsq_fill_colset(in, a->icols);
    for (size_t c = 0; c < a->icols.size(); c++)
        if (a->icols[c].type == TSQ_BYTES) colset_use_heap(a, in, (int)c, row0[c]);
    tsq_status s = agg_batch(a, in, sg.staged);
    hipError_t e = hipStreamSynchronize(a->ctx->stream);
    tmp.release();
    tmp2.release();
    a->in_rows += sg.staged;
    sg.reset();
    if (s != TSQ_OK) return s;
    if (e != hipSuccess) return tsq_fail(&a->hdr, TSQ_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return agg_heap_gc(a);
}

}  // namespace

TSQ_API tsq_status tsq_agg_create(tsq_ctx* ctx, const tsq_agg_cfg* cfg, tsq_agg** out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !cfg || !out) return tsq_fail(nullptr, TSQ_ERR_INVALID, "tsq_agg_create: NULL argument");
    *out = nullptr;
    tsq_handle_hdr* ch = &ctx->hdr;
    if (cfg->n_group_keys < 0 || cfg->n_group_keys > TSQ_MAX_GROUP_KEYS) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "0..4 group keys supported");
    if (cfg->n_aggs < 1 || cfg->n_aggs > TSQ_MAX_AGGS) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "1..16 aggregate functions supported");
    if (cfg->n_input_cols < 1 || cfg->n_input_cols > TSQ_MAX_COLS) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "1..16 input columns supported");
    bool has_str = false;
    for (int c = 0; c < cfg->n_input_cols; c++) {
        if (cfg->input_types[c] < TSQ_I64 || cfg->input_types[c] > TSQ_BYTES) return tsq_fail(ch, TSQ_ERR_INVALID, "unknown input column type");
        has_str |= cfg->input_types[c] == TSQ_BYTES;
    }
    for (int k = 0; k < cfg->n_group_keys; k++) {
        if (cfg->group_key_col[k] < 0 || cfg->group_key_col[k] >= cfg->n_input_cols) return tsq_fail(ch, TSQ_ERR_INVALID, "group key column out of range");
        if (cfg->group_key_type[k] != cfg->input_types[cfg->group_key_col[k]]) return tsq_fail(ch, TSQ_ERR_INVALID, "group key type mismatch");
    }
    std::unique_ptr<tsq_agg> a(new tsq_agg());
    a->hdr.magic = TSQ_MAGIC_AGG;
    a->ctx = ctx;
    a->cfg = *cfg;
    if (a->cfg.max_chunk_size <= 0) a->cfg.max_chunk_size = 1024;
    a->plan.n_keys = cfg->n_group_keys;
    a->plan.n_aggs = cfg->n_aggs;
    for (int k = 0; k < cfg->n_group_keys; k++) a->plan.key_col[k] = cfg->group_key_col[k];
    a->multi = cfg->n_group_keys > 1;
    for (int k = 0; k < cfg->n_group_keys; k++) a->multi |= cfg->group_key_type[k] == TSQ_BYTES;  // a string key is verified by its bytes
    a->has_str = has_str;
    {
        const int64_t tb = tsq_knob(ctx, TSQ_KNOB_AGG_TAG_BITS, 0);
        a->test_tag_bits = (tb > 0 && tb < 64) ? (uint32_t)tb : 0;
    }
    for (int i = 0; i < cfg->n_aggs; i++) {
        const tsq_agg_func& f = cfg->aggs[i];
        if (f.func < TSQ_AGG_COUNT || f.func > TSQ_AGG_FIRSTROW) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "unknown aggregate function");
        if (f.mode < TSQ_MODE_COMPLETE || f.mode > TSQ_MODE_PARTIAL2) return tsq_fail(ch, TSQ_ERR_INVALID, "bad aggregate mode");
        const bool merge = f.mode == TSQ_MODE_FINAL || f.mode == TSQ_MODE_PARTIAL2;
        if (f.arg_col >= cfg->n_input_cols || f.arg_col < -1) return tsq_fail(ch, TSQ_ERR_INVALID, "aggregate argument column out of range");
        if (f.arg_col < 0 && !(f.func == TSQ_AGG_COUNT && !merge)) return tsq_fail(ch, TSQ_ERR_INVALID, "only COUNT may take a constant argument");
        if (f.func == TSQ_AGG_AVG && merge && (f.arg_col2 < 0 || f.arg_col2 >= cfg->n_input_cols))
            return tsq_fail(ch, TSQ_ERR_INVALID, "AVG in final mode needs (count, sum) columns");
        if (f.arg_type < TSQ_I64 || f.arg_type > TSQ_BYTES) return tsq_fail(ch, TSQ_ERR_INVALID, "unknown aggregate argument type");
        if (f.arg_type == TSQ_BYTES && (f.func == TSQ_AGG_SUM || f.func == TSQ_AGG_AVG))
            return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "SUM/AVG of a string argument (the planner casts it to double first): fall back to the Go operator");
        if (f.func != TSQ_AGG_COUNT) {
            const int vc = (f.func == TSQ_AGG_AVG && merge) ? f.arg_col2 : f.arg_col;
            if (cfg->input_types[vc] != f.arg_type) return tsq_fail(ch, TSQ_ERR_INVALID, "aggregate arg_type does not match its input column");
        }
        a->plan.f[i] = f;
        // output schema
        const bool partial_out = f.mode == TSQ_MODE_PARTIAL1 || f.mode == TSQ_MODE_PARTIAL2;
        const bool real = f.arg_type == TSQ_F32 || f.arg_type == TSQ_F64;
        switch (f.func) {
            case TSQ_AGG_COUNT: a->out_types.push_back(TSQ_I64); break;
            case TSQ_AGG_SUM: a->out_types.push_back(real ? TSQ_F64 : TSQ_I64); break;  // base_func.go:119-131
            case TSQ_AGG_AVG:
                if (partial_out) a->out_types.push_back(TSQ_I64);
                a->out_types.push_back(real ? TSQ_F64 : TSQ_I64);
                break;
            default: a->out_types.push_back(f.arg_type); break;
        }
        a->out_arg_col.resize(a->out_types.size(), -1);
        if (f.func != TSQ_AGG_COUNT && f.func != TSQ_AGG_SUM && f.func != TSQ_AGG_AVG && f.arg_type == TSQ_BYTES) a->out_arg_col.back() = f.arg_col;
    }
    a->n_out = (int)a->out_types.size();
    {   // LDS pre-aggregation plan: one group key, raw-argument modes, every aggregate expressible in <= 5 LDS words
        AfPlan& fp = a->fplan;
        memset(&fp, 0, sizeof fp);
        bool ok = cfg->n_group_keys >= 1 && cfg->n_group_keys <= TSQ_DAAGG_MAXK && !has_str;  // LDS words are fixed width
        fp.n_aggs = cfg->n_aggs;
        if (ok) {
            fp.key_col = cfg->group_key_col[0];
            fp.key_type = cfg->group_key_type[0];
        }
        // several key columns: integers only, and only through the packed route (their cells become the fields of one word)
        a->mk_n = ok && cfg->n_group_keys > 1 ? cfg->n_group_keys : 0;
        for (int k = 0; k < a->mk_n && ok; k++) {
            a->mk_col[k] = cfg->group_key_col[k];
            if (cfg->group_key_type[k] != TSQ_I64 && cfg->group_key_type[k] != TSQ_U64) ok = false;
        }
        memset(a->da_keys.fr_key, 0xff, sizeof a->da_keys.fr_key);
        for (int i = 0; i < cfg->n_aggs && ok; i++) {
            const tsq_agg_func& f = cfg->aggs[i];
            AfAgg& g = fp.f[i];
            g.func = f.func;
            g.type = f.arg_type;
            g.v = g.w = -1;
            if (f.mode != TSQ_MODE_COMPLETE && f.mode != TSQ_MODE_PARTIAL1) { ok = false; break; }
            if (f.func == TSQ_AGG_FIRSTROW) {  // only firstrow(group key): its value is the key itself
                if (a->mk_n) {
                    ok = false;
                    for (int k = 0; k < a->mk_n; k++)
                        if (f.arg_col == a->mk_col[k]) {
                            a->da_keys.fr_key[i] = (int8_t)k;
                            ok = true;
                        }
                } else {
                    ok = f.arg_col == fp.key_col;
                }
                continue;
            }
            if (f.arg_col >= 0) {
                int v = 0;
                while (v < fp.V && fp.vcol[v] != f.arg_col) v++;
                if (v == fp.V) {
                    if (fp.V == TSQ_RADIX_MAXV) { ok = false; break; }
                    fp.vcol[fp.V] = f.arg_col;
                    fp.vtype[fp.V] = cfg->input_types[f.arg_col];
                    fp.V++;
                }
                g.v = v;
            }
            const bool real = f.arg_type == TSQ_F32 || f.arg_type == TSQ_F64;
            int words = 1;
            if (f.func == TSQ_AGG_SUM) words = real ? 1 : 2;
            if (f.func == TSQ_AGG_AVG) words = real ? 2 : 3;
            if (fp.W + words > TSQ_AF_MAXW) { ok = false; break; }
            g.w = fp.W;
            for (int k = 0; k < words; k++) fp.init[fp.W + k] = 0;
            if (f.func == TSQ_AGG_MIN) fp.init[fp.W] = ~0ull;
            fp.W += words;
        }
        if (fp.W == 0) ok = false;
        if (ok) af_fill_wdesc(fp);
        if (!ok) a->mk_n = 0;
        a->fast_ok = ok;
    }
    TSQ_HIP(ch, hipSetDevice(ctx->device));
    tsq_handle_hdr* h = &a->hdr;
    a->icols.resize(cfg->n_input_cols);
    a->heap.resize(cfg->n_input_cols);
    for (int c = 0; c < cfg->n_input_cols; c++) a->icols[c].type = a->heap[c].type = cfg->input_types[c];
    tsq_status s = a->counters.reserve(ctx, h, 64);
    if (s == TSQ_OK) {
        hipError_t e = hipMemsetAsync(a->counters.p, 0, 64, ctx->stream);
        if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
    }
    // several integer key columns, fixed-width inputs: the composite-key child may take this aggregate (wide_setup decides at the first
    // batch); this operator's own table then only holds exception rows and starts small
    a->wide_ok = cfg->n_group_keys >= 2 && !has_str && tsq_knob(ctx, TSQ_KNOB_AGG_WIDE_KEYS, 1) != 0;
    for (int k = 0; k < cfg->n_group_keys && a->wide_ok; k++)
        if (cfg->group_key_type[k] != TSQ_I64 && cfg->group_key_type[k] != TSQ_U64) a->wide_ok = false;
    // string keys, or several key columns of integers and strings: the dictionary of group keys may take this aggregate (kd_setup decides
    // at the first batch, after the composite-key route): key cells that fit a key record, 8-byte argument columns that travel
    {
        bool ok = cfg->n_group_keys >= 1 && (cfg->n_group_keys >= 2 || cfg->group_key_type[0] == TSQ_BYTES) && tsq_knob(ctx, TSQ_KNOB_KEYREC, 1) != 0;
        for (int k = 0; k < cfg->n_group_keys && ok; k++) {
            const int32_t t = cfg->group_key_type[k];
            ok = t == TSQ_I64 || t == TSQ_U64 || t == TSQ_BYTES;
        }
        a->kd_npay = 0;
        auto travel = [&](int c) {
            if (c < 0) return true;
            const int32_t t = cfg->input_types[c];
            if (t != TSQ_I64 && t != TSQ_U64 && t != TSQ_F64) return false;
            for (int v = 0; v < a->kd_npay; v++)
                if (a->kd_paycol[v] == c) return true;
            if (a->kd_npay == TSQ_KR_MAXPAY) return false;
            a->kd_paycol[a->kd_npay++] = c;
            return true;
        };
        for (int i = 0; i < cfg->n_aggs && ok; i++) {
            const tsq_agg_func& f = cfg->aggs[i];
            if (f.func == TSQ_AGG_FIRSTROW) {  // only of a group key column: its value comes back from the dictionary
                bool is_key = false;
                for (int k = 0; k < cfg->n_group_keys; k++) is_key = is_key || f.arg_col == cfg->group_key_col[k];
                ok = is_key;
                continue;
            }
            const bool merge = f.mode == TSQ_MODE_FINAL || f.mode == TSQ_MODE_PARTIAL2;
            ok = travel(f.arg_col) && (!(f.func == TSQ_AGG_AVG && merge) || travel(f.arg_col2));
        }
        int fixed_bytes = 0;  // what the cells need at least: 9 bytes per integer cell, flag + length per string cell
        for (int k = 0; k < cfg->n_group_keys; k++) fixed_bytes += cfg->group_key_type[k] == TSQ_BYTES ? 2 : 9;
        a->kd_ok = ok && cfg->n_input_cols < TSQ_MAX_COLS && fixed_bytes <= TSQ_KR_BYTES;
        for (int k = 0; k < cfg->n_group_keys; k++) a->has_str_key = a->has_str_key || cfg->group_key_type[k] == TSQ_BYTES;
    }
    uint64_t cap = cfg->n_group_keys == 0 ? 16 : ((cfg->est_groups > 0 && !a->wide_ok && !a->kd_ok) ? (uint64_t)cfg->est_groups * 2 + 16 : (1u << 16));
    if (s == TSQ_OK) s = alloc_table(a.get(), a->tb, cap);
    if (s == TSQ_OK) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
    }
    if (s != TSQ_OK) {
        tsq_fail(ch, s, a->hdr.err);
        tsq_agg_destroy(a.release());
        return s;
    }
    *out = a.release();
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_push(tsq_agg* a, const tsq_col* cols, int32_t n_cols, int64_t nrows) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(a, TSQ_MAGIC_AGG));
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG) return TSQ_ERR_INVALID;
    TSQ_TRY(agg_cancelled(a));
    if (a->finished) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "push after finish");
    if (nrows < 0 || (!cols && nrows > 0)) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "bad arguments");
    if (nrows == 0) return TSQ_OK;
    bool dev = false;
    TSQ_TRY(tsq_validate_cols(&a->hdr, cols, n_cols, a->cfg.n_input_cols, a->cfg.input_types, nrows, &dev));
    TSQ_HIP(&a->hdr, hipSetDevice(a->ctx->device));
    if (a->in_rows == 0 && a->stage.staged == 0) a->host_mode = !dev;
    if (dev) {
        TSQ_TRY(agg_flush(a));
        tsq_colset all;
        tsq_colset_from_cols(all, cols, n_cols);
        if (a->has_str) {  // the operator keeps the var-len cells (device to device) and works on its own copy
            DevBuf t1, t2;
            tsq_status hs = TSQ_OK;
            for (int c = 0; c < n_cols && hs == TSQ_OK; c++) {
                if (cols[c].type != TSQ_BYTES) continue;
                int64_t row0 = 0;
                hs = heap_append(a, c, cols[c].data, cols[c].offsets, cols[c].null_bitmap, nrows, true, t1, t2, &row0);
                if (hs == TSQ_OK) colset_use_heap(a, all, c, row0);
            }
            t1.release();
            t2.release();
            TSQ_TRY(hs);
        }
        const int64_t slice = 256 << 20;  // device batches: every batch ends with a merge of its partial groups and two host syncs
        for (int64_t off = 0; off < nrows; off += slice) {
            const int64_t n = std::min<int64_t>(slice, nrows - off);
            tsq_colset s;
            tsq_colset_slice(s, all, off);
            TSQ_TRY(agg_batch(a, s, n));
            a->in_rows += n;
        }
        return agg_heap_gc(a);
    }
    if (a->stage.cap == 0) {
        int64_t batch = 4 << 20;  // host chunks are aggregated in device batches of this many rows (test knob: TSQ_AGG_BATCH_ROWS)
        if (a->ctx->knob[TSQ_KNOB_AGG_BATCH_ROWS] != TSQ_KNOB_DEFAULT) batch = std::max<int64_t>(1024, (a->ctx->knob[TSQ_KNOB_AGG_BATCH_ROWS] + 63) & ~63LL);
        TSQ_TRY(a->stage.init(&a->hdr, n_cols, a->cfg.input_types, batch));
    }
    int64_t off = 0;
    while (off < nrows) {
        int64_t n = std::min<int64_t>(nrows - off, a->stage.room());
        a->stage.add(cols, off, n, nullptr);
        off += n;
        if (a->stage.room() == 0) TSQ_TRY(agg_flush(a));
    }
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_finish(tsq_agg* a) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(a, TSQ_MAGIC_AGG));
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG) return TSQ_ERR_INVALID;
    TSQ_TRY(agg_cancelled(a));
    if (a->finished) return TSQ_OK;
    tsq_ctx* ctx = a->ctx;
    tsq_handle_hdr* h = &a->hdr;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    TSQ_TRY(agg_flush(a));
    TSQ_TRY(side_join(a));  // (the last batches of the dense packed route may still be on the side stream)
    // every group in the dense state of the packed route and none in the table: the rows come straight from the cells (k_dense_finalize)
    const bool dense_direct = a->dense_state == 1 && a->dense_rows > 0 && a->groups == 0 && !a->multi && a->plan.n_keys == 1 && a->mk_n <= 1 &&
                              a->wide_state != 1 && !a->stream && tsq_knob(ctx, TSQ_KNOB_DENSE_DIRECT, 1) != 0;
    if (!dense_direct) TSQ_TRY(da_dense_flush(a));
    TSQ_TRY(pg_pend_flush(a));
    // every group in the partitioned sub-tables and none in the table (no spilled or exception row): the rows come straight from the slots
    const bool pg_direct = a->pg_state == 1 && a->pg_rows > 0 && a->groups == 0 && !a->multi && a->plan.n_keys == 1 && a->mk_n <= 1 && a->wide_state != 1 && !a->stream;
    if (!pg_direct) TSQ_TRY(pg_flush(a));
    // empty input without GROUP BY: exactly one row of defaults (aggregate.go:572-574,
    // builder.go:517-539): COUNT -> 0, everything else NULL.  The NULL-group slot (cap+1) is the
    // single group of a key-less aggregate; claim it so that finalize emits it.
    if (a->plan.n_keys == 0 && a->groups == 0) {
        unsigned long long one = 1;
        // (StreamAggExec: groups live in slots [0, groups) — the default row is group 0)
        TSQ_HIP(h, hipMemcpy((char*)a->tb.tag.p + (a->stream ? 0 : (a->tb.cap + 1) * 8), &one, 8, hipMemcpyHostToDevice));
        a->groups = 1;
    }
    // the groups of the composite-key child come after this operator's own (exception) groups
    int64_t g_child = 0;
    if (a->wide_state == 1) {
        const tsq_status cs = tsq_agg_finish(a->wide);
        if (cs != TSQ_OK) return tsq_fail(h, cs, a->wide->hdr.err);
        g_child = a->wide->out_rows;
    }
    int64_t g_own = a->groups;
    if (dense_direct) {  // the groups = the touched cells
        const uint64_t nwords = ((uint64_t)1 << a->da_dm.b) >> 5;
        TSQ_HIP(h, hipMemsetAsync((char*)a->counters.p + 6 * 8, 0, 8, ctx->stream));
        hipLaunchKernelGGL(k_dense_count, dim3(std::min<int>(tsq_grid_for(ctx, (int64_t)nwords, 256), ctx->num_cus)), dim3(256), 0, ctx->stream, a->dense_touch.as<uint32_t>(), nwords,
                           a->counters.as<unsigned long long>() + 6);
        TSQ_HIP(h, hipGetLastError());
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 9, (char*)a->counters.p + 6 * 8, 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        g_own = (int64_t)ctx->pinned[9];
        a->st.kernel_launches++;
    }
    if (pg_direct) {  // the groups = the occupied slots: the sub-tables' counts, summed on the host (<= 128 KB)
        const size_t subs = (size_t)1 << (a->pg_pbits + a->pg_sbits);
        std::vector<uint32_t> used(subs);
        TSQ_HIP(h, hipMemcpyAsync(used.data(), a->pg_used.p, subs * 4, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        g_own = 0;
        for (uint32_t u : used) g_own += u;
    }
    const int64_t g = g_own + g_child;
    a->odata.resize(a->n_out);
    a->onn.resize(a->n_out);
    a->obitmap.resize(a->n_out);
    FinalArgs fa;
    memset(&fa, 0, sizeof fa);
    fill_agg_table(a, a->tb, fa.t);
    fa.plan = a->plan;
    for (int k = 0; k < a->plan.n_keys; k++) fa.key_type[k] = a->cfg.group_key_type[k];
    for (int oc = 0; oc < a->n_out; oc++) {
        TSQ_TRY(a->odata[oc].reserve(ctx, h, (size_t)g * tsq_elem_size(a->out_types[oc]) + 16));
        TSQ_TRY(a->onn[oc].reserve(ctx, h, (size_t)g + 16));
        fa.out_data[oc] = a->odata[oc].p;
        fa.out_notnull[oc] = a->onn[oc].as<uint8_t>();
    }
    fa.counters = a->counters.as<unsigned long long>();
    fa.ordered = a->stream ? 1 : 0;
    fa.ordered_groups = (uint64_t)g_own;
    TSQ_HIP(h, hipMemsetAsync((char*)a->counters.p + 3 * 8, 0, 16, ctx->stream));
    if (dense_direct || pg_direct) {
        DenseFinalArgs da;
        memset(&da, 0, sizeof da);
        da.fplan = a->fplan;
        da.plan = a->plan;
        da.dm = a->da_dm;
        da.key_type = a->cfg.group_key_type[0];
        if (pg_direct) {
            for (int k = 0; k < a->fplan.W; k++) da.dense_w[k] = a->pg_w[k].as<unsigned long long>();
            da.pg_key = a->pg_key.as<unsigned long long>();
            da.ncells = (uint64_t)af_slots(a->fplan) << (a->pg_pbits + a->pg_sbits);
        } else {
            for (int k = 0; k < a->fplan.W; k++) da.dense_w[k] = a->dense_w[k].as<unsigned long long>();
            da.dense_touch = a->dense_touch.as<uint32_t>();
            da.ncells = (uint64_t)1 << a->da_dm.b;
        }
        for (int oc = 0; oc < a->n_out; oc++) {
            da.out_data[oc] = fa.out_data[oc];
            da.out_notnull[oc] = fa.out_notnull[oc];
        }
        da.counters = fa.counters;
        hipLaunchKernelGGL(k_dense_finalize, dim3(tsq_grid_for(ctx, (int64_t)da.ncells, 256, TSQ_FINAL_CHUNK / 256)), dim3(256), 0, ctx->stream, da);
        if (dense_direct) a->dense_flushes++;
    } else {
        int grid = tsq_grid_for(ctx, a->stream ? std::max<int64_t>(g_own, 1) : (int64_t)a->tb.cap + 2, 256, TSQ_FINAL_CHUNK / 256);
        hipLaunchKernelGGL(k_agg_finalize, dim3(grid), dim3(256), 0, ctx->stream, fa);
    }
    TSQ_HIP(h, hipGetLastError());
    a->st.kernel_launches++;
    if (g_child > 0) {  // rows [g_own, g) of every output column: copied from the child, or decoded from its composite key
        tsq_agg* c = a->wide;
        WideDecodeArgs da;
        memset(&da, 0, sizeof da);
        da.f = a->wide_f;
        da.d = c->odata[0].as<uint64_t>();
        da.n = g_child;
        for (int oc = 0; oc < a->n_out; oc++) {
            const int32_t src = a->wide_child_out[oc];
            const size_t es = tsq_elem_size(a->out_types[oc]);
            if (src < 0) {
                da.key_of[da.n_out] = -1 - src;
                da.out[da.n_out] = a->odata[oc].as<uint64_t>() + g_own;
                da.out_nn[da.n_out] = a->onn[oc].as<uint8_t>() + g_own;
                da.n_out++;
            } else {
                TSQ_HIP(h, hipMemcpyAsync((char*)a->odata[oc].p + (size_t)g_own * es, c->odata[src].p, (size_t)g_child * es, hipMemcpyDeviceToDevice, ctx->stream));
                TSQ_HIP(h, hipMemcpyAsync(a->onn[oc].as<uint8_t>() + g_own, c->onn[src].p, (size_t)g_child, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        if (da.n_out && a->kd_mode) {  // the key columns come back from the dictionary records
            KdDecodeArgs ka;
            memset(&ka, 0, sizeof ka);
            ka.id = da.d;
            ka.n = g_child;
            ka.dloc = a->kd_dloc.as<uint32_t>();
            ka.drec = a->kd_drec.as<unsigned long long>();
            ka.n_keys = a->plan.n_keys;
            ka.len_bits = TSQ_REF_LEN_BITS;
            for (int k = 0; k < a->plan.n_keys; k++) ka.key_is_str[k] = a->cfg.group_key_type[k] == TSQ_BYTES ? 1 : 0;
            // (two FIRST_ROWs of one key column: the decode kernel fills one output per key column, the others are copies)
            int first_of[TSQ_MAX_GROUP_KEYS];
            for (int k = 0; k < TSQ_MAX_GROUP_KEYS; k++) first_of[k] = -1;
            for (int o = 0; o < da.n_out; o++) {
                const int k = da.key_of[o];
                if (first_of[k] < 0) {
                    first_of[k] = o;
                    ka.out[k] = da.out[o];
                    ka.out_nn[k] = da.out_nn[o];
                }
            }
            hipLaunchKernelGGL(k_kd_decode, dim3(tsq_grid_for(ctx, g_child, 256)), dim3(256), 0, ctx->stream, ka);
            TSQ_HIP(h, hipGetLastError());
            for (int o = 0; o < da.n_out; o++) {
                const int f0 = first_of[da.key_of[o]];
                if (f0 == o) continue;
                TSQ_HIP(h, hipMemcpyAsync(da.out[o], da.out[f0], (size_t)g_child * 8, hipMemcpyDeviceToDevice, ctx->stream));
                TSQ_HIP(h, hipMemcpyAsync(da.out_nn[o], da.out_nn[f0], (size_t)g_child, hipMemcpyDeviceToDevice, ctx->stream));
            }
            a->st.kernel_launches++;
        } else if (da.n_out) {
            hipLaunchKernelGGL(k_agg_wide_decode, dim3(tsq_grid_for(ctx, g_child, 256)), dim3(256), 0, ctx->stream, da);
            TSQ_HIP(h, hipGetLastError());
            a->st.kernel_launches++;
        }
    }
    for (int oc = 0; oc < a->n_out; oc++) {
        TSQ_TRY(a->obitmap[oc].reserve(ctx, h, tsq_bitmap_bytes(g) + 16));
        TSQ_TRY(tsq_launch_pack_bitmap(ctx, h, a->onn[oc].as<uint8_t>(), a->obitmap[oc].as<uint8_t>(), g));
    }
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned, (char*)a->counters.p + 3 * 8, 24, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    const int64_t emitted = (int64_t)ctx->pinned[0];
    const bool overflow = ctx->pinned[1] != 0;
    if (ctx->pinned[2]) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "aggregate: a string cell longer than 16 MiB - 2 bytes: fall back to the Go operator");
    if (overflow) return tsq_fail(h, TSQ_ERR_OVERFLOW_BIGINT, "BIGINT value is out of range in 'sum/avg' (func_sum.go:133-137)");
    if (emitted != g_own) return tsq_fail(h, TSQ_ERR_HIP, "internal: finalize emitted " + std::to_string(emitted) + " groups, expected " + std::to_string(g_own));
    // var-len output columns: references -> offsets + bytes
    a->ooffs.resize(a->n_out);
    a->obytes.resize(a->n_out);
    a->onbytes.assign(a->n_out, 0);
    for (int oc = 0; oc < a->n_out; oc++) {
        if (a->out_types[oc] != TSQ_BYTES) continue;
        TSQ_TRY(a->ooffs[oc].reserve(ctx, h, (size_t)(g + 1) * 8 + 64));
        TSQ_HIP(h, hipMemsetAsync(a->ooffs[oc].p, 0, 8, ctx->stream));
        if (g == 0) continue;
        RefOutArgs ra;
        memset(&ra, 0, sizeof ra);
        ra.refs = a->odata[oc].as<unsigned long long>();
        ra.notnull = a->onn[oc].as<uint8_t>();
        ra.rows = g;
        ra.heap = (const uint8_t*)a->heap[a->out_arg_col[oc]].data.p;
        ra.heap_child = a->kd_mode ? (const uint8_t*)a->kd_drec.p : nullptr;
        ra.child_from = a->kd_mode ? g_own : g;
        ra.out_offs = a->ooffs[oc].as<int64_t>();
        hipLaunchKernelGGL(k_ref_len, dim3(tsq_grid_for(ctx, g, 256)), dim3(256), 0, ctx->stream, ra);
        TSQ_HIP(h, hipGetLastError());
        DevBuf scratch;
        tsq_status s = tsq_launch_scan64(ctx, h, ra.out_offs, g, scratch);
        if (s == TSQ_OK) {
            hipError_t e = hipMemcpyAsync(ctx->pinned + 8, ra.out_offs + g, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, std::string("var-len output: ") + hipGetErrorString(e));
        }
        scratch.release();
        TSQ_TRY(s);
        const int64_t nbytes = (int64_t)ctx->pinned[8];
        a->onbytes[oc] = nbytes;
        TSQ_TRY(a->obytes[oc].reserve(ctx, h, (size_t)nbytes + 64));
        ra.out_data = a->obytes[oc].as<uint8_t>();
        if (nbytes > 0) {
            if (nbytes / g > 32) hipLaunchKernelGGL(k_ref_copy<true>, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, ra);
            else hipLaunchKernelGGL(k_ref_copy<false>, dim3(tsq_grid_for(ctx, g, 256)), dim3(256), 0, ctx->stream, ra);
            TSQ_HIP(h, hipGetLastError());
        }
        a->st.kernel_launches += 5;
    }
    a->out_rows = g;
    a->out_cursor = 0;
    a->st.out_rows = g;
    if (a->host_mode) {
        a->hdata.resize(a->n_out);
        a->hbitmap.resize(a->n_out);
        a->hoffs.resize(a->n_out);
        for (int oc = 0; oc < a->n_out; oc++) {
            const bool var = a->out_types[oc] == TSQ_BYTES;
            const size_t bytes = var ? (size_t)a->onbytes[oc] : (size_t)g * tsq_elem_size(a->out_types[oc]);
            TSQ_TRY(a->hdata[oc].reserve(h, bytes + 16));
            TSQ_TRY(a->hbitmap[oc].reserve(h, tsq_bitmap_bytes(g) + 16));
            if (var) {
                TSQ_TRY(a->hoffs[oc].reserve(h, (size_t)(g + 1) * 8 + 16));
                TSQ_HIP(h, hipMemcpyAsync(a->hoffs[oc].p, a->ooffs[oc].p, (size_t)(g + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
                a->st.d2h_bytes += (g + 1) * 8;
            }
            if (bytes) TSQ_HIP(h, hipMemcpyAsync(a->hdata[oc].p, var ? a->obytes[oc].p : a->odata[oc].p, bytes, hipMemcpyDeviceToHost, ctx->stream));
            TSQ_HIP(h, hipMemcpyAsync(a->hbitmap[oc].p, a->obitmap[oc].p, tsq_bitmap_bytes(g), hipMemcpyDeviceToHost, ctx->stream));
            a->st.d2h_bytes += bytes;
        }
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        a->out_on_host = true;
    }
    a->finished = true;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_set_fast(tsq_agg* a, int32_t mode) {
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG) return TSQ_ERR_INVALID;
    if (mode < TSQ_AGGFAST_AUTO || mode > TSQ_AGGFAST_FORCE) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "mode must be -1 (auto), 0 (off) or 1 (force)");
    a->fast_mode = mode;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_set_stream(tsq_agg* a, int32_t on) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(a, TSQ_MAGIC_AGG));
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG) return TSQ_ERR_INVALID;
    if (a->in_rows > 0 || a->stage.staged > 0 || a->finished) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "stream mode must be chosen before the first row");
    const bool want = on != 0;
    if (want && !a->multi && !a->tb.gknull.p) {  // the open group's NULL flags: one byte per group (the hash table of a one-key plan has none)
        TSQ_HIP(&a->hdr, hipSetDevice(a->ctx->device));
        TSQ_TRY(a->tb.gknull.reserve(a->ctx, &a->hdr, a->tb.cap + 2));
        TSQ_HIP(&a->hdr, hipMemsetAsync(a->tb.gknull.p, 0, a->tb.cap + 2, a->ctx->stream));
    }
    a->stream = want;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_num_groups(tsq_agg* a, int64_t* out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(a, TSQ_MAGIC_AGG));
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG || !out) return TSQ_ERR_INVALID;
    *out = a->finished ? a->out_rows : a->groups + (a->wide_state == 1 ? a->wide->groups : 0);
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_pull(tsq_agg* a, tsq_col* out_cols, int32_t n_cols, int64_t cap_rows, int64_t* nrows_out, int32_t* eos) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(a, TSQ_MAGIC_AGG));
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG) return TSQ_ERR_INVALID;
    if (!nrows_out || !eos) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "NULL out pointer");
    *nrows_out = 0;
    *eos = 0;
    TSQ_TRY(agg_cancelled(a));
    if (!a->finished) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "pull before finish");
    if (n_cols != a->n_out) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "pull: wrong number of output columns");
    const int64_t n = std::min<int64_t>(cap_rows, a->out_rows - a->out_cursor);
    if (n <= 0) { *eos = 1; return TSQ_OK; }
    TSQ_HIP(&a->hdr, hipSetDevice(a->ctx->device));
    for (int oc = 0; oc < a->n_out; oc++) {
        tsq_col& o = out_cols[oc];
        const int es = tsq_elem_size(a->out_types[oc]);
        const bool var = a->out_types[oc] == TSQ_BYTES;
        const bool odev = o.flags & TSQ_COL_DEVICE;
        if (!o.data || !o.null_bitmap) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "pull: out column needs data and null_bitmap buffers");
        if (var && !o.offsets) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "pull: var-len out column needs an offsets buffer (tsq_agg_peek tells the data bytes)");
        if (a->out_on_host && !odev) {
            if (var) {  // cells [cursor, cursor + n): their bytes, and the offsets moved to start at 0
                const int64_t* so = (const int64_t*)a->hoffs[oc].p + a->out_cursor;
                memcpy(o.data, (const char*)a->hdata[oc].p + so[0], (size_t)(so[n] - so[0]));
                for (int64_t i = 0; i <= n; i++) o.offsets[i] = so[i] - so[0];
            } else {
                tsq_host_copy(o.data, (const char*)a->hdata[oc].p + (size_t)a->out_cursor * es, (size_t)n * es);
            }
            const uint8_t* src = (const uint8_t*)a->hbitmap[oc].p;
            if ((a->out_cursor & 7) == 0) memcpy(o.null_bitmap, src + (a->out_cursor >> 3), tsq_bitmap_bytes(n));
            else {
                memset(o.null_bitmap, 0, tsq_bitmap_bytes(n));
                for (int64_t i = 0; i < n; i++) {
                    const int64_t s = a->out_cursor + i;
                    if ((src[s >> 3] >> (s & 7)) & 1) o.null_bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
                }
            }
        } else if (!a->out_on_host && odev) {
            if (a->out_cursor & 7) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "device pull: cap_rows must keep the cursor a multiple of 8");
            if (var) {
                const int64_t* so = a->ooffs[oc].as<int64_t>() + a->out_cursor;
                TSQ_HIP(&a->hdr, hipMemcpyAsync(a->ctx->pinned + 44, so, 8, hipMemcpyDeviceToHost, a->ctx->stream));
                TSQ_HIP(&a->hdr, hipMemcpyAsync(a->ctx->pinned + 45, so + n, 8, hipMemcpyDeviceToHost, a->ctx->stream));
                TSQ_HIP(&a->hdr, hipStreamSynchronize(a->ctx->stream));
                const int64_t b0 = (int64_t)a->ctx->pinned[44], b1 = (int64_t)a->ctx->pinned[45];
                if (b1 > b0) TSQ_HIP(&a->hdr, hipMemcpyAsync(o.data, (const char*)a->obytes[oc].p + b0, (size_t)(b1 - b0), hipMemcpyDeviceToDevice, a->ctx->stream));
                TSQ_TRY(tsq_launch_offsets_rebase(a->ctx, &a->hdr, o.offsets, so, n + 1, -b0));
            } else {
                TSQ_HIP(&a->hdr, hipMemcpyAsync(o.data, (const char*)a->odata[oc].p + (size_t)a->out_cursor * es, (size_t)n * es, hipMemcpyDeviceToDevice, a->ctx->stream));
            }
            TSQ_HIP(&a->hdr, hipMemcpyAsync(o.null_bitmap, a->obitmap[oc].as<uint8_t>() + (a->out_cursor >> 3), tsq_bitmap_bytes(n), hipMemcpyDeviceToDevice, a->ctx->stream));
        } else {
            return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "pull: output placement (host/device) must match the pushes");
        }
        o.length = n;
        o.type = a->out_types[oc];
        o.elem_size = var ? -1 : es;
    }
    if (!a->out_on_host) TSQ_HIP(&a->hdr, hipStreamSynchronize(a->ctx->stream));
    a->out_cursor += n;
    *nrows_out = n;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_peek(tsq_agg* a, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_out, int32_t n_cols) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(a, TSQ_MAGIC_AGG));
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG) return TSQ_ERR_INVALID;
    if (!nrows_out || !bytes_out) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "NULL out pointer");
    if (!a->finished) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "peek before finish");
    if (n_cols != a->n_out) return tsq_fail(&a->hdr, TSQ_ERR_INVALID, "peek: wrong number of output columns");
    *nrows_out = 0;
    for (int oc = 0; oc < n_cols; oc++) bytes_out[oc] = 0;
    TSQ_TRY(agg_cancelled(a));
    const int64_t n = std::min<int64_t>(cap_rows, a->out_rows - a->out_cursor);
    if (n <= 0) return TSQ_OK;
    *nrows_out = n;
    TSQ_HIP(&a->hdr, hipSetDevice(a->ctx->device));
    for (int oc = 0; oc < n_cols; oc++) {
        if (a->out_types[oc] != TSQ_BYTES) continue;
        if (a->out_on_host) {
            const int64_t* so = (const int64_t*)a->hoffs[oc].p + a->out_cursor;
            bytes_out[oc] = so[n] - so[0];
        } else {
            const int64_t* so = a->ooffs[oc].as<int64_t>() + a->out_cursor;
            TSQ_HIP(&a->hdr, hipMemcpyAsync(a->ctx->pinned + 44, so, 8, hipMemcpyDeviceToHost, a->ctx->stream));
            TSQ_HIP(&a->hdr, hipMemcpyAsync(a->ctx->pinned + 45, so + n, 8, hipMemcpyDeviceToHost, a->ctx->stream));
            TSQ_HIP(&a->hdr, hipStreamSynchronize(a->ctx->stream));
            bytes_out[oc] = (int64_t)a->ctx->pinned[45] - (int64_t)a->ctx->pinned[44];
        }
    }
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_cancel(tsq_agg* a) {
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG) return TSQ_ERR_INVALID;
    a->cancelled.store(1);
    if (a->wide) a->wide->cancelled.store(1);
    return TSQ_OK;
}

TSQ_API tsq_status tsq_agg_stats(tsq_agg* a, tsq_stats* out) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(a, TSQ_MAGIC_AGG));
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG || !out) return TSQ_ERR_INVALID;
    a->st.probe_rows = a->in_rows;
    a->st.table_buckets = (int64_t)a->tb.cap;
    a->st.radix_batches = a->fast_batches;
    a->st.radix_overflow_rows = a->fast_fallbacks;
    a->st.packed_key_bits = a->packed_batches > 0 ? (int32_t)a->da_dm.b : 0;
    a->st.dense_flushes = (int32_t)a->dense_flushes;
    a->st.side_stream_batches = (int32_t)a->side_batches;
    a->st.table_slice_bits = (a->packed_batches > 0 && a->mk_n <= 1 && a->fplan.V == 1) ? (int32_t)a->da_paybytes * 8 : 0;
    if (a->wide_state == 1) {  // the composite-key child did the work: its batches / packed range, the rows that stayed here as exceptions
        tsq_stats cs;
        if (tsq_agg_stats(a->wide, &cs) == TSQ_OK) {
            a->st.radix_batches = cs.radix_batches;
            a->st.packed_key_bits = cs.packed_key_bits;
        }
        a->st.build_handed_back_rows = a->wide_exception_rows;
        a->st.build_partitioned = a->kd_mode ? 3 : 2;  // (aggregate: 2 = several key columns composed into one 64-bit key, 3 = group keys through the dictionary of key records)
    }
    if (a->pg_batches > 0 && a->wide_state != 1) a->st.build_partitioned = 4;  // (aggregate: 4 = the groups lived in partitioned, LDS-sized sub-tables: tsq_aggfast.h K7p)
    else if (a->wide_state == 1 && a->wide && a->wide->pg_batches > 0 && !a->kd_mode) a->st.dense_flushes = -4;  // (... of the composite-key child)
    a->st.heap_bytes = 0;
    for (const ColStore& hs : a->heap)
        if (hs.type == TSQ_BYTES) a->st.heap_bytes = std::max<int64_t>(a->st.heap_bytes, hs.nbytes);
    a->st.heap_compactions = a->heap_gcs;
    *out = a->st;
    return TSQ_OK;
}

TSQ_API void tsq_agg_destroy(tsq_agg* a) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(a, TSQ_MAGIC_AGG));
    if (!a || a->hdr.magic != TSQ_MAGIC_AGG) return;
    (void)hipSetDevice(a->ctx->device);
    (void)hipStreamSynchronize(a->ctx->stream);
    if (a->side) {
        (void)hipStreamSynchronize(a->side);
        (void)hipStreamDestroy(a->side);
        for (hipEvent_t e : {a->side_part, a->side_ev[0], a->side_ev[1]})
            if (e) (void)hipEventDestroy(e);
    }
    for (DevBuf* b : {&a->r2keys, &a->r2ctl, &a->r2vend, &a->r2okeys}) b->release();
    for (auto& b : a->r2pay) b.release();
    for (auto& b : a->r2opay) b.release();
    if (a->wide) {
        tsq_agg_destroy(a->wide);
        a->wide = nullptr;
    }
    for (DevBuf* b : {&a->wide_d, &a->wide_okrows, &a->wide_excrows}) b->release();
    for (DevBuf* b : {&a->kd_counts, &a->kd_pstart, &a->kd_rec, &a->kd_ids, &a->kd_flags, &a->kd_paynn, &a->kd_norec, &a->kd_ridx, &a->kd_gid, &a->kd_drec,
                      &a->kd_dids, &a->kd_dcount, &a->kd_dloc, &a->kd_ctl})
        b->release();
    for (auto& b : a->kd_pay) b.release();
    for (auto& b : a->kd_paybm) b.release();
    a->tb.release();
    a->counters.release();
    a->retry[0].release();
    a->retry[1].release();
    a->slot_of.release();
    a->sa_cnt.release();
    a->hotkeys.release();
    a->stage.release();
    for (auto& c : a->icols) c.release();
    for (auto& c : a->heap) c.release();
    for (auto& b : a->ooffs) b.release();
    for (auto& b : a->obytes) b.release();
    for (auto& b : a->hoffs) b.release();
    for (auto& b : a->odata) b.release();
    for (auto& b : a->onn) b.release();
    for (auto& b : a->obitmap) b.release();
    for (auto& b : a->hdata) b.release();
    for (auto& b : a->hbitmap) b.release();
    a->fkey.release();
    for (auto& b : a->fw) b.release();
    a->pg_key.release();
    for (auto& b : a->pg_w) b.release();
    a->pg_used.release();
    for (auto& cs : a->pg_pend) cs.release();
    a->fctl.release();
    a->fexc.release();
    a->rkeys.release();
    a->rctl.release();
    a->rvend.release();
    for (auto& b : a->dense_w) b.release();
    a->dense_touch.release();
    a->rokeys.release();
    for (auto& b : a->rpay) b.release();
    for (auto& b : a->ropay) b.release();
    a->hdr.magic = 0;
    delete a;
}
