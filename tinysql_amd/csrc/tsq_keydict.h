// tsq_keydict.h — GROUP BY string keys / several key columns: a DICTIONARY of group keys (round 5; device code, included by tsq_agg.hip).
//
// HashAggExec keys its partial results by the encoded group key (executor/aggregate.go:332-350 getGroupKey -> codec.HashGroupKey,
// util/codec/codec.go:700-760).  The several-column upsert of tsq_agg.hip does that literally — hash the cells, find the slot in a
// table in HBM, compare the cells through references into a heap of strings: several random lines per row, 4.3e9 rows/s for 16-byte
// string keys.  Here a row's key cells become a 32-byte KEY RECORD (tsq_keyrec.h: flag byte + word, or flag 2 + length + bytes; a NULL
// cell is the NilFlag byte — GROUP BY makes NULL a group), the rows are hash-partitioned by the record with their 8-byte argument cells
// travelling along (k_kr_hist / k_kr_scatter), and ONE workgroup per partition turns every record into a dense GROUP ID:
//   the partition's part of the dictionary (<= TSQ_KR_CAP records, kept in HBM next to their ids) is indexed in LDS (18-bit tag + place,
//   as the join's probe kernel does); a record found there takes that id; a record that is not draws a place, writes its bytes and
//   publishes the entry (k_kd_assign); ids of new keys = a range drawn from one device counter per partition and batch.
// The ids (one BIGINT UNSIGNED column, in partition order) and the travelled argument columns are the batch of a CHILD aggregate
// GROUP BY id — an integer key with a dense range: the packed route (tsq_daagg.h) takes it.  At the end the child's groups get their key
// columns back from the dictionary records.  A record that does not fit 32 bytes, a partition whose dictionary is full (it stays full:
// a key is either in the dictionary for good or never) make EXCEPTION
// rows: they take the several-column upsert into this operator's own table, so a group lives in exactly one of the two places.
// Bytes per row: the key cells twice (histogram + scatter) + 32 B record written and read + 8 B per argument cell three times + 8 B id.
#ifndef TSQ_KEYDICT_H
#define TSQ_KEYDICT_H

#include "tsq_keyrec.h"

#define TSQ_KD_NT 1024
#define TSQ_KD_PUB 0x80000000u      // ridx: this record PUBLISHED the dictionary place in the low bits (it hands out the id)
#define TSQ_KD_EXC 0xffffffffu      // ridx: the key is not in the dictionary and found no place
#define TSQ_KD_DEFER 0xfffffffeu    // ridx: the walk met a reserved slot with this key's tag: look again after the barrier
#define TSQ_KD_RES 0x3fffu          // place code of a reserved slot (places are < TSQ_KR_CAP = 12288)
#define TSQ_KD_DEAD 0x3ffeu         // place code of a slot whose reservation drew no place: walks pass it, it never turns EMPTY again

struct KdArgs {
    const unsigned long long* prec;  // the batch's records, partition order: `stride` words each (4: records; 8: the 64-byte slots of
                                     // k_kr_scatter — record, travelling cells, source row | NOT-NULL bits)
    uint32_t stride;
    int32_t n_pay;                   // slots: the travelling cells leave as columns (partition order), with the row ids and the NOT-NULL bytes
    uint64_t* pay_dst[3];
    uint8_t* pay_nn;
    uint32_t* ids;
    const uint32_t* pstart;          // [P + 1]
    uint32_t P;
    unsigned long long* drec;        // dictionary: [P][TSQ_KR_CAP] records ...
    uint32_t* dids;                  // ... their group ids ...
    uint32_t* dcount;                // ... [P] places in use
    uint32_t* dloc;                  // [id] -> p * TSQ_KR_CAP + place
    unsigned long long* next_id;     // device counter of the ids handed out
    uint32_t* ridx;                  // [records of the batch] scratch: the dictionary place of a record's key
    uint64_t* gid;                   // [records of the batch] out: group id, ~0 = exception row
    unsigned long long* counters;    // [0] += exception rows
};

// is the dictionary record at q equal to w?  Another wave of this workgroup may have written it a moment ago: workgroup-scope loads (the
// waves of a workgroup share their CU's vector L1, which a store of the CU goes through — nothing to invalidate, but the compiler must not
// hoist or merge these loads).  Agent scope would be wrong for the price: its release writes the XCD's whole L2 back (3 ms per batch)
__device__ __forceinline__ bool kd_drec_equal(const unsigned long long* q, const uint64_t (&w)[4]) {
    const uint64_t a0 = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint64_t a1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint64_t a2 = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint64_t a3 = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return a0 == w[0] && a1 == w[1] && a2 == w[2] && a3 == w[3];
}

// the 18-bit tag of a record's hash — never all ones: (all ones, TSQ_KD_RES) would be the empty slot's code, and a slot reserved for such
// a key would look empty to the next row that brings it (1e6 keys: four of them, each a group emitted twice — found by the group count)
__device__ __forceinline__ uint32_t kd_tag(uint64_t h) {
    const uint32_t t = (uint32_t)(h >> 14) & 0x3ffffu;
    return t == 0x3ffffu ? 0x3fffeu : t;
}
// One record through the partition's index: found -> its place; an empty slot at the end of the walk -> RESERVE it with a CAS (tag +
// TSQ_KD_RES: nobody can compare against it yet), draw a place of the partition's dictionary, write the record's bytes there, and only
// then put the real entry (tag, place) into the slot — whoever sees a place can compare against complete bytes.  A walk that meets a
// reserved slot with its own tag cannot tell yet whether that is its key: TSQ_KD_DEFER (the caller looks again after a barrier, when
// every reservation has become an entry).
// No place left: the row is an exception row.  A reservation that drew no place becomes a DEAD slot and never EMPTY again (ADVICE r5:
// another key may have walked PAST the reserved slot and settled behind it — were the slot emptied, a later row with that key would
// stop at it, find no place and become an exception row although its key lives in the dictionary: one group in two places).  A walk
// that ends at an empty slot reads the draw counter first and reserves nothing once the places are gone, so DEAD slots come only from
// threads already between that check and their draw: fewer than the workgroup's threads, and the index keeps
// TSQ_KR_SLOTS - TSQ_KR_CAP = 4096 slots beyond the places — a walk always meets an empty slot.
__device__ __forceinline__ uint32_t kd_find_or_insert(uint32_t* s_tab, uint32_t* s_draw, unsigned long long* drec_p, uint32_t nd, const ulonglong2& x, const ulonglong2& y) {
    const uint64_t w[4] = {x.x, x.y, y.x, y.y};
    const uint64_t h = kr_hash(w);
    const uint32_t tag = kd_tag(h);
    uint32_t slot = (uint32_t)h & (TSQ_KR_SLOTS - 1);
    for (;;) {
        // acquire: the record bytes behind a published place are read after the place (free at workgroup scope: no cache invalidate)
        uint32_t e = __hip_atomic_load(&s_tab[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (e == 0xffffffffu) {
            if (__hip_atomic_load(s_draw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= TSQ_KR_CAP - nd) return TSQ_KD_EXC;  // full (for good)
            e = atomicCAS(&s_tab[slot], 0xffffffffu, (tag << 14) | TSQ_KD_RES);
            if (e == 0xffffffffu) {
                const uint32_t k = atomicAdd(s_draw, 1u);
                if (k >= TSQ_KR_CAP - nd) {  // the last places went between the check and the draw
                    __hip_atomic_store(&s_tab[slot], (tag << 14) | TSQ_KD_DEAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return TSQ_KD_EXC;
                }
                const uint32_t place = nd + k;
                ulonglong2* d = reinterpret_cast<ulonglong2*>(drec_p + (size_t)place * 4);
                d[0] = x;
                d[1] = y;
                __threadfence_block();  // the bytes are on their way through the CU's L1 before the place can be seen
                __hip_atomic_store(&s_tab[slot], (tag << 14) | place, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                return place | TSQ_KD_PUB;
            }
        }
        if ((e >> 14) == tag) {
            const uint32_t pl = e & 0x3fffu;
            if (pl == TSQ_KD_RES) return TSQ_KD_DEFER;
            if (pl != TSQ_KD_DEAD && kd_drec_equal(drec_p + (size_t)pl * 4, w)) return pl;
        }
        slot = (slot + 1) & (TSQ_KR_SLOTS - 1);
    }
}

// One workgroup per partition.  A partition whose places are used up is full for good: a key is either in the dictionary from some
// batch on or never — so a group lives either in the child (by its id) or in the parent's own table (its rows are exception rows).
static __global__ void __launch_bounds__(TSQ_KD_NT) k_kd_assign(KdArgs a) {
    __shared__ uint32_t s_tab[TSQ_KR_SLOTS];  // tag18 | place (14 bits)
    __shared__ uint32_t s_draw, s_defer;
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x;
    uint32_t exc = 0;
    for (uint32_t p = blockIdx.x; p < a.P; p += gridDim.x) {
        const uint64_t p0 = a.pstart[p], p1 = a.pstart[p + 1];
        if (p1 == p0) continue;  // (block-uniform)
        const uint32_t nd = a.dcount[p];
        unsigned long long* const drec_p = a.drec + (size_t)p * TSQ_KR_CAP * 4;
        const size_t dbase = (size_t)p * TSQ_KR_CAP;
        __syncthreads();  // the previous partition is done with the index
        for (uint32_t i = tid; i < TSQ_KR_SLOTS; i += TSQ_KD_NT) s_tab[i] = 0xffffffffu;
        if (tid == 0) { s_draw = 0; s_defer = 0; }
        __syncthreads();
        for (uint32_t i = tid; i < nd; i += TSQ_KD_NT) {
            const ulonglong2* s = reinterpret_cast<const ulonglong2*>(drec_p + (size_t)i * 4);
            const ulonglong2 x = s[0], y = s[1];
            const uint64_t w[4] = {x.x, x.y, y.x, y.y};
            const uint64_t h = kr_hash(w);
            const uint32_t entry = (kd_tag(h) << 14) | i;
            uint32_t slot = (uint32_t)h & (TSQ_KR_SLOTS - 1);
            while (atomicCAS(&s_tab[slot], 0xffffffffu, entry) != 0xffffffffu) slot = (slot + 1) & (TSQ_KR_SLOTS - 1);
        }
        __syncthreads();
        // ---- pass 1: find or insert; deferred records look again after a barrier (a thread owns the same records in every sweep)
        uint32_t deferred = 0;
        for (uint64_t r = p0 + tid; r < p1; r += TSQ_KD_NT) {
            const ulonglong2* s = reinterpret_cast<const ulonglong2*>(a.prec + r * a.stride);
            const uint32_t res = kd_find_or_insert(s_tab, &s_draw, drec_p, nd, s[0], s[1]);
            a.ridx[r] = res;
            deferred += res == TSQ_KD_DEFER ? 1u : 0u;
        }
        for (;;) {
            if (deferred) atomicAdd(&s_defer, deferred);
            __syncthreads();
            const uint32_t any = s_defer;
            __syncthreads();
            if (!any) break;  // (block-uniform)
            if (tid == 0) s_defer = 0;
            __syncthreads();
            if (deferred) {
                deferred = 0;
                for (uint64_t r = p0 + tid; r < p1; r += TSQ_KD_NT) {
                    if (a.ridx[r] != TSQ_KD_DEFER) continue;
                    const ulonglong2* s = reinterpret_cast<const ulonglong2*>(a.prec + r * a.stride);
                    const uint32_t res = kd_find_or_insert(s_tab, &s_draw, drec_p, nd, s[0], s[1]);
                    a.ridx[r] = res;
                    deferred += res == TSQ_KD_DEFER ? 1u : 0u;
                }
            }
        }
        if (tid == 0) {
            const uint32_t room = TSQ_KR_CAP - nd;
            const uint32_t nn = s_draw < room ? s_draw : room;
            s_base = nn ? atomicAdd(a.next_id, (unsigned long long)nn) : 0ull;
            a.dcount[p] = nd + nn;
        }
        __syncthreads();
        // ---- pass 2: ids (a place drawn in this batch: the partition's range of the device counter)
        const unsigned long long base = s_base;
        for (uint64_t r = p0 + tid; r < p1; r += TSQ_KD_NT) {
            const uint32_t ri = a.ridx[r];
            uint64_t id = ~0ull;
            if (ri != TSQ_KD_EXC) {
                const uint32_t place = ri & 0x3fffu;
                id = place < nd ? (uint64_t)a.dids[dbase + place] : base + (place - nd);
                if (ri & TSQ_KD_PUB) {
                    a.dids[dbase + place] = (uint32_t)id;
                    a.dloc[id] = (uint32_t)(dbase + place);
                }
            }
            a.gid[r] = id;
            exc += id == ~0ull ? 1u : 0u;
            if (a.stride == 8) {  // the slot's second half: the travelling cells become columns (consecutive r: coalesced)
                const ulonglong2* s = reinterpret_cast<const ulonglong2*>(a.prec + r * 8 + 4);
                const ulonglong2 c01 = s[0], c2t = s[1];
                if (a.n_pay > 0) a.pay_dst[0][r] = c01.x;
                if (a.n_pay > 1) a.pay_dst[1][r] = c01.y;
                if (a.n_pay > 2) a.pay_dst[2][r] = c2t.x;
                a.ids[r] = (uint32_t)c2t.y;
                if (a.pay_nn) a.pay_nn[r] = (uint8_t)(c2t.y >> 32);
            }
        }
    }
    const uint64_t e = wave_sum_u64(exc);
    if ((tid & 63u) == 0 && e) atomicAdd(&a.counters[0], (unsigned long long)e);
}

// the key columns of the child's groups, read back from the dictionary records: an integer cell is its word, a string cell becomes a
// reference (offset of its bytes inside the dictionary buffer | length) that the var-len output pass resolves like a heap reference
struct KdDecodeArgs {
    const uint64_t* id;              // [n] the child's groups: their ids
    int64_t n;
    const uint32_t* dloc;
    const unsigned long long* drec;
    int32_t n_keys;
    int32_t key_is_str[TSQ_MAX_GROUP_KEYS];
    uint64_t* out[TSQ_MAX_GROUP_KEYS];     // nullptr: this key column is not an output column
    uint8_t* out_nn[TSQ_MAX_GROUP_KEYS];
    uint32_t len_bits;                     // reference = byte offset << len_bits | length
};
static __global__ void __launch_bounds__(256) k_kd_decode(KdDecodeArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const size_t loc = a.dloc[a.id[i]];
        const uint8_t* rec = reinterpret_cast<const uint8_t*>(a.drec + loc * 4);
        uint32_t at = 0;
        for (int k = 0; k < a.n_keys; k++) {
            uint64_t word;
            uint32_t off, len;
            const uint32_t flag = kr_parse_cell(rec, &at, a.key_is_str[k] != 0, &word, &off, &len);
            if (!a.out[k]) continue;
            // a string cell: a reference to its bytes inside the dictionary buffer
            a.out[k][i] = flag == 0 ? 0ull : (a.key_is_str[k] ? (((uint64_t)(loc * TSQ_KR_BYTES + off) << a.len_bits) | len) : word);
            a.out_nn[k][i] = flag == 0 ? 0 : 1;
        }
    }
}
// rows of the batch that became exceptions, by their SOURCE row (the parent's upsert reads the caller's columns), and the positions
// (partition order) of the others for the child
struct KdListArgs {
    const uint64_t* gid;
    const uint32_t* ids;   // position -> source row
    int64_t n;
    uint32_t* ok_pos;
    uint32_t* exc_rows;
    unsigned long long* cursors;  // [0] ok, [1] exceptions
};
static __global__ void __launch_bounds__(256) k_kd_lists(KdListArgs a) {
    __shared__ uint32_t s_base[2], s_cnt[2];
    for (int64_t base = (int64_t)blockIdx.x * 256; base < a.n; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        const bool in = i < a.n, exc = in && a.gid[i] == ~0ull, ok = in && !exc;
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
        if (ok) a.ok_pos[s_base[0] + wo + (uint32_t)__popcll(mo & below)] = (uint32_t)i;
        if (exc) a.exc_rows[s_base[1] + we + (uint32_t)__popcll(me & below)] = a.ids[i];
        __syncthreads();
    }
}
// NOT-NULL bytes (bit v of a mask byte per record) -> the packed bitmap of travelling column v
static __global__ void __launch_bounds__(256) k_kd_nn_bitmap(const uint8_t* mask, int64_t n, uint32_t v, uint8_t* bitmap) {
    const int64_t nbytes = (n + 7) >> 3;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nbytes; b += (int64_t)gridDim.x * 256) {
        uint32_t m = 0;
        for (int k = 0; k < 8 && b * 8 + k < n; k++) m |= ((mask[b * 8 + k] >> v) & 1u) << k;
        bitmap[b] = (uint8_t)m;
    }
}

#endif
