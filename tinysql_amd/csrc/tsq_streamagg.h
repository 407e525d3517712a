// tsq_streamagg.h — StreamAggExec: aggregation over input that arrives ORDERED by the group keys (device code + launch
// plumbing; included by tsq_agg.hip after its table / plan / update helpers).
//
// BASELINE.json's north star names StreamAggExec next to HashAggExec.  The reference ships only the plan name
// (planner/core/cbo_test.go:200,204,212 "StreamAgg"), no executor: what is built here is the operator the planner means — the
// child delivers rows sorted by the group-by items (a SortExec, an index scan, a merge join), rows of one group are ADJACENT, a
// group is closed by the first row with another key, groups come out in input order.  Aggregate functions, modes, NULL rules,
// overflow and the output schema are HashAggExec's (executor/aggregate.go:307-350, 559-588; aggfuncs/*.go): the same AggPlan,
// the same per-group state arrays, the same finalize kernel — only "which group does this row belong to" changes:
//
//   hash aggregate : slot = find-or-claim(group key) in an open-addressed table           (k_agg_update)
//   stream         : slot = groups before this batch + (number of group HEADS in rows [0, r]) - 1
//
// so no table is probed and FIRST_ROW is the true first row of the group.  A batch is three launches:
//   k_sa_count   : head[r] = key(r) != key(r - 1)  (row 0 against the OPEN group of the previous batch: the key cells kept in
//                  the table's gkey arrays) — heads per 2048-row chunk
//   k_sa_scan    : exclusive scan of the chunk counts (one workgroup)
//   k_sa_update  : heads again, block scan -> slot per row; values are reduced PER WAVE over runs of equal slots (segmented
//                  shuffle scan: lanes hold consecutive rows), the last lane of a run adds the run's partial result to the
//                  group's state with one atomic per aggregate — 1 atomic per (wave, group, aggregate) instead of 1 per row
// Keys are compared as HashAggExec compares them (group_key_word: NULL == NULL, +0.0 == -0.0, util/codec/codec.go:713-746;
// strings byte-wise).  Algorithmic bytes per row: the key cells twice (count + update pass) + every argument cell once.
#ifndef TSQ_STREAMAGG_H
#define TSQ_STREAMAGG_H

#include "tsq_wavescan.h"

#define TSQ_SA_NT 256
#define TSQ_SA_CHUNK 2048  // rows per workgroup step: 8 tiles of 256 rows

struct StreamAggArgs {
    AggArgs u;                 // in, plan, t (the group table: arrays indexed by group number), nrows, counters
    uint64_t groups_before;    // groups in the table before this batch; the last of them is still open
    uint32_t* chunk_cnt;       // [nchunks + 1]: heads per chunk -> (k_sa_scan) exclusive prefix, [nchunks] = heads of the batch
    uint32_t nchunks;
};

// does row r of the batch carry the same group key as row q (q = r - 1)?
__device__ __forceinline__ bool sa_same_rows(const AggArgs& a, int64_t r, int64_t q) {
    for (int k = 0; k < a.plan.n_keys; k++) {
        const int c = a.plan.key_col[k];
        const bool nr = tsq_is_null(a.in.nulls[c], r), nq = tsq_is_null(a.in.nulls[c], q);
        if (nr != nq) return false;
        if (nr) continue;
        if (a.in.type[c] == TSQ_BYTES) {
            const int64_t o1 = a.in.offs[c][r], n1 = a.in.offs[c][r + 1] - o1, o2 = a.in.offs[c][q], n2 = a.in.offs[c][q + 1] - o2;
            if (n1 != n2) return false;
            const uint8_t* d = (const uint8_t*)a.in.data[c];
            if (tsq_cmp_bytes(d + o1, (uint32_t)n1, d + o2, (uint32_t)n2) != 0) return false;
        } else if (group_key_word(a.in, c, r) != group_key_word(a.in, c, q)) {
            return false;
        }
    }
    return true;
}
// ... as the open group in slot s (its key cells were stored by the row that opened it)
__device__ __forceinline__ bool sa_same_as_slot(const AggArgs& a, int64_t r, uint64_t s) {
    const uint32_t nullmask = a.t.gknull[s];
    for (int k = 0; k < a.plan.n_keys; k++) {
        const int c = a.plan.key_col[k];
        const bool nr = tsq_is_null(a.in.nulls[c], r);
        if (nr != (((nullmask >> k) & 1u) != 0)) return false;
        if (nr) continue;
        if (a.in.type[c] == TSQ_BYTES) {
            if (!ref_equal(a.in.data[c], a.t.gkey[k][s], str_ref(a.in, c, r, &a.counters[5]))) return false;
        } else if (a.t.gkey[k][s] != group_key_word(a.in, c, r)) {
            return false;
        }
    }
    return true;
}
__device__ __forceinline__ bool sa_is_head(const StreamAggArgs& a, int64_t r) {
    if (r > 0) return !sa_same_rows(a.u, r, r - 1);
    return a.groups_before == 0 || !sa_same_as_slot(a.u, 0, a.groups_before - 1);
}

__global__ void __launch_bounds__(TSQ_SA_NT) k_sa_count(StreamAggArgs a) {
    __shared__ uint32_t s_cnt;
    for (uint32_t ch = blockIdx.x; ch < a.nchunks; ch += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        uint32_t mine = 0;
        const int64_t lo = (int64_t)ch * TSQ_SA_CHUNK;
#pragma unroll
        for (int i = 0; i < TSQ_SA_CHUNK / TSQ_SA_NT; i++) {
            const int64_t r = lo + (int64_t)i * TSQ_SA_NT + threadIdx.x;
            if (r < a.u.nrows && sa_is_head(a, r)) mine++;
        }
        for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_cnt, mine);
        __syncthreads();
        if (threadIdx.x == 0) a.chunk_cnt[ch] = s_cnt;
        __syncthreads();
    }
}

// exclusive scan of chunk_cnt[0 .. n) in place, chunk_cnt[n] = total (one workgroup of 1024 threads)
__global__ void __launch_bounds__(1024) k_sa_scan(uint32_t* cnt, uint32_t n) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_run;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? cnt[i] : 0u;
        uint32_t x = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t before = s_run;
        for (uint32_t w = 0; w < wave; w++) before += s_w[w];
        if (i < n) cnt[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_run = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[n] = s_run;
}

__device__ __forceinline__ void sa_add128_to(unsigned long long* lo, unsigned long long* hi, uint64_t vlo, int64_t vhi) {
    const unsigned long long old = atomicAdd(lo, (unsigned long long)vlo);
    const long long d = vhi + ((old + vlo < old) ? 1 : 0);
    if (d) atomicAdd(hi, (unsigned long long)d);
}

__global__ void __launch_bounds__(TSQ_SA_NT) k_sa_update(StreamAggArgs a) {
    __shared__ uint32_t s_w[TSQ_SA_NT / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const AggArgs& u = a.u;
    for (uint32_t ch = blockIdx.x; ch < a.nchunks; ch += gridDim.x) {
        uint64_t run = a.groups_before + a.chunk_cnt[ch];  // group heads before the current tile (+ the groups before the batch)
        const int64_t lo = (int64_t)ch * TSQ_SA_CHUNK;
        for (int i = 0; i < TSQ_SA_CHUNK / TSQ_SA_NT; i++) {
            const int64_t r = lo + (int64_t)i * TSQ_SA_NT + threadIdx.x;
            const bool live = r < u.nrows;
            const bool head = live && sa_is_head(a, r);
            const unsigned long long hm = __ballot(head);
            if (lane == 0) s_w[wave] = (uint32_t)__popcll(hm);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (uint32_t w = 0; w < TSQ_SA_NT / 64; w++) {
                const uint32_t c = s_w[w];
                before += w < wave ? c : 0u;
                total += c;
            }
            __syncthreads();
            // heads in rows [0, r] of the batch, inclusive -> the row's group (row 0 of a batch that continues the open group: heads = 0)
            const uint64_t slot = run + before + (uint64_t)__popcll(hm & ((2ull << lane) - 1ull)) - 1ull;
            run += total;
            // runs inside the wave: a run ends at lane 63, at the last live row, or before the next head
            const bool tail = live && (lane == 63 || r + 1 >= u.nrows || ((hm >> (lane + 1)) & 1ull));
            const uint32_t cm = sa_cond_mask(head || !live, lane);
            if (head) {  // the row that opens the group keeps its key cells (the next batch compares against them) and marks the slot live
                u.t.tag[slot] = 1ull;
                uint32_t nullmask = 0;
                for (int k = 0; k < u.plan.n_keys; k++) {
                    const int c = u.plan.key_col[k];
                    const bool isn = tsq_is_null(u.in.nulls[c], r);
                    nullmask |= isn ? (1u << k) : 0u;
                    u.t.gkey[k][slot] = isn ? 0ull : (u.in.type[c] == TSQ_BYTES ? str_ref(u.in, c, r, &u.counters[5]) : group_key_word(u.in, c, r));
                }
                u.t.gknull[slot] = (uint8_t)nullmask;
            }
            for (int ai = 0; ai < u.plan.n_aggs; ai++) {
                const tsq_agg_func f = u.plan.f[ai];
                const AggState st = u.t.st[ai];
                const bool merge = f.mode == TSQ_MODE_FINAL || f.mode == TSQ_MODE_PARTIAL2;
                const bool arg_null = !live || (f.arg_col >= 0 ? tsq_is_null(u.in.nulls[f.arg_col], r) : false);
                switch (f.func) {
                    case TSQ_AGG_COUNT: {  // func_count.go:33-119
                        uint64_t v = arg_null ? 0ull : (merge ? ((const unsigned long long*)u.in.data[f.arg_col])[r] : 1ull);
                        v = sa_scan_add(v, cm);
                        if (tail && v) atomicAdd(&st.acc[slot], (unsigned long long)v);
                        break;
                    }
                    case TSQ_AGG_SUM:  // func_sum.go:60-154
                    case TSQ_AGG_AVG: {  // func_avg.go:62-128,164-216
                        const bool avg = f.func == TSQ_AGG_AVG;
                        const int vc = (avg && merge) ? f.arg_col2 : f.arg_col;
                        const bool vnull = arg_null || (avg && merge && tsq_is_null(u.in.nulls[vc], r));
                        uint64_t n = vnull ? 0ull : ((avg && merge) ? ((const unsigned long long*)u.in.data[f.arg_col])[r] : 1ull);
                        const uint64_t seen = sa_scan_add(vnull ? 0ull : 1ull, cm);
                        if (avg) n = sa_scan_add(n, cm);
                        if (is_real_type(f.arg_type)) {
                            double v = 0.0;
                            if (!vnull) v = u.in.type[vc] == TSQ_F32 ? (double)((const float*)u.in.data[vc])[r] : ((const double*)u.in.data[vc])[r];
                            v = sa_scan_addf(v, cm);
                            if (tail && seen) atomicAdd((double*)&st.acc[slot], v);
                        } else {
                            const int64_t x = vnull ? 0 : ((const int64_t*)u.in.data[vc])[r];
                            uint64_t vlo = (uint64_t)x;
                            int64_t vhi = x < 0 ? -1 : 0;
                            sa_scan_add128(vlo, vhi, cm);
                            if (tail && seen) sa_add128_to(&st.acc[slot], &st.aux[slot], vlo, vhi);
                        }
                        if (tail && seen) {
                            if (avg) atomicAdd(&st.cnt[slot], (unsigned long long)n);
                            else st.seen[slot] = 1;
                        }
                        break;
                    }
                    case TSQ_AGG_MAX:  // func_max_min.go:81-117 (+uint/float variants)
                    case TSQ_AGG_MIN: {
                        const bool mx = f.func == TSQ_AGG_MAX;
                        if (f.arg_type == TSQ_BYTES) {  // maxMin4String: row by row, the reference of the best string so far
                            if (arg_null) break;
                            const unsigned long long mine = str_ref(u.in, f.arg_col, r, &u.counters[5]);
                            unsigned long long cur = __hip_atomic_load(&st.acc[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            for (;;) {
                                if (cur != TSQ_REF_NONE) {
                                    const int c = ref_cmp(u.in.data[f.arg_col], mine, cur);
                                    if (mx ? c <= 0 : c >= 0) break;
                                }
                                const unsigned long long prev = atomicCAS(&st.acc[slot], cur, mine);
                                if (prev == cur) break;
                                cur = prev;
                            }
                            st.seen[slot] = 1;
                            break;
                        }
                        const uint64_t seen = sa_scan_add(arg_null ? 0ull : 1ull, cm);
                        uint64_t v = arg_null ? (mx ? 0ull : ~0ull) : ord_image(u.in, f.arg_col, f.arg_type, r);
                        v = mx ? sa_scan_max(v, cm) : sa_scan_min(v, cm);
                        if (tail && seen) {
                            if (mx) atomicMax(&st.acc[slot], (unsigned long long)v);
                            else atomicMin(&st.acc[slot], (unsigned long long)v);
                            st.seen[slot] = 1;
                        }
                        break;
                    }
                    case TSQ_AGG_FIRSTROW:  // func_first_row.go:67-81 — here the TRUE first row of the group
                        if (!head) break;
                        st.acc[slot] = arg_null ? 0ull : agg_cell(u.in, f.arg_col, r, &u.counters[5]);
                        st.seen[slot] = arg_null ? 0 : 1;
                        break;
                }
            }
        }
    }
}

#endif
