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
#define TSQ_SA_CHUNK 4096  // rows per WAVE: its stripe, walked 64 rows at a time

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

// heads per chunk: a chunk = TSQ_SA_CHUNK consecutive rows = the stripe ONE WAVE walks in k_sa_update
__global__ void __launch_bounds__(TSQ_SA_NT) k_sa_count(StreamAggArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * (TSQ_SA_NT / 64), w0 = blockIdx.x * (TSQ_SA_NT / 64) + (threadIdx.x >> 6);
    for (uint32_t ch = w0; ch < a.nchunks; ch += waves) {
        const int64_t lo = (int64_t)ch * TSQ_SA_CHUNK;
        uint32_t mine = 0;
        for (int i = 0; i < TSQ_SA_CHUNK / 64; i++) {
            const int64_t r = lo + (int64_t)i * 64 + lane;
            if (r < a.u.nrows && sa_is_head(a, r)) mine++;
        }
        for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if (lane == 0) a.chunk_cnt[ch] = mine;
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

// The partial result of ONE aggregate over a run of rows, as four words: A = count | sum lo | sum (double bits) | max / min image,
// B = sum hi, C = rows that carried a value, D = AVG's count.  `kind`: how two partials combine and how one is added to the group.
enum { SA_K_NONE = 0, SA_K_COUNT = 1, SA_K_SUMI = 2, SA_K_SUMR = 3, SA_K_MAX = 4, SA_K_MIN = 5 };
struct SaPart {
    uint64_t A, B, C, D;
};
__device__ __forceinline__ void sa_combine(int kind, SaPart& x, const SaPart& p) {  // x := x (+) p, p = the earlier rows of the same run
    switch (kind) {
        case SA_K_COUNT: x.A += p.A; break;
        case SA_K_SUMI: {
            const uint64_t lo = x.A + p.A;
            x.B = (uint64_t)((int64_t)x.B + (int64_t)p.B + (lo < x.A ? 1 : 0));
            x.A = lo;
            x.C += p.C;
            x.D += p.D;
            break;
        }
        case SA_K_SUMR: x.A = tsq_f64_bits(tsq_bits_f64(p.A) + tsq_bits_f64(x.A)); x.C += p.C; x.D += p.D; break;
        case SA_K_MAX: if (p.C && (!x.C || p.A > x.A)) x.A = p.A; x.C += p.C; break;
        case SA_K_MIN: if (p.C && (!x.C || p.A < x.A)) x.A = p.A; x.C += p.C; break;
        default: break;
    }
}
__device__ __forceinline__ void sa_apply(int kind, bool avg, const AggState& st, uint64_t slot, const SaPart& x) {
    switch (kind) {
        case SA_K_COUNT: if (x.A) atomicAdd(&st.acc[slot], (unsigned long long)x.A); break;
        case SA_K_SUMI:
            if (!x.C) break;
            sa_add128_to(&st.acc[slot], &st.aux[slot], x.A, (int64_t)x.B);
            if (avg) atomicAdd(&st.cnt[slot], (unsigned long long)x.D);
            else st.seen[slot] = 1;
            break;
        case SA_K_SUMR:
            if (!x.C) break;
            atomicAdd((double*)&st.acc[slot], tsq_bits_f64(x.A));
            if (avg) atomicAdd(&st.cnt[slot], (unsigned long long)x.D);
            else st.seen[slot] = 1;
            break;
        case SA_K_MAX: if (x.C) { atomicMax(&st.acc[slot], (unsigned long long)x.A); st.seen[slot] = 1; } break;
        case SA_K_MIN: if (x.C) { atomicMin(&st.acc[slot], (unsigned long long)x.A); st.seen[slot] = 1; } break;
        default: break;
    }
}

// One WAVE walks a stripe of TSQ_SA_CHUNK consecutive rows, 64 at a time.  Inside a step the runs of equal groups are reduced with
// segmented shuffle scans; the LAST run of a step stays open — its partial result waits in LDS (s_pend) and is combined into the
// first run of the next step when that step's first row is no head — so a group costs one set of device atomics per STRIPE it
// touches (a device-scope atomic goes to memory: ~2.5e9 of them per second chip-wide were 1.8 ms per 1e8 rows when every step of
// every wave added its runs to the groups), not per 64 rows.  No block barrier: the waves of a workgroup share nothing.
__global__ void __launch_bounds__(TSQ_SA_NT) k_sa_update(StreamAggArgs a) {
    __shared__ unsigned long long s_pend[TSQ_SA_NT / 64][TSQ_MAX_AGGS][4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const AggArgs& u = a.u;
    volatile unsigned long long (*pend)[4] = s_pend[wave];
    const uint32_t waves = gridDim.x * (TSQ_SA_NT / 64), w0 = blockIdx.x * (TSQ_SA_NT / 64) + wave;
    for (uint32_t ch = w0; ch < a.nchunks; ch += waves) {
        uint64_t run = a.groups_before + a.chunk_cnt[ch];  // group heads before the current step (+ the groups before the batch)
        const int64_t lo = (int64_t)ch * TSQ_SA_CHUNK;
        bool pend_valid = false;   // (wave-uniform) a run of the previous step is still open ...
        uint64_t pend_slot = 0;    // ... for this group
        for (int i = 0; i < TSQ_SA_CHUNK / 64; i++) {
            const int64_t r = lo + (int64_t)i * 64 + lane;
            if (lo + (int64_t)i * 64 >= u.nrows) break;  // (wave-uniform)
            const bool live = r < u.nrows;
            const bool head = live && sa_is_head(a, r);
            const unsigned long long hm = __ballot(head);
            const unsigned long long lm = __ballot(live);
            // heads in rows [0, r] of the batch, inclusive -> the row's group (row 0 of a batch that continues the open group: heads = 0)
            const uint64_t slot = run + (uint64_t)__popcll(hm & ((2ull << lane) - 1ull)) - 1ull;
            run += (uint64_t)__popcll(hm);
            // runs inside the step: a run ends at the last live lane or before the next head
            const uint32_t last_live = 63u - (uint32_t)__builtin_clzll(lm);
            const bool tail = live && (lane == last_live || ((hm >> (lane + 1)) & 1ull));
            const uint32_t cm = sa_cond_mask(head || !live, lane);
            const unsigned long long hle = hm & ((2ull << lane) - 1ull);
            const uint32_t run_rows = lane + 1u - (hle ? 63u - (uint32_t)__builtin_clzll(hle) : 0u);  // rows of the lane's run inside the step, up to the lane
            // the open run of the previous step: continued by this step's first run (lane 0 is no head), or closed now
            const bool cont = pend_valid && !(hm & 1ull);
            const bool flush = pend_valid && !cont;
            // the first run's tail combines the waiting partial; the last run's tail (the last live lane) waits itself when rows of this
            // stripe follow
            const unsigned long long tm = __ballot(tail);
            const bool first_tail = tail && lane == (uint32_t)__builtin_ctzll(tm);
            const bool more = i + 1 < TSQ_SA_CHUNK / 64 && lo + (int64_t)(i + 1) * 64 < u.nrows;
            const bool defer = more && tail && lane == last_live;
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
                const bool no_nulls = f.arg_col < 0 || u.in.nulls[f.arg_col] == nullptr;  // (wave-uniform: a kernel argument)
                const bool avg = f.func == TSQ_AGG_AVG;
                int kind = SA_K_NONE;
                SaPart x{0, 0, 0, 0};
                switch (f.func) {
                    case TSQ_AGG_COUNT:  // func_count.go:33-119
                        kind = SA_K_COUNT;
                        if (!merge && no_nulls) x.A = run_rows;
                        else x.A = sa_scan_add(arg_null ? 0ull : (merge ? ((const unsigned long long*)u.in.data[f.arg_col])[r] : 1ull), cm);
                        break;
                    case TSQ_AGG_SUM:    // func_sum.go:60-154
                    case TSQ_AGG_AVG: {  // func_avg.go:62-128,164-216
                        const int vc = (avg && merge) ? f.arg_col2 : f.arg_col;
                        const bool vnull = arg_null || (avg && merge && tsq_is_null(u.in.nulls[vc], r));
                        const bool plain = no_nulls && !(avg && merge);  // every live row of the run carries a value: counts are run lengths
                        x.C = plain ? (uint64_t)run_rows : sa_scan_add(vnull ? 0ull : 1ull, cm);
                        if (avg) x.D = plain ? (uint64_t)run_rows : sa_scan_add(vnull ? 0ull : ((avg && merge) ? ((const unsigned long long*)u.in.data[f.arg_col])[r] : 1ull), cm);
                        if (is_real_type(f.arg_type)) {
                            kind = SA_K_SUMR;
                            double v = 0.0;
                            if (!vnull) v = u.in.type[vc] == TSQ_F32 ? (double)((const float*)u.in.data[vc])[r] : ((const double*)u.in.data[vc])[r];
                            x.A = tsq_f64_bits(sa_scan_addf(v, cm));
                        } else {
                            kind = SA_K_SUMI;
                            const int64_t v = vnull ? 0 : ((const int64_t*)u.in.data[vc])[r];
                            uint64_t vlo = (uint64_t)v;
                            int64_t vhi = v < 0 ? -1 : 0;
                            // 64 addends below 2^56 in magnitude cannot leave 64 bits: one 64-bit scan; any larger cell in the wave: 128 bits
                            if (__ballot(v > (1ll << 56) || v < -(1ll << 56)) == 0ull) {
                                vlo = sa_scan_add(vlo, cm);
                                vhi = (int64_t)vlo < 0 ? -1 : 0;
                            } else {
                                sa_scan_add128(vlo, vhi, cm);
                            }
                            x.A = vlo;
                            x.B = (uint64_t)vhi;
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
                        kind = mx ? SA_K_MAX : SA_K_MIN;
                        x.C = no_nulls ? (uint64_t)run_rows : sa_scan_add(arg_null ? 0ull : 1ull, cm);
                        const uint64_t v = arg_null ? (mx ? 0ull : ~0ull) : ord_image(u.in, f.arg_col, f.arg_type, r);
                        x.A = mx ? sa_scan_max(v, cm) : sa_scan_min(v, cm);
                        break;
                    }
                    case TSQ_AGG_FIRSTROW:  // func_first_row.go:67-81 — here the TRUE first row of the group
                        if (!head) break;
                        st.acc[slot] = arg_null ? 0ull : agg_cell(u.in, f.arg_col, r, &u.counters[5]);
                        st.seen[slot] = arg_null ? 0 : 1;
                        break;
                }
                if (kind == SA_K_NONE) continue;  // (wave-uniform: the plan decides)
                SaPart p{0, 0, 0, 0};
                if (pend_valid && (first_tail || (flush && lane == 0))) {
                    p.A = pend[ai][0]; p.B = pend[ai][1]; p.C = pend[ai][2]; p.D = pend[ai][3];
                }
                if (flush && lane == 0) sa_apply(kind, avg, st, pend_slot, p);  // the group of the previous step closed at the step boundary
                if (cont && first_tail) sa_combine(kind, x, p);
                if (defer) {
                    pend[ai][0] = x.A; pend[ai][1] = x.B; pend[ai][2] = x.C; pend[ai][3] = x.D;
                } else if (tail) {
                    sa_apply(kind, avg, st, slot, x);
                }
            }
            // the run that waits after this step (if any): the last live lane's
            pend_valid = more;
            pend_slot = (uint64_t)__shfl((unsigned long long)slot, (int)last_live, 64);
        }
        // (the last step of a stripe defers nothing: `more` is false there)
    }
}

// ---------------------------------------------------------------- the same walk with per-LANE accumulators (plans with <= 4 reducing aggregates)
// k_sa_update reduces every step across the lanes (segmented shuffle scans: ~30 LDS-crossbar operations per 64 rows) although, with
// runs of hundreds of rows, most steps lie INSIDE one run and need no cross-lane work at all: 2.2 ms per 1e8 rows, ten times the
// time k_sa_count needs to read the same keys.  Here a lane keeps the partial result of the open run over ITS rows in registers
// (acc[q], q < NA: the plan's reducing aggregates, listed by the host); a step without a group head adds its row to them and is
// done.  Only a step that contains heads pays: the open run's lane partials are reduced across the wave (xor butterfly), the
// step's own runs are scanned as in k_sa_update, closed runs go to their groups, the step's last run becomes the new open run.
struct SaRed {
    int32_t n;
    int32_t agg[4];  // indexes into plan.f[]
};
__device__ __forceinline__ int sa_kind_of(const tsq_agg_func& f) {
    switch (f.func) {
        case TSQ_AGG_COUNT: return SA_K_COUNT;
        case TSQ_AGG_SUM:
        case TSQ_AGG_AVG: return is_real_type(f.arg_type) ? SA_K_SUMR : SA_K_SUMI;
        case TSQ_AGG_MAX: return f.arg_type == TSQ_BYTES ? SA_K_NONE : SA_K_MAX;
        case TSQ_AGG_MIN: return f.arg_type == TSQ_BYTES ? SA_K_NONE : SA_K_MIN;
        default: return SA_K_NONE;
    }
}
// what row r alone contributes to aggregate f (live rows only)
__device__ __forceinline__ SaPart sa_row_part(const AggArgs& u, const tsq_agg_func& f, int kind, int64_t r) {
    SaPart x{0, 0, 0, 0};
    const bool merge = f.mode == TSQ_MODE_FINAL || f.mode == TSQ_MODE_PARTIAL2;
    const bool arg_null = f.arg_col >= 0 ? tsq_is_null(u.in.nulls[f.arg_col], r) : false;
    switch (kind) {
        case SA_K_COUNT:
            x.A = arg_null ? 0ull : (merge ? ((const unsigned long long*)u.in.data[f.arg_col])[r] : 1ull);
            break;
        case SA_K_SUMI:
        case SA_K_SUMR: {
            const bool avg = f.func == TSQ_AGG_AVG;
            const int vc = (avg && merge) ? f.arg_col2 : f.arg_col;
            const bool vnull = arg_null || (avg && merge && tsq_is_null(u.in.nulls[vc], r));
            if (vnull) break;
            x.C = 1;
            x.D = (avg && merge) ? ((const unsigned long long*)u.in.data[f.arg_col])[r] : 1ull;
            if (kind == SA_K_SUMR) {
                x.A = tsq_f64_bits(u.in.type[vc] == TSQ_F32 ? (double)((const float*)u.in.data[vc])[r] : ((const double*)u.in.data[vc])[r]);
            } else {
                const int64_t v = ((const int64_t*)u.in.data[vc])[r];
                x.A = (uint64_t)v;
                x.B = (uint64_t)(v < 0 ? -1ll : 0ll);
            }
            break;
        }
        case SA_K_MAX:
        case SA_K_MIN:
            if (arg_null) break;
            x.C = 1;
            x.A = ord_image(u.in, f.arg_col, f.arg_type, r);
            break;
        default: break;
    }
    return x;
}
__device__ __forceinline__ SaPart sa_wave_reduce(int kind, SaPart x) {  // every lane gets the combination over the wave
    for (int o = 32; o; o >>= 1) {
        SaPart y;
        y.A = __shfl_xor(x.A, o, 64);
        y.B = __shfl_xor(x.B, o, 64);
        y.C = __shfl_xor(x.C, o, 64);
        y.D = __shfl_xor(x.D, o, 64);
        sa_combine(kind, x, y);
    }
    return x;
}
// segmented inclusive scan of a partial under the step's run boundaries (cm: sa_cond_mask)
__device__ __forceinline__ SaPart sa_seg_scan(int kind, SaPart x, uint32_t cm) {
    for (int k = 0; k < 6; k++) {
        SaPart y;
        y.A = __shfl_up(x.A, 1 << k, 64);
        y.B = __shfl_up(x.B, 1 << k, 64);
        y.C = __shfl_up(x.C, 1 << k, 64);
        y.D = __shfl_up(x.D, 1 << k, 64);
        if ((cm >> k) & 1u) sa_combine(kind, x, y);
    }
    return x;
}
// what the step loop needs of one reducing aggregate, read from the kernel arguments ONCE per wave (the first version looked the plan up
// for every 64-row step: 285 scalar instructions per step — the CU's one scalar unit was the kernel's bottleneck, SQ counters in
// profiles/r05_streamagg_sq.txt)
struct SaQ {
    const void* v;        // the value column (SUM / AVG / MAX / MIN), or the count column of a merging COUNT
    const uint8_t* vn;
    const void* c;        // AVG in a merge mode: its count column
    const uint8_t* cn;
    int32_t vtype;        // TSQ_* of the value cells as stored
    int32_t otype;        // MAX / MIN: the type whose order image is taken
    int kind;
    bool avg, merge;
};
__device__ __forceinline__ SaQ sa_q_of(const AggArgs& u, const tsq_agg_func& f) {
    SaQ q;
    q.kind = sa_kind_of(f);
    q.avg = f.func == TSQ_AGG_AVG;
    q.merge = f.mode == TSQ_MODE_FINAL || f.mode == TSQ_MODE_PARTIAL2;
    const int vc = (q.avg && q.merge) ? f.arg_col2 : f.arg_col;
    q.v = vc >= 0 ? u.in.data[vc] : nullptr;
    q.vn = vc >= 0 ? u.in.nulls[vc] : nullptr;
    q.vtype = vc >= 0 ? u.in.type[vc] : TSQ_I64;
    q.otype = f.arg_type;
    q.c = (q.avg && q.merge) ? u.in.data[f.arg_col] : nullptr;
    q.cn = (q.avg && q.merge) ? u.in.nulls[f.arg_col] : nullptr;
    return q;
}
__device__ __forceinline__ SaPart sa_row_part_q(const SaQ& q, int64_t r) {
    SaPart x{0, 0, 0, 0};
    const bool vnull = q.v != nullptr && tsq_is_null(q.vn, r);
    switch (q.kind) {
        case SA_K_COUNT:
            x.A = vnull ? 0ull : (q.merge ? ((const unsigned long long*)q.v)[r] : 1ull);
            break;
        case SA_K_SUMI:
        case SA_K_SUMR: {
            if (vnull || (q.c != nullptr && tsq_is_null(q.cn, r))) break;
            x.C = 1;
            x.D = q.c != nullptr ? ((const unsigned long long*)q.c)[r] : 1ull;
            if (q.kind == SA_K_SUMR) {
                x.A = tsq_f64_bits(q.vtype == TSQ_F32 ? (double)((const float*)q.v)[r] : ((const double*)q.v)[r]);
            } else {
                const int64_t v = ((const int64_t*)q.v)[r];
                x.A = (uint64_t)v;
                x.B = (uint64_t)(v < 0 ? -1ll : 0ll);
            }
            break;
        }
        case SA_K_MAX:
        case SA_K_MIN: {
            if (vnull) break;
            x.C = 1;
            switch (q.otype) {  // (ord_image of tsq_agg.hip, on the column read above)
                case TSQ_I64: x.A = ((const uint64_t*)q.v)[r] ^ 0x8000000000000000ULL; break;
                case TSQ_U64: x.A = ((const uint64_t*)q.v)[r]; break;
                default: {
                    const double f = q.vtype == TSQ_F32 ? (double)((const float*)q.v)[r] : ((const double*)q.v)[r];
                    const uint64_t b = tsq_f64_bits(f);
                    x.A = (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
                }
            }
            break;
        }
        default: break;
    }
    return x;
}
// KINDS: the aggregates' kinds (SA_K_*, 4 bits each, aggregate q at bits 4q) known at compile time for the commonest plans — the
// switches over the kind fold away (they were scalar compares and branches in every step); 0: read them from the plan
template <int NA, uint32_t KINDS>
__global__ void __launch_bounds__(TSQ_SA_NT) k_sa_update_lanes(StreamAggArgs a, SaRed red) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const AggArgs& u = a.u;
    int kind[NA];
    bool avg[NA];
    SaQ qd[NA];
    AggState qst[NA];
#pragma unroll
    for (int q = 0; q < NA; q++) {
        qd[q] = sa_q_of(u, u.plan.f[red.agg[q]]);
        kind[q] = KINDS ? (int)((KINDS >> (4 * q)) & 15u) : qd[q].kind;
        qd[q].kind = kind[q];
        avg[q] = qd[q].avg;
        qst[q] = u.t.st[red.agg[q]];
    }
    // one fixed-width key column (the common plan): the head test reads two words and two bitmap bits, no plan walk
    const bool key1 = u.plan.n_keys == 1 && u.in.type[u.plan.key_col[0]] != TSQ_BYTES;
    const int kc0 = u.plan.n_keys > 0 ? u.plan.key_col[0] : 0;
    const void* kdata = u.in.data[kc0];
    const uint8_t* knulls = u.in.nulls[kc0];
    const bool kreal = u.in.type[kc0] == TSQ_F32 || u.in.type[kc0] == TSQ_F64;
    const bool k32 = u.in.type[kc0] == TSQ_F32;
    auto key_word1 = [&](int64_t row) -> uint64_t {
        if (!kreal) return ((const uint64_t*)kdata)[row];
        const double f = k32 ? (double)((const float*)kdata)[row] : ((const double*)kdata)[row];
        const uint64_t b = tsq_f64_bits(f);
        return f >= 0 ? (b | 0x8000000000000000ULL) : ~b;  // (group_key_word: float.go:22-30)
    };
    auto is_head = [&](int64_t row) -> bool {
        if (!key1 || row == 0) return sa_is_head(a, row);
        const bool n1 = tsq_is_null(knulls, row), n0 = tsq_is_null(knulls, row - 1);
        if (n1 != n0) return true;
        return !n1 && key_word1(row) != key_word1(row - 1);
    };
    const uint32_t waves = gridDim.x * (TSQ_SA_NT / 64), w0 = blockIdx.x * (TSQ_SA_NT / 64) + wave;
    for (uint32_t ch = w0; ch < a.nchunks; ch += waves) {
        uint64_t run = a.groups_before + a.chunk_cnt[ch];
        const int64_t lo = (int64_t)ch * TSQ_SA_CHUNK;
        SaPart acc[NA];
#pragma unroll
        for (int q = 0; q < NA; q++) acc[q] = SaPart{0, 0, 0, 0};
        bool open_valid = false;  // (wave-uniform) acc holds the lane partials of an open run ...
        uint64_t open_slot = 0;   // ... of this group
        auto close_open = [&]() {  // the open run's total -> its group (lane 0 applies)
#pragma unroll
            for (int q = 0; q < NA; q++) {
                const SaPart p = sa_wave_reduce(kind[q], acc[q]);
                if (lane == 0) sa_apply(kind[q], avg[q], qst[q], open_slot, p);
                acc[q] = SaPart{0, 0, 0, 0};
            }
        };
        for (int i = 0; i < TSQ_SA_CHUNK / 64; i++) {
            if (lo + (int64_t)i * 64 >= u.nrows) break;  // (wave-uniform)
            const int64_t r = lo + (int64_t)i * 64 + lane;
            const bool live = r < u.nrows;
            const bool head = live && is_head(r);
            const unsigned long long hm = __ballot(head);
            const uint64_t slot = run + (uint64_t)__popcll(hm & ((2ull << lane) - 1ull)) - 1ull;
            run += (uint64_t)__popcll(hm);
            if (head) {  // the row that opens the group keeps its key cells (the next batch compares against them), marks the slot live and
                         // is every FIRST_ROW's row (func_first_row.go:67-81)
                u.t.tag[slot] = 1ull;
                uint32_t nullmask = 0;
                for (int k = 0; k < u.plan.n_keys; k++) {
                    const int c = u.plan.key_col[k];
                    const bool isn = tsq_is_null(u.in.nulls[c], r);
                    nullmask |= isn ? (1u << k) : 0u;
                    u.t.gkey[k][slot] = isn ? 0ull : (u.in.type[c] == TSQ_BYTES ? str_ref(u.in, c, r, &u.counters[5]) : group_key_word(u.in, c, r));
                }
                u.t.gknull[slot] = (uint8_t)nullmask;
                for (int ai = 0; ai < u.plan.n_aggs; ai++) {
                    const tsq_agg_func f = u.plan.f[ai];
                    if (f.func != TSQ_AGG_FIRSTROW) continue;
                    const bool arg_null = tsq_is_null(u.in.nulls[f.arg_col], r);
                    u.t.st[ai].acc[slot] = arg_null ? 0ull : agg_cell(u.in, f.arg_col, r, &u.counters[5]);
                    u.t.st[ai].seen[slot] = arg_null ? 0 : 1;
                }
            }
            SaPart own[NA];
#pragma unroll
            for (int q = 0; q < NA; q++) own[q] = live ? sa_row_part_q(qd[q], r) : SaPart{0, 0, 0, 0};
            if (hm == 0ull) {  // the step lies inside the open run (row 0 of a batch that continues the open group included: open_slot from `run`)
                if (!open_valid) { open_valid = true; open_slot = run - 1ull; }
#pragma unroll
                for (int q = 0; q < NA; q++) sa_combine(kind[q], acc[q], own[q]);
                continue;
            }
            if (hm == 1ull) {  // a group starts exactly at the step: the open run closes, the step is the new one
                if (open_valid) close_open();
#pragma unroll
                for (int q = 0; q < NA; q++) acc[q] = own[q];
                open_valid = true;
                open_slot = run - 1ull;
                continue;
            }
            // heads inside the step: scan its runs; the first run continues the open one (unless lane 0 is a head), the last run stays open
            const unsigned long long lm = __ballot(live);
            const uint32_t last_live = 63u - (uint32_t)__builtin_clzll(lm);
            const bool tail = live && (lane == last_live || ((hm >> (lane + 1)) & 1ull));
            const uint32_t cm = sa_cond_mask(head || !live, lane);
            const unsigned long long tm = __ballot(tail);
            const bool first_tail = tail && lane == (uint32_t)__builtin_ctzll(tm);
            const bool cont = open_valid && !(hm & 1ull);
            const bool close_first = open_valid && (hm & 1ull);
#pragma unroll
            for (int q = 0; q < NA; q++) {
                SaPart p{0, 0, 0, 0};
                if (open_valid) p = sa_wave_reduce(kind[q], acc[q]);
                if (close_first && lane == 0) sa_apply(kind[q], avg[q], qst[q], open_slot, p);
                SaPart x = sa_seg_scan(kind[q], own[q], cm);
                if (cont && first_tail) sa_combine(kind[q], x, p);
                if (tail && lane != last_live) sa_apply(kind[q], avg[q], qst[q], slot, x);
                acc[q] = (lane == last_live) ? x : SaPart{0, 0, 0, 0};  // the step's last run is the open run now (its total sits in one lane)
            }
            open_valid = true;
            open_slot = run - 1ull;
        }
        if (open_valid) close_open();
    }
}

#endif
