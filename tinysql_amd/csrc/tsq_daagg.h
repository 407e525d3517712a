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
//
// Replaces (reference): HashAggPartialWorker.updatePartialResult + getGroupKey + getPartialResult
// (executor/aggregate.go:332-410) with LDS as the partial worker's map; the partial groups are merged by k_agg_merge
// (consumeIntermData, aggregate.go:424-427).  Algorithmic bytes: 16 B per row.
#ifndef TSQ_DAAGG_H
#define TSQ_DAAGG_H

#include "tsq_aggfast.h"
#include "tsq_dajoin.h"

#define TSQ_DAAGG_MAX_BITS 23

struct DaAggStore {
    uint16_t* ent;        // [P * 8 * cap] entries
    uint64_t* pay[TSQ_RADIX_MAXV];  // [P * 8 * cap] argument cells travelling with the entry
    uint32_t* cursor;     // [8][P]
    uint32_t* valid_end;  // [8][P]
    uint32_t* ovf_row;    // overflow list (runs that did not fit their region): source rows, re-read and handled row by row
    uint32_t* ovf_count;
    uint32_t ovf_cap;
    uint32_t bits, ebits, cap;
};
struct DaAggSrc {
    const void* kdata;
    const uint8_t* knulls;
    const void* vdata[TSQ_RADIX_MAXV];
    const uint8_t* vnulls[TSQ_RADIX_MAXV];
    int32_t vtype[TSQ_RADIX_MAXV];
    int64_t nrows;
    uint32_t* exc_rows;   // rows the LDS stage cannot take (NULL key / NULL argument / key outside the packed range)
    uint32_t* exc_count;
};
__device__ __forceinline__ uint32_t daagg_region_len(const DaAggStore& st, uint32_t P, uint32_t p, uint32_t r) {
    const uint32_t c = r * P + p;
    uint32_t len = st.cursor[c];
    const uint32_t ve = st.valid_end[c];
    len = len < ve ? len : ve;
    return len < st.cap ? len : st.cap;
}

// K5e — partition of (packed key entry, argument cells).  The structure of k_da_partition (tsq_dajoin.h) with V payload columns
// staged through LDS next to the words; a run that does not fit its region sends its ROWS to the exception list (they are
// aggregated row by row: exact under any skew).
template <int NT, int K, int V>
__global__ void __launch_bounds__(NT) k_daagg_partition(DaAggSrc src, DaDomain dm, DaAggStore st) {
    constexpr int T = NT * K;
    constexpr int MAXPER = (TSQ_RADIX_MAX_P + NT - 1) / NT;
    static_assert(T <= 65536 && (K % 2) == 0 && V >= 0 && V <= TSQ_RADIX_MAXV, "tile");
    __shared__ uint32_t s_u[T];
    __shared__ uint32_t s_row[T];
    __shared__ uint64_t s_pay[V ? V : 1][V ? T : 1];
    __shared__ uint32_t s_hist[TSQ_RADIX_MAX_P];
    __shared__ uint32_t s_delta[TSQ_RADIX_MAX_P];
    __shared__ uint32_t s_wsum[NT / 64];
    __shared__ uint32_t s_flag;
    const uint32_t tid = threadIdx.x;
    const uint32_t P = 1u << st.bits, ebits = st.ebits, emask = (1u << ebits) - 1u;
    const uint32_t r = tsq_xcc_id();
    const uint32_t per = P >= (uint32_t)NT ? P / NT : 1u;
    if (tid == 0) s_flag = 0;
    const int64_t ntiles = (src.nrows + T - 1) / T;
    bool wide = src.knulls == nullptr;
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
            const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>((const uint64_t*)src.kdata + base);
#pragma unroll
            for (int j = 0; j < K / 2; j++) {
                const ulonglong2 x = s2[j * NT + tid];
                k[2 * j] = x.x;
                k[2 * j + 1] = x.y;
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
                u[j] = da_word(dm, k[j]);
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
                    bool isnull = tsq_is_null(src.knulls, base + pos);
#pragma unroll
                    for (int v = 0; v < V; v++) isnull |= tsq_is_null(src.vnulls[v], base + pos);
                    if (!isnull) u[j] = da_word(dm, ((const uint64_t*)src.kdata)[base + pos]);
                    if (u[j] == TSQ_DA_NONE) except((uint32_t)base + pos);
                    else {
#pragma unroll
                        for (int v = 0; v < V; v++)
                            pay[v][j] = src.vtype[v] == TSQ_F32 ? (uint64_t)((const uint32_t*)src.vdata[v])[base + pos] : ((const uint64_t*)src.vdata[v])[base + pos];
                    }
                }
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
#pragma unroll
        for (int q = 0; q < MAXPER; q++) {
            if ((uint32_t)q < per && p0 + q < P) {
                const uint32_t p = p0 + q, cnt = c[q], offs = run;
                run += cnt;
                uint32_t flag = 0;
                if (cnt) {
                    if (g[q] + cnt > st.cap) {
                        flag = 1;
                        atomicMin(&st.valid_end[r * P + p], g[q]);
                        s_flag = 1;
                    }
                    s_delta[p] = (p * 8u + r) * st.cap + g[q] - offs;
                }
                s_hist[p] = offs | (flag << 31);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K; j++)
            if (u[j] != TSQ_DA_NONE) {
                const uint32_t d = (s_hist[u[j] >> ebits] & 0x7fffffffu) + rk[j];
                s_u[d] = u[j];
                s_row[d] = (uint32_t)base + (full ? (((uint32_t)(j >> 1) * NT + tid) * 2 + (j & 1)) : ((uint32_t)j * NT + tid));
#pragma unroll
                for (int v = 0; v < V; v++) s_pay[v][d] = pay[v][j];
            }
        __syncthreads();
        const bool any_ovf = s_flag != 0;
        for (uint32_t i = tid; i < total; i += NT) {
            const uint32_t w = s_u[i], p = w >> ebits;
            if (!any_ovf || !(s_hist[p] >> 31)) {
                const uint32_t d = s_delta[p] + i;
                st.ent[d] = (uint16_t)(w & emask);
#pragma unroll
                for (int v = 0; v < V; v++) st.pay[v][d] = s_pay[v][i];
            } else {
                except(s_row[i]);  // rare (skewed keys): the run did not fit its region — the row is aggregated row by row
            }
        }
        __syncthreads();
    }
}

// K7d — direct-addressed LDS pre-aggregation of one partition: words [k][e], one non-returning LDS atomic per word and row.
// A cell that received a row is marked in a bitmap; the partial groups (key = kmin + unmix(p : e)) leave through ONE returning
// device atomic per workgroup and a block scan, as in k_agg_lds.
struct DaAggLdsArgs {
    AfPlan plan;
    AfPartials out;
    DaAggStore st;
    DaDomain dm;
};
template <int W, int CELLS>
__global__ void __launch_bounds__(TSQ_AF_NT) k_agg_da(DaAggLdsArgs a) {
    constexpr int U = 4;
    uint32_t wd[W];
#pragma unroll
    for (int k = 0; k < W; k++) wd[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.plan.wdesc[k]);
    __shared__ unsigned long long s_w[W][CELLS];
    __shared__ uint32_t s_touch[CELLS / 32];
    __shared__ uint32_t s_base, s_wsum[TSQ_AF_NT / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t P = 1u << a.st.bits;
    for (uint32_t p = blockIdx.x; p < P; p += gridDim.x) {
        __syncthreads();
        for (uint32_t i = tid; i < (uint32_t)CELLS; i += TSQ_AF_NT) {
#pragma unroll
            for (int k = 0; k < W; k++) s_w[k][i] = a.plan.init[k];
        }
        for (uint32_t i = tid; i < (uint32_t)CELLS / 32; i += TSQ_AF_NT) s_touch[i] = 0;
        __syncthreads();
        auto apply = [&](uint32_t e, uint64_t c0, uint64_t c1) {
            atomicOr(&s_touch[e >> 5], 1u << (e & 31u));
#pragma unroll
            for (int k = 0; k < W; k++) {
                const uint32_t d = wd[k];
                const uint64_t cell = (d & 8u) ? c1 : c0;
                const int32_t type = (int32_t)(d >> 4);
                switch (d & 7u) {
                    case AF_W_ADD1: atomicAdd(&s_w[k][e], 1ull); break;
                    case AF_W_ADD_REAL: atomicAdd(reinterpret_cast<double*>(&s_w[k][e]), af_real(cell, type)); break;
                    case AF_W_ADD_LO32: atomicAdd(&s_w[k][e], (unsigned long long)(cell & 0xffffffffull)); break;
                    case AF_W_ADD_HI32: atomicAdd(&s_w[k][e], (unsigned long long)((long long)cell >> 32)); break;
                    case AF_W_MAX: atomicMax(&s_w[k][e], (unsigned long long)af_ord_image(cell, type)); break;
                    default: atomicMin(&s_w[k][e], (unsigned long long)af_ord_image(cell, type)); break;
                }
            }
        };
        for (uint32_t r = 0; r < 8; r++) {
            const uint32_t len = daagg_region_len(a.st, P, p, r);
            const size_t base = (size_t)(p * 8u + r) * a.st.cap;
            for (uint32_t i0 = tid; i0 < len; i0 += TSQ_AF_NT * U) {
                uint32_t e[U];
                uint64_t cells[U][TSQ_RADIX_MAXV];
#pragma unroll
                for (int x = 0; x < U; x++) {
                    const uint32_t i = i0 + (uint32_t)x * TSQ_AF_NT;
                    const uint32_t ic = i < len ? i : 0u;  // (a load that has nothing to fetch reads the region's first slot)
                    e[x] = a.st.ent[base + ic];
#pragma unroll
                    for (int v = 0; v < TSQ_RADIX_MAXV; v++) cells[x][v] = v < a.plan.V ? a.st.pay[v][base + ic] : 0ull;
                }
#pragma unroll
                for (int x = 0; x < U; x++)
                    if (i0 + (uint32_t)x * TSQ_AF_NT < len) apply(e[x], cells[x][0], cells[x][1]);
            }
        }
        __syncthreads();
        // ---- emit the touched cells
        uint32_t mine = 0;
        for (uint32_t i = tid; i < (uint32_t)CELLS / 32; i += TSQ_AF_NT) mine += (uint32_t)__popc(s_touch[i]);
        uint32_t used;
        (void)block_excl_scan<TSQ_AF_NT>(mine, s_wsum, &used);
        __syncthreads();
        if (tid == 0) s_base = used ? __hip_atomic_fetch_add(a.out.count, used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        __syncthreads();
        uint32_t running = s_base;
        for (uint32_t i0 = 0; i0 < (uint32_t)CELLS; i0 += TSQ_AF_NT) {
            const uint32_t i = i0 + tid;
            const bool occ = (s_touch[i >> 5] >> (i & 31u)) & 1u;
            uint32_t total;
            const uint32_t ex = block_excl_scan<TSQ_AF_NT>(occ ? 1u : 0u, s_wsum, &total);
            const uint32_t o = running + ex;
            running += total;
            __syncthreads();  // s_wsum is reused by the next pass
            if (occ && o < a.out.cap) {
                const uint32_t uu = (p << a.st.ebits) | i;
                a.out.key[o] = a.dm.kmin + (uint64_t)tsq_da_unmix(uu, a.dm.s, a.dm.mask);
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

#endif
