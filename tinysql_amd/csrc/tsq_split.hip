// tsq_split.hip — hash-radix redistribute step for the multi-GPU path (gfx950).
//
// CPU analogue in the reference: the partial->final shuffle of HashAggExec
// (executor/aggregate.go:352-356) and the round-robin dispatch of probe chunks
// (executor/join.go:219).  Here rows are split by rank(key) into n_parts contiguous runs; run p
// is then sent to GPU p with one RCCL all-to-all over xGMI by the caller (one process per GPU).
// Two passes over the key column, one over the payload columns:
//   K5a  histogram  : per-lane rank, block-level LDS counters, one global atomic per (block, part)
//   K5b  scatter    : wave ballot per part -> one atomicAdd per (wave, part) claims the run, lanes
//                     write their row at base + popcount prefix (wavefront match compaction).
#include "tsq_stage.h"

#define TSQ_SPLIT_MAX_PARTS 64

struct SplitArgs {
    tsq_colset in;
    int32_t key_col;
    int32_t key_mode;   // 0: join-key equality (codec.go:212-240)  1: group-key equality (codec.go:713-746)
    int32_t n_parts;
    int64_t nrows;
    unsigned long long* cursors;  // [n_parts] running output positions (pre-loaded with run offsets)
    void* out_data[TSQ_MAX_COLS];
    uint8_t* out_notnull[TSQ_MAX_COLS];
};

__device__ __forceinline__ uint32_t split_rank(const SplitArgs& a, int64_t row) {
    const int c = a.key_col;
    if (tsq_is_null(a.in.nulls[c], row)) return 0;
    uint64_t w;
    if (a.key_mode == 1 && (a.in.type[c] == TSQ_F32 || a.in.type[c] == TSQ_F64)) {
        double f = a.in.type[c] == TSQ_F32 ? (double)((const float*)a.in.data[c])[row] : ((const double*)a.in.data[c])[row];
        uint64_t u = tsq_f64_bits(f);
        w = f >= 0 ? (u | 0x8000000000000000ULL) : ~u;
    } else {
        uint32_t flag;
        w = tsq_key_word(a.in.data[c], a.in.type[c], row, &flag);
    }
    return tsq_key_rank(w, (uint32_t)a.n_parts);
}

__global__ void __launch_bounds__(256) k_split_hist(SplitArgs a, unsigned long long* counts) {
    __shared__ unsigned int lc[TSQ_SPLIT_MAX_PARTS];
    if (threadIdx.x < TSQ_SPLIT_MAX_PARTS) lc[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.nrows; r += stride) atomicAdd(&lc[split_rank(a, r)], 1u);
    __syncthreads();
    if (threadIdx.x < a.n_parts && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)lc[threadIdx.x]);
}

__global__ void __launch_bounds__(256) k_split_scatter(SplitArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nround = (a.nrows + 63) & ~(int64_t)63;
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nround; r += stride) {
        const bool active = r < a.nrows;
        const uint32_t rank = active ? split_rank(a, r) : 0xffffffffu;
        uint64_t pos = 0;
        for (int p = 0; p < a.n_parts; p++) {
            const unsigned long long m = __ballot(rank == (uint32_t)p);
            if (!m) continue;
            const int leader = __ffsll((long long)m) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(&a.cursors[p], (unsigned long long)__popcll(m));
            base = __shfl(base, leader, 64);
            if (rank == (uint32_t)p) pos = base + __popcll(m & ((1ull << lane) - 1));
        }
        if (!active) continue;
        for (int c = 0; c < a.in.n; c++) {
            const bool nn = !tsq_is_null(a.in.nulls[c], r);
            if (a.in.type[c] == TSQ_F32) ((uint32_t*)a.out_data[c])[pos] = ((const uint32_t*)a.in.data[c])[r];
            else ((uint64_t*)a.out_data[c])[pos] = ((const uint64_t*)a.in.data[c])[r];
            if (a.out_notnull[c]) a.out_notnull[c][pos] = nn ? 1 : 0;
        }
    }
}

TSQ_API tsq_status tsq_radix_split(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode, int64_t nrows,
                                   int32_t n_parts, tsq_col* out_cols, int64_t* counts_out) {
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (!cols || !out_cols || !counts_out || n_cols < 1 || n_cols > TSQ_MAX_COLS || key_col < 0 || key_col >= n_cols || nrows < 0)
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: bad arguments");
    if (n_parts < 1 || n_parts > TSQ_SPLIT_MAX_PARTS) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: 1..64 parts");
    if (key_mode != 0 && key_mode != 1) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: key_mode must be 0 (join) or 1 (group)");
    for (int c = 0; c < n_cols; c++) {
        if (!(cols[c].flags & TSQ_COL_DEVICE) || !(out_cols[c].flags & TSQ_COL_DEVICE))
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: columns must be device resident");
        if (cols[c].type < TSQ_I64 || cols[c].type > TSQ_F64) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "tsq_radix_split: var-len column");
        if (cols[c].null_bitmap && !out_cols[c].null_bitmap) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: nullable column needs an output bitmap");
        if (nrows > 0 && (!cols[c].data || !out_cols[c].data)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: NULL data pointer");
    }
    for (int p = 0; p < n_parts; p++) counts_out[p] = 0;
    if (nrows == 0) return TSQ_OK;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    SplitArgs a;
    memset(&a, 0, sizeof a);
    tsq_colset_from_cols(a.in, cols, n_cols);
    a.key_col = key_col;
    a.key_mode = key_mode;
    a.n_parts = n_parts;
    a.nrows = nrows;
    DevBuf cur;
    std::vector<DevBuf> nn(n_cols);
    auto cleanup = [&]() {
        cur.release();
        for (auto& b : nn) b.release();
    };
    tsq_status s = cur.reserve(ctx, h, 2 * TSQ_SPLIT_MAX_PARTS * 8);
    if (s != TSQ_OK) { cleanup(); return s; }
    unsigned long long* counts_d = cur.as<unsigned long long>();
    a.cursors = counts_d + TSQ_SPLIT_MAX_PARTS;
    hipError_t e = hipMemsetAsync(counts_d, 0, 2 * TSQ_SPLIT_MAX_PARTS * 8, ctx->stream);
    const int grid = tsq_grid_for(ctx, nrows, 256);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_split_hist, dim3(grid), dim3(256), 0, ctx->stream, a, counts_d);
        e = hipGetLastError();
    }
    unsigned long long hc[TSQ_SPLIT_MAX_PARTS], off[TSQ_SPLIT_MAX_PARTS];
    if (e == hipSuccess) e = hipMemcpyAsync(hc, counts_d, n_parts * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { cleanup(); return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_radix_split: ") + hipGetErrorString(e)); }
    unsigned long long acc = 0;
    for (int p = 0; p < n_parts; p++) { off[p] = acc; acc += hc[p]; counts_out[p] = (int64_t)hc[p]; }
    e = hipMemcpyAsync(a.cursors, off, n_parts * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { cleanup(); return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_radix_split: ") + hipGetErrorString(e)); }
    for (int c = 0; c < n_cols; c++) {
        a.out_data[c] = out_cols[c].data;
        if (cols[c].null_bitmap) {
            s = nn[c].reserve(ctx, h, (size_t)nrows + 16);
            if (s != TSQ_OK) { cleanup(); return s; }
            a.out_notnull[c] = nn[c].as<uint8_t>();
        }
    }
    hipLaunchKernelGGL(k_split_scatter, dim3(grid), dim3(256), 0, ctx->stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) { cleanup(); return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_radix_split: ") + hipGetErrorString(e)); }
    for (int c = 0; c < n_cols; c++) {
        if (a.out_notnull[c]) s = tsq_launch_pack_bitmap(ctx, h, a.out_notnull[c], out_cols[c].null_bitmap, nrows);
        else if (out_cols[c].null_bitmap) {
            e = hipMemsetAsync(out_cols[c].null_bitmap, 0xff, tsq_bitmap_bytes(nrows), ctx->stream);
            if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
        }
        if (s != TSQ_OK) { cleanup(); return s; }
        out_cols[c].length = nrows;
        out_cols[c].type = cols[c].type;
        out_cols[c].elem_size = tsq_elem_size(cols[c].type);
    }
    e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_radix_split: ") + hipGetErrorString(e));
    return TSQ_OK;
}
