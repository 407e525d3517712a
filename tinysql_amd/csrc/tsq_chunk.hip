// tsq_chunk.hip — chunk-level data movement between GPU operators (gfx950).
//
// tsq_chunk_compact replaces what the reference does when an operator hands a FILTERED chunk on: SelectionExec copies the
// selected rows into the output chunk (executor/executor.go:393-438) / Column.CopyReconstruct gathers by the selection
// vector (util/chunk/column.go:504-552).  With device-resident chunks this is the hand-off between a GPU Selection and a
// GPU join / aggregate: no D2H, the dense chunk never leaves HBM.
//   K12a k_compact_count  : selected rows per workgroup (contiguous rows per workgroup)
//   K12b k_compact_scatter: exclusive bases by k_scan (one workgroup) -> LDS cursor per workgroup, wave ballot + popcount
//                           prefix, every column's cell copied to its dense position; NULL flags as bytes -> k_pack_bitmap
// Algorithmic bytes: 1 B flag + (8 B read + 8 B written per SELECTED cell).  Row order is preserved inside a wave and a
// workgroup's rows stay together, i.e. the output is the input order up to a permutation inside 256-row tiles — operators
// downstream (join probe, aggregate) are order-insensitive; Projection/Selection order guarantees of the reference
// (executor/projection.go:187-207) are kept by the host-chunk path, which does not use this entry point.
#include "tsq_stage.h"

struct CompactArgs {
    tsq_colset in;
    const uint8_t* selected;  // one byte per row (Go []bool)
    int64_t nrows;
    int64_t rows_per_block;
    unsigned long long* block_base;  // in: per-workgroup counts -> exclusive bases
    void* out_data[TSQ_MAX_COLS];
    uint8_t* out_notnull[TSQ_MAX_COLS];
    unsigned long long* total;
};

__global__ void __launch_bounds__(256) k_compact_count(CompactArgs a) {
    __shared__ unsigned int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t hi = lo + a.rows_per_block;
    hi = hi < a.nrows ? hi : a.nrows;
    unsigned int c = 0;
    for (int64_t r = lo + threadIdx.x; r < hi; r += 256) c += a.selected[r] ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0) a.block_base[blockIdx.x] = s_n;
}
// exclusive scan of n per-workgroup counts; *total = sum
__global__ void __launch_bounds__(1024) k_compact_scan(unsigned long long* v, int n, unsigned long long* total) {
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
    unsigned long long pre = 0, all = 0;
    for (int w = 0; w < 16; w++) {
        if (w < (int)(threadIdx.x >> 6)) pre += s_w[w];
        all += s_w[w];
    }
    unsigned long long run = pre + x - sum;
    for (int i = lo; i < lo + per && i < n; i++) {
        const unsigned long long c = v[i];
        v[i] = run;
        run += c;
    }
    if (threadIdx.x == 0) *total = all;
}
__global__ void __launch_bounds__(256) k_compact_scatter(CompactArgs a) {
    __shared__ unsigned long long s_cur;
    if (threadIdx.x == 0) s_cur = a.block_base[blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t lo = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t hi = lo + a.rows_per_block;
    hi = hi < a.nrows ? hi : a.nrows;
    if (hi < lo) hi = lo;
    const int64_t round = lo + ((hi - lo + 63) & ~(int64_t)63);
    for (int64_t r = lo + threadIdx.x; r < round; r += 256) {
        const bool sel = r < hi && a.selected[r];
        const unsigned long long m = __ballot(sel);
        if (!m) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&s_cur, (unsigned long long)__popcll(m));
        base = __shfl(base, 0, 64);
        if (!sel) continue;
        const unsigned long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
        for (int c = 0; c < a.in.n; c++) {
            if (a.in.type[c] == TSQ_F32) ((uint32_t*)a.out_data[c])[pos] = ((const uint32_t*)a.in.data[c])[r];
            else ((uint64_t*)a.out_data[c])[pos] = ((const uint64_t*)a.in.data[c])[r];
            if (a.out_notnull[c]) a.out_notnull[c][pos] = tsq_is_null(a.in.nulls[c], r) ? 0 : 1;
        }
    }
}

TSQ_API tsq_status tsq_chunk_compact(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, int64_t nrows, const uint8_t* selected,
                                     tsq_col* out_cols, int64_t* nrows_out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (!cols || !out_cols || !nrows_out || !selected || n_cols < 1 || n_cols > TSQ_MAX_COLS || nrows < 0)
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_compact: bad arguments");
    *nrows_out = 0;
    for (int c = 0; c < n_cols; c++) {
        if (!(cols[c].flags & TSQ_COL_DEVICE) || !(out_cols[c].flags & TSQ_COL_DEVICE))
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_compact: columns (and the selected[] flags) must be device resident");
        if (cols[c].type < TSQ_I64 || cols[c].type > TSQ_F64) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "tsq_chunk_compact: var-len column");
        if (cols[c].null_bitmap && !out_cols[c].null_bitmap) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_compact: nullable column needs an output bitmap");
        if (nrows > 0 && (!cols[c].data || !out_cols[c].data)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_compact: NULL data pointer");
    }
    if (nrows == 0) return TSQ_OK;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    CompactArgs a;
    memset(&a, 0, sizeof a);
    tsq_colset_from_cols(a.in, cols, n_cols);
    a.selected = selected;
    a.nrows = nrows;
    const int grid = tsq_grid_for(ctx, nrows, 256);
    a.rows_per_block = (((nrows + grid - 1) / grid) + 63) & ~(int64_t)63;
    DevBuf base;
    std::vector<DevBuf> nn(n_cols);
    auto cleanup = [&]() {
        base.release();
        for (auto& b : nn) b.release();
    };
    tsq_status s = base.reserve(ctx, h, (size_t)grid * 8 + 64);
    for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
        a.out_data[c] = out_cols[c].data;
        if (cols[c].null_bitmap) {
            s = nn[c].reserve(ctx, h, (size_t)nrows + 64);
            a.out_notnull[c] = nn[c].as<uint8_t>();
        }
    }
    if (s != TSQ_OK) { cleanup(); return s; }
    a.block_base = base.as<unsigned long long>();
    a.total = a.block_base + grid;
    hipLaunchKernelGGL(k_compact_count, dim3(grid), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, ctx->stream, a.block_base, grid, a.total);
    hipLaunchKernelGGL(k_compact_scatter, dim3(grid), dim3(256), 0, ctx->stream, a);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 16, a.total, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { cleanup(); return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_chunk_compact: ") + hipGetErrorString(e)); }
    const int64_t n_out = (int64_t)ctx->pinned[16];
    for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
        if (a.out_notnull[c]) s = tsq_launch_pack_bitmap(ctx, h, a.out_notnull[c], out_cols[c].null_bitmap, n_out);
        else if (out_cols[c].null_bitmap && n_out > 0) {
            e = hipMemsetAsync(out_cols[c].null_bitmap, 0xff, tsq_bitmap_bytes(n_out), ctx->stream);
            if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
        }
        out_cols[c].length = n_out;
        out_cols[c].type = cols[c].type;
        out_cols[c].elem_size = tsq_elem_size(cols[c].type);
    }
    if (s == TSQ_OK) {
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
    }
    cleanup();
    if (s != TSQ_OK) return s;
    *nrows_out = n_out;
    return TSQ_OK;
}
