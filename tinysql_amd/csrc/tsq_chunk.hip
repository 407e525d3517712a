// tsq_chunk.hip — chunk-level data movement between GPU operators (gfx950).
//
// tsq_chunk_compact replaces what the reference does when an operator hands a FILTERED chunk on: SelectionExec copies the
// selected rows into the output chunk (executor/executor.go:393-438) / Column.CopyReconstruct gathers by the selection
// vector (util/chunk/column.go:504-552).  With device-resident chunks this is the hand-off between a GPU Selection and a
// GPU join / aggregate: no D2H, the dense chunk never leaves HBM.
//   K12a k_compact_count  : selected rows per WAVE (every wave owns a contiguous run of rows)
//   K12b k_compact_scatter: exclusive bases by k_compact_scan (one workgroup) -> a running cursor per wave in a register, wave
//                           ballot + popcount prefix, every column's cell copied to its dense position; NULL flags as bytes ->
//                           k_pack_bitmap
// Algorithmic bytes: 1 B flag + (8 B read + 8 B written per SELECTED cell).  ROW ORDER IS PRESERVED: a wave walks its rows in
// order and the waves' output ranges follow each other — what SelectionExec / CopyReconstruct guarantee (a Limit above a
// Selection, a keep-order table scan: store/mockstore/mocktikv/executor.go:360-390, :472-507).  (The first version handed out
// positions from one LDS cursor per workgroup: the four waves raced for it and the output was the input order only up to a
// permutation inside 256-row tiles.)
#include "tsq_stage.h"

struct CompactArgs {
    tsq_colset in;
    const uint8_t* selected;  // one byte per row (Go []bool)
    int64_t nrows;
    int64_t rows_per_wave;           // rows of one wave's contiguous run (a multiple of 64)
    unsigned long long* block_base;  // per-wave counts -> exclusive bases (4 per workgroup)
    void* out_data[TSQ_MAX_COLS];
    uint8_t* out_notnull[TSQ_MAX_COLS];
    int64_t* out_offs[TSQ_MAX_COLS];  // var-len columns: the scatter leaves the cell lengths here, a scan makes them offsets
    uint32_t* src_row;                // var-len columns: source row of every dense row (for the byte copy)
    unsigned long long* total;
};

__global__ void __launch_bounds__(256) k_compact_count(CompactArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // this wave's run of rows
    const int64_t lo = u * a.rows_per_wave;
    int64_t hi = lo + a.rows_per_wave;
    hi = hi < a.nrows ? hi : a.nrows;
    unsigned int c = 0;
    for (int64_t r = lo + lane; r < hi; r += 64) c += a.selected[r] ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) a.block_base[u] = c;
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
    const int lane = threadIdx.x & 63;
    const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t lo = u * a.rows_per_wave;  // a multiple of 64
    int64_t hi = lo + a.rows_per_wave;
    hi = hi < a.nrows ? hi : a.nrows;
    unsigned long long cur = a.block_base[u];  // first output row of this wave's run (wave uniform)
    for (int64_t r = lo + lane; r - lane < hi; r += 64) {
        const bool sel = r < hi && a.selected[r];
        const unsigned long long m = __ballot(sel);
        const unsigned long long pos = cur + __popcll(m & ((1ull << lane) - 1ull));
        cur += (unsigned long long)__popcll(m);
        if (!sel) continue;
        if (a.src_row) a.src_row[pos] = (uint32_t)r;
        for (int c = 0; c < a.in.n; c++) {
            if (a.in.type[c] == TSQ_BYTES) a.out_offs[c][pos] = a.in.offs[c][r + 1] - a.in.offs[c][r];
            else if (a.in.type[c] == TSQ_F32) ((uint32_t*)a.out_data[c])[pos] = ((const uint32_t*)a.in.data[c])[r];
            else ((uint64_t*)a.out_data[c])[pos] = ((const uint64_t*)a.in.data[c])[r];
            if (a.out_notnull[c]) a.out_notnull[c][pos] = tsq_is_null(a.in.nulls[c], r) ? 0 : 1;
        }
    }
}
// the bytes of the selected cells of one var-len column (Column.CopyReconstruct of a var-len column, column.go:504-552): one cell per
// lane for short cells, one per wave (64 lanes on consecutive bytes) for long ones
template <bool WAVE>
__global__ void __launch_bounds__(256) k_compact_varlen(const uint8_t* src, const int64_t* src_offs, const uint32_t* src_row, const int64_t* out_offs,
                                                        uint8_t* dst, int64_t n_out) {
    const int lane = threadIdx.x & 63;
    const int64_t me = WAVE ? ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6 : (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = WAVE ? ((int64_t)gridDim.x * blockDim.x) >> 6 : (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = me; p < n_out; p += step) {
        const uint8_t* s = src + src_offs[src_row[p]];
        uint8_t* d = dst + out_offs[p];
        const int64_t n = out_offs[p + 1] - out_offs[p];
        if (WAVE) {
            for (int64_t i = lane; i < n; i += 64) d[i] = s[i];
        } else {
            tsq_copy_cell(d, s, n);
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
        if (cols[c].type < TSQ_I64 || cols[c].type > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_compact: unknown column type");
        if (cols[c].null_bitmap && !out_cols[c].null_bitmap) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_compact: nullable column needs an output bitmap");
        if (cols[c].type == TSQ_BYTES) {
            // a var-len output column needs offsets[nrows + 1] and a data array as large as the input's (a selection never grows)
            if (nrows > 0 && (!cols[c].offsets || !out_cols[c].offsets)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_compact: var-len column needs offsets on both sides");
            if (nrows >= 0xffffffffLL) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "tsq_chunk_compact: var-len chunk beyond 2^32 rows");
        } else if (nrows > 0 && (!cols[c].data || !out_cols[c].data)) {
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_compact: NULL data pointer");
        }
    }
    if (nrows == 0) {
        for (int c = 0; c < n_cols; c++)
            if (cols[c].type == TSQ_BYTES && out_cols[c].offsets) TSQ_HIP(h, hipMemsetAsync(out_cols[c].offsets, 0, 8, ctx->stream));
        return TSQ_OK;
    }
    TSQ_HIP(h, hipSetDevice(ctx->device));
    CompactArgs a;
    memset(&a, 0, sizeof a);
    tsq_colset_from_cols(a.in, cols, n_cols);
    a.selected = selected;
    a.nrows = nrows;
    const int grid = tsq_grid_for(ctx, nrows, 256);
    const int n_runs = grid * 4;
    a.rows_per_wave = (((nrows + n_runs - 1) / n_runs) + 63) & ~(int64_t)63;
    DevBuf base, srow, scratch;
    std::vector<DevBuf> nn(n_cols);
    auto cleanup = [&]() {
        base.release();
        srow.release();
        scratch.release();
        for (auto& b : nn) b.release();
    };
    tsq_status s = base.reserve(ctx, h, (size_t)n_runs * 8 + 64);
    bool any_var = false;
    for (int c = 0; c < n_cols; c++) any_var = any_var || cols[c].type == TSQ_BYTES;
    if (s == TSQ_OK && any_var) {
        s = srow.reserve(ctx, h, (size_t)nrows * 4 + 64);
        a.src_row = srow.as<uint32_t>();
    }
    for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
        a.out_data[c] = out_cols[c].data;
        a.out_offs[c] = cols[c].type == TSQ_BYTES ? out_cols[c].offsets : nullptr;
        if (cols[c].null_bitmap) {
            s = nn[c].reserve(ctx, h, (size_t)nrows + 64);
            a.out_notnull[c] = nn[c].as<uint8_t>();
        }
    }
    if (s != TSQ_OK) { cleanup(); return s; }
    a.block_base = base.as<unsigned long long>();
    a.total = a.block_base + n_runs;
    hipLaunchKernelGGL(k_compact_count, dim3(grid), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, ctx->stream, a.block_base, n_runs, a.total);
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
        out_cols[c].elem_size = cols[c].type == TSQ_BYTES ? -1 : tsq_elem_size(cols[c].type);
        if (cols[c].type == TSQ_BYTES && s == TSQ_OK) {  // lengths -> offsets, then the bytes
            s = tsq_launch_scan64(ctx, h, out_cols[c].offsets, n_out, scratch);
            if (s == TSQ_OK && n_out > 0) {
                e = hipMemcpyAsync(ctx->pinned + 17, out_cols[c].offsets + n_out, 8, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
                const int64_t nbytes = (int64_t)ctx->pinned[17];
                if (s == TSQ_OK && nbytes > 0) {
                    if (nbytes / n_out > 32)
                        hipLaunchKernelGGL(k_compact_varlen<true>, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, (const uint8_t*)cols[c].data, cols[c].offsets,
                                           a.src_row, out_cols[c].offsets, (uint8_t*)out_cols[c].data, n_out);
                    else
                        hipLaunchKernelGGL(k_compact_varlen<false>, dim3(tsq_grid_for(ctx, n_out, 256)), dim3(256), 0, ctx->stream, (const uint8_t*)cols[c].data,
                                           cols[c].offsets, a.src_row, out_cols[c].offsets, (uint8_t*)out_cols[c].data, n_out);
                    e = hipGetLastError();
                    if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, hipGetErrorString(e));
                }
            }
        }
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
