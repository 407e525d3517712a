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
#include "tsq_radix.h"

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
    // var-len (payload) columns: the cell's length goes to out_len[c][pos] (its scan = the output offsets), its source position to
    // out_pos[c][pos]; the bytes follow in a second kernel (tsq_launch_var_copy)
    int64_t* out_len[TSQ_MAX_COLS];
    int64_t* out_pos[TSQ_MAX_COLS];
};

__device__ __forceinline__ uint32_t split_rank(const SplitArgs& a, int64_t row) {
    const int c = a.key_col;
    if (tsq_is_null(a.in.nulls[c], row)) return 0;
    uint64_t w;
    if (a.in.type[c] == TSQ_BYTES) {  // a string key: equal bytes rank alike (string join keys: codec.go:233-235; group keys: :739-742)
        const int64_t lo = a.in.offs[c][row];
        return tsq_key_rank(tsq_hash_bytes((const uint8_t*)a.in.data[c] + lo, a.in.offs[c][row + 1] - lo), (uint32_t)a.n_parts);
    }
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
            if (a.in.type[c] == TSQ_BYTES) {
                const int64_t lo = a.in.offs[c][r];
                a.out_pos[c][pos] = lo;
                a.out_len[c][pos] = nn ? a.in.offs[c][r + 1] - lo : 0;  // a NULL cell has no bytes
            } else if (a.in.type[c] == TSQ_F32) ((uint32_t*)a.out_data[c])[pos] = ((const uint32_t*)a.in.data[c])[r];
            else ((uint64_t*)a.out_data[c])[pos] = ((const uint64_t*)a.in.data[c])[r];
            if (a.out_notnull[c]) a.out_notnull[c][pos] = nn ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------- fast path: 8-byte columns without NULL bitmaps
// K5h: per-part row counts.  Every lane ranks its keys; for each part the wave adds popcount(ballot) on the scalar
// unit; one LDS add per (wave, part) and one device atomic per (workgroup, part) at the very end.
__global__ void __launch_bounds__(256) k_rank_hist(const uint64_t* keys, int64_t nrows, int32_t key_mode_f64, uint32_t n_parts, uint32_t* counts) {
    __shared__ uint32_t lc[TSQ_SPLIT_MAX_PARTS];
    if (threadIdx.x < TSQ_SPLIT_MAX_PARTS) lc[threadIdx.x] = 0;
    __syncthreads();
    uint32_t wc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // n_parts <= 8 on this path (one node = 8 GPUs)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nround = (nrows + 63) & ~(int64_t)63;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nround; r += stride) {
        uint32_t rank = 0xffffffffu;
        if (r < nrows) {
            uint64_t w = keys[r];
            if (key_mode_f64) {
                const double f = tsq_bits_f64(w);
                w = f >= 0 ? (w | 0x8000000000000000ULL) : ~w;
            }
            rank = tsq_key_rank(w, n_parts);
        }
#pragma unroll
        for (uint32_t p = 0; p < 8; p++)
            if (p < n_parts) wc[p] += (uint32_t)__popcll(__ballot(rank == p));
    }
    if ((threadIdx.x & 63) == 0)
        for (uint32_t p = 0; p < n_parts; p++)
            if (wc[p]) atomicAdd(&lc[p], wc[p]);
    __syncthreads();
    if (threadIdx.x < n_parts && lc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], lc[threadIdx.x]);
}
// exclusive scan of <= 64 counts -> region bases; also widens the counts for the host
__global__ void k_rank_scan(const uint32_t* counts, uint32_t n_parts, uint32_t* base, unsigned long long* counts64) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t p = 0; p < n_parts; p++) {
            base[p] = acc;
            acc += counts[p];
            counts64[p] = counts[p];
        }
    }
}

// returns TSQ_OK with *done = true when the fast path ran
static tsq_status split_fast(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode, int64_t nrows, int32_t n_parts,
                             tsq_col* out_cols, int64_t* counts_out, bool* done) {
    *done = false;
    tsq_handle_hdr* h = &ctx->hdr;
    if (n_parts > 8 || n_cols > 1 + TSQ_RADIX_MAXV || nrows >= 0xffffffffLL || nrows < (1 << 16)) return TSQ_OK;
    for (int c = 0; c < n_cols; c++)
        if (cols[c].null_bitmap || out_cols[c].null_bitmap || cols[c].type == TSQ_F32 || cols[c].type == TSQ_BYTES) return TSQ_OK;
    if (key_mode == 0 && cols[key_col].type == TSQ_F64) {}  // join-key word of a double is its bits: fine
    const bool f64_image = key_mode == 1 && cols[key_col].type == TSQ_F64;
    if (f64_image) return TSQ_OK;  // the partition kernel's wide path reads raw key words; group-key float images take the general path
    DevBuf ctl;
    TSQ_TRY(ctl.reserve(ctx, h, 4096));
    // layout: counts[64] u32 | base[64] u32 | cursor[64] u32 | valid_end[64] u32 | ovf_count u32 | counts64[64] u64
    uint32_t* counts = ctl.as<uint32_t>();
    uint32_t* base = counts + 64;
    uint32_t* cursor = counts + 128;
    uint32_t* vend = counts + 192;
    uint32_t* ovfc = counts + 256;
    unsigned long long* counts64 = (unsigned long long*)(counts + 320);
    hipError_t e = hipMemsetAsync(ctl.p, 0, 4096, ctx->stream);
    const int grid = tsq_grid_for(ctx, nrows, 256);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rank_hist, dim3(grid), dim3(256), 0, ctx->stream, (const uint64_t*)cols[key_col].data, nrows, 0, (uint32_t)n_parts, counts);
        hipLaunchKernelGGL(k_rank_scan, dim3(1), dim3(64), 0, ctx->stream, counts, (uint32_t)n_parts, base, counts64);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { ctl.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_radix_split: ") + hipGetErrorString(e)); }
    RadixSrc src;
    memset(&src, 0, sizeof src);
    src.data = cols[key_col].data;
    src.type = cols[key_col].type;
    src.nrows = nrows;
    src.key_kind = 0;
    RadixStore st;
    memset(&st, 0, sizeof st);
    st.keys = (uint64_t*)out_cols[key_col].data;
    int V = 0;
    for (int c = 0; c < n_cols; c++) {
        if (c == key_col) continue;
        src.vdata[V] = cols[c].data;
        src.vtype[V] = cols[c].type;
        st.pay[V] = (uint64_t*)out_cols[c].data;
        V++;
    }
    st.cursor = cursor;
    st.valid_end = vend;
    st.ovf_count = ovfc;
    st.region_base = base;
    st.rank_parts = (uint32_t)n_parts;
    st.R = 1;
    st.cap = 0xffffffffu;
    const int K = V == 0 ? 16 : (V == 1 ? 8 : 4), T = 1024 * K;
    const int pgrid = (int)std::min<int64_t>((nrows + T - 1) / T, ctx->num_cus);
    if (V == 0) hipLaunchKernelGGL((k_radix_partition<1024, 16, 4, 0, false>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
    else if (V == 1) hipLaunchKernelGGL((k_radix_partition<1024, 8, 4, 1, false>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
    else hipLaunchKernelGGL((k_radix_partition<1024, 4, 4, 2, false>), dim3(pgrid), dim3(1024), 0, ctx->stream, src, st);
    e = hipGetLastError();
    unsigned long long hc[TSQ_SPLIT_MAX_PARTS];
    if (e == hipSuccess) e = hipMemcpyAsync(hc, counts64, n_parts * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    ctl.release();
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_radix_split: ") + hipGetErrorString(e));
    for (int p = 0; p < n_parts; p++) counts_out[p] = (int64_t)hc[p];
    for (int c = 0; c < n_cols; c++) {
        out_cols[c].length = nrows;
        out_cols[c].type = cols[c].type;
        out_cols[c].elem_size = 8;
    }
    *done = true;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_radix_split(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, int32_t key_col, int32_t key_mode, int64_t nrows,
                                   int32_t n_parts, tsq_col* out_cols, int64_t* counts_out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (!cols || !out_cols || !counts_out || n_cols < 1 || n_cols > TSQ_MAX_COLS || key_col < 0 || key_col >= n_cols || nrows < 0)
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: bad arguments");
    if (n_parts < 1 || n_parts > TSQ_SPLIT_MAX_PARTS) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: 1..64 parts");
    if (key_mode != 0 && key_mode != 1) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: key_mode must be 0 (join) or 1 (group)");
    for (int c = 0; c < n_cols; c++) {
        if (!(cols[c].flags & TSQ_COL_DEVICE) || !(out_cols[c].flags & TSQ_COL_DEVICE))
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: columns must be device resident");
        if (cols[c].type < TSQ_I64 || cols[c].type > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: unknown column type");
        if (cols[c].type == TSQ_BYTES && (!cols[c].offsets || !out_cols[c].offsets)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: a var-len column needs offsets");
        if (cols[c].null_bitmap && !out_cols[c].null_bitmap) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: nullable column needs an output bitmap");
        if (nrows > 0 && cols[c].type != TSQ_BYTES && (!cols[c].data || !out_cols[c].data)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: NULL data pointer");
    }
    for (int p = 0; p < n_parts; p++) counts_out[p] = 0;
    if (nrows == 0) return TSQ_OK;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    {   // LDS-staged tile sort with exact region bases (tsq_radix.h) when the columns allow it
        bool done = false;
        TSQ_TRY(split_fast(ctx, cols, n_cols, key_col, key_mode, nrows, n_parts, out_cols, counts_out, &done));
        if (done) return TSQ_OK;
    }
    SplitArgs a;
    memset(&a, 0, sizeof a);
    tsq_colset_from_cols(a.in, cols, n_cols);
    a.key_col = key_col;
    a.key_mode = key_mode;
    a.n_parts = n_parts;
    a.nrows = nrows;
    DevBuf cur, scan_scratch;
    std::vector<DevBuf> nn(n_cols), vpos(n_cols);
    auto cleanup = [&]() {
        cur.release();
        scan_scratch.release();
        for (auto& b : nn) b.release();
        for (auto& b : vpos) b.release();
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
        if (cols[c].type == TSQ_BYTES) {
            s = vpos[c].reserve(ctx, h, (size_t)nrows * 8 + 64);
            if (s != TSQ_OK) { cleanup(); return s; }
            a.out_pos[c] = vpos[c].as<int64_t>();
            a.out_len[c] = out_cols[c].offsets;  // lengths first, their scan in place
        }
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
        out_cols[c].elem_size = cols[c].type == TSQ_BYTES ? -1 : tsq_elem_size(cols[c].type);
        if (cols[c].type == TSQ_BYTES) {  // lengths -> offsets[nrows + 1], then the bytes of every cell to its new place
            s = tsq_launch_scan64(ctx, h, out_cols[c].offsets, nrows, scan_scratch);
            if (s == TSQ_OK) {
                e = hipMemcpyAsync(ctx->pinned, out_cols[c].offsets + nrows, 8, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_radix_split: ") + hipGetErrorString(e));
            }
            const int64_t total = (int64_t)ctx->pinned[0];
            if (s == TSQ_OK && total > 0 && !out_cols[c].data) s = tsq_fail(h, TSQ_ERR_INVALID, "tsq_radix_split: NULL data pointer");
            if (s == TSQ_OK) s = tsq_launch_var_copy(ctx, h, (const uint8_t*)cols[c].data, a.out_pos[c], out_cols[c].offsets, nrows, total, (uint8_t*)out_cols[c].data);
            if (s != TSQ_OK) { cleanup(); return s; }
        }
    }
    e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_radix_split: ") + hipGetErrorString(e));
    return TSQ_OK;
}
