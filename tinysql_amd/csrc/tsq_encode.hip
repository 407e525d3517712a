// tsq_encode.hip — chunk rows -> the RowsData byte string of a coprocessor response, on the GPU (SURVEY.md §8 f, rank 4).
//
// Replaces the per-row, per-datum encode loop of the storage side: codec.EncodeValue of every output value
// (store/mockstore/mocktikv/aggregate.go:96-113 for partial aggregates, util/rowcodec/decoder.go:252-322 for scanned rows) and the
// concatenation of the requested values of a row (cop_handler_dag.go:414-425, appendRow :512-519) — the inverse of
// tsq_rows_decode.  Rows are independent but their bytes are packed back to back, so the byte position of a row is a prefix
// sum over the value lengths:
//   K16a k_enc_size : every workgroup owns a contiguous range of 256-row tiles and adds up the encoded bytes of its rows
//   K16b k_enc_scan : <= 1024 workgroup totals -> the byte position where each workgroup starts
//   K16c k_enc_emit : per tile: row lengths again, block-wide exclusive scan -> row positions (also handed to the caller: the
//                     response is cut into chunks of 64 rows, cop_handler_dag.go:510-519), every lane writes its row's datums
//                     into an LDS image of the tile, and the workgroup copies the image to its place in the output with aligned
//                     16-byte stores (tsq_enc_copy_plan: the few bytes shared with a neighbouring tile's vector go one by one).
// HBM-bound byte work, no MFMA.  Algorithmic bytes per value: 8 B read + its datum bytes written (the columns are read twice).
#include "tsq_internal.h"
#include "tsq_radix.h"
#include "tsq_encode_dp.h"

#define ENC_NT 256
#define ENC_MAXWG 1024

struct EncArgs {
    const void* data[TSQ_MAX_COLS];
    const uint8_t* nulls[TSQ_MAX_COLS];  // nullptr: no NULLs
    const int64_t* offs[TSQ_MAX_COLS];   // var-len columns: offsets[nrows + 1] into data
    uint32_t lds_bytes;                  // LDS image of the launch: a tile that does not fit is written to the output directly
    int32_t type[TSQ_MAX_COLS];
    uint32_t comparable;                 // bit c: column c in the EncodeKey form
    int32_t n_cols;
    int64_t nrows, n_tiles, tiles_per_wg;
    int32_t n_wg;
    unsigned long long* wg_bytes;        // [n_wg + 1]: totals (K16a), then exclusive starts + grand total (K16b)
    uint8_t* out;
    int64_t* row_offsets;                // nullptr or [nrows + 1]
};

namespace {

// the 8 bytes column c stores for row r as the encoder wants them (a float32 widened to its double image) + its NOT NULL bit
__device__ __forceinline__ uint64_t enc_load(const EncArgs& a, int c, int64_t r, bool* notnull) {
    const uint8_t* bm = a.nulls[c];
    *notnull = bm ? ((bm[r >> 3] >> (r & 7)) & 1) != 0 : true;
    if (a.type[c] == TSQ_F32) {
        const double d = (double)((const float*)a.data[c])[r];
        uint64_t b;
        memcpy(&b, &d, 8);
        return b;
    }
    return ((const uint64_t*)a.data[c])[r];
}

__device__ __forceinline__ uint32_t enc_row_len(const EncArgs& a, int64_t r) {
    uint32_t len = 0;
    for (int c = 0; c < a.n_cols; c++) {
        if (a.type[c] == TSQ_BYTES) {  // NilFlag, or compactBytesFlag + varint(n) + the n bytes
            const uint8_t* bm = a.nulls[c];
            const bool nn = bm ? ((bm[r >> 3] >> (r & 7)) & 1) != 0 : true;
            const uint64_t n = (uint64_t)(a.offs[c][r + 1] - a.offs[c][r]);
            if ((a.comparable >> c) & 1u) len += nn ? (uint32_t)tsq_enc_membytes_len(n) : 1u;
            else len += nn ? tsq_enc_str_hdr_len(n) + (uint32_t)n : 1u;
            continue;
        }
        bool nn;
        const uint64_t bits = enc_load(a, c, r, &nn);
        len += tsq_enc_len(a.type[c], (a.comparable >> c) & 1u, bits, nn);
    }
    return len;
}

__global__ void __launch_bounds__(ENC_NT) k_enc_size(EncArgs a) {
    __shared__ unsigned long long s_sum[ENC_NT / 64];
    const uint32_t tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * a.tiles_per_wg, t1 = t0 + a.tiles_per_wg < a.n_tiles ? t0 + a.tiles_per_wg : a.n_tiles;
    unsigned long long sum = 0;
    for (int64_t t = t0; t < t1; t++) {
        const int64_t r = t * ENC_NT + tid;
        if (r < a.nrows) sum += enc_row_len(a, r);
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((tid & 63u) == 0) s_sum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
        for (int w = 0; w < ENC_NT / 64; w++) tot += s_sum[w];
        a.wg_bytes[blockIdx.x] = tot;
    }
}

// <= 1024 workgroup totals: one thread turns them into exclusive starts; [n_wg] = the size of the whole byte string
__global__ void k_enc_scan(EncArgs a) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long run = 0;
        for (int b = 0; b < a.n_wg; b++) {
            const unsigned long long v = a.wg_bytes[b];
            a.wg_bytes[b] = run;
            run += v;
        }
        a.wg_bytes[a.n_wg] = run;
    }
}

__global__ void __launch_bounds__(ENC_NT) k_enc_emit(EncArgs a) {
    extern __shared__ uint4 s_img[];  // the tile's bytes at [skew, skew + T)
    __shared__ uint32_t s_wsum[ENC_NT / 64];
    uint8_t* img = (uint8_t*)s_img;
    const uint32_t tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * a.tiles_per_wg, t1 = t0 + a.tiles_per_wg < a.n_tiles ? t0 + a.tiles_per_wg : a.n_tiles;
    int64_t base = (int64_t)a.wg_bytes[blockIdx.x];
    for (int64_t t = t0; t < t1; t++) {
        const int64_t r = t * ENC_NT + tid;
        const bool live = r < a.nrows;
        const uint32_t len = live ? enc_row_len(a, r) : 0u;
        uint32_t T;
        const uint32_t ex = block_excl_scan<ENC_NT>(len, s_wsum, &T);
        const tsq_enc_copy plan = tsq_enc_copy_plan((uint64_t)(uintptr_t)a.out, base, T);
        // a tile whose bytes do not fit the LDS image (rows with long strings) is written to its place directly, row by row
        const bool staged = (size_t)plan.skew + T + 16 <= a.lds_bytes;  // workgroup uniform
        uint8_t* dst = staged ? img + plan.skew + ex : a.out + base + ex;
        if (live) {
            if (a.row_offsets) a.row_offsets[r] = base + (int64_t)ex;
            uint32_t pos = 0;
            for (int c = 0; c < a.n_cols; c++) {
                uint64_t lo;
                uint32_t hi;
                if (a.type[c] == TSQ_BYTES) {
                    const uint8_t* bm = a.nulls[c];
                    const bool nn = bm ? ((bm[r >> 3] >> (r & 7)) & 1) != 0 : true;
                    if (!nn) { dst[pos++] = 0; continue; }  // NilFlag
                    const int64_t s0 = a.offs[c][r];
                    const uint64_t n = (uint64_t)(a.offs[c][r + 1] - s0);
                    if ((a.comparable >> c) & 1u) {  // bytesFlag + the groups of 8 bytes with their markers
                        const uint8_t* src = (const uint8_t*)a.data[c] + s0;
                        const uint32_t m = (uint32_t)tsq_enc_membytes_len(n) - 1u;
                        dst[pos++] = 1;
                        for (uint32_t i = 0; i < m; i++) dst[pos + i] = tsq_enc_membytes_at(src, n, i);
                        pos += m;
                        continue;
                    }
                    const uint32_t hn = tsq_enc_str_hdr(n, &lo, &hi);
                    for (uint32_t i = 0; i < hn; i++) dst[pos + i] = (uint8_t)(i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8)));
                    pos += hn;
                    const uint8_t* src = (const uint8_t*)a.data[c] + s0;
                    for (uint64_t i = 0; i < n; i++) dst[pos + i] = src[i];
                    pos += (uint32_t)n;
                    continue;
                }
                bool nn;
                const uint64_t bits = enc_load(a, c, r, &nn);
                const uint32_t n = tsq_enc_bytes(a.type[c], (a.comparable >> c) & 1u, bits, nn, &lo, &hi);
                for (uint32_t i = 0; i < n; i++) dst[pos + i] = (uint8_t)(i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8)));
                pos += n;
            }
        }
        __syncthreads();
        if (staged) {
            uint8_t* g = a.out + base - plan.skew;  // 16-byte aligned by construction
            if (tid < 16 && plan.skew + tid < plan.head_end) g[plan.skew + tid] = img[plan.skew + tid];
            if (tid >= 16 && tid < 32 && plan.tail_lo + (tid - 16) < plan.tail_end) g[plan.tail_lo + (tid - 16)] = img[plan.tail_lo + (tid - 16)];
            for (uint32_t i = plan.body_lo + tid; i < plan.body_hi; i += ENC_NT) ((uint4*)g)[i] = s_img[i];
        }
        __syncthreads();  // the image (and s_wsum) are reused by the next tile
        base += T;
    }
    if (tid == 0 && a.row_offsets && t1 == a.n_tiles && t1 > t0) a.row_offsets[a.nrows] = base;  // the workgroup that owns the last tile
}

}  // namespace

// ====================================================================== host side
TSQ_API tsq_status tsq_rows_encode(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, const uint32_t* col_flags, int64_t nrows, uint8_t* out,
                                   int64_t cap_bytes, uint32_t out_flags, int64_t* row_offsets, int64_t* bytes_out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (bytes_out) *bytes_out = 0;
    if (!bytes_out || !cols || nrows < 0 || cap_bytes < 0 || (cap_bytes > 0 && !out)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_encode: bad arguments");
    if (n_cols < 1 || n_cols > TSQ_MAX_COLS) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "1..16 columns supported");
    bool in_dev = false, in_host = false, any_host_var = false;
    for (int c = 0; c < n_cols; c++) {
        if (cols[c].type < TSQ_I64 || cols[c].type > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_encode: unknown column type");
        if (cols[c].length < nrows) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_encode: column shorter than nrows");
        if (cols[c].type == TSQ_BYTES) {
            if (nrows > 0 && !cols[c].offsets) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_encode: var-len column without offsets");
            if (!(cols[c].flags & TSQ_COL_DEVICE)) any_host_var = true;
        } else if (nrows > 0 && !cols[c].data) {
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_encode: column data == NULL");
        }
        (cols[c].flags & TSQ_COL_DEVICE) ? in_dev = true : in_host = true;
    }
    if (in_dev && in_host) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_encode: mixed host/device columns");
    TSQ_HIP(h, hipSetDevice(ctx->device));
    if (nrows == 0) {  // an empty response still has its first row boundary
        if (row_offsets) {
            if (out_flags & TSQ_COL_DEVICE) TSQ_HIP(h, hipMemsetAsync(row_offsets, 0, sizeof(row_offsets[0]), ctx->stream));
            else row_offsets[0] = 0;
        }
        return TSQ_OK;
    }
    const bool out_dev = out_flags & TSQ_COL_DEVICE;
    EncArgs a;
    memset(&a, 0, sizeof a);
    a.n_cols = n_cols;
    a.nrows = nrows;
    a.n_tiles = (nrows + ENC_NT - 1) / ENC_NT;
    {   // workgroup b owns tiles [b * R, (b + 1) * R): <= 1024 workgroups, up to 8 per CU
        const int64_t want = std::min<int64_t>(std::min<int64_t>(a.n_tiles, (int64_t)ctx->num_cus * 8), ENC_MAXWG);
        a.tiles_per_wg = (a.n_tiles + want - 1) / want;
        a.n_wg = (int32_t)((a.n_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg);
    }
    DevBuf dwg, dout, doffs, ddata[TSQ_MAX_COLS], dbm[TSQ_MAX_COLS], dco[TSQ_MAX_COLS];
    auto release_all = [&]() {
        for (DevBuf* b : {&dwg, &dout, &doffs}) b->release();
        for (int c = 0; c < TSQ_MAX_COLS; c++) { ddata[c].release(); dbm[c].release(); dco[c].release(); }
    };
    (void)any_host_var;
    auto fail = [&](tsq_status st) { release_all(); return st; };
    tsq_status s = dwg.reserve(ctx, h, ((size_t)a.n_wg + 1) * 8 + 64);
    hipError_t e = hipSuccess;
    uint32_t row_max = 0;
    bool any_var = false;
    for (int c = 0; c < n_cols && s == TSQ_OK && e == hipSuccess; c++) {
        a.type[c] = cols[c].type;
        const bool cmp = col_flags && (col_flags[c] & TSQ_ENC_COMPARABLE);
        if (cmp) a.comparable |= 1u << c;
        const bool var = cols[c].type == TSQ_BYTES;
        any_var = any_var || var;
        row_max += var ? (cmp ? 40u : 32u) : ((cols[c].type == TSQ_F32 || cols[c].type == TSQ_F64 || cmp) ? 9u : TSQ_ENC_MAX_VALUE);  // (strings: a guess; tiles that do not fit go direct)
        if (in_dev) {
            a.data[c] = cols[c].data;
            a.nulls[c] = cols[c].null_bitmap;
            a.offs[c] = var ? cols[c].offsets : nullptr;
        } else if (var) {
            // a host var-len column: its offsets and the bytes they span (the offsets may start anywhere: the kernel indexes data with them)
            const int64_t b0 = cols[c].offsets[0], b1 = cols[c].offsets[nrows];
            if (b1 < b0) { s = tsq_fail(h, TSQ_ERR_INVALID, "var-len column: offsets must not decrease"); break; }
            s = dco[c].reserve(ctx, h, ((size_t)nrows + 1) * 8 + 64);
            if (s == TSQ_OK) s = ddata[c].reserve(ctx, h, (size_t)(b1 - b0) + 64);
            if (s == TSQ_OK) e = hipMemcpyAsync(dco[c].p, cols[c].offsets, ((size_t)nrows + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
            if (s == TSQ_OK && e == hipSuccess && b1 > b0) e = hipMemcpyAsync(ddata[c].p, (const uint8_t*)cols[c].data + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, ctx->stream);
            a.data[c] = (const uint8_t*)ddata[c].p - b0;  // data[offsets[r]] addresses the staged copy
            a.offs[c] = dco[c].as<int64_t>();
            if (s == TSQ_OK && e == hipSuccess && cols[c].null_bitmap) {
                s = dbm[c].reserve(ctx, h, tsq_bitmap_bytes(nrows) + 64);
                if (s == TSQ_OK) e = hipMemcpyAsync(dbm[c].p, cols[c].null_bitmap, tsq_bitmap_bytes(nrows), hipMemcpyHostToDevice, ctx->stream);
                a.nulls[c] = dbm[c].as<uint8_t>();
            }
        } else {
            const size_t es = (size_t)tsq_elem_size(cols[c].type);
            s = ddata[c].reserve(ctx, h, (size_t)nrows * es + 64);
            if (s == TSQ_OK) e = hipMemcpyAsync(ddata[c].p, cols[c].data, (size_t)nrows * es, hipMemcpyHostToDevice, ctx->stream);
            a.data[c] = ddata[c].p;
            if (s == TSQ_OK && e == hipSuccess && cols[c].null_bitmap) {
                s = dbm[c].reserve(ctx, h, tsq_bitmap_bytes(nrows) + 64);
                if (s == TSQ_OK) e = hipMemcpyAsync(dbm[c].p, cols[c].null_bitmap, tsq_bitmap_bytes(nrows), hipMemcpyHostToDevice, ctx->stream);
                a.nulls[c] = dbm[c].as<uint8_t>();
            }
        }
    }
    if (s != TSQ_OK) return fail(s);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rows_encode(H2D): ") + hipGetErrorString(e)));
    a.wg_bytes = dwg.as<unsigned long long>();
    // pass 1 + scan: the size of the byte string is known before a byte is written
    hipLaunchKernelGGL(k_enc_size, dim3(a.n_wg), dim3(ENC_NT), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_enc_scan, dim3(1), dim3(64), 0, ctx->stream, a);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned, a.wg_bytes + a.n_wg, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rows_encode: ") + hipGetErrorString(e)));
    const int64_t total = (int64_t)ctx->pinned[0];
    *bytes_out = total;
    if (total > cap_bytes) return fail(tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_encode: output buffer too small (*bytes_out = bytes needed)"));
    if (out_dev) {
        a.out = out;
        a.row_offsets = row_offsets;
    } else {
        s = dout.reserve(ctx, h, (size_t)total + 64);
        if (s == TSQ_OK && row_offsets) s = doffs.reserve(ctx, h, ((size_t)nrows + 1) * 8 + 64);
        if (s != TSQ_OK) return fail(s);
        a.out = dout.as<uint8_t>();
        a.row_offsets = row_offsets ? doffs.as<int64_t>() : nullptr;
    }
    size_t lds = (((size_t)ENC_NT * row_max + 15 + 16 + 15) / 16) * 16;  // the tile at any skew, whole vectors
    if (any_var) lds = std::min<size_t>(std::max<size_t>(lds, 32 * 1024), 64 * 1024);  // tiles with longer strings are written directly
    if (lds > 64 * 1024) return fail(tsq_fail(h, TSQ_ERR_UNSUPPORTED, "tsq_rows_encode: row too wide for the LDS tile"));
    a.lds_bytes = (uint32_t)lds;
    hipLaunchKernelGGL(k_enc_emit, dim3(a.n_wg), dim3(ENC_NT), lds, ctx->stream, a);
    e = hipGetLastError();
    if (e == hipSuccess && !out_dev) {
        if (total > 0) e = hipMemcpyAsync(out, a.out, (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && row_offsets) e = hipMemcpyAsync(row_offsets, a.row_offsets, ((size_t)nrows + 1) * 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release_all();
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rows_encode: ") + hipGetErrorString(e));
    return TSQ_OK;
}
