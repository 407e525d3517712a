// tsq_stage.h — chunk hand-off between the Go-side volcano interface and HBM.
//
// The reference moves ≤1024-row chunks between operators (util/chunk/chunk.go:31-46,
// tidb_max_chunk_size).  A PCIe DMA per 8 KiB column chunk would be latency bound, so host
// chunks are accumulated in PINNED staging memory (cgo forbids keeping Go pointers, so a copy
// is mandatory anyway) and flushed to HBM in batches of up to millions of rows; device-resident
// chunks (TSQ_COL_DEVICE) skip staging.  ColStore is the device analogue of chunk.List
// (util/chunk/list.go:22-38): one growing contiguous array per column.
#ifndef TSQ_STAGE_H
#define TSQ_STAGE_H

#include "tsq_internal.h"

struct ColStore {  // one device-resident column that grows by appends
    DevBuf data, nulls;
    DevBuf offs;         // TSQ_BYTES: offsets[rows + 1] into data (util/chunk/column.go:28-34)
    int64_t nbytes = 0;  // TSQ_BYTES: data bytes stored
    bool has_nulls = false;
    int64_t rows = 0;
    int32_t type = TSQ_I64;
    int elem() const { return tsq_elem_size(type); }
    void clear() {  // keep the buffers, forget the rows (per-batch probe staging)
        rows = 0;
        nbytes = 0;
        has_nulls = false;
    }
    void release() {
        data.release();
        nulls.release();
        offs.release();
        clear();
    }
};

// defined in tsq_ctx.hip
tsq_status tsq_launch_append_bits(tsq_ctx* ctx, tsq_handle_hdr* h, uint8_t* dst, int64_t dst_off, const uint8_t* src_dev, int64_t n);
tsq_status tsq_launch_offsets_rebase(tsq_ctx* ctx, tsq_handle_hdr* h, int64_t* dst, const int64_t* src_dev, int64_t n, int64_t delta);
tsq_status tsq_launch_scan64(tsq_ctx* ctx, tsq_handle_hdr* h, int64_t* v, int64_t n, DevBuf& scratch);
// cell r = bytes [pos[r], pos[r] + offs[r + 1] - offs[r]) of `data` -> out + offs[r] (tsq_decodec.hip; one cell per lane, or per wave when cells are long)
tsq_status tsq_launch_var_copy(tsq_ctx* ctx, tsq_handle_hdr* h, const uint8_t* data, const int64_t* pos, const int64_t* offs, int64_t rows, int64_t total_bytes,
                               uint8_t* out);

inline tsq_status tsq_col_append_bitmap(tsq_ctx* ctx, tsq_handle_hdr* h, ColStore& cs, const uint8_t* bitmap, int64_t n, bool src_dev, DevBuf& tmp_bits) {
    if (bitmap && !cs.has_nulls) {
        TSQ_TRY(cs.nulls.reserve(ctx, h, tsq_bitmap_bytes(cs.rows + n) + 64));
        TSQ_HIP(h, hipMemsetAsync(cs.nulls.p, 0xff, tsq_bitmap_bytes(cs.rows) + 8, ctx->stream));
        cs.has_nulls = true;
    }
    if (cs.has_nulls) {
        TSQ_TRY(cs.nulls.reserve(ctx, h, tsq_bitmap_bytes(cs.rows + n) + 64, true, tsq_bitmap_bytes(cs.rows) + 8));
        const uint8_t* src = bitmap;
        if (bitmap && !src_dev) {
            TSQ_TRY(tmp_bits.reserve(ctx, h, tsq_bitmap_bytes(n) + 8));
            TSQ_HIP(h, hipMemcpyAsync(tmp_bits.p, bitmap, tsq_bitmap_bytes(n), hipMemcpyHostToDevice, ctx->stream));
            src = tmp_bits.as<uint8_t>();
        }
        TSQ_TRY(tsq_launch_append_bits(ctx, h, cs.nulls.as<uint8_t>(), cs.rows, src, n));
    }
    return TSQ_OK;
}

// append n rows of a VAR-LEN column: `offsets` are the source's (n + 1 entries, any base), `data` its data array
inline tsq_status tsq_col_append_varlen(tsq_ctx* ctx, tsq_handle_hdr* h, ColStore& cs, const void* data, const int64_t* offsets, const uint8_t* bitmap,
                                        int64_t n, bool src_dev, DevBuf& tmp_bits, DevBuf& tmp_offs) {
    if (n <= 0) return TSQ_OK;
    int64_t o0 = 0, on = 0;
    if (src_dev) {
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 40, offsets, 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 41, offsets + n, 8, hipMemcpyDeviceToHost, ctx->stream));
        TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
        o0 = (int64_t)ctx->pinned[40];
        on = (int64_t)ctx->pinned[41];
    } else {
        o0 = offsets[0];
        on = offsets[n];
    }
    const int64_t bytes = on - o0;
    if (bytes < 0) return tsq_fail(h, TSQ_ERR_INVALID, "var-len column: offsets must not decrease");
    TSQ_TRY(cs.data.reserve(ctx, h, (size_t)(cs.nbytes + bytes) + 64, true, (size_t)cs.nbytes));
    if (bytes) TSQ_HIP(h, hipMemcpyAsync((char*)cs.data.p + cs.nbytes, (const char*)data + o0, (size_t)bytes, src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    TSQ_TRY(cs.offs.reserve(ctx, h, (size_t)(cs.rows + n + 1) * 8 + 64, true, (size_t)(cs.rows + 1) * 8));
    if (cs.rows == 0) {  // offsets[0] = the bytes already there (0 for a fresh column; the live strings of a compacted aggregate heap)
        if (cs.nbytes == 0) TSQ_HIP(h, hipMemsetAsync(cs.offs.p, 0, 8, ctx->stream));
        else {
            ctx->pinned[42] = (uint64_t)cs.nbytes;
            TSQ_HIP(h, hipMemcpyAsync(cs.offs.p, ctx->pinned + 42, 8, hipMemcpyHostToDevice, ctx->stream));
            TSQ_HIP(h, hipStreamSynchronize(ctx->stream));  // (the pinned word is reused)
        }
    }
    const int64_t* src = offsets;
    if (!src_dev) {
        TSQ_TRY(tmp_offs.reserve(ctx, h, (size_t)(n + 1) * 8 + 64));
        TSQ_HIP(h, hipMemcpyAsync(tmp_offs.p, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        src = tmp_offs.as<int64_t>();
    }
    TSQ_TRY(tsq_launch_offsets_rebase(ctx, h, cs.offs.as<int64_t>() + cs.rows + 1, src + 1, n, cs.nbytes - o0));
    TSQ_TRY(tsq_col_append_bitmap(ctx, h, cs, bitmap, n, src_dev, tmp_bits));
    cs.nbytes += bytes;
    cs.rows += n;
    return TSQ_OK;
}

// append n rows of one column (host or device source) to a device column store
inline tsq_status tsq_col_append(tsq_ctx* ctx, tsq_handle_hdr* h, ColStore& cs, const void* data, const uint8_t* bitmap,
                                 int64_t n, bool src_dev, DevBuf& tmp_bits) {
    const int es = cs.elem();
    TSQ_TRY(cs.data.reserve(ctx, h, (size_t)(cs.rows + n) * es + 64, true, (size_t)cs.rows * es));
    TSQ_HIP(h, hipMemcpyAsync((char*)cs.data.p + (size_t)cs.rows * es, data, (size_t)n * es,
                              src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    TSQ_TRY(tsq_col_append_bitmap(ctx, h, cs, bitmap, n, src_dev, tmp_bits));
    cs.rows += n;
    return TSQ_OK;
}

inline void tsq_fill_colset(tsq_colset& cs, const std::vector<ColStore>& cols) {
    memset(&cs, 0, sizeof(cs));
    cs.n = (int32_t)cols.size();
    for (int c = 0; c < cs.n; c++) {
        cs.data[c] = cols[c].data.p;
        cs.nulls[c] = cols[c].has_nulls ? cols[c].nulls.as<uint8_t>() : nullptr;
        cs.offs[c] = cols[c].type == TSQ_BYTES ? cols[c].offs.as<int64_t>() : nullptr;
        cs.type[c] = cols[c].type;
    }
}
inline void tsq_colset_from_cols(tsq_colset& cs, const tsq_col* cols, int32_t n_cols) {
    memset(&cs, 0, sizeof(cs));
    cs.n = n_cols;
    for (int c = 0; c < n_cols; c++) {
        cs.data[c] = cols[c].data;
        cs.nulls[c] = cols[c].null_bitmap;
        cs.offs[c] = cols[c].type == TSQ_BYTES ? cols[c].offsets : nullptr;
        cs.type[c] = cols[c].type;
    }
}
// slice [off, off+n) of a colset; off must be a multiple of 8 when any column has a bitmap
inline void tsq_colset_slice(tsq_colset& out, const tsq_colset& in, int64_t off) {
    out = in;
    for (int c = 0; c < in.n; c++) {
        if (in.type[c] == TSQ_BYTES) out.offs[c] = in.offs[c] + off;  // the offsets index rows, the data array keeps its base
        else out.data[c] = (const char*)in.data[c] + (size_t)off * tsq_elem_size(in.type[c]);
        if (in.nulls[c]) out.nulls[c] = in.nulls[c] + (off >> 3);
    }
}

// validates a pushed chunk against a schema; *is_dev = columns are device resident
inline tsq_status tsq_validate_cols(tsq_handle_hdr* h, const tsq_col* cols, int32_t n_cols, int32_t expect, const int32_t* types,
                                    int64_t nrows, bool* is_dev) {
    if (n_cols != expect) return tsq_fail(h, TSQ_ERR_INVALID, "column count does not match the operator schema");
    bool dev = false, host = false;
    for (int c = 0; c < n_cols; c++) {
        if (cols[c].type != types[c]) return tsq_fail(h, TSQ_ERR_INVALID, "column type does not match the operator schema");
        if (cols[c].length < nrows) return tsq_fail(h, TSQ_ERR_INVALID, "column shorter than nrows");
        if (cols[c].type == TSQ_BYTES) {
            if (nrows > 0 && !cols[c].offsets) return tsq_fail(h, TSQ_ERR_INVALID, "var-len column without offsets");
        } else if (nrows > 0 && !cols[c].data) {
            return tsq_fail(h, TSQ_ERR_INVALID, "column data == NULL");
        }
        (cols[c].flags & TSQ_COL_DEVICE) ? dev = true : host = true;
    }
    if (dev && host) return tsq_fail(h, TSQ_ERR_INVALID, "mixing host and device columns in one push");
    *is_dev = dev;
    return TSQ_OK;
}

// pinned accumulation of small host chunks
struct HostStage {
    int64_t cap = 0, staged = 0;
    std::vector<int32_t> types;
    std::vector<PinnedBuf> data, nulls;
    std::vector<PinnedBuf> offs;       // TSQ_BYTES columns: staged offsets[staged + 1]
    std::vector<int64_t> nbytes;       // TSQ_BYTES columns: staged data bytes (data[c] grows by doubling)
    std::vector<bool> null_any;
    PinnedBuf sel;
    bool sel_any = false;
    bool failed = false;               // a var-len data buffer could not grow (reported by the next flush)

    tsq_status init(tsq_handle_hdr* h, int32_t ncols, const int32_t* t, int64_t capacity) {
        cap = capacity;
        staged = 0;
        types.assign(t, t + ncols);
        data.resize(ncols);
        nulls.resize(ncols);
        offs.resize(ncols);
        nbytes.assign(ncols, 0);
        null_any.assign(ncols, false);
        failed = false;
        for (int c = 0; c < ncols; c++) {
            if (types[c] == TSQ_BYTES) {
                TSQ_TRY(offs[c].reserve(h, (size_t)(cap + 1) * 8));
                ((int64_t*)offs[c].p)[0] = 0;
                TSQ_TRY(data[c].reserve(h, (size_t)1 << 20));
            } else {
                TSQ_TRY(data[c].reserve(h, (size_t)cap * tsq_elem_size(types[c])));
            }
            TSQ_TRY(nulls[c].reserve(h, tsq_bitmap_bytes(cap) + 8));
        }
        TSQ_TRY(sel.reserve(h, (size_t)cap));
        sel_any = false;
        return TSQ_OK;
    }
    int64_t room() const { return cap - staged; }
    const uint8_t* bitmap(int c) const { return null_any[c] ? (const uint8_t*)nulls[c].p : nullptr; }
    // copies rows [src_off, src_off+n) of the caller's chunk; nothing of the caller is retained
    void add(const tsq_col* cols, int64_t src_off, int64_t n, const uint8_t* selected) {
        const int ncols = (int)types.size();
        for (int c = 0; c < ncols; c++) {
            if (types[c] == TSQ_BYTES) {  // the bytes of the n cells + their offsets, rebased onto what is already staged
                const int64_t* so = cols[c].offsets + src_off;
                const int64_t bytes = so[n] - so[0];
                if ((size_t)(nbytes[c] + bytes) > data[c].cap) {
                    PinnedBuf bigger;
                    size_t want = std::max<size_t>((size_t)(nbytes[c] + bytes), data[c].cap * 2);
                    if (bigger.reserve(nullptr, want) != TSQ_OK) {
                        failed = true;
                        continue;
                    }
                    memcpy(bigger.p, data[c].p, (size_t)nbytes[c]);
                    data[c].release();
                    data[c] = bigger;
                }
                if (bytes > 0) memcpy((char*)data[c].p + nbytes[c], (const char*)cols[c].data + so[0], (size_t)bytes);
                int64_t* d = (int64_t*)offs[c].p + staged;
                for (int64_t i = 1; i <= n; i++) d[i] = nbytes[c] + (so[i] - so[0]);
                nbytes[c] += bytes;
            } else {
                const int es = tsq_elem_size(types[c]);
                tsq_stage_copy((char*)data[c].p + (size_t)staged * es, (const char*)cols[c].data + (size_t)src_off * es, (size_t)n * es);
            }
            uint8_t* bm = (uint8_t*)nulls[c].p;
            if (cols[c].null_bitmap && !null_any[c]) {  // first bitmap seen: earlier staged rows are NOT NULL
                memset(bm, 0xff, tsq_bitmap_bytes(staged) + 1);
                null_any[c] = true;
            }
            if (null_any[c]) {
                const uint8_t* sb = cols[c].null_bitmap;
                for (int64_t i = 0; i < n; i++) {
                    const int64_t s = src_off + i, d = staged + i;
                    const bool one = sb ? ((sb[s >> 3] >> (s & 7)) & 1) : true;
                    if (one) bm[d >> 3] |= (uint8_t)(1u << (d & 7));
                    else bm[d >> 3] &= (uint8_t)~(1u << (d & 7));
                }
            }
        }
        if (selected && !sel_any) {
            memset(sel.p, 1, (size_t)staged);
            sel_any = true;
        }
        if (sel_any) {
            if (selected) memcpy((uint8_t*)sel.p + staged, selected + src_off, (size_t)n);
            else memset((uint8_t*)sel.p + staged, 1, (size_t)n);
        }
        staged += n;
    }
    void reset() {
        staged = 0;
        sel_any = false;
        for (size_t c = 0; c < null_any.size(); c++) null_any[c] = false;
        for (size_t c = 0; c < nbytes.size(); c++) nbytes[c] = 0;
    }
    // flush helper: append the staged rows of column c to a device column store
    tsq_status append_to(tsq_ctx* ctx, tsq_handle_hdr* h, int c, ColStore& dst, DevBuf& tmp_bits, DevBuf& tmp_offs) {
        if (failed) return tsq_fail(h, TSQ_ERR_OOM_DEVICE, "pinned staging for a var-len column could not grow");
        if (types[c] == TSQ_BYTES) return tsq_col_append_varlen(ctx, h, dst, data[c].p, (const int64_t*)offs[c].p, bitmap(c), staged, false, tmp_bits, tmp_offs);
        return tsq_col_append(ctx, h, dst, data[c].p, bitmap(c), staged, false, tmp_bits);
    }
    void release() {
        for (auto& b : data) b.release();
        for (auto& b : nulls) b.release();
        for (auto& b : offs) b.release();
        sel.release();
        cap = 0;
        staged = 0;
    }
};

#endif
