// tsq_wire.hip — the chunk wire format on device chunks (SURVEY.md §8 a/A "wire Codec", f rank 2).
//
// Replaces chunk.Codec.Encode / DecodeToChunk (util/chunk/codec.go:42-143) and the row-window append of chunk.Decoder
// (codec.go:246-353: Reset / Decode / ReuseIntermChk).  A wire chunk is its columns one after the other, each
//     u32 length | u32 nullCount | [bitmap, when nullCount > 0] | [(length + 1) offsets, var-len only] | data
// with no padding: every piece sits at an arbitrary byte position.  The work is a byte stream at HBM rate:
//   k_wire_notnull   population counts of the bitmaps (nullCount, which also decides whether a bitmap goes on the wire)
//   k_wire_walk      one lane walks the column headers of a DEVICE-resident buffer (a host buffer is walked by the caller's thread
//                    with the same function, tsq_wire_dp.h)
//   k_wire_move      ONE launch moves all pieces of all columns: 16-byte destination-aligned vectors with unaligned 16-byte loads,
//                    headers, bitmaps appended at a bit offset (Decoder.decodeColumn's shift-and-or, codec.go:325-343), offsets rebased
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>

#include "../../include/tsq.h"
#include "tsq_internal.h"
#include "tsq_wire_dp.h"

namespace {

constexpr int WIRE_MAX_SEGS = TSQ_MAX_COLS * 4;

struct WireSeg {
    const uint8_t* src;  // WM_BITS: null = every bit set (a column that travelled without its bitmap)
    uint8_t* dst;
    int64_t n;           // COPY: bytes; BITS: rows appended; OFFS: offsets written; HDR: 8
    int64_t imm;         // HDR: the eight bytes; OFFS: delta added to every offset; BITS: rows already in the destination
};

struct WireMoveArgs {
    WireSeg seg[WIRE_MAX_SEGS];
    int32_t mode[WIRE_MAX_SEGS];
    int32_t blk_first[WIRE_MAX_SEGS + 1];  // first workgroup of every piece
    int32_t n_segs;
};

__global__ void __launch_bounds__(256) k_wire_move(WireMoveArgs a) {
    int s = 0;
    while (s + 1 < a.n_segs && (int)blockIdx.x >= a.blk_first[s + 1]) s++;  // uniform: a handful of scalar compares
    const WireSeg g = a.seg[s];
    tsq_wire_move_lane(a.mode[s], g.src, g.dst, g.n, g.imm, (int64_t)blockIdx.x - a.blk_first[s], (int)threadIdx.x);
}

// NOT NULL rows of the first n rows of up to 16 bitmaps (blockIdx.y = column); counts[c] accumulates
struct WireCountArgs {
    const uint8_t* bm[TSQ_MAX_COLS];
    unsigned long long* counts;
    int64_t n;
};
__global__ void __launch_bounds__(256) k_wire_notnull(WireCountArgs a) {
    const uint8_t* bm = a.bm[blockIdx.y];
    if (!bm) return;
    const int64_t nbytes = (a.n + 7) >> 3;
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * 256) {
        uint32_t v = bm[i];
        if (i == nbytes - 1 && (a.n & 7)) v &= (1u << (a.n & 7)) - 1u;
        c += __popc(v);
    }
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(a.counts + blockIdx.y, c);
}

struct WireWalkArgs {
    const uint8_t* buf;
    int64_t n_bytes, first, max_rows;
    int32_t elem[TSQ_MAX_COLS];
    int32_t n_cols;
    uint64_t* out;  // [4 n_cols]
};
__global__ void k_wire_walk(WireWalkArgs a) {
    if (threadIdx.x == 0 && blockIdx.x == 0) tsq_wire_walk(a.buf, a.n_bytes, a.elem, a.n_cols, a.first, a.max_rows, a.out);
}

struct MovePlan {
    WireMoveArgs a;
    int grid = 0;
    void add(int mode, const void* src, void* dst, int64_t n, int64_t imm) {
        int64_t blocks = 1;
        if (mode == WM_COPY) blocks = std::max<int64_t>(1, (n + TSQ_WIRE_BLOCK_BYTES - 1) / TSQ_WIRE_BLOCK_BYTES);  // (head bytes shift the vectors by < 1 block)
        else if (mode == WM_BITS) blocks = std::max<int64_t>(1, (((imm + n + 7) >> 3) - (imm >> 3) + 4095) / 4096);
        else if (mode == WM_OFFS) blocks = std::max<int64_t>(1, (n + 2047) / 2048);
        if (mode != WM_HDR && n <= 0) return;
        const int i = a.n_segs++;
        a.seg[i].src = (const uint8_t*)src;
        a.seg[i].dst = (uint8_t*)dst;
        a.seg[i].n = n;
        a.seg[i].imm = imm;
        a.mode[i] = mode;
        a.blk_first[i] = grid;
        grid += (int)blocks;
        a.blk_first[i + 1] = grid;
    }
};

int wire_elem(int32_t type) { return type == TSQ_BYTES ? -1 : tsq_elem_size(type); }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- Codec.Encode
TSQ_API tsq_status tsq_chunk_encode(tsq_ctx* ctx, const tsq_col* cols, int32_t n_cols, int64_t nrows, uint8_t* out, int64_t cap_bytes,
                                    uint32_t out_flags, int64_t* bytes_out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (bytes_out) *bytes_out = 0;
    if (!cols || !bytes_out || nrows < 0 || cap_bytes < 0 || (cap_bytes > 0 && !out)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_encode: bad arguments");
    if (n_cols < 1 || n_cols > TSQ_MAX_COLS) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "1..16 columns supported");
    if (nrows >= (1LL << 32)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_encode: the wire format carries the row count in 32 bits");
    for (int c = 0; c < n_cols; c++) {
        if (cols[c].type < TSQ_I64 || cols[c].type > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_encode: unknown column type");
        if (((cols[c].flags ^ cols[0].flags) & TSQ_COL_DEVICE) != 0) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_encode: mixed host/device columns");
        if (cols[c].type == TSQ_BYTES ? !cols[c].offsets : (nrows > 0 && !cols[c].data)) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_encode: column without data / offsets");
    }
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const bool in_dev = cols[0].flags & TSQ_COL_DEVICE, out_dev = out_flags & TSQ_COL_DEVICE;
    DevBuf sdata[TSQ_MAX_COLS], sbm[TSQ_MAX_COLS], soffs[TSQ_MAX_COLS], srebase[TSQ_MAX_COLS], dout, dcnt;
    auto fail = [&](tsq_status st) {
        for (int c = 0; c < TSQ_MAX_COLS; c++) { sdata[c].release(); sbm[c].release(); soffs[c].release(); srebase[c].release(); }
        dout.release();
        dcnt.release();
        return st;
    };
    // the columns in HBM (host columns are staged), and the data bytes of the var-len ones
    const void* data[TSQ_MAX_COLS];
    const uint8_t* bm[TSQ_MAX_COLS];
    const int64_t* offs[TSQ_MAX_COLS];
    int64_t data_bytes[TSQ_MAX_COLS];
    int64_t off0[TSQ_MAX_COLS] = {0};  // a var-len VIEW (DeviceColumn.view, tsq_colset_slice) has offsets that start anywhere: the wire's start at 0
    tsq_status s = dcnt.reserve(ctx, h, TSQ_MAX_COLS * 8 + 64);
    if (s != TSQ_OK) return fail(s);
    hipError_t e = hipMemsetAsync(dcnt.p, 0, TSQ_MAX_COLS * 8, ctx->stream);
    const size_t nbm = tsq_bitmap_bytes(nrows);
    for (int c = 0; c < n_cols && s == TSQ_OK && e == hipSuccess; c++) {
        const bool var = cols[c].type == TSQ_BYTES;
        data_bytes[c] = var ? 0 : nrows * tsq_elem_size(cols[c].type);
        if (in_dev) {
            data[c] = cols[c].data;
            bm[c] = cols[c].null_bitmap;
            offs[c] = var ? cols[c].offsets : nullptr;
            if (var) e = hipMemcpyAsync(ctx->pinned + c, cols[c].offsets + nrows, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (var && e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 2 * TSQ_MAX_COLS + c, cols[c].offsets, 8, hipMemcpyDeviceToHost, ctx->stream);
            continue;
        }
        if (var) {
            off0[c] = cols[c].offsets[0];
            data_bytes[c] = cols[c].offsets[nrows] - off0[c];
            if (off0[c] < 0 || data_bytes[c] < 0 || (data_bytes[c] > 0 && !cols[c].data)) return fail(tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_encode: bad offsets"));
            s = soffs[c].reserve(ctx, h, ((size_t)nrows + 1) * 8 + 64);
            if (s == TSQ_OK) e = hipMemcpyAsync(soffs[c].p, cols[c].offsets, ((size_t)nrows + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
        }
        offs[c] = var ? soffs[c].as<int64_t>() : nullptr;
        if (s == TSQ_OK) s = sdata[c].reserve(ctx, h, (size_t)data_bytes[c] + 64);
        if (s == TSQ_OK && e == hipSuccess && data_bytes[c] > 0)
            e = hipMemcpyAsync(sdata[c].p, (const uint8_t*)cols[c].data + off0[c], (size_t)data_bytes[c], hipMemcpyHostToDevice, ctx->stream);
        data[c] = sdata[c].p;
        bm[c] = nullptr;
        if (cols[c].null_bitmap && nrows > 0) {
            if (s == TSQ_OK) s = sbm[c].reserve(ctx, h, nbm + 64);
            if (s == TSQ_OK && e == hipSuccess) e = hipMemcpyAsync(sbm[c].p, cols[c].null_bitmap, nbm, hipMemcpyHostToDevice, ctx->stream);
            bm[c] = sbm[c].as<uint8_t>();
        }
    }
    if (s != TSQ_OK) return fail(s);
    // nullCount of every column (Column.nullCount, column.go:117-128)
    WireCountArgs ca;
    memset(&ca, 0, sizeof ca);
    bool any_bm = false;
    for (int c = 0; c < n_cols; c++) { ca.bm[c] = nrows > 0 ? bm[c] : nullptr; any_bm = any_bm || ca.bm[c]; }
    ca.counts = dcnt.as<unsigned long long>();
    ca.n = nrows;
    if (e == hipSuccess && any_bm) {
        const int gx = (int)std::min<int64_t>(ctx->num_cus * 4, std::max<int64_t>(1, ((int64_t)nbm + 255) / 256));
        hipLaunchKernelGGL(k_wire_notnull, dim3(gx, n_cols), dim3(256), 0, ctx->stream, ca);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + TSQ_MAX_COLS, dcnt.p, TSQ_MAX_COLS * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_chunk_encode: ") + hipGetErrorString(e)));
    int64_t nulls[TSQ_MAX_COLS], total = 0;
    for (int c = 0; c < n_cols; c++) {
        if (in_dev && cols[c].type == TSQ_BYTES) {
            off0[c] = (int64_t)ctx->pinned[2 * TSQ_MAX_COLS + c];
            data_bytes[c] = (int64_t)ctx->pinned[c] - off0[c];
            if (off0[c] < 0 || data_bytes[c] < 0) return fail(tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_encode: bad offsets"));
            data[c] = (const uint8_t*)data[c] + off0[c];  // the bytes of rows [0, nrows) of the view
        }
        nulls[c] = ca.bm[c] ? nrows - (int64_t)ctx->pinned[TSQ_MAX_COLS + c] : 0;
        total += 8 + (nulls[c] > 0 ? (int64_t)nbm : 0) + (cols[c].type == TSQ_BYTES ? (nrows + 1) * 8 : 0) + data_bytes[c];
    }
    *bytes_out = total;
    if (cap_bytes == 0) return fail(TSQ_OK);  // size query
    if (total > cap_bytes) return fail(tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_encode: out buffer too small (*bytes_out holds the bytes needed)"));
    uint8_t* w = out;
    if (!out_dev) {
        s = dout.reserve(ctx, h, (size_t)total + 64);
        if (s != TSQ_OK) return fail(s);
        w = dout.as<uint8_t>();
    }
    {   // offsets that do not start at 0 are rebased into an aligned scratch first (WM_OFFS wants an aligned destination; the wire
        // position of the offsets is arbitrary): the wire chunk then carries offsets from 0 and only the view's bytes
        MovePlan pre;
        memset(&pre.a, 0, sizeof pre.a);
        for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
            if (cols[c].type != TSQ_BYTES || off0[c] == 0) continue;
            s = srebase[c].reserve(ctx, h, ((size_t)nrows + 1) * 8 + 64);
            if (s != TSQ_OK) break;
            pre.add(WM_OFFS, (const uint8_t*)offs[c], (uint8_t*)srebase[c].p, nrows + 1, -off0[c]);
            offs[c] = srebase[c].as<int64_t>();
        }
        if (s != TSQ_OK) return fail(s);
        if (pre.grid > 0) {
            hipLaunchKernelGGL(k_wire_move, dim3(pre.grid), dim3(256), 0, ctx->stream, pre.a);
            e = hipGetLastError();
            if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_chunk_encode(rebase): ") + hipGetErrorString(e)));
        }
    }
    MovePlan mp;
    memset(&mp.a, 0, sizeof mp.a);
    int64_t pos = 0;
    for (int c = 0; c < n_cols; c++) {
        mp.add(WM_HDR, nullptr, w + pos, 8, (int64_t)((uint64_t)nrows | ((uint64_t)nulls[c] << 32)));
        pos += 8;
        // the bitmap through the bit mover (no shift): the bits beyond the last row are cleared — a Go Column never has them set
        // (appendNullBitmap, column.go:113-125), a device column of this library may (bitmaps preset to all ones)
        if (nulls[c] > 0) { mp.add(WM_BITS, bm[c], w + pos, nrows, 0); pos += (int64_t)nbm; }
        if (cols[c].type == TSQ_BYTES) { mp.add(WM_COPY, offs[c], w + pos, (nrows + 1) * 8, 0); pos += (nrows + 1) * 8; }
        mp.add(WM_COPY, data[c], w + pos, data_bytes[c], 0);
        pos += data_bytes[c];
    }
    hipLaunchKernelGGL(k_wire_move, dim3(mp.grid), dim3(256), 0, ctx->stream, mp.a);
    e = hipGetLastError();
    if (e == hipSuccess && !out_dev) e = hipMemcpyAsync(out, w, (size_t)total, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && (!out_dev || !in_dev)) e = hipStreamSynchronize(ctx->stream);  // staging buffers go back to the pool
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_chunk_encode(move): ") + hipGetErrorString(e)));
    return fail(TSQ_OK);
}

// ------------------------------------------------------------------------------------- Codec.DecodeToChunk / Decoder.Decode
namespace {
struct WireView {  // the walk of one buffer
    int64_t rows[TSQ_MAX_COLS], nulls[TSQ_MAX_COLS], data_bytes[TSQ_MAX_COLS], off_first[TSQ_MAX_COLS], off_last[TSQ_MAX_COLS];
    int64_t bitmap_pos[TSQ_MAX_COLS], offs_pos[TSQ_MAX_COLS], data_pos[TSQ_MAX_COLS];
    int64_t consumed, take;
};

tsq_status wire_view(tsq_ctx* ctx, const char* who, const uint8_t* buf, int64_t n_bytes, bool in_dev, const int32_t* col_types, int32_t n_cols, int64_t first,
                     int64_t max_rows, WireView& v) {
    tsq_handle_hdr* h = &ctx->hdr;
    int32_t elem[TSQ_MAX_COLS];
    for (int c = 0; c < n_cols; c++) elem[c] = wire_elem(col_types[c]);
    uint64_t words[TSQ_MAX_COLS * 4];
    if (in_dev) {
        WireWalkArgs wa;
        memset(&wa, 0, sizeof wa);
        wa.buf = buf;
        wa.n_bytes = n_bytes;
        wa.first = first;
        wa.max_rows = max_rows;
        wa.n_cols = n_cols;
        for (int c = 0; c < n_cols; c++) wa.elem[c] = elem[c];
        wa.out = ctx->dscratch;
        hipLaunchKernelGGL(k_wire_walk, dim3(1), dim3(64), 0, ctx->stream, wa);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned, ctx->dscratch, (size_t)n_cols * 32, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
        memcpy(words, ctx->pinned, (size_t)n_cols * 32);
    } else {
        tsq_wire_walk(buf, n_bytes, elem, n_cols, first, max_rows, words);
    }
    int64_t pos = 0;
    for (int c = 0; c < n_cols; c++) {
        if (words[4 * c] == ~0ull) return tsq_fail(h, TSQ_ERR_INVALID, std::string(who) + ": the buffer ends inside column " + std::to_string(c) + " (or its offsets are damaged)");
        v.rows[c] = (int64_t)(words[4 * c] & 0xffffffffull);
        v.nulls[c] = (int64_t)(words[4 * c] >> 32);
        v.data_bytes[c] = (int64_t)words[4 * c + 1];
        v.off_first[c] = (int64_t)words[4 * c + 2];
        v.off_last[c] = (int64_t)words[4 * c + 3];
        if (v.rows[c] != v.rows[0]) return tsq_fail(h, TSQ_ERR_INVALID, std::string(who) + ": columns of different lengths");
        pos += 8;
        v.bitmap_pos[c] = -1;
        if (v.nulls[c] > 0) { v.bitmap_pos[c] = pos; pos += (v.rows[c] + 7) / 8; }
        v.offs_pos[c] = -1;
        if (elem[c] < 0) {
            v.offs_pos[c] = pos;
            pos += (v.rows[c] + 1) * 8;
            if (v.off_first[c] < 0 || v.off_last[c] < v.off_first[c] || v.off_last[c] > v.data_bytes[c])
                return tsq_fail(h, TSQ_ERR_INVALID, std::string(who) + ": offsets of column " + std::to_string(c) + " are damaged");
        }
        v.data_pos[c] = pos;
        pos += v.data_bytes[c];
    }
    v.consumed = pos;
    const int64_t f = std::min(first, v.rows[0]);
    v.take = std::min(max_rows, v.rows[0] - f);
    return TSQ_OK;
}

tsq_status wire_check(tsq_ctx* ctx, const char* who, const uint8_t* buf, int64_t n_bytes, const int32_t* col_types, int32_t n_cols, int64_t first, int64_t max_rows) {
    tsq_handle_hdr* h = &ctx->hdr;
    if (!col_types || n_bytes < 0 || (n_bytes > 0 && !buf) || first < 0 || max_rows < 0) return tsq_fail(h, TSQ_ERR_INVALID, std::string(who) + ": bad arguments");
    if (n_cols < 1 || n_cols > TSQ_MAX_COLS) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "1..16 columns supported");
    for (int c = 0; c < n_cols; c++)
        if (col_types[c] < TSQ_I64 || col_types[c] > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, std::string(who) + ": unknown column type");
    if (first & 7) return tsq_fail(h, TSQ_ERR_INVALID, std::string(who) + ": first_row must be a multiple of 8 (Decoder.Decode, codec.go:258-260)");
    return TSQ_OK;
}
}  // namespace

// the offsets of the window [first, first + take] of every var-len column must not decrease: its endpoints lie inside the data
// (wire_view), so every cell then does — a damaged or hostile chunk is refused before a byte of the destination changes (the
// reference slices out of range and panics, codec.go:314-320)
struct WireOffsCheckArgs {
    const uint8_t* offs[TSQ_MAX_COLS];  // at any byte position
    int64_t n;                          // pairs to compare
    uint32_t* bad;                      // bit c: column c
};
__global__ void __launch_bounds__(256) k_wire_check_offs(WireOffsCheckArgs a) {
    const uint8_t* p = a.offs[blockIdx.y];
    if (!p) return;
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < a.n; k += (int64_t)gridDim.x * 256)
        bad |= ((const tsq_wire_i64u*)(p + (k + 1) * 8))->v < ((const tsq_wire_i64u*)(p + k * 8))->v;
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(a.bad, 1u << blockIdx.y);
}

TSQ_API tsq_status tsq_chunk_decode_peek(tsq_ctx* ctx, const uint8_t* buf, int64_t n_bytes, uint32_t data_flags, const int32_t* col_types, int32_t n_cols,
                                         int64_t first_row, int64_t max_rows, int64_t* rows_total_out, int64_t* nrows_out, int64_t* bytes_out,
                                         int64_t* bytes_consumed) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    TSQ_TRY(wire_check(ctx, "tsq_chunk_decode_peek", buf, n_bytes, col_types, n_cols, first_row, max_rows));
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    WireView v;
    TSQ_TRY(wire_view(ctx, "tsq_chunk_decode_peek", buf, n_bytes, data_flags & TSQ_COL_DEVICE, col_types, n_cols, first_row, max_rows, v));
    if (rows_total_out) *rows_total_out = v.rows[0];
    if (nrows_out) *nrows_out = v.take;
    if (bytes_consumed) *bytes_consumed = v.consumed;
    if (bytes_out)
        for (int c = 0; c < n_cols; c++) bytes_out[c] = col_types[c] == TSQ_BYTES ? v.off_last[c] - v.off_first[c] : 0;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_chunk_decode(tsq_ctx* ctx, const uint8_t* buf, int64_t n_bytes, uint32_t data_flags, const int32_t* col_types, int32_t n_cols,
                                    int64_t first_row, int64_t max_rows, tsq_col* out_cols, int64_t* nrows_out, int64_t* bytes_consumed) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (nrows_out) *nrows_out = 0;
    if (bytes_consumed) *bytes_consumed = 0;
    TSQ_TRY(wire_check(ctx, "tsq_chunk_decode", buf, n_bytes, col_types, n_cols, first_row, max_rows));
    if (!out_cols || !nrows_out) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_decode: bad arguments");
    for (int c = 0; c < n_cols; c++) {
        if (((out_cols[c].flags ^ out_cols[0].flags) & TSQ_COL_DEVICE) != 0) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_decode: mixed host/device outputs");
        if (out_cols[c].length != out_cols[0].length || out_cols[c].length < 0) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_decode: out columns of different lengths");
        if (!out_cols[c].null_bitmap || (col_types[c] == TSQ_BYTES && !out_cols[c].offsets))
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_decode: out columns need null_bitmap buffers (a var-len column: offsets too)");
    }
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const bool in_dev = data_flags & TSQ_COL_DEVICE, out_dev = out_cols[0].flags & TSQ_COL_DEVICE;
    WireView v;
    TSQ_TRY(wire_view(ctx, "tsq_chunk_decode", buf, n_bytes, in_dev, col_types, n_cols, first_row, max_rows, v));
    const int64_t dst_rows = out_cols[0].length, take = v.take, first = std::min(first_row, v.rows[0]);
    if (bytes_consumed) *bytes_consumed = v.consumed;
    *nrows_out = take;
    DevBuf dbuf, sdata[TSQ_MAX_COLS], sbm[TSQ_MAX_COLS], soffs[TSQ_MAX_COLS];
    auto fail = [&](tsq_status st) {
        dbuf.release();
        for (int c = 0; c < TSQ_MAX_COLS; c++) { sdata[c].release(); sbm[c].release(); soffs[c].release(); }
        return st;
    };
    hipError_t e = hipSuccess;
    tsq_status s = TSQ_OK;
    const uint8_t* w = buf;
    if (!in_dev && v.consumed > 0) {
        s = dbuf.reserve(ctx, h, (size_t)v.consumed + 64);
        if (s != TSQ_OK) return fail(s);
        e = hipMemcpyAsync(dbuf.p, buf, (size_t)v.consumed, hipMemcpyHostToDevice, ctx->stream);
        w = dbuf.as<uint8_t>();
    }
    // where the appended rows go: offsets continue from offsets[dst_rows] of the destination (0 for an empty one)
    int64_t base[TSQ_MAX_COLS];
    for (int c = 0; c < n_cols; c++) base[c] = 0;
    if (dst_rows > 0) {
        bool any = false;
        for (int c = 0; c < n_cols && e == hipSuccess; c++) {
            if (col_types[c] != TSQ_BYTES) continue;
            if (out_dev) { e = hipMemcpyAsync(ctx->pinned + c, out_cols[c].offsets + dst_rows, 8, hipMemcpyDeviceToHost, ctx->stream); any = true; }
            else base[c] = out_cols[c].offsets[dst_rows];
        }
        if (any && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (any && e == hipSuccess)
            for (int c = 0; c < n_cols; c++)
                if (col_types[c] == TSQ_BYTES) base[c] = (int64_t)ctx->pinned[c];
    }
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_chunk_decode: ") + hipGetErrorString(e)));
    {
        WireOffsCheckArgs oc;
        memset(&oc, 0, sizeof oc);
        bool any = false;
        for (int c = 0; c < n_cols; c++)
            if (col_types[c] == TSQ_BYTES && take > 0) { oc.offs[c] = w + v.offs_pos[c] + first * 8; any = true; }
        if (any) {
            oc.n = take;
            oc.bad = (uint32_t*)(ctx->dscratch + 40);
            e = hipMemsetAsync(oc.bad, 0, 8, ctx->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_wire_check_offs, dim3((unsigned)std::min<int64_t>(ctx->num_cus * 4, (take + 255) / 256), n_cols), dim3(256), 0, ctx->stream, oc);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 40, oc.bad, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_chunk_decode(check): ") + hipGetErrorString(e)));
            const uint32_t bad = (uint32_t)ctx->pinned[40];
            if (bad) {
                int c = 0;
                while (!((bad >> c) & 1)) c++;
                *nrows_out = 0;
                return fail(tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_decode: offsets of column " + std::to_string(c) + " are damaged (they decrease)"));
            }
        }
    }
    const int b = (int)(dst_rows & 7);
    const int64_t bm_first = dst_rows >> 3, bm_bytes = ((dst_rows + take + 7) >> 3) - bm_first;
    MovePlan mp;
    memset(&mp.a, 0, sizeof mp.a);
    for (int c = 0; c < n_cols && s == TSQ_OK && e == hipSuccess; c++) {
        const bool var = col_types[c] == TSQ_BYTES;
        const int es = var ? 0 : tsq_elem_size(col_types[c]);
        const int64_t nb = var ? v.off_last[c] - v.off_first[c] : take * es;
        const int64_t src_at = v.data_pos[c] + (var ? v.off_first[c] : first * es);
        const int64_t dst_at = var ? base[c] : dst_rows * es;
        if (nb > 0 && !out_cols[c].data) return fail(tsq_fail(h, TSQ_ERR_INVALID, "tsq_chunk_decode: out column without a data buffer"));
        uint8_t *ddata = (uint8_t*)out_cols[c].data + dst_at, *dbm = out_cols[c].null_bitmap;
        int64_t* doffs = var ? out_cols[c].offsets + dst_rows : nullptr;
        if (!out_dev) {  // host outputs: the same pieces in HBM first, at the same alignment
            s = sdata[c].reserve(ctx, h, (size_t)nb + 64);
            if (s == TSQ_OK) s = sbm[c].reserve(ctx, h, (size_t)bm_bytes + 64);
            if (s == TSQ_OK && var) s = soffs[c].reserve(ctx, h, ((size_t)take + 1) * 8 + 64);
            if (s != TSQ_OK) break;
            ddata = sdata[c].as<uint8_t>();
            dbm = sbm[c].as<uint8_t>() - bm_first;  // (only bytes [bm_first, bm_first + bm_bytes) are touched)
            doffs = var ? soffs[c].as<int64_t>() : nullptr;
            if (b && take > 0) e = hipMemcpyAsync(sbm[c].p, out_cols[c].null_bitmap + bm_first, 1, hipMemcpyHostToDevice, ctx->stream);
        }
        if (take > 0) {
            mp.add(WM_BITS, v.bitmap_pos[c] >= 0 ? w + v.bitmap_pos[c] + (first >> 3) : nullptr, dbm, take, dst_rows);
            mp.add(WM_COPY, w + src_at, ddata, nb, 0);
        }
        if (var) {
            // offsets[dst_rows + 1 + i] = wire offsets[first + 1 + i] + (base - wire offsets[first])   (codec.go:314-320); an empty
            // destination also gets its offsets[0] = 0
            if (dst_rows == 0) mp.add(WM_OFFS, w + v.offs_pos[c] + first * 8, (uint8_t*)doffs, take + 1, base[c] - v.off_first[c]);
            else mp.add(WM_OFFS, w + v.offs_pos[c] + (first + 1) * 8, (uint8_t*)(doffs + 1), take, base[c] - v.off_first[c]);
        }
    }
    if (s != TSQ_OK) return fail(s);
    if (e == hipSuccess && mp.grid > 0) {
        hipLaunchKernelGGL(k_wire_move, dim3(mp.grid), dim3(256), 0, ctx->stream, mp.a);
        e = hipGetLastError();
    }
    if (!out_dev) {
        for (int c = 0; c < n_cols && e == hipSuccess; c++) {
            const bool var = col_types[c] == TSQ_BYTES;
            const int es = var ? 0 : tsq_elem_size(col_types[c]);
            const int64_t nb = var ? v.off_last[c] - v.off_first[c] : take * es;
            if (nb > 0) e = hipMemcpyAsync((uint8_t*)out_cols[c].data + (var ? base[c] : dst_rows * es), sdata[c].p, (size_t)nb, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && take > 0) e = hipMemcpyAsync(out_cols[c].null_bitmap + bm_first, sbm[c].p, (size_t)bm_bytes, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && var) {
                if (dst_rows == 0) e = hipMemcpyAsync(out_cols[c].offsets, soffs[c].p, ((size_t)take + 1) * 8, hipMemcpyDeviceToHost, ctx->stream);
                else if (take > 0) e = hipMemcpyAsync(out_cols[c].offsets + dst_rows + 1, soffs[c].as<int64_t>() + 1, (size_t)take * 8, hipMemcpyDeviceToHost, ctx->stream);
            }
        }
    }
    if (e == hipSuccess && (!out_dev || !in_dev)) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_chunk_decode(move): ") + hipGetErrorString(e)));
    for (int c = 0; c < n_cols; c++) out_cols[c].length = dst_rows + take;
    return fail(TSQ_OK);
}
