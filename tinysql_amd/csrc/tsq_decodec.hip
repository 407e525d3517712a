// tsq_decodec.hip — coprocessor response CHUNKS -> chunk columns on the GPU, var-len columns included (SURVEY.md §8 f, rank 2).
//
// Replaces selectResult.readRowsData (distsql/select_result.go:139-155) + codec.Decoder.DecodeOne (util/codec/codec.go:623-690)
// for a response whose tipb.Chunks are known: the storage side cuts the rows into chunks of 64 (cop_handler_dag.go:510-519), each
// an independent RowsData byte string.  tsq_rows_decode (tsq_decode.hip) parses ONE byte string without any boundary and owes its
// speed to a bound on the value length (<= 11 bytes); a compact-bytes datum — what a varchar / blob column arrives as — has no such
// bound.  Here the chunk boundaries carry the parallelism instead: ONE LANE walks ONE chunk, value after value.
//   K13d k_decc_count : values per chunk, the first offending value in stream order (the reference's error), datum kinds against
//                       the column types
//   (scan)            : rows per chunk -> first output row of every chunk (tsq_launch_scan64)
//   K13e k_decc_emit  : the same walk again; value i of chunk k goes to column i % n_cols, row base[k] + i / n_cols; a string cell
//                       leaves (absolute position, length) — its bytes are copied by K13f once the column's offsets exist
//   K13f k_decc_var_copy : lengths -> scan = offsets[n + 1], then the bytes (one cell per lane / per wave)
// HBM-bound byte work, no MFMA.  A lane's loads are its own chunk's lines (64 distinct lines per wave instruction), so this
// route runs at the L2 request rate, not at the streaming rate — it is the route for schemas the bounded-lookahead parse cannot
// take; all-fixed-width responses keep tsq_rows_decode.  Algorithmic bytes: the encoded bytes + 8 B per value (+ the string bytes
// a second time).
#include "tsq_stage.h"
#include "tsq_decode_dp.h"

struct DeccArgs {
    const uint8_t* data;
    int64_t n_bytes;
    const int64_t* chunk_offs;  // [n_chunks + 1]
    int64_t n_chunks;
    int32_t n_cols;
    int32_t col_type[TSQ_MAX_COLS];
    int64_t* rows;              // [n_chunks + 1]: complete rows per chunk (K13d), then their exclusive scan
    unsigned long long* err;    // min over (chunk << 32 | min(value ordinal, 2^28 - 1) << 4 | code); ~0 = none
    // K13e
    int64_t err_chunk, err_rows;  // chunk that holds the first error (n_chunks: none) and the complete rows before it inside that chunk
    void* out_data[TSQ_MAX_COLS];       // fixed-width columns
    uint8_t* out_notnull[TSQ_MAX_COLS]; // one byte per row
    int64_t* ref_pos[TSQ_MAX_COLS];     // var-len columns: where the cell's bytes start in `data`
    int64_t* out_offs[TSQ_MAX_COLS];    // var-len columns: the cell's length (the scan makes offsets of them)
    // index keys (tsq_indexkeys_decode): every "chunk" is one key = `prefix` bytes, n_key_cols datums, then the handle datum or nothing
    int32_t prefix, n_key_cols, pk_status;
    const uint8_t* vals;        // the pairs' values: the handle of a key that does not carry it (8 bytes big endian)
    const int64_t* val_offs;    // [n_chunks + 1]
    int64_t n_val_bytes;
};

namespace {

// the 12 bytes at data[p ..], as three little-endian words; bytes at or beyond `end` read as zero.  Only aligned 8-byte words that
// contain at least one byte of [0, n_bytes) are touched.
__device__ __forceinline__ void decc_fetch12(const uint8_t* data, int64_t n_bytes, int64_t p, int64_t end, uint32_t* b0, uint32_t* b1, uint32_t* b2) {
    const uintptr_t addr = (uintptr_t)(data + p), a = addr & ~(uintptr_t)7, lim = (uintptr_t)(data + n_bytes);
    const uint32_t sh = (uint32_t)(addr & 7) * 8u;
    const uint64_t w0 = *reinterpret_cast<const uint64_t*>(a);
    const uint64_t w1 = a + 8 < lim ? *reinterpret_cast<const uint64_t*>(a + 8) : 0ull;
    const uint64_t w2 = a + 16 < lim ? *reinterpret_cast<const uint64_t*>(a + 16) : 0ull;
    uint64_t lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;  // bytes p .. p + 7
    uint64_t hi = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;  // bytes p + 8 .. p + 15
    const int64_t valid = end - p;                            // >= 1
    if (valid < 8) { lo &= (1ull << (8 * valid)) - 1; hi = 0; }
    else if (valid < 12) hi &= (1ull << (8 * (valid - 8))) - 1;
    *b0 = (uint32_t)lo;
    *b1 = (uint32_t)(lo >> 32);
    *b2 = (uint32_t)hi;
}

__device__ __forceinline__ bool decc_chunk_range(const DeccArgs& a, int64_t k, int64_t* lo, int64_t* hi) {
    *lo = a.chunk_offs[k];
    *hi = a.chunk_offs[k + 1];
    return *lo >= 0 && *hi >= *lo && *hi <= a.n_bytes;
}

__global__ void __launch_bounds__(256) k_decc_count(DeccArgs a) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < a.n_chunks; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo, hi;
        int code = DEC_OK;
        int64_t vals = 0;
        if (!decc_chunk_range(a, k, &lo, &hi)) {
            code = DEC_ROW_CUT;  // chunk boundaries that run backwards or past the bytes: nothing of this chunk can be read
        } else {
            int32_t col = 0;
            for (int64_t p = lo; p < hi;) {
                uint32_t b0, b1, b2;
                decc_fetch12(a.data, a.n_bytes, p, hi, &b0, &b1, &b2);
                tsq_decc_val v;
                code = tsq_decc_value(b0, b1, b2, (uint64_t)(hi - p), &v);
                if (code == DEC_VARLEN) code = tsq_decc_membytes(a.data + p, (uint64_t)(hi - p), &v);
                uint64_t bits;
                if (code == DEC_OK && !tsq_decc_store(a.col_type[col], v, &bits)) code = DEC_KIND_MISMATCH;
                if (code != DEC_OK) break;
                p += (int64_t)v.len;
                vals++;
                col = col + 1 == a.n_cols ? 0 : col + 1;
            }
            // a last row that ends early is noticed by the DecodeOne call after its last value (codec.go:624-626)
            if (code == DEC_OK && vals % a.n_cols != 0) code = DEC_ROW_CUT;
        }
        a.rows[k] = vals / a.n_cols;
        if (code != DEC_OK) {
            const unsigned long long ord = (unsigned long long)(vals < (1 << 28) - 1 ? vals : (1 << 28) - 1);
            atomicMin(a.err, ((unsigned long long)k << 32) | (ord << 4) | (unsigned long long)code);
        }
    }
}

__global__ void __launch_bounds__(256) k_decc_emit(DeccArgs a) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < a.n_chunks && k <= a.err_chunk; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo, hi;
        if (!decc_chunk_range(a, k, &lo, &hi)) continue;
        const int64_t base = a.rows[k];
        const int64_t want = k == a.err_chunk ? a.err_rows : a.rows[k + 1] - base;  // complete rows of this chunk that are handed over
        int64_t r = 0;
        int32_t col = 0;
        for (int64_t p = lo; p < hi && r < want;) {
            uint32_t b0, b1, b2;
            decc_fetch12(a.data, a.n_bytes, p, hi, &b0, &b1, &b2);
            tsq_decc_val v;
            int code = tsq_decc_value(b0, b1, b2, (uint64_t)(hi - p), &v);
            const bool grouped = code == DEC_VARLEN;
            if (grouped) code = tsq_decc_membytes(a.data + p, (uint64_t)(hi - p), &v);
            if (code != DEC_OK) break;  // (K13d has seen it: rows before it only)
            uint64_t bits;
            (void)tsq_decc_store(a.col_type[col], v, &bits);
            const int64_t row = base + r;
            const int32_t t = a.col_type[col];
            if (t == TSQ_BYTES) {
                a.ref_pos[col][row] = (p + (int64_t)v.data_at) | (grouped ? TSQ_DECC_GROUPED : 0);
                a.out_offs[col][row] = v.kind == DECV_BYTES ? (int64_t)v.bits : 0;  // a NULL cell has no bytes
            } else if (t == TSQ_F32) {
                ((uint32_t*)a.out_data[col])[row] = (uint32_t)bits;
            } else {
                ((uint64_t*)a.out_data[col])[row] = bits;
            }
            a.out_notnull[col][row] = v.kind != DECV_NULL ? 1 : 0;
            p += (int64_t)v.len;
            col++;
            if (col == a.n_cols) { col = 0; r++; }
        }
    }
}

// one datum at p (DecodeOne, codec.go:623-690); *grouped: a memcomparable string
__device__ __forceinline__ int decc_one(const DeccArgs& a, int64_t p, int64_t hi, tsq_decc_val* v, bool* grouped) {
    uint32_t b0, b1, b2;
    decc_fetch12(a.data, a.n_bytes, p, hi, &b0, &b1, &b2);
    int code = tsq_decc_value(b0, b1, b2, (uint64_t)(hi - p), v);
    *grouped = code == DEC_VARLEN;
    if (*grouped) code = tsq_decc_membytes(a.data + p, (uint64_t)(hi - p), v);
    return code;
}

// K13g: index keys.  tablecodec.DecodeIndexKV (tablecodec.go:376-434) of pair k: CutIndexKeyNew skips the 19-byte prefix
// ('t' tableID "_i" indexID) and cuts n_key_cols datums; if bytes remain they are the handle datum (a non-unique index, or a unique
// one whose key holds a NULL), otherwise the pair's VALUE is the handle (DecodeIndexValueAsHandle, tablecodec.go:456-465: 8 bytes
// big endian).  Row k = pair k; the first offending pair in key order goes to a.err (the pairs before it are the result).
__global__ void __launch_bounds__(256) k_idx_walk(DeccArgs a) {
    constexpr bool EMIT = true;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < a.n_chunks; k += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo, hi;
        int code = DEC_OK;
        int32_t col = 0;
        if (!decc_chunk_range(a, k, &lo, &hi) || hi - lo < a.prefix) {
            code = DEC_ROW_CUT;
        } else {
            int64_t p = lo + a.prefix;
            for (; col < a.n_key_cols; col++) {
                if (p >= hi) { code = DEC_ROW_CUT; break; }  // peek / DecodeOne on an empty slice: "invalid encoded key"
                tsq_decc_val v;
                bool grouped;
                code = decc_one(a, p, hi, &v, &grouped);
                uint64_t bits = 0;
                if (code == DEC_OK && !tsq_decc_store(a.col_type[col], v, &bits)) code = DEC_KIND_MISMATCH;
                if (code != DEC_OK) break;
                if (EMIT) {
                    const int32_t t = a.col_type[col];
                    if (t == TSQ_BYTES) {
                        a.ref_pos[col][k] = (p + (int64_t)v.data_at) | (grouped ? TSQ_DECC_GROUPED : 0);
                        a.out_offs[col][k] = v.kind == DECV_BYTES ? (int64_t)v.bits : 0;
                    } else if (t == TSQ_F32) {
                        ((uint32_t*)a.out_data[col])[k] = (uint32_t)bits;
                    } else {
                        ((uint64_t*)a.out_data[col])[k] = bits;
                    }
                    a.out_notnull[col][k] = v.kind != DECV_NULL ? 1 : 0;
                }
                p += (int64_t)v.len;
            }
            if (code == DEC_OK && a.pk_status != 0) {
                uint64_t handle = 0;
                if (p < hi) {  // the handle travels in the key (values = append(values, b), tablecodec.go:411-414)
                    tsq_decc_val v;
                    bool grouped;
                    code = decc_one(a, p, hi, &v, &grouped);
                    if (code == DEC_OK && (v.kind == DECV_NULL || v.kind == DECV_BYTES)) code = DEC_KIND_MISMATCH;  // (an int datum in every key the reference writes)
                    handle = v.bits;
                } else {
                    const int64_t vlo = a.val_offs ? a.val_offs[k] : 0, vhi = a.val_offs ? a.val_offs[k + 1] : -1;
                    if (!a.vals || vlo < 0 || vhi - vlo < 8 || vhi > a.n_val_bytes) code = DEC_NO_HANDLE;
                    else
                        for (int i = 0; i < 8; i++) handle = (handle << 8) | a.vals[vlo + i];  // binary.Read(buf, binary.BigEndian, &h)
                }
                if (EMIT && code == DEC_OK) {
                    ((uint64_t*)a.out_data[col])[k] = handle;
                    a.out_notnull[col][k] = 1;
                }
            }
        }
        // the first offending pair in key order decides (row k = pair k: the emitting walk needs no validating pass before it)
        if (code != DEC_OK) atomicMin(a.err, ((unsigned long long)k << 32) | ((unsigned long long)col << 4) | (unsigned long long)code);
    }
}

struct DeccVarArgs {
    const uint8_t* data;
    const int64_t* pos;
    const int64_t* offs;  // [rows + 1]
    int64_t rows;
    uint8_t* out;
};
template <bool WAVE>
__global__ void __launch_bounds__(256) k_decc_var_copy(DeccVarArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = WAVE ? gtid >> 6 : gtid, step = WAVE ? nthr >> 6 : nthr;
    for (int64_t r = first; r < a.rows; r += step) {
        const int64_t n = a.offs[r + 1] - a.offs[r];
        if (n == 0) continue;
        const bool grouped = (a.pos[r] & TSQ_DECC_GROUPED) != 0;
        const uint8_t* s = a.data + (a.pos[r] & ~TSQ_DECC_GROUPED);
        uint8_t* d = a.out + a.offs[r];
        if (grouped) {  // a memcomparable cell: one marker byte follows every 8 data bytes
            for (int64_t i = WAVE ? lane : 0; i < n; i += WAVE ? 64 : 1) d[i] = s[i + (i >> 3)];
        } else if (!WAVE) {
            tsq_copy_cell(d, s, n);
        } else {  // head up to an 8-byte boundary of the destination, then 8 bytes per lane, then the tail
            int64_t head = (8 - ((uintptr_t)d & 7)) & 7;
            head = head < n ? head : n;
            if (lane < head) d[lane] = s[lane];
            const int64_t words = (n - head) >> 3;
            for (int64_t w = lane; w < words; w += 64) {
                uint64_t x;
                memcpy(&x, s + head + w * 8, 8);
                *reinterpret_cast<uint64_t*>(d + head + w * 8) = x;
            }
            const int64_t done = head + words * 8;
            if (done + lane < n) d[done + lane] = s[done + lane];
        }
    }
}

}  // namespace

// cells (pos[r], offs[r + 1] - offs[r]) of `data` -> out + offs[r]: shared with the var-len columns of tsq_radix_split (tsq_stage.h)
tsq_status tsq_launch_var_copy(tsq_ctx* ctx, tsq_handle_hdr* h, const uint8_t* data, const int64_t* pos, const int64_t* offs, int64_t rows, int64_t total_bytes,
                               uint8_t* out) {
    if (rows <= 0 || total_bytes <= 0) return TSQ_OK;
    DeccVarArgs va;
    va.data = data;
    va.pos = pos;
    va.offs = offs;
    va.rows = rows;
    va.out = out;
    if (total_bytes / rows > 32) hipLaunchKernelGGL(k_decc_var_copy<true>, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, va);
    else hipLaunchKernelGGL(k_decc_var_copy<false>, dim3(tsq_grid_for(ctx, rows, 256)), dim3(256), 0, ctx->stream, va);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("var copy: ") + hipGetErrorString(e));
    return TSQ_OK;
}

// ====================================================================== host side
namespace {
struct IdxMode {  // tsq_indexkeys_decode: the extra inputs of the key walk
    int32_t n_key_cols, pk_status;
    const uint8_t* vals;
    int64_t n_val_bytes;
    const int64_t* val_offs;
};
}  // namespace

static tsq_status decc_decode(tsq_ctx* ctx, const std::string& who, const uint8_t* rows_data, int64_t n_bytes, const int64_t* chunk_offsets, int64_t n_chunks,
                              uint32_t data_flags, int32_t n_cols, const int32_t* col_types, tsq_col* out_cols, int64_t cap_rows, int64_t* nrows_out,
                              const IdxMode* idx) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (nrows_out) *nrows_out = 0;
    if (!nrows_out || !col_types || !out_cols || n_bytes < 0 || n_chunks < 0 || cap_rows < 0 || (n_bytes > 0 && !rows_data) || (n_chunks > 0 && !chunk_offsets))
        return tsq_fail(h, TSQ_ERR_INVALID, who + ": bad arguments");
    if (n_cols < 1 || n_cols > TSQ_MAX_COLS) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "1..16 columns supported");
    if (n_chunks >= (1LL << 31)) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "more than 2^31 response chunks per call");
    bool any_var = false;
    for (int c = 0; c < n_cols; c++) {
        if (col_types[c] < TSQ_I64 || col_types[c] > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, who + ": unknown column type");
        const bool var = col_types[c] == TSQ_BYTES;
        any_var = any_var || var;
        if (!out_cols[c].null_bitmap || (var ? (!out_cols[c].offsets || (n_bytes > 0 && !out_cols[c].data)) : !out_cols[c].data))
            return tsq_fail(h, TSQ_ERR_INVALID, who + ": out columns need data and null_bitmap buffers (a var-len column: offsets too)");
        if (((out_cols[c].flags ^ out_cols[0].flags) & TSQ_COL_DEVICE) != 0) return tsq_fail(h, TSQ_ERR_INVALID, who + ": mixed host/device outputs");
    }
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const bool in_dev = data_flags & TSQ_COL_DEVICE, out_dev = out_cols[0].flags & TSQ_COL_DEVICE;
    auto empty_result = [&]() -> tsq_status {  // no rows: a var-len column still has its first offset
        for (int c = 0; c < n_cols; c++) {
            out_cols[c].length = 0;
            if (col_types[c] != TSQ_BYTES) continue;
            if (out_dev) TSQ_HIP(h, hipMemsetAsync(out_cols[c].offsets, 0, 8, ctx->stream));
            else out_cols[c].offsets[0] = 0;
        }
        return TSQ_OK;
    };
    if (n_chunks == 0 || n_bytes == 0) return empty_result();
    DeccArgs a;
    memset(&a, 0, sizeof a);
    a.n_bytes = n_bytes;
    a.n_chunks = n_chunks;
    a.n_cols = n_cols;
    for (int c = 0; c < n_cols; c++) a.col_type[c] = col_types[c];
    DevBuf dbytes, dco, drows, derr, scratch, dvals, dvo, ddata[TSQ_MAX_COLS], dnn[TSQ_MAX_COLS], dbm[TSQ_MAX_COLS], dpos[TSQ_MAX_COLS], dvoffs[TSQ_MAX_COLS];
    auto release_all = [&]() {
        for (DevBuf* b : {&dbytes, &dco, &drows, &derr, &scratch, &dvals, &dvo}) b->release();
        for (int c = 0; c < TSQ_MAX_COLS; c++) { ddata[c].release(); dnn[c].release(); dbm[c].release(); dpos[c].release(); dvoffs[c].release(); }
    };
    auto fail = [&](tsq_status st) { release_all(); return st; };
    tsq_status s = drows.reserve(ctx, h, ((size_t)n_chunks + 1) * 8 + 64);
    if (s == TSQ_OK) s = derr.reserve(ctx, h, 64);
    hipError_t e = hipSuccess;
    if (s == TSQ_OK && !in_dev) {
        s = dbytes.reserve(ctx, h, (size_t)n_bytes + 64);
        if (s == TSQ_OK) s = dco.reserve(ctx, h, ((size_t)n_chunks + 1) * 8 + 64);
        if (s == TSQ_OK) e = hipMemcpyAsync(dbytes.p, rows_data, (size_t)n_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (s == TSQ_OK && e == hipSuccess) e = hipMemcpyAsync(dco.p, chunk_offsets, ((size_t)n_chunks + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
        a.data = dbytes.as<uint8_t>();
        a.chunk_offs = dco.as<int64_t>();
    } else {
        a.data = rows_data;
        a.chunk_offs = chunk_offsets;
    }
    if (idx) {
        a.prefix = 19;  // prefixLen + idLen: 't' | tableID | "_i" | indexID (tablecodec.go:372-374)
        a.n_key_cols = idx->n_key_cols;
        a.pk_status = idx->pk_status;
        a.vals = idx->vals;
        a.val_offs = idx->val_offs;
        a.n_val_bytes = idx->n_val_bytes;
        if (s == TSQ_OK && e == hipSuccess && !in_dev && idx->vals && idx->val_offs) {
            s = dvals.reserve(ctx, h, (size_t)idx->n_val_bytes + 64);
            if (s == TSQ_OK) s = dvo.reserve(ctx, h, ((size_t)n_chunks + 1) * 8 + 64);
            if (s == TSQ_OK && idx->n_val_bytes > 0) e = hipMemcpyAsync(dvals.p, idx->vals, (size_t)idx->n_val_bytes, hipMemcpyHostToDevice, ctx->stream);
            if (s == TSQ_OK && e == hipSuccess) e = hipMemcpyAsync(dvo.p, idx->val_offs, ((size_t)n_chunks + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
            a.vals = dvals.as<uint8_t>();
            a.val_offs = dvo.as<int64_t>();
        }
    }
    if (s != TSQ_OK) return fail(s);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string(who + "(H2D): ") + hipGetErrorString(e)));
    a.rows = drows.as<int64_t>();
    a.err = derr.as<unsigned long long>();
    e = hipMemsetAsync(a.err, 0xff, 8, ctx->stream);
    const int grid = tsq_grid_for(ctx, n_chunks, 256);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string(who + ": ") + hipGetErrorString(e)));
    uint64_t errw = ~0ull;
    int64_t rows = n_chunks;  // index keys: row k = pair k, one walk (k_idx_walk) writes the rows and finds the first offending pair
    if (!idx) {
        hipLaunchKernelGGL(k_decc_count, dim3(grid), dim3(256), 0, ctx->stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string(who + "(count): ") + hipGetErrorString(e)));
        s = tsq_launch_scan64(ctx, h, a.rows, n_chunks, scratch);  // rows[k] = first output row of chunk k, rows[n_chunks] = all rows
        if (s != TSQ_OK) return fail(s);
        e = hipMemcpyAsync(ctx->pinned, a.err, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 1, a.rows + n_chunks, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string(who + ": ") + hipGetErrorString(e)));
        errw = ctx->pinned[0];
        rows = (int64_t)ctx->pinned[1];
    }
    int code = DEC_OK;
    a.err_chunk = n_chunks;
    a.err_rows = 0;
    if (errw != ~0ull) {  // the rows before the first offending value: whole chunks before its chunk + the complete rows before it inside
        code = (int)(errw & 15);
        a.err_chunk = (int64_t)(errw >> 32);
        e = hipMemcpyAsync(ctx->pinned + 2, a.rows + a.err_chunk, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 3, a.rows + a.err_chunk + 1, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string(who + ": ") + hipGetErrorString(e)));
        a.err_rows = (int64_t)ctx->pinned[3] - (int64_t)ctx->pinned[2];  // K13d counted the complete rows before the error
        rows = (int64_t)ctx->pinned[2] + a.err_rows;
    }
    if (rows > cap_rows) {
        release_all();
        *nrows_out = rows;
        return tsq_fail(h, TSQ_ERR_INVALID, who + ": output columns too small (*nrows_out = rows needed)");
    }
    if (rows > 0) {
        for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
            const bool var = col_types[c] == TSQ_BYTES;
            s = dnn[c].reserve(ctx, h, (size_t)rows + 64);
            if (s == TSQ_OK && var) s = dpos[c].reserve(ctx, h, (size_t)rows * 8 + 64);
            if (s == TSQ_OK && !out_dev) {
                s = ddata[c].reserve(ctx, h, (var ? (size_t)n_bytes : (size_t)rows * tsq_elem_size(col_types[c])) + 64);
                if (s == TSQ_OK) s = dbm[c].reserve(ctx, h, tsq_bitmap_bytes(rows) + 64);
                if (s == TSQ_OK && var) s = dvoffs[c].reserve(ctx, h, ((size_t)rows + 1) * 8 + 64);
            }
            a.out_notnull[c] = dnn[c].as<uint8_t>();
            a.out_data[c] = var ? nullptr : (out_dev ? out_cols[c].data : ddata[c].p);
            a.ref_pos[c] = var ? dpos[c].as<int64_t>() : nullptr;
            a.out_offs[c] = var ? (out_dev ? out_cols[c].offsets : dvoffs[c].as<int64_t>()) : nullptr;
        }
        if (s != TSQ_OK) return fail(s);
        if (idx) hipLaunchKernelGGL(k_idx_walk, dim3(grid), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL(k_decc_emit, dim3(grid), dim3(256), 0, ctx->stream, a);
        e = hipGetLastError();
        if (idx && e == hipSuccess) {  // the pairs before the first offending one
            e = hipMemcpyAsync(ctx->pinned, a.err, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess && ctx->pinned[0] != ~0ull) {
                code = (int)(ctx->pinned[0] & 15);
                rows = (int64_t)(ctx->pinned[0] >> 32);
            }
        }
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string(who + "(emit): ") + hipGetErrorString(e)));
        int64_t var_bytes[TSQ_MAX_COLS] = {0};
        for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
            uint8_t* bm = out_dev ? out_cols[c].null_bitmap : dbm[c].as<uint8_t>();
            s = tsq_launch_pack_bitmap(ctx, h, a.out_notnull[c], bm, rows);
            if (s != TSQ_OK || col_types[c] != TSQ_BYTES) continue;
            s = tsq_launch_scan64(ctx, h, a.out_offs[c], rows, scratch);
            if (s != TSQ_OK) break;
            e = hipMemcpyAsync(ctx->pinned + 4, a.out_offs[c] + rows, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { s = tsq_fail(h, TSQ_ERR_HIP, std::string(who + "(var scan): ") + hipGetErrorString(e)); break; }
            var_bytes[c] = (int64_t)ctx->pinned[4];
            s = tsq_launch_var_copy(ctx, h, a.data, a.ref_pos[c], a.out_offs[c], rows, var_bytes[c], out_dev ? (uint8_t*)out_cols[c].data : ddata[c].as<uint8_t>());
        }
        if (s != TSQ_OK) return fail(s);
        if (!out_dev) {
            for (int c = 0; c < n_cols && e == hipSuccess; c++) {
                if (col_types[c] == TSQ_BYTES) {
                    e = hipMemcpyAsync(out_cols[c].offsets, dvoffs[c].p, ((size_t)rows + 1) * 8, hipMemcpyDeviceToHost, ctx->stream);
                    if (e == hipSuccess && var_bytes[c] > 0) e = hipMemcpyAsync(out_cols[c].data, ddata[c].p, (size_t)var_bytes[c], hipMemcpyDeviceToHost, ctx->stream);
                } else {
                    e = hipMemcpyAsync(out_cols[c].data, ddata[c].p, (size_t)rows * tsq_elem_size(col_types[c]), hipMemcpyDeviceToHost, ctx->stream);
                }
                if (e == hipSuccess) e = hipMemcpyAsync(out_cols[c].null_bitmap, dbm[c].p, tsq_bitmap_bytes(rows), hipMemcpyDeviceToHost, ctx->stream);
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string(who + "(D2H): ") + hipGetErrorString(e)));
    } else {
        s = empty_result();
        if (s != TSQ_OK) return fail(s);
    }
    for (int c = 0; c < n_cols; c++) {
        out_cols[c].length = rows;
        out_cols[c].type = col_types[c];
        out_cols[c].elem_size = col_types[c] == TSQ_BYTES ? -1 : tsq_elem_size(col_types[c]);
    }
    release_all();
    *nrows_out = rows;
    switch (code) {
        case DEC_OK: return TSQ_OK;
        case DEC_ROW_CUT: return tsq_fail(h, TSQ_ERR_INVALID, "invalid encoded key");                        // codec.go:625
        case DEC_INSUFFICIENT: return tsq_fail(h, TSQ_ERR_INVALID, "insufficient bytes to decode value");   // number.go:46,122; bytes.go:156-158
        case DEC_OVERFLOW: return tsq_fail(h, TSQ_ERR_INVALID, "value larger than 64 bits");                // number.go:120
        case DEC_BAD_MARKER: return tsq_fail(h, TSQ_ERR_INVALID, "invalid marker byte");                    // bytes.go:89-91
        case DEC_BAD_PADDING: return tsq_fail(h, TSQ_ERR_INVALID, "invalid padding byte");                  // bytes.go:103-107
        case DEC_KIND_MISMATCH: return tsq_fail(h, TSQ_ERR_INVALID, "datum kind does not match the column type");
        case DEC_NO_HANDLE: return tsq_fail(h, TSQ_ERR_INVALID, "no handle in index key or value");                // tablecodec.go:449-453, 456-465
        default: return tsq_fail(h, TSQ_ERR_INVALID, "invalid encoded key flag");                           // codec.go:683
    }
}

TSQ_API tsq_status tsq_rows_decode_chunks(tsq_ctx* ctx, const uint8_t* rows_data, int64_t n_bytes, const int64_t* chunk_offsets, int64_t n_chunks,
                                          uint32_t data_flags, int32_t n_cols, const int32_t* col_types, tsq_col* out_cols, int64_t cap_rows,
                                          int64_t* nrows_out) {
    return decc_decode(ctx, "tsq_rows_decode_chunks", rows_data, n_bytes, chunk_offsets, n_chunks, data_flags, n_cols, col_types, out_cols, cap_rows, nrows_out,
                       nullptr);
}

// mocktikv's indexScanExec (executor.go:191-320): tablecodec.DecodeIndexKV of every pair of an index range
TSQ_API tsq_status tsq_indexkeys_decode(tsq_ctx* ctx, const uint8_t* keys, int64_t n_bytes, const int64_t* key_offsets, int64_t n_keys, const uint8_t* values,
                                        int64_t n_value_bytes, const int64_t* value_offsets, uint32_t data_flags, int32_t n_index_cols, const int32_t* col_types,
                                        int32_t pk_status, tsq_col* out_cols, int64_t* nkeys_out) {
    if (!ctx) return TSQ_ERR_INVALID;
    if (n_index_cols < 0 || pk_status < 0 || pk_status > 2 || n_value_bytes < 0 || (n_value_bytes > 0 && !values))
        return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_indexkeys_decode: bad arguments");
    const int32_t n_cols = n_index_cols + (pk_status != 0 ? 1 : 0);
    if (col_types && pk_status != 0 && n_cols >= 1 && n_cols <= TSQ_MAX_COLS && col_types[n_cols - 1] != (pk_status == 2 ? TSQ_U64 : TSQ_I64))
        return tsq_fail(&ctx->hdr, TSQ_ERR_INVALID, "tsq_indexkeys_decode: the handle column is TSQ_I64 (PrimaryKeyIsSigned) or TSQ_U64 (PrimaryKeyIsUnsigned)");
    IdxMode im;
    im.n_key_cols = n_index_cols;
    im.pk_status = pk_status;
    im.vals = values;
    im.n_val_bytes = n_value_bytes;
    im.val_offs = value_offsets;
    return decc_decode(ctx, "tsq_indexkeys_decode", keys, n_bytes, key_offsets, n_keys, data_flags, n_cols, col_types, out_cols, n_keys, nkeys_out, &im);
}
