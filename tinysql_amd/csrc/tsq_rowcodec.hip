// tsq_rowcodec.hip — stored rows (rowcodec v2, one KV value per row) -> chunk columns on the GPU (SURVEY.md §8 f, rank 4).
//
// Replaces the per-row loop around rowcodec.ChunkDecoder.DecodeToChunk (util/rowcodec/decoder.go:158-238) that a table scan
// runs over its KV values (storage side: BytesDecoder.DecodeToBytes, decoder.go:252-322, called from mocktikv's tableScanExec,
// store/mockstore/mocktikv/executor.go:124-196, followed on the SQL side by readRowsData + DecodeOne).
//
// Unlike the coprocessor response (tsq_decode.hip) the row boundaries are known here (one KV value per row), so rows are
// independent: K15 `k_rowcodec_decode` gives every lane one row.  HBM-bound byte work, no MFMA:
//   * a workgroup takes 256 consecutive rows; their bytes are ONE contiguous span of `values`, copied into LDS with 16-byte
//     loads (consecutive lanes, consecutive addresses) — the per-row parse (header, binary search of the column id, 1/2/4/8-byte
//     value) then reads LDS bytes instead of issuing 64 scattered global loads per wave instruction;
//   * column c of row r is stored at out[c][r]: consecutive lanes write consecutive 8-byte slots;
//   * the null bitmap of 64 rows is one wave ballot, written as 8 bytes by lanes 0..7 (bit 1 = NOT NULL, column.go:89-92);
//   * a tile wider than the LDS budget (rows with long strings next to the requested columns) is parsed from global memory.
// Algorithmic bytes per row: its stored bytes + 8 B per output value (+ 1/8 B bitmap).
#include "tsq_stage.h"
#include "tsq_rowcodec_dp.h"

#define RC_NT 256
// LDS tile of one workgroup: sized per call from the average row length (dynamic shared memory) — a fixed 48 KB tile allowed
// only three workgroups (12 waves) per CU, and between its two barriers a workgroup either loads or parses
#define RC_LDS_MIN (8 * 1024)
#define RC_LDS_DEFAULT_MIN (24 * 1024)  // six workgroups per CU: measured best (0.77 ms per 2.5e7 rows; eight: 0.81, four: 1.18, three: 1.11)
#define RC_LDS_MAX (64 * 1024)

struct RcArgs {
    const uint8_t* bytes;
    const int64_t* offsets;
    const int64_t* handles;
    int64_t nrows, n_bytes;
    int32_t n_cols;
    tsq_rowcodec_col cols[TSQ_MAX_COLS];
    void* out[TSQ_MAX_COLS];
    uint8_t* out_bm[TSQ_MAX_COLS];
    unsigned long long* err;  // min over (row << 4 | code); ~0 = no error
    uint32_t lds_bytes;       // dynamic shared memory of the launch
    uint32_t fast_layout;     // 1: waves whose rows share one layout resolve the columns once (0 only through the tuning knob)
};

namespace {

// the row's bytes inside the staged tile: single bytes as LDS byte reads, anything wider as three aligned words + a funnel shift
struct RcLds {
    const uint32_t* w;  // the tile
    uint32_t base;      // where the row starts in it
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return ((const uint8_t*)w)[base + i]; }
    __device__ __forceinline__ uint64_t le(uint32_t p, uint32_t) const {
        const uint32_t q = base + p, i = q >> 2;
        return tsq_rc_funnel(w[i], w[i + 1], w[i + 2], q);  // up to 11 bytes past q: the tile has 16 bytes of slack
    }
};
// the row's bytes in global memory (tiles that do not fit the LDS budget): exactly the bytes asked for
struct RcGlobal {
    const uint8_t* p;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return p[i]; }
    __device__ __forceinline__ uint64_t le(uint32_t q, uint32_t n) const { return tsq_rc_le_bytes(*this, q, n); }
};

// the rows of one tile, one per lane: parse, decode every requested column, store value + bitmap bits, report the first error
template <class R>
__device__ __forceinline__ void rc_rows(const RcArgs& a, const R& rd, uint32_t len, bool live, bool bad_offsets, int64_t handle,
                                        int64_t r, int64_t r0, int64_t bm_bytes, uint32_t tid) {
    int code = RC_OK;
    tsq_rc_row row = {0, 0, 0, 0, 0, 0, 0};
    if (live) code = bad_offsets ? RC_MALFORMED /* offsets not non-decreasing inside [0, n_bytes] */ : tsq_rc_parse(rd, len, &row);
    for (int c = 0; c < a.n_cols; c++) {
        uint64_t bits = 0;
        bool notnull = false;
        if (live && code == RC_OK)
            code = tsq_rc_column(rd, row, a.cols[c].col_id, a.cols[c].type, a.cols[c].flags, a.cols[c].def_bits, handle, &bits, &notnull);
        if (live) {
            if (a.cols[c].type == TSQ_F32) ((uint32_t*)a.out[c])[r] = (uint32_t)bits;
            else ((uint64_t*)a.out[c])[r] = bits;
        }
        // 64 rows = one ballot = 8 bitmap bytes, written by lanes 0..7 (bit 1 = NOT NULL; rows past the end contribute zeros)
        const unsigned long long m = __ballot(notnull);
        const uint32_t lane = tid & 63u;
        const int64_t byte_at = ((r0 + (int64_t)(tid & ~63u)) >> 3) + lane;
        if (lane < 8 && byte_at < bm_bytes) a.out_bm[c][byte_at] = (uint8_t)(m >> (8 * lane));
    }
    if (live && code != RC_OK) atomicMin(a.err, ((unsigned long long)r << 4) | (unsigned long long)code);
}

__device__ __forceinline__ uint64_t rc_first_lane(uint64_t v) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}

// the staged tile: when every row of the wave has the layout of the wave's first row (same header, same ids — the usual case
// inside one table) the column search runs once on that shared signature (uniform values: scalar unit) and a lane only reads its
// offsets and values; otherwise the wave takes the general path above.  Both give the reference's result for every row.
__device__ __forceinline__ void rc_rows_lds(const RcArgs& a, const RcLds& rd, uint32_t len, bool live, bool bad_offsets, int64_t handle, int64_t r,
                                            int64_t r0, int64_t bm_bytes, uint32_t tid) {
    uint64_t hdr = 0, ids8 = 0;
    const bool cand = live && !bad_offsets && tsq_rc_signature(rd, len, &hdr, &ids8);
    const uint64_t hdr0 = rc_first_lane(cand ? hdr : 0), ids0 = rc_first_lane(cand ? ids8 : 0);
    const bool same = cand && hdr == hdr0 && ids8 == ids0;
    const bool fast = a.fast_layout && (hdr0 & 0xffu) == TSQ_RC_CODEC_VER && __ballot(same) == __ballot(live);
    if (!fast) {
        rc_rows(a, rd, len, live, bad_offsets, handle, r, r0, bm_bytes, tid);
        return;
    }
    const uint32_t nn = (uint32_t)(hdr0 >> 16) & 0xffffu, nl = (uint32_t)(hdr0 >> 32) & 0xffffu;
    uint64_t o_lo = 0, o_hi = 0;
    if (live) tsq_rc_fast_offsets(rd, nn, 6 + nn + nl, &o_lo, &o_hi);
    int code = RC_OK;
    for (int c = 0; c < a.n_cols; c++) {
        uint64_t bits = 0;
        bool notnull = false;
        if (live && code == RC_OK)
            code = tsq_rc_fast_column(rd, len, hdr0, ids0, o_lo, o_hi, a.cols[c].col_id, a.cols[c].type, a.cols[c].flags, a.cols[c].def_bits, handle, &bits,
                                      &notnull);
        if (live) {
            if (a.cols[c].type == TSQ_F32) ((uint32_t*)a.out[c])[r] = (uint32_t)bits;
            else ((uint64_t*)a.out[c])[r] = bits;
        }
        const unsigned long long m = __ballot(notnull);
        const uint32_t lane = tid & 63u;
        const int64_t byte_at = ((r0 + (int64_t)(tid & ~63u)) >> 3) + lane;
        if (lane < 8 && byte_at < bm_bytes) a.out_bm[c][byte_at] = (uint8_t)(m >> (8 * lane));
    }
    if (live && code != RC_OK) atomicMin(a.err, ((unsigned long long)r << 4) | (unsigned long long)code);
}

__global__ void __launch_bounds__(RC_NT) k_rowcodec_decode(RcArgs a) {
    extern __shared__ uint4 s_tile[];
    const uint32_t tid = threadIdx.x;
    const int64_t n_tiles = (a.nrows + RC_NT - 1) / RC_NT;
    const int64_t bm_bytes = (a.nrows + 7) / 8;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t r0 = t * RC_NT, r1 = r0 + RC_NT < a.nrows ? r0 + RC_NT : a.nrows;
        const int64_t r = r0 + tid;
        const bool live = r < r1;
        const int64_t tile_lo = a.offsets[r0], tile_hi = a.offsets[r1];
        int64_t lo = 0, hi = 0;
        if (live) {
            lo = a.offsets[r];
            hi = a.offsets[r + 1];
        }
        const tsq_rc_plan plan = tsq_rc_tile_plan((uint64_t)(uintptr_t)a.bytes, tile_lo, tile_hi, a.n_bytes, a.lds_bytes);
        if (plan.staged) {
            const uint4* src = (const uint4*)(a.bytes + plan.copy_from);  // 16-byte aligned by construction
            for (uint32_t i = tid; i < plan.n_vec; i += RC_NT) s_tile[i] = src[i];
        }
        __syncthreads();
        const bool bad_offsets = live && (lo < tile_lo || hi < lo || hi > tile_hi || tile_hi > a.n_bytes || hi - lo > 0x7fffffffLL);
        const bool ok_row = live && !bad_offsets;
        const uint32_t len = ok_row ? (uint32_t)(hi - lo) : 0u;
        const int64_t handle = (live && a.handles) ? a.handles[r] : 0;
        // plan.staged is the same for the whole workgroup: each branch is entered by complete waves (the ballots inside need
        // that), and each instantiation reads through a pointer of one address space (LDS reads, not flat ones)
        if (plan.staged) {
            RcLds rd;
            rd.w = (const uint32_t*)s_tile;
            rd.base = ok_row ? plan.skew + (uint32_t)(lo - tile_lo) : 0u;
            rc_rows_lds(a, rd, len, live, bad_offsets, handle, r, r0, bm_bytes, tid);
        } else {
            RcGlobal rd;
            rd.p = a.bytes + (ok_row ? lo : 0);
            rc_rows(a, rd, len, live, bad_offsets, handle, r, r0, bm_bytes, tid);
        }
        __syncthreads();  // the tile is overwritten by the next iteration
    }
}

// The software-pipelined form (the default): a workgroup walks its tiles t, t + G, t + 2G, ... and keeps TWO things in flight
// while it parses tile t out of LDS: the bytes of tile t + G (in registers: KV 16-byte vectors per lane, written to LDS at the
// top of the next iteration) and the row boundaries of tile t + 2G (which the loads of the iteration after need).  The plain
// kernel above pays offsets -> bytes -> parse as three dependent latencies per tile.
struct RcMeta {
    int64_t tile_lo, tile_hi, lo, hi;
};
__device__ __forceinline__ RcMeta rc_load_meta(const RcArgs& a, int64_t t, int64_t n_tiles, uint32_t tid) {
    RcMeta m = {0, 0, 0, 0};
    if (t < n_tiles) {
        const int64_t r0 = t * RC_NT, r1 = r0 + RC_NT < a.nrows ? r0 + RC_NT : a.nrows;
        m.tile_lo = a.offsets[r0];
        m.tile_hi = a.offsets[r1];
        if (r0 + tid < r1) {
            m.lo = a.offsets[r0 + tid];
            m.hi = a.offsets[r0 + tid + 1];
        }
    }
    return m;
}

template <int KV>
__global__ void __launch_bounds__(RC_NT, KV <= 6 ? 6 : 1) k_rowcodec_decode_pipe(RcArgs a) {
    extern __shared__ uint4 s_tile[];
    const uint32_t tid = threadIdx.x;
    const int64_t n_tiles = (a.nrows + RC_NT - 1) / RC_NT, G = gridDim.x;
    const int64_t bm_bytes = (a.nrows + 7) / 8;
    const tsq_rc_plan none = {0, 0, 0, 0};
    uint4 regs[KV];
    int64_t t = blockIdx.x;
    RcMeta m0 = rc_load_meta(a, t, n_tiles, tid), m1 = rc_load_meta(a, t + G, n_tiles, tid);
    tsq_rc_plan p0 = t < n_tiles ? tsq_rc_tile_plan((uint64_t)(uintptr_t)a.bytes, m0.tile_lo, m0.tile_hi, a.n_bytes, a.lds_bytes) : none;
#pragma unroll
    for (int i = 0; i < KV; i++) regs[i] = make_uint4(0, 0, 0, 0);
    if (p0.staged) {
        const uint4* src = (const uint4*)(a.bytes + p0.copy_from);
        const uint32_t last = p0.n_vec ? p0.n_vec - 1 : 0u;  // lanes past the end re-read the last vector (never stored)
#pragma unroll
        for (int i = 0; i < KV; i++) regs[i] = src[min((uint32_t)i * RC_NT + tid, last)];
    }
    while (t < n_tiles) {
        if (p0.staged) {
#pragma unroll
            for (int i = 0; i < KV; i++)
                if ((uint32_t)i * RC_NT + tid < p0.n_vec) s_tile[(uint32_t)i * RC_NT + tid] = regs[i];
        }
        __syncthreads();
        // in flight during the parse: boundaries of tile t + 2G, bytes of tile t + G
        const RcMeta m2 = rc_load_meta(a, t + 2 * G, n_tiles, tid);
        const tsq_rc_plan p1 = t + G < n_tiles ? tsq_rc_tile_plan((uint64_t)(uintptr_t)a.bytes, m1.tile_lo, m1.tile_hi, a.n_bytes, a.lds_bytes) : none;
        if (p1.staged) {
            const uint4* src = (const uint4*)(a.bytes + p1.copy_from);
            const uint32_t last = p1.n_vec ? p1.n_vec - 1 : 0u;
#pragma unroll
            for (int i = 0; i < KV; i++) regs[i] = src[min((uint32_t)i * RC_NT + tid, last)];
        }
        const int64_t r0 = t * RC_NT, r1 = r0 + RC_NT < a.nrows ? r0 + RC_NT : a.nrows;
        const int64_t r = r0 + tid;
        const bool live = r < r1;
        const int64_t lo = m0.lo, hi = m0.hi, tile_lo = m0.tile_lo, tile_hi = m0.tile_hi;
        const bool bad_offsets = live && (lo < tile_lo || hi < lo || hi > tile_hi || tile_hi > a.n_bytes || hi - lo > 0x7fffffffLL);
        const bool ok_row = live && !bad_offsets;
        const uint32_t len = ok_row ? (uint32_t)(hi - lo) : 0u;
        const int64_t handle = (live && a.handles) ? a.handles[r] : 0;
        if (p0.staged) {
            RcLds rd;
            rd.w = (const uint32_t*)s_tile;
            rd.base = ok_row ? p0.skew + (uint32_t)(lo - tile_lo) : 0u;
            rc_rows_lds(a, rd, len, live, bad_offsets, handle, r, r0, bm_bytes, tid);
        } else {
            RcGlobal rd;
            rd.p = a.bytes + (ok_row ? lo : 0);
            rc_rows(a, rd, len, live, bad_offsets, handle, r, r0, bm_bytes, tid);
        }
        __syncthreads();  // every wave is done with the tile before the next one is written over it
        m0 = m1;
        m1 = m2;
        p0 = p1;
        t += G;
    }
}

// ---- var-len (string / blob) columns: chk.AppendBytes(colIdx, colData) (decoder.go:226-228).  The decode kernels leave a
// REFERENCE per row in place of the cell — (start inside the row) << 32 | length, 0 for a NULL — because where a cell's bytes go
// depends on the lengths of all the cells before it.  K15b turns the references into lengths, tsq_launch_scan64 into the column's
// offsets, K15c copies the bytes out of `values` (one row per lane for short cells, one per wave for long ones).
struct RcVarArgs {
    const uint8_t* bytes;     // values
    const int64_t* offsets;   // row boundaries in values
    const uint64_t* ref;      // [rows]
    int64_t rows;
    int64_t* out_offs;        // [rows + 1]: lengths, then (after the scan) the offsets of the column
    uint8_t* out_data;
    const uint8_t* def_pool;  // default strings of the scan's columns: a reference with bit 63 set points here (bits 62..32: where)
};
__global__ void __launch_bounds__(256) k_rowcodec_var_len(RcVarArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.rows; r += (int64_t)gridDim.x * blockDim.x)
        a.out_offs[r] = (int64_t)(uint32_t)a.ref[r];
}
// A TypeBit column (decoder.go:229-231): the value is stored as an unsigned int; its cell is the last byteSize = (Flen + 7) / 8 bytes
// of the value in BIG endian (types.NewBinaryLiteralFromUint, types/binary_literal.go:57-69).  The row kernel decodes such a column
// as a TSQ_U64 column into a scratch array; here the cells' lengths (byteSize, 0 for NULL) and, after the scan, their bytes.
struct RcBitArgs {
    const uint64_t* vals;    // [rows] the decoded unsigned values
    const uint8_t* bitmap;   // packed NOT NULL bits of the column
    int64_t rows;
    int32_t byte_size;       // 1..8
    int64_t* out_offs;       // [rows + 1]
    uint8_t* out_data;
};
__global__ void __launch_bounds__(256) k_rowcodec_bit_len(RcBitArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.rows; r += (int64_t)gridDim.x * blockDim.x)
        a.out_offs[r] = ((a.bitmap[r >> 3] >> (r & 7)) & 1) ? a.byte_size : 0;
}
__global__ void __launch_bounds__(256) k_rowcodec_bit_cells(RcBitArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.rows; r += (int64_t)gridDim.x * blockDim.x) {
        if (a.out_offs[r + 1] == a.out_offs[r]) continue;
        const uint64_t v = a.vals[r];
        uint8_t* d = a.out_data + a.out_offs[r];
        for (int i = 0; i < a.byte_size; i++) d[i] = (uint8_t)(v >> (8 * (a.byte_size - 1 - i)));  // buf[8 - byteSize:] of BigEndian.PutUint64
    }
}

template <bool WAVE>
__global__ void __launch_bounds__(256) k_rowcodec_var_copy(RcVarArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = WAVE ? gtid >> 6 : gtid, step = WAVE ? nthr >> 6 : nthr;
    for (int64_t r = first; r < a.rows; r += step) {
        const uint64_t ref = a.ref[r];
        const int64_t n = (int64_t)(uint32_t)ref;
        if (n == 0) continue;
        const uint8_t* s = (ref >> 63) ? a.def_pool + (int64_t)((ref >> 32) & 0x7fffffffu) : a.bytes + a.offsets[r] + (int64_t)(ref >> 32);
        uint8_t* d = a.out_data + a.out_offs[r];
        if (!WAVE) {
            tsq_copy_cell(d, s, n);
        } else {  // head up to an 8-byte boundary of the destination, then 8 bytes per lane, then the tail
            int64_t head = (8 - ((uintptr_t)d & 7)) & 7;
            head = head < n ? head : n;
            if (lane < head) d[lane] = s[lane];
            const int64_t words = (n - head) >> 3;
            for (int64_t w = lane; w < words; w += 64) {
                uint64_t x;
                memcpy(&x, s + head + w * 8, 8);  // the source is not aligned with the destination
                *reinterpret_cast<uint64_t*>(d + head + w * 8) = x;
            }
            const int64_t done = head + words * 8;
            if (done + lane < n) d[done + lane] = s[done + lane];
        }
    }
}

}  // namespace

// ====================================================================== host side
TSQ_API tsq_status tsq_rowcodec_decode(tsq_ctx* ctx, const uint8_t* values, int64_t n_bytes, const int64_t* offsets, const int64_t* handles,
                                       int64_t nrows, uint32_t data_flags, int32_t n_cols, const tsq_rowcodec_col* cols, tsq_col* out_cols,
                                       int64_t* nrows_out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (nrows_out) *nrows_out = 0;
    if (!nrows_out || !cols || !out_cols || nrows < 0 || n_bytes < 0 || (nrows > 0 && (!offsets || (n_bytes > 0 && !values))))
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowcodec_decode: bad arguments");
    if (n_cols < 1 || n_cols > TSQ_MAX_COLS) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "1..16 columns supported");
    if (nrows >= (1LL << 40)) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "more than 2^40 rows per call");
    bool any_handle = false, any_var = false;
    for (int c = 0; c < n_cols; c++) {
        if (cols[c].type < TSQ_I64 || cols[c].type > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowcodec_decode: unknown column type");
        if ((cols[c].flags & TSQ_RC_HANDLE) && cols[c].type != TSQ_I64 && cols[c].type != TSQ_U64)
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowcodec_decode: the handle column is an integer column");
        if (cols[c].type == TSQ_BYTES && (cols[c].flags & TSQ_RC_HAS_DEFAULT) && (cols[c].def_len < 0 || cols[c].def_len > 0x7fffffffLL || (cols[c].def_len > 0 && !cols[c].def_bytes)))
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowcodec_decode: a var-len column with a default needs def_bytes / def_len");
        if ((cols[c].flags & TSQ_RC_BIT) && (cols[c].type != TSQ_BYTES || TSQ_RC_BIT_SIZE(cols[c].flags) < 1 || TSQ_RC_BIT_SIZE(cols[c].flags) > 8 ||
                                             ((cols[c].flags & TSQ_RC_HAS_DEFAULT) && cols[c].def_len != (int64_t)TSQ_RC_BIT_SIZE(cols[c].flags))))
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowcodec_decode: a bit column is a TSQ_BYTES column of 1..8 bytes (its default: a literal of that size)");
        any_handle = any_handle || (cols[c].flags & TSQ_RC_HANDLE);
        any_var = any_var || cols[c].type == TSQ_BYTES;
        // a var-len output column: offsets[nrows + 1] and room for n_bytes data bytes (a cell is a piece of its row)
        const bool var = cols[c].type == TSQ_BYTES;
        if (!out_cols[c].null_bitmap || (var ? (!out_cols[c].offsets || (n_bytes > 0 && !out_cols[c].data)) : !out_cols[c].data))
            return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowcodec_decode: out columns need data and null_bitmap buffers (a var-len column: offsets too)");
        if (((out_cols[c].flags ^ out_cols[0].flags) & TSQ_COL_DEVICE) != 0) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowcodec_decode: mixed host/device outputs");
    }
    if (nrows == 0) {  // an empty scan: a var-len column still has its first offset
        for (int c = 0; c < n_cols; c++) {
            if (cols[c].type != TSQ_BYTES) continue;
            if (out_cols[c].flags & TSQ_COL_DEVICE) TSQ_HIP(h, hipMemsetAsync(out_cols[c].offsets, 0, 8, ctx->stream));
            else out_cols[c].offsets[0] = 0;
        }
        return TSQ_OK;
    }
    if (any_handle && !handles) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rowcodec_decode: a handle column needs handles[]");
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const bool in_dev = data_flags & TSQ_COL_DEVICE, out_dev = out_cols[0].flags & TSQ_COL_DEVICE;
    RcArgs a;
    memset(&a, 0, sizeof a);
    a.nrows = nrows;
    a.n_bytes = n_bytes;
    a.n_cols = n_cols;
    DevBuf dbytes, doffs, dhandles, derr, scratch, ddef, ddata[TSQ_MAX_COLS], dbm[TSQ_MAX_COLS], dref[TSQ_MAX_COLS], dvoffs[TSQ_MAX_COLS];
    auto release_all = [&]() {
        for (DevBuf* b : {&dbytes, &doffs, &dhandles, &derr, &scratch, &ddef}) b->release();
        for (int c = 0; c < TSQ_MAX_COLS; c++) { ddata[c].release(); dbm[c].release(); dref[c].release(); dvoffs[c].release(); }
    };
    auto fail = [&](tsq_status st) { release_all(); return st; };
    tsq_status s = derr.reserve(ctx, h, 64);
    hipError_t e = hipSuccess;
    if (s == TSQ_OK && !in_dev) {
        // host rows: one H2D copy per array (the caller has gathered the KV values of a scan batch into one buffer)
        s = dbytes.reserve(ctx, h, (size_t)n_bytes + 64);
        if (s == TSQ_OK) s = doffs.reserve(ctx, h, ((size_t)nrows + 1) * 8 + 64);
        if (s == TSQ_OK && any_handle) s = dhandles.reserve(ctx, h, (size_t)nrows * 8 + 64);
        if (s == TSQ_OK && n_bytes > 0) e = hipMemcpyAsync(dbytes.p, values, (size_t)n_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (s == TSQ_OK && e == hipSuccess) e = hipMemcpyAsync(doffs.p, offsets, ((size_t)nrows + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
        if (s == TSQ_OK && e == hipSuccess && any_handle) e = hipMemcpyAsync(dhandles.p, handles, (size_t)nrows * 8, hipMemcpyHostToDevice, ctx->stream);
        a.bytes = dbytes.as<uint8_t>();
        a.offsets = doffs.as<int64_t>();
        a.handles = any_handle ? dhandles.as<int64_t>() : nullptr;
    } else {
        a.bytes = values;
        a.offsets = offsets;
        a.handles = any_handle ? handles : nullptr;
    }
    if (s != TSQ_OK) return fail(s);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode(H2D): ") + hipGetErrorString(e)));
    // default strings (defDatum of a var-len column, decoder.go:186-194): one small pool in HBM; the column's def_bits becomes a cell
    // reference into it — (1 << 63) | where << 32 | length — which the kernel hands out like any default (tsq_rc_column)
    std::string pool;
    int64_t def_len[TSQ_MAX_COLS] = {0};
    for (int c = 0; c < n_cols; c++) {
        a.cols[c] = cols[c];
        a.cols[c].def_bytes = nullptr;
        if (cols[c].flags & TSQ_RC_BIT) {  // decoded as the unsigned int it is stored as; its cells are made afterwards (k_rowcodec_bit_*)
            a.cols[c].type = TSQ_U64;
            uint64_t v = 0;
            if (cols[c].flags & TSQ_RC_HAS_DEFAULT)
                for (int64_t i = 0; i < cols[c].def_len; i++) v = (v << 8) | cols[c].def_bytes[i];  // the literal is big endian
            a.cols[c].def_bits = v;
            continue;
        }
        if (cols[c].type != TSQ_BYTES || !(cols[c].flags & TSQ_RC_HAS_DEFAULT)) continue;
        a.cols[c].def_bits = (1ull << 63) | ((uint64_t)pool.size() << 32) | (uint64_t)cols[c].def_len;
        def_len[c] = cols[c].def_len;
        pool.append((const char*)cols[c].def_bytes, (size_t)cols[c].def_len);
    }
    if (pool.size() >= (1ull << 31)) return fail(tsq_fail(h, TSQ_ERR_UNSUPPORTED, "tsq_rowcodec_decode: default strings of 2 GB"));
    if (!pool.empty()) {
        s = ddef.reserve(ctx, h, pool.size() + 64);
        if (s != TSQ_OK) return fail(s);
        e = hipMemcpyAsync(ddef.p, pool.data(), pool.size(), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (pool is a local)
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode(H2D): ") + hipGetErrorString(e)));
    }
    for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
        const bool var = cols[c].type == TSQ_BYTES;
        if (var) s = dref[c].reserve(ctx, h, (size_t)nrows * 8 + 64);  // the kernel leaves (start, length) references here (a bit column: its values)
        if (s == TSQ_OK && !out_dev) {
            s = ddata[c].reserve(ctx, h, (var ? (size_t)n_bytes + (size_t)nrows * (size_t)def_len[c] : (size_t)nrows * tsq_elem_size(cols[c].type)) + 64);
            if (s == TSQ_OK) s = dbm[c].reserve(ctx, h, tsq_bitmap_bytes(nrows) + 64);
            if (s == TSQ_OK && var) s = dvoffs[c].reserve(ctx, h, ((size_t)nrows + 1) * 8 + 64);
        }
        a.out[c] = var ? dref[c].p : (out_dev ? out_cols[c].data : ddata[c].p);
        a.out_bm[c] = out_dev ? out_cols[c].null_bitmap : dbm[c].as<uint8_t>();
    }
    if (s != TSQ_OK) return fail(s);
    a.err = derr.as<unsigned long long>();
    e = hipMemsetAsync(a.err, 0xff, 8, ctx->stream);
    if (e == hipSuccess) {
        const int64_t n_tiles = (nrows + RC_NT - 1) / RC_NT;
        // tile = 256 average rows + 25 % + the alignment skew, in 4 KB steps; a tile that does not fit (rows far above the
        // average) is parsed from global memory by its workgroup
        int64_t want = (n_bytes / nrows) * RC_NT;
        want = std::max<int64_t>(((want + want / 4 + 512 + 4095) / 4096) * 4096, RC_LDS_DEFAULT_MIN);
        if (ctx->knob[TSQ_KNOB_ROWCODEC_LDS_KB] != TSQ_KNOB_DEFAULT) want = ctx->knob[TSQ_KNOB_ROWCODEC_LDS_KB] * 1024;  // tuning knobs for tools/bench_rowcodec.py
        a.fast_layout = tsq_knob(ctx, TSQ_KNOB_ROWCODEC_FAST_LAYOUT, 1) == 0 ? 0u : 1u;
        a.lds_bytes = (uint32_t)std::min<int64_t>(std::max<int64_t>(want, RC_LDS_MIN), RC_LDS_MAX);
        const int64_t wg_per_cu = std::min<int64_t>(8, (160 * 1024) / a.lds_bytes);  // 160 KB of LDS and 32 waves per CU
        const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)ctx->num_cus * wg_per_cu);
        if (tsq_knob(ctx, TSQ_KNOB_ROWCODEC_PIPELINE, 1) == 0) {
            hipLaunchKernelGGL(k_rowcodec_decode, dim3(grid), dim3(RC_NT), a.lds_bytes + 16, ctx->stream, a);  // + slack for the word reads
        } else if (a.lds_bytes <= 24 * 1024) {  // KV = 16-byte vectors a lane may hold for the next tile = tile bytes / (16 * 256)
            hipLaunchKernelGGL(k_rowcodec_decode_pipe<6>, dim3(grid), dim3(RC_NT), a.lds_bytes + 16, ctx->stream, a);
        } else if (a.lds_bytes <= 32 * 1024) {
            hipLaunchKernelGGL(k_rowcodec_decode_pipe<8>, dim3(grid), dim3(RC_NT), a.lds_bytes + 16, ctx->stream, a);
        } else {
            hipLaunchKernelGGL(k_rowcodec_decode_pipe<16>, dim3(grid), dim3(RC_NT), a.lds_bytes + 16, ctx->stream, a);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned, a.err, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode: ") + hipGetErrorString(e)));
    const uint64_t errw = ctx->pinned[0];
    int code = RC_OK;
    int64_t rows = nrows;
    if (errw != ~0ull) {
        code = (int)(errw & 15);
        rows = (int64_t)(errw >> 4);
    }
    // var-len columns of the rows before the first offending one: references -> lengths -> offsets (scan) -> bytes
    int64_t var_bytes[TSQ_MAX_COLS] = {0};
    for (int c = 0; c < n_cols && any_var; c++) {
        if (cols[c].type != TSQ_BYTES) continue;
        RcVarArgs va;
        va.bytes = a.bytes;
        va.offsets = a.offsets;
        va.ref = dref[c].as<uint64_t>();
        va.rows = rows;
        va.out_offs = out_dev ? out_cols[c].offsets : dvoffs[c].as<int64_t>();
        va.out_data = out_dev ? (uint8_t*)out_cols[c].data : ddata[c].as<uint8_t>();
        va.def_pool = ddef.as<uint8_t>();
        if (cols[c].flags & TSQ_RC_BIT) {
            RcBitArgs ba;
            ba.vals = dref[c].as<uint64_t>();
            ba.bitmap = a.out_bm[c];
            ba.rows = rows;
            ba.byte_size = (int32_t)TSQ_RC_BIT_SIZE(cols[c].flags);
            ba.out_offs = va.out_offs;
            ba.out_data = va.out_data;
            if (rows > 0) hipLaunchKernelGGL(k_rowcodec_bit_len, dim3(tsq_grid_for(ctx, rows, 256)), dim3(256), 0, ctx->stream, ba);
            tsq_status bs = tsq_launch_scan64(ctx, h, ba.out_offs, rows, scratch);
            if (bs != TSQ_OK) return fail(bs);
            if (rows > 0) hipLaunchKernelGGL(k_rowcodec_bit_cells, dim3(tsq_grid_for(ctx, rows, 256)), dim3(256), 0, ctx->stream, ba);
            e = hipMemcpyAsync(ctx->pinned + 1, ba.out_offs + rows, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode(bit cells): ") + hipGetErrorString(e)));
            var_bytes[c] = (int64_t)ctx->pinned[1];
            continue;
        }
        if (rows > 0) {
            hipLaunchKernelGGL(k_rowcodec_var_len, dim3(tsq_grid_for(ctx, rows, 256)), dim3(256), 0, ctx->stream, va);
            e = hipGetLastError();
            if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode(var len): ") + hipGetErrorString(e)));
        }
        const tsq_status vs = tsq_launch_scan64(ctx, h, va.out_offs, rows, scratch);
        if (vs != TSQ_OK) return fail(vs);
        e = hipMemcpyAsync(ctx->pinned + 1, va.out_offs + rows, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode(var scan): ") + hipGetErrorString(e)));
        var_bytes[c] = (int64_t)ctx->pinned[1];
        if (var_bytes[c] > 0) {
            if (var_bytes[c] / rows > 32) hipLaunchKernelGGL(k_rowcodec_var_copy<true>, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, va);
            else hipLaunchKernelGGL(k_rowcodec_var_copy<false>, dim3(tsq_grid_for(ctx, rows, 256)), dim3(256), 0, ctx->stream, va);
            e = hipGetLastError();
            if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode(var copy): ") + hipGetErrorString(e)));
        }
    }
    if (any_var) {
        e = hipStreamSynchronize(ctx->stream);  // the references and `values` staging are released below
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode(var): ") + hipGetErrorString(e)));
    }
    // hand the rows before the first offending one over (the reference has appended them to the chunk by then)
    if (!out_dev && (rows > 0 || any_var)) {
        for (int c = 0; c < n_cols && e == hipSuccess; c++) {
            if (cols[c].type == TSQ_BYTES) {
                e = hipMemcpyAsync(out_cols[c].offsets, dvoffs[c].p, ((size_t)rows + 1) * 8, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess && var_bytes[c] > 0) e = hipMemcpyAsync(out_cols[c].data, ddata[c].p, (size_t)var_bytes[c], hipMemcpyDeviceToHost, ctx->stream);
            } else if (rows > 0) {
                e = hipMemcpyAsync(out_cols[c].data, ddata[c].p, (size_t)rows * tsq_elem_size(cols[c].type), hipMemcpyDeviceToHost, ctx->stream);
            }
            if (e == hipSuccess && rows > 0) e = hipMemcpyAsync(out_cols[c].null_bitmap, dbm[c].p, tsq_bitmap_bytes(rows), hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rowcodec_decode(D2H): ") + hipGetErrorString(e)));
    }
    for (int c = 0; c < n_cols; c++) {
        out_cols[c].length = rows;
        out_cols[c].type = cols[c].type;
        out_cols[c].elem_size = cols[c].type == TSQ_BYTES ? -1 : tsq_elem_size(cols[c].type);
    }
    release_all();
    *nrows_out = rows;
    switch (code) {
        case RC_OK: return TSQ_OK;
        case RC_BAD_VERSION: return tsq_fail(h, TSQ_ERR_INVALID, "invalid codec version");                 // row.go:54-56
        case RC_SHORT_FLOAT: return tsq_fail(h, TSQ_ERR_INVALID, "insufficient bytes to decode value");  // number.go:84-86 via float.go:42-46
        default: return tsq_fail(h, TSQ_ERR_INVALID, "malformed row");  // the reference panics (index / slice bounds out of range)
    }
}
