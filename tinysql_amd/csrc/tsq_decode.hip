// tsq_decode.hip — coprocessor-response rows -> columns on the GPU (gfx950).  SURVEY.md §8 (f) rank 2.
//
// Replaces selectResult.readRowsData (distsql/select_result.go:139-155) + codec.Decoder.DecodeOne
// (util/codec/codec.go:623-690): the response is ONE byte string, row after row, every value a flag byte followed by a
// varint (flag 8/9, number.go:107-130 over Go's encoding/binary), 8 big-endian bytes (flag 3/4/5: comparable int / uint /
// memcomparable float, number.go:24-90, float.go:22-46) or nothing (flag 0 = NULL).  There are no row or value
// boundaries in the stream, so the reference decodes it with one sequential loop; where a value starts depends on every
// value before it.
//
// Parallel formulation (speculative parsing): a value is at most 11 bytes long, so a block of bytes can be entered at
// only 11 different offsets.  For every 64-byte sub-block and each of the 11 entry offsets we compute the exit offset
// into the next sub-block and the number of values that start inside (an "exit map", 11 x 4 bits, and 11 counts).
// Maps compose associatively, so
//   K13a k_dec_map   : tile (16 KB in LDS): value length at every byte position (as if a value started there, all lanes
//                      busy), 11 cheap pointer-chasing walks per sub-block, composition of the 256 sub-block maps ->
//                      tile map + tile counts
//   K13b k_dec_scan  : one workgroup composes the tile maps -> true entry offset and first value ordinal of every tile
//   K13c k_dec_emit  : per tile again: lengths, sub-block maps, composition from the now known entry -> every sub-block
//                      walks its true path, decodes its ~12 values and stores them at (row, column) = divmod(ordinal)
// Errors follow the reference in stream order: the FIRST offending value decides (invalid flag, value cut by the end of
// the buffer, varint longer than 64 bits, a row that ends early); atomicMin over (ordinal, code) finds it.
// Algorithmic bytes: encoded bytes read once + 8 B written per value (the implementation reads the bytes twice).
#include "tsq_stage.h"

#define TSQ_DEC_TB 16384      // bytes per tile
#define TSQ_DEC_SB 64         // bytes per sub-block
#define TSQ_DEC_NSB (TSQ_DEC_TB / TSQ_DEC_SB)
#define TSQ_DEC_HALO 16
#define TSQ_DEC_NT 256        // threads per workgroup = sub-blocks per tile

enum { DEC_OK = 0, DEC_ROW_CUT = 1, DEC_INSUFFICIENT = 2, DEC_OVERFLOW = 3, DEC_BAD_FLAG = 4, DEC_VARLEN = 5 };

struct DecArgs {
    const uint8_t* data;
    int64_t n_bytes;
    int64_t n_tiles;
    unsigned long long* tile_map;   // [n_tiles] 11 x 4-bit exit offsets
    uint16_t* tile_cnt;             // [n_tiles][12] values starting in the tile, per entry offset
    uint8_t* tile_entry;            // [n_tiles] true entry offset (k_dec_scan)
    unsigned long long* tile_base;  // [n_tiles + 1] ordinal of the first value starting in the tile; [n_tiles] = total
    // emit
    int32_t n_cols;
    int32_t col_type[TSQ_MAX_COLS];
    void* out_data[TSQ_MAX_COLS];
    uint8_t* out_notnull[TSQ_MAX_COLS];  // one byte per row
    int64_t cap_rows;
    unsigned long long* result;  // [0] = min over errors of (ordinal << 4 | code), [1] = byte offset of value number cap_rows * n_cols
};

// length of the value that would start at LDS position p (flag + payload).  A varint has at most 10 bytes; one whose
// 10th byte still has the continuation bit is an overflow (binary.Uvarint) and is given the maximal length 11 as well.
__device__ __forceinline__ uint32_t dec_len_at(const uint8_t* s, uint32_t p) {
    const uint8_t f = s[p];
    if (f == 3 || f == 4 || f == 5) return 9;
    if (f == 8 || f == 9) {
        uint32_t k = 1;
        while (k < 10 && (s[p + k] & 0x80)) k++;
        return k + 1;
    }
    return 1;  // NULL, or a flag that is an error if this position is ever reached on the true path
}

// loads tile `t` (+halo, zero padded past n_bytes) into s_bytes and fills s_len; returns the number of valid bytes
__device__ __forceinline__ uint32_t dec_load_tile(const DecArgs& a, int64_t t, uint8_t* s_bytes, uint8_t* s_len) {
    const int64_t t0 = t * TSQ_DEC_TB;
    const int64_t left = a.n_bytes - t0;
    const uint32_t valid = left < TSQ_DEC_TB ? (uint32_t)left : (uint32_t)TSQ_DEC_TB;
    const uint32_t avail = left < TSQ_DEC_TB + TSQ_DEC_HALO ? (uint32_t)left : (uint32_t)(TSQ_DEC_TB + TSQ_DEC_HALO);
    // 16-byte loads where the source allows it (tiles start at multiples of 16 KB of a 16-byte aligned buffer)
    const bool aligned = (((uintptr_t)a.data) & 15) == 0;
    for (uint32_t i = threadIdx.x * 16; i < TSQ_DEC_TB + TSQ_DEC_HALO; i += TSQ_DEC_NT * 16) {
        if (aligned && i + 16 <= avail) {
            *(uint4*)(s_bytes + i) = *(const uint4*)(a.data + t0 + i);
        } else {
            for (uint32_t j = 0; j < 16; j++) s_bytes[i + j] = i + j < avail ? a.data[t0 + i + j] : (uint8_t)0;
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < TSQ_DEC_TB; p += TSQ_DEC_NT) s_len[p] = (uint8_t)dec_len_at(s_bytes, p);
    __syncthreads();
    return valid;
}

// exit map and counts of sub-block `sb` (one thread): 11 walks over the precomputed lengths, advanced in lockstep so that
// the 11 dependent LDS reads of a round are in flight together.  Counts are packed four per word (cnt_out[3]).
__device__ __forceinline__ void dec_subblock_map(const uint8_t* s_len, uint32_t sb, uint32_t valid, unsigned long long* map_out, uint32_t* cnt_out) {
    const uint32_t lo = sb * TSQ_DEC_SB, end = lo + TSQ_DEC_SB;
    const uint32_t lim = end < valid ? end : valid;
    uint32_t pos[11], c[11];
#pragma unroll
    for (int e = 0; e < 11; e++) { pos[e] = lo + e; c[e] = 0; }
    bool any = lo < lim;
    while (any) {
        any = false;
#pragma unroll
        for (int e = 0; e < 11; e++) {
            if (pos[e] < lim) {
                pos[e] += s_len[pos[e]];
                c[e]++;
                any |= pos[e] < lim;
            }
        }
    }
    unsigned long long m = 0;
    uint32_t w[3] = {0, 0, 0};
#pragma unroll
    for (int e = 0; e < 11; e++) {
        const uint32_t ex = pos[e] >= end ? pos[e] - end : 0u;  // < 11
        m |= (unsigned long long)ex << (4 * e);
        w[e >> 2] |= c[e] << (8 * (e & 3));
    }
    *map_out = m;
    cnt_out[0] = w[0];
    cnt_out[1] = w[1];
    cnt_out[2] = w[2];
}
// count of entry offset `state` out of the packed words (no data-dependent address: the loads do not wait for `state`)
__device__ __forceinline__ uint32_t dec_cnt_of(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t state) {
    const uint32_t w = state < 4 ? w0 : (state < 8 ? w1 : w2);
    return (w >> (8 * (state & 3))) & 255u;
}

__global__ void __launch_bounds__(TSQ_DEC_NT) k_dec_map(DecArgs a) {
    __shared__ __align__(16) uint8_t s_bytes[TSQ_DEC_TB + TSQ_DEC_HALO];
    __shared__ uint8_t s_len[TSQ_DEC_TB];
    __shared__ unsigned long long s_map[TSQ_DEC_NSB];
    __shared__ uint32_t s_cnt[TSQ_DEC_NSB][3];
    __shared__ uint32_t s_exit[11];
    for (int64_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        const uint32_t valid = dec_load_tile(a, t, s_bytes, s_len);
        dec_subblock_map(s_len, threadIdx.x, valid, &s_map[threadIdx.x], s_cnt[threadIdx.x]);
        __syncthreads();
        if (threadIdx.x < 11) {  // lane e follows entry offset e through the 256 sub-blocks (ALU-only dependency chain)
            uint32_t state = threadIdx.x, total = 0;
            for (uint32_t i = 0; i < TSQ_DEC_NSB; i++) {
                total += dec_cnt_of(s_cnt[i][0], s_cnt[i][1], s_cnt[i][2], state);
                state = (uint32_t)(s_map[i] >> (4 * state)) & 15u;
            }
            s_exit[threadIdx.x] = state;
            a.tile_cnt[t * 12 + threadIdx.x] = (uint16_t)total;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long m = 0;
            for (int e = 0; e < 11; e++) m |= (unsigned long long)s_exit[e] << (4 * e);
            a.tile_map[t] = m;
        }
        __syncthreads();
    }
}

// one workgroup: thread i owns a contiguous run of tiles; run maps are composed through LDS
__global__ void __launch_bounds__(1024) k_dec_scan(DecArgs a) {
    __shared__ unsigned long long s_map[1024];
    __shared__ unsigned long long s_cnt[1024][11];
    __shared__ uint8_t s_entry[1024];
    __shared__ unsigned long long s_base[1025];
    const int64_t per = (a.n_tiles + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * per;
    const int64_t hi = lo + per < a.n_tiles ? lo + per : a.n_tiles;
    {
        uint32_t st[11];
        unsigned long long cn[11];
        for (int e = 0; e < 11; e++) { st[e] = e; cn[e] = 0; }
        for (int64_t t = lo; t < hi; t++) {
            const unsigned long long m = a.tile_map[t];
            const uint16_t* c = a.tile_cnt + t * 12;
#pragma unroll
            for (int e = 0; e < 11; e++) {
                uint32_t cc = 0;  // c[st[e]] without a data-dependent address
#pragma unroll
                for (int q = 0; q < 11; q++) cc = st[e] == (uint32_t)q ? c[q] : cc;
                cn[e] += cc;
                st[e] = (uint32_t)(m >> (4 * st[e])) & 15u;
            }
        }
        unsigned long long m = 0;
        for (int e = 0; e < 11; e++) { m |= (unsigned long long)st[e] << (4 * e); s_cnt[threadIdx.x][e] = cn[e]; }
        s_map[threadIdx.x] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t state = 0;
        unsigned long long base = 0;
        for (int i = 0; i < 1024; i++) {
            s_entry[i] = (uint8_t)state;
            s_base[i] = base;
            base += s_cnt[i][state];
            state = (uint32_t)(s_map[i] >> (4 * state)) & 15u;
        }
        s_base[1024] = base;
        a.tile_base[a.n_tiles] = base;
    }
    __syncthreads();
    uint32_t state = s_entry[threadIdx.x];
    unsigned long long base = s_base[threadIdx.x];
    for (int64_t t = lo; t < hi; t++) {
        a.tile_entry[t] = (uint8_t)state;
        a.tile_base[t] = base;
        base += a.tile_cnt[t * 12 + state];
        state = (uint32_t)(a.tile_map[t] >> (4 * state)) & 15u;
    }
}

__device__ __forceinline__ void dec_error(const DecArgs& a, unsigned long long ordinal, int code) {
    atomicMin(&a.result[0], (ordinal << 4) | (unsigned long long)code);
}

__global__ void __launch_bounds__(TSQ_DEC_NT) k_dec_emit(DecArgs a) {
    __shared__ __align__(16) uint8_t s_bytes[TSQ_DEC_TB + TSQ_DEC_HALO];
    __shared__ uint8_t s_len[TSQ_DEC_TB];
    __shared__ unsigned long long s_map[TSQ_DEC_NSB];
    __shared__ uint32_t s_cnt[TSQ_DEC_NSB][3];
    __shared__ uint8_t s_entry[TSQ_DEC_NSB];
    __shared__ uint32_t s_base[TSQ_DEC_NSB];
    const unsigned long long limit = (unsigned long long)a.cap_rows * (unsigned long long)a.n_cols;  // values wanted
    const uint32_t ncols = (uint32_t)a.n_cols;
    for (int64_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        const unsigned long long tile_base = a.tile_base[t];
        if (tile_base > limit) continue;  // block-uniform: everything in this tile lies beyond cap_rows
        const uint32_t valid = dec_load_tile(a, t, s_bytes, s_len);
        dec_subblock_map(s_len, threadIdx.x, valid, &s_map[threadIdx.x], s_cnt[threadIdx.x]);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t state = a.tile_entry[t], base = 0;
            for (uint32_t i = 0; i < TSQ_DEC_NSB; i++) {
                s_entry[i] = (uint8_t)state;
                s_base[i] = base;
                base += dec_cnt_of(s_cnt[i][0], s_cnt[i][1], s_cnt[i][2], state);
                state = (uint32_t)(s_map[i] >> (4 * state)) & 15u;
            }
        }
        __syncthreads();
        {
            const uint32_t lo = threadIdx.x * TSQ_DEC_SB, end = lo + TSQ_DEC_SB;
            const uint32_t lim = end < valid ? end : valid;
            uint32_t pos = lo + s_entry[threadIdx.x];
            unsigned long long ord = tile_base + s_base[threadIdx.x];
            const int64_t t0 = t * TSQ_DEC_TB;
            while (pos < lim) {
                const uint32_t len = s_len[pos];
                if (ord == limit) a.result[1] = (unsigned long long)(t0 + pos);  // first byte that is not consumed
                if (ord < limit) {
                    const uint8_t f = s_bytes[pos];
                    uint64_t bits = 0;
                    bool isnull = false, real = false;
                    int err = DEC_OK;
                    if (t0 + pos + len > a.n_bytes) {
                        err = DEC_INSUFFICIENT;  // the value is cut by the end of the buffer (number.go:45,115-123)
                    } else if (f == 3 || f == 4 || f == 5) {
                        uint64_t u = 0;
#pragma unroll
                        for (int i = 0; i < 8; i++) u = (u << 8) | s_bytes[pos + 1 + i];
                        if (f == 3) bits = u ^ 0x8000000000000000ULL;  // DecodeCmpUintToInt (number.go:29-31)
                        else if (f == 4) bits = u;
                        else {  // decodeCmpUintToFloat (float.go:32-40)
                            bits = (u & 0x8000000000000000ULL) ? (u & ~0x8000000000000000ULL) : ~u;
                            real = true;
                        }
                    } else if (f == 8 || f == 9) {
                        // binary.Uvarint: a 10th byte with the continuation bit (an 11th byte would be needed) or above 1 is
                        // an overflow ("value larger than 64 bits", number.go:119-121)
                        if (len == 11 && s_bytes[pos + 10] > 1) err = DEC_OVERFLOW;
                        else {
                            uint64_t x = 0;
                            for (uint32_t i = 0; i + 1 < len; i++) x |= (uint64_t)(s_bytes[pos + 1 + i] & 0x7f) << (7 * i);
                            bits = f == 8 ? ((x >> 1) ^ (0 - (x & 1))) : x;  // zig-zag (binary.Varint)
                        }
                    } else if (f == 0) {
                        isnull = true;
                    } else {
                        err = (f == 1 || f == 2) ? DEC_VARLEN : DEC_BAD_FLAG;
                    }
                    if (err != DEC_OK) {
                        dec_error(a, ord, err);
                    } else {
                        const unsigned long long row = ord / ncols;
                        const uint32_t col = (uint32_t)(ord - row * ncols);
                        if (a.col_type[col] == TSQ_F32) {
                            uint32_t w = 0;
                            if (!isnull) {
                                if (real) { const float f32 = (float)tsq_bits_f64(bits); memcpy(&w, &f32, 4); }
                                else w = (uint32_t)bits;
                            }
                            ((uint32_t*)a.out_data[col])[row] = w;
                        } else {
                            ((uint64_t*)a.out_data[col])[row] = isnull ? 0ull : bits;
                        }
                        a.out_notnull[col][row] = isnull ? 0 : 1;
                    }
                }
                pos += len;
                ord++;
            }
        }
        __syncthreads();
    }
}

// ====================================================================== host side
TSQ_API tsq_status tsq_rows_decode(tsq_ctx* ctx, const uint8_t* rows_data, int64_t n_bytes, uint32_t data_flags, int32_t n_cols,
                                   const int32_t* col_types, tsq_col* out_cols, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_consumed) {
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (nrows_out) *nrows_out = 0;
    if (bytes_consumed) *bytes_consumed = 0;
    if (!nrows_out || !bytes_consumed || !col_types || !out_cols || n_bytes < 0 || cap_rows < 0 || (n_bytes > 0 && !rows_data))
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_decode: bad arguments");
    if (n_cols < 1 || n_cols > TSQ_MAX_COLS) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "1..16 columns supported");
    for (int c = 0; c < n_cols; c++) {
        if (col_types[c] < TSQ_I64 || col_types[c] > TSQ_F64) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "var-len column: decode it with the Go decoder");
        if (!out_cols[c].data || !out_cols[c].null_bitmap) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_decode: out columns need data and null_bitmap buffers");
        if (((out_cols[c].flags ^ out_cols[0].flags) & TSQ_COL_DEVICE) != 0) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_decode: mixed host/device outputs");
    }
    if (n_bytes == 0 || cap_rows == 0) return TSQ_OK;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const bool in_dev = data_flags & TSQ_COL_DEVICE, out_dev = out_cols[0].flags & TSQ_COL_DEVICE;
    // at most one value per byte: rows that can exist at all
    const int64_t max_rows = std::max<int64_t>(1, std::min<int64_t>(cap_rows, n_bytes / n_cols));
    DecArgs a;
    memset(&a, 0, sizeof a);
    a.n_bytes = n_bytes;
    a.n_tiles = (n_bytes + TSQ_DEC_TB - 1) / TSQ_DEC_TB;
    a.n_cols = n_cols;
    a.cap_rows = cap_rows;
    DevBuf dbytes, dmap, dcnt, dentry, dbase, dres, ddata[TSQ_MAX_COLS], dnn[TSQ_MAX_COLS], dbm[TSQ_MAX_COLS];
    auto release_all = [&]() {
        for (DevBuf* b : {&dbytes, &dmap, &dcnt, &dentry, &dbase, &dres}) b->release();
        for (int c = 0; c < TSQ_MAX_COLS; c++) { ddata[c].release(); dnn[c].release(); dbm[c].release(); }
    };
    tsq_status s = TSQ_OK;
    auto fail = [&](tsq_status st) { release_all(); return st; };
    if (!in_dev) {
        s = dbytes.reserve(ctx, h, (size_t)n_bytes + 64);
        if (s != TSQ_OK) return fail(s);
        hipError_t e = hipMemcpyAsync(dbytes.p, rows_data, (size_t)n_bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("hipMemcpyAsync(rows): ") + hipGetErrorString(e)));
        a.data = dbytes.as<uint8_t>();
    } else {
        a.data = rows_data;
    }
    if (s == TSQ_OK) s = dmap.reserve(ctx, h, (size_t)a.n_tiles * 8 + 64);
    if (s == TSQ_OK) s = dcnt.reserve(ctx, h, (size_t)a.n_tiles * 24 + 64);
    if (s == TSQ_OK) s = dentry.reserve(ctx, h, (size_t)a.n_tiles + 64);
    if (s == TSQ_OK) s = dbase.reserve(ctx, h, ((size_t)a.n_tiles + 1) * 8 + 64);
    if (s == TSQ_OK) s = dres.reserve(ctx, h, 64);
    for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
        a.col_type[c] = col_types[c];
        s = dnn[c].reserve(ctx, h, (size_t)max_rows + 64);
        if (s == TSQ_OK && !out_dev) s = ddata[c].reserve(ctx, h, (size_t)max_rows * tsq_elem_size(col_types[c]) + 64);
        if (s == TSQ_OK && !out_dev) s = dbm[c].reserve(ctx, h, tsq_bitmap_bytes(max_rows) + 64);
        a.out_data[c] = out_dev ? out_cols[c].data : ddata[c].p;
        a.out_notnull[c] = dnn[c].as<uint8_t>();
    }
    if (s != TSQ_OK) return fail(s);
    a.tile_map = dmap.as<unsigned long long>();
    a.tile_cnt = dcnt.as<uint16_t>();
    a.tile_entry = dentry.as<uint8_t>();
    a.tile_base = dbase.as<unsigned long long>();
    a.result = dres.as<unsigned long long>();
    ctx->pinned[0] = ~0ull;             // no error
    ctx->pinned[1] = (uint64_t)n_bytes; // everything consumed unless value number cap_rows * n_cols exists
    hipError_t e = hipMemcpyAsync(a.result, ctx->pinned, 16, hipMemcpyHostToDevice, ctx->stream);
    const int grid = (int)std::min<int64_t>(a.n_tiles, (int64_t)ctx->num_cus * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_dec_map, dim3(grid), dim3(TSQ_DEC_NT), 0, ctx->stream, a);
        hipLaunchKernelGGL(k_dec_scan, dim3(1), dim3(1024), 0, ctx->stream, a);
        hipLaunchKernelGGL(k_dec_emit, dim3(grid), dim3(TSQ_DEC_NT), 0, ctx->stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 2, a.result, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 4, a.tile_base + a.n_tiles, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rows_decode: ") + hipGetErrorString(e)));
    const uint64_t errw = ctx->pinned[2], cut = ctx->pinned[3], total = ctx->pinned[4];
    const uint64_t limit = (uint64_t)cap_rows * (uint64_t)n_cols;
    int64_t rows = (int64_t)(std::min<uint64_t>(total, limit) / (uint64_t)n_cols);
    int code = DEC_OK;
    uint64_t err_ord = ~0ull;
    if (errw != ~0ull) { err_ord = errw >> 4; code = (int)(errw & 15); }
    // a last row that ends early is noticed by the DecodeOne call after its last value (codec.go:624-626) — unless an
    // earlier value was already in error
    if (total < limit && total % (uint64_t)n_cols != 0 && err_ord >= total) { code = DEC_ROW_CUT; err_ord = total; }
    if (code != DEC_OK) rows = (int64_t)(err_ord / (uint64_t)n_cols);
    // hand the complete rows over (also in the error case: the reference has appended them to the chunk by then)
    if (rows > 0) {
        for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
            uint8_t* bm = out_dev ? out_cols[c].null_bitmap : dbm[c].as<uint8_t>();
            s = tsq_launch_pack_bitmap(ctx, h, a.out_notnull[c], bm, rows);
            if (s == TSQ_OK && !out_dev) {
                hipError_t e2 = hipMemcpyAsync(out_cols[c].data, ddata[c].p, (size_t)rows * tsq_elem_size(col_types[c]), hipMemcpyDeviceToHost, ctx->stream);
                if (e2 == hipSuccess) e2 = hipMemcpyAsync(out_cols[c].null_bitmap, bm, tsq_bitmap_bytes(rows), hipMemcpyDeviceToHost, ctx->stream);
                if (e2 != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rows_decode(D2H): ") + hipGetErrorString(e2));
            }
            out_cols[c].length = rows;
        }
        if (s == TSQ_OK) {
            hipError_t e2 = hipStreamSynchronize(ctx->stream);
            if (e2 != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rows_decode: ") + hipGetErrorString(e2));
        }
    }
    release_all();
    if (s != TSQ_OK) return s;
    *nrows_out = rows;
    if (code == DEC_OK) {
        *bytes_consumed = (int64_t)((total > limit) ? cut : (uint64_t)n_bytes);
        return TSQ_OK;
    }
    switch (code) {
        case DEC_ROW_CUT: return tsq_fail(h, TSQ_ERR_INVALID, "invalid encoded key");                          // codec.go:625
        case DEC_INSUFFICIENT: return tsq_fail(h, TSQ_ERR_INVALID, "insufficient bytes to decode value");     // number.go:46,122
        case DEC_OVERFLOW: return tsq_fail(h, TSQ_ERR_INVALID, "value larger than 64 bits");                  // number.go:120
        case DEC_VARLEN: return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "bytes datum in a fixed-width column: decode this response with the Go decoder");
        default: return tsq_fail(h, TSQ_ERR_INVALID, "invalid encoded key flag");                             // codec.go:683
    }
}
