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
// only 11 different offsets.  For every 32-byte sub-block and each of the 11 entry offsets we need the exit offset into
// the next sub-block and the number of values that start inside (an "exit map", 11 x 4 bits, and 11 counts); maps compose
// associatively.  Per sub-block this is ONE backward pass over its 32 positions — (exit, count)[p] = p + len[p] leaves the
// sub-block ? (p + len[p] - 32, 1) : (exit, count + 1)[p + len[p]] — i.e. one step per byte, no divergence (walking the
// 11 entry offsets forward costs 11 steps per VALUE and the slowest of 64 x 11 walks per wave: 5x more instructions,
// measured).  Workgroup b owns the contiguous tiles [b*R, (b+1)*R):
//   K13a k_dec_map   : per 8 KB tile in LDS: value length at every byte position (as if a value started there), the
//                      backward pass, composition of the 256 sub-block maps (20 segments x 11 entry offsets in parallel,
//                      then 20 sequential steps) -> tile map + counts; the workgroup composes its R tile maps
//   K13b k_dec_scan  : one thread composes the <= 1024 workgroup maps -> entry offset and first value ordinal per workgroup
//   K13c k_dec_emit  : workgroup b replays its tile maps from its entry; per tile: lengths, backward pass, composition
//                      from the known entry -> every sub-block walks its TRUE path (about five values), decodes them
//                      and stores them at (row, column) = divmod(ordinal, n_cols)
// LDS rows are padded (36 bytes per 32-byte sub-block) so that 64 lanes working in 64 different sub-blocks fall into
// different banks.
// Errors follow the reference in stream order: the FIRST offending value decides (invalid flag, value cut by the end of
// the buffer, varint longer than 64 bits, a row that ends early); atomicMin over (ordinal, code) finds it.
// Algorithmic bytes: encoded bytes read once + 8 B written per value (the implementation reads the bytes twice).
#include "tsq_stage.h"
#include "tsq_decode_dp.h"

#define TSQ_DEC_TB 8192       // bytes per tile
#define TSQ_DEC_SB 32         // bytes per sub-block
#define TSQ_DEC_NSB (TSQ_DEC_TB / TSQ_DEC_SB)
#define TSQ_DEC_NT 256        // threads per workgroup = sub-blocks per tile
#define TSQ_DEC_BSTR 36       // s_bytes: bytes per sub-block row (32 + 4 pad: 9 words, odd)
#define TSQ_DEC_NSEG 20       // composition segments per tile
#define TSQ_DEC_SEGLEN 13     // sub-blocks per segment (20 * 13 >= 256)
#define TSQ_DEC_MAXWG 1024


struct DecArgs {
    const uint8_t* data;
    int64_t n_bytes;
    int64_t n_tiles;
    int64_t tiles_per_wg;           // R
    int32_t n_wg;                   // workgroups that own tiles (grid of map and emit)
    unsigned long long* sb_map;     // [n_tiles * 256] exit map of every sub-block (written by K13a, read by K13c)
    uint32_t* sb_cnt;               // [n_tiles * 256][3] packed counts of every sub-block
    unsigned long long* wg_map;     // [n_wg]
    uint32_t* wg_cnt;               // [n_wg][12]
    uint32_t* wg_entry;             // [n_wg] true entry offset of the workgroup's first tile (k_dec_scan)
    unsigned long long* wg_base;    // [n_wg + 1] ordinal of the first value of the workgroup's range; [n_wg] = total
    // emit
    int32_t n_cols;
    int32_t col_type[TSQ_MAX_COLS];
    void* out_data[TSQ_MAX_COLS];
    uint8_t* out_notnull[TSQ_MAX_COLS];  // one byte per row (packed afterwards), or
    uint32_t* out_bm32[TSQ_MAX_COLS];    // the packed bitmap itself, preset to all ones: a NULL clears its bit (atomicAnd)
    int64_t cap_rows;
    unsigned long long* result;  // [0] = min over errors of (ordinal << 4 | code), [1] = byte offset of value number cap_rows * n_cols
};

// LDS state of one tile
struct DecTile {
    uint8_t bytes[(TSQ_DEC_NSB + 1) * TSQ_DEC_BSTR];  // stream bytes, padded rows (+ one halo row)
    unsigned long long map[TSQ_DEC_NSB];              // per sub-block: 11 x 4-bit exit offsets
    uint32_t cnt[TSQ_DEC_NSB][3];                     // per sub-block: 11 counts, four per word
    uint32_t seg_exit[TSQ_DEC_NSEG][12];              // per segment and entry offset
    uint32_t seg_cnt[TSQ_DEC_NSEG][12];
};
__device__ __forceinline__ uint32_t dec_bidx(uint32_t p) { return (p >> 5) * TSQ_DEC_BSTR + (p & 31u); }

// count of entry offset `state` out of the packed words (no data-dependent address: the loads do not wait for `state`)
__device__ __forceinline__ uint32_t dec_cnt_of(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t state) {
    const uint32_t w = state < 4 ? w0 : (state < 8 ? w1 : w2);
    return (w >> (8 * (state & 3))) & 255u;
}

// Loads tile `t` (+ 11 bytes of halo, zero padded past n_bytes) and computes the exit map and counts of every sub-block.
// Thread sb owns sub-block sb: its 32 bytes + 11 halo bytes live in REGISTERS (11 conflict-free LDS words), the value
// length at each position comes from the flag byte and a 43-bit mask of continuation bits (count-trailing-zeros instead
// of a byte loop), and the backward pass keeps the last 11 results packed in a 128-bit register window (the successor
// p + len[p] is at most 11 positions ahead): fully unrolled, no LDS traffic, no divergence.  (Doing both with LDS byte reads and an LDS-resident
// state array cost 56 % of k_dec_map.)  Returns the number of valid bytes of the tile.
template <bool FROM_MAPS>
__device__ __forceinline__ uint32_t dec_prepare_tile(const DecArgs& a, int64_t t, DecTile& T) {
    const int64_t t0 = t * TSQ_DEC_TB;
    const int64_t left = a.n_bytes - t0;
    const uint32_t valid = left < TSQ_DEC_TB ? (uint32_t)left : (uint32_t)TSQ_DEC_TB;
    const uint32_t avail = left < TSQ_DEC_TB + 16 ? (uint32_t)left : (uint32_t)(TSQ_DEC_TB + 16);
    const bool aligned = (((uintptr_t)a.data) & 15) == 0;
    // global -> LDS: 16-byte global loads (a tile starts at a multiple of 8 KB), 4-byte LDS stores into the padded rows.
    // (Prefetching the next tile into registers while this one is processed was measured and dropped: 12 more live VGPRs
    // cost more than the hidden latency, k_dec_emit 1.74 -> 2.09 ms.)
    for (uint32_t i = threadIdx.x * 16; i < TSQ_DEC_TB + 16; i += TSQ_DEC_NT * 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (aligned && i + 16 <= avail) {
            const uint4 v = *(const uint4*)(a.data + t0 + i);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            for (uint32_t j = 0; j < 16; j++) w[j >> 2] |= (i + j < avail ? (uint32_t)a.data[t0 + i + j] : 0u) << (8 * (j & 3));
        }
        uint32_t* d = (uint32_t*)(T.bytes + dec_bidx(i));  // 16 bytes never straddle a 32-byte row
        d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = w[3];
    }
    __syncthreads();
    if (FROM_MAPS) {  // K13c: the maps were computed by K13a — 20 bytes per sub-block read back instead of a second backward pass
        const size_t g = (size_t)t * TSQ_DEC_NSB + threadIdx.x;
        T.map[threadIdx.x] = a.sb_map[g];
        T.cnt[threadIdx.x][0] = a.sb_cnt[g * 3 + 0];
        T.cnt[threadIdx.x][1] = a.sb_cnt[g * 3 + 1];
        T.cnt[threadIdx.x][2] = a.sb_cnt[g * 3 + 2];
    } else {
        const uint32_t sb = threadIdx.x, lo = sb * TSQ_DEC_SB;
        const uint32_t lim = lo + TSQ_DEC_SB <= valid ? (uint32_t)TSQ_DEC_SB : (valid > lo ? valid - lo : 0u);
        const uint32_t* rw = (const uint32_t*)(T.bytes + sb * TSQ_DEC_BSTR);
        uint32_t w[11];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = rw[i];
#pragma unroll
        for (int i = 0; i < 3; i++) w[8 + i] = rw[9 + i];  // first 12 bytes of the next row (word 8 of a row is padding)
        unsigned long long m;
        uint32_t cw[3];
        tsq_dec_subblock(w, lim, &m, cw);
        T.map[sb] = m;
        T.cnt[sb][0] = cw[0];
        T.cnt[sb][1] = cw[1];
        T.cnt[sb][2] = cw[2];
        const size_t g = (size_t)t * TSQ_DEC_NSB + sb;
        a.sb_map[g] = m;
        a.sb_cnt[g * 3 + 0] = cw[0];
        a.sb_cnt[g * 3 + 1] = cw[1];
        a.sb_cnt[g * 3 + 2] = cw[2];
    }
    __syncthreads();
    // 20 segments x 11 entry offsets in parallel: exit offset and count of every (segment, entry)
    if (threadIdx.x < TSQ_DEC_NSEG * 11) {
        const uint32_t seg = threadIdx.x / 11, e = threadIdx.x - seg * 11;
        uint32_t state = e, total = 0;
        const uint32_t i1 = seg * TSQ_DEC_SEGLEN + TSQ_DEC_SEGLEN < TSQ_DEC_NSB ? seg * TSQ_DEC_SEGLEN + TSQ_DEC_SEGLEN : TSQ_DEC_NSB;
        for (uint32_t i = seg * TSQ_DEC_SEGLEN; i < i1; i++) {
            total += dec_cnt_of(T.cnt[i][0], T.cnt[i][1], T.cnt[i][2], state);
            state = (uint32_t)(T.map[i] >> (4 * state)) & 15u;
        }
        T.seg_exit[seg][e] = state;
        T.seg_cnt[seg][e] = total;
    }
    __syncthreads();
    return valid;
}

__global__ void __launch_bounds__(TSQ_DEC_NT) k_dec_map(DecArgs a) {
    __shared__ __align__(16) DecTile T;
    __shared__ uint32_t s_exit[11];
    const int64_t lo = (int64_t)blockIdx.x * a.tiles_per_wg;
    const int64_t hi = lo + a.tiles_per_wg < a.n_tiles ? lo + a.tiles_per_wg : a.n_tiles;
    // lanes 0..10 follow entry offset e of the workgroup's RANGE through the segments of all its tiles
    uint32_t wstate = threadIdx.x, wtotal = 0;
    for (int64_t t = lo; t < hi; t++) {
        dec_prepare_tile<false>(a, t, T);
        if (threadIdx.x < 11) {
            for (int sg = 0; sg < TSQ_DEC_NSEG; sg++) {
                wtotal += T.seg_cnt[sg][wstate];
                wstate = T.seg_exit[sg][wstate];
            }
        }
        // the next tile's prepare overwrites T.seg_* only after two barriers that lanes 0..10 take part in
    }
    if (threadIdx.x < 11) {
        s_exit[threadIdx.x] = wstate;
        a.wg_cnt[blockIdx.x * 12 + threadIdx.x] = wtotal;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = 0;
        for (int e = 0; e < 11; e++) m |= (unsigned long long)s_exit[e] << (4 * e);
        a.wg_map[blockIdx.x] = m;
    }
}

// <= 1024 workgroup maps: one thread walks them (maps and counts staged in LDS)
__global__ void __launch_bounds__(1024) k_dec_scan(DecArgs a) {
    __shared__ unsigned long long s_map[TSQ_DEC_MAXWG];
    __shared__ uint32_t s_cnt[TSQ_DEC_MAXWG][12];
    for (int i = threadIdx.x; i < a.n_wg; i += 1024) {
        s_map[i] = a.wg_map[i];
        for (int e = 0; e < 11; e++) s_cnt[i][e] = a.wg_cnt[i * 12 + e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t state = 0;
        unsigned long long base = 0;
        for (int i = 0; i < a.n_wg; i++) {
            a.wg_entry[i] = state;
            a.wg_base[i] = base;
            base += s_cnt[i][state];
            state = (uint32_t)(s_map[i] >> (4 * state)) & 15u;
        }
        a.wg_base[a.n_wg] = base;
    }
}

// length of the value that starts at tile position p (flag + payload), from the LDS bytes: used on the true path only
// (~5 values per sub-block).  Same rule as in dec_prepare_tile: a varint has at most 10 bytes, overflow gets length 11.
__device__ __forceinline__ uint32_t dec_len_at(const uint8_t* bytes, uint32_t p) {
    const uint8_t f = bytes[dec_bidx(p)];
    if (f == 3 || f == 4 || f == 5) return 9;
    if (f == 8 || f == 9) {
        uint32_t k = 1;
        while (k < 10 && (bytes[dec_bidx(p + k)] & 0x80)) k++;
        return k + 1;
    }
    return 1;
}

__device__ __forceinline__ void dec_error(const DecArgs& a, unsigned long long ordinal, int code) {
    atomicMin(&a.result[0], (ordinal << 4) | (unsigned long long)code);
}

// decodes the value at tile position pos (ordinal ord) and stores it; t0 = byte offset of the tile in the stream.
// The 12 bytes starting at pos are fetched as four aligned LDS words + a byte funnel shift; big-endian payloads are two
// byte swaps, a varint is eight shift-and-mask terms cut to its length — no per-byte loop.
__device__ __forceinline__ void dec_value(const DecArgs& a, const uint8_t* bytes, uint32_t pos, uint32_t len, unsigned long long ord, unsigned long long row, uint32_t col,
                                          int64_t t0) {
    const uint32_t* W = (const uint32_t*)bytes;
    const uint32_t q = pos >> 2, sh = pos & 3u;
    uint32_t w[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) w[k] = W[((q + k) >> 3) * (TSQ_DEC_BSTR / 4) + ((q + k) & 7u)];  // word 8 of a row is padding
    const uint32_t b0 = __builtin_amdgcn_alignbyte(w[1], w[0], sh), b1 = __builtin_amdgcn_alignbyte(w[2], w[1], sh),
                   b2 = __builtin_amdgcn_alignbyte(w[3], w[2], sh);
    uint64_t bits = 0;
    bool isnull = false, real = false;
    int err = t0 + pos + len > a.n_bytes ? DEC_INSUFFICIENT  // the value is cut by the end of the buffer (number.go:45,115-123)
                                         : tsq_dec_value(b0, b1, b2, len, &bits, &isnull, &real);
    if (err != DEC_OK) {
        dec_error(a, ord, err);
        return;
    }
    if (a.col_type[col] == TSQ_F32) {
        uint32_t w32 = 0;
        if (!isnull) {
            if (real) { const float f32 = (float)tsq_bits_f64(bits); memcpy(&w32, &f32, 4); }
            else w32 = (uint32_t)bits;
        }
        ((uint32_t*)a.out_data[col])[row] = w32;
    } else {
        ((uint64_t*)a.out_data[col])[row] = isnull ? 0ull : bits;
    }
    if (a.out_bm32[col]) {
        if (isnull) atomicAnd(&a.out_bm32[col][row >> 5], ~(1u << (row & 31)));  // NULLs are the rare case: no flag byte per value, no pack pass
    } else {
        a.out_notnull[col][row] = isnull ? 0 : 1;
    }
}

__global__ void __launch_bounds__(TSQ_DEC_NT) k_dec_emit(DecArgs a) {
    __shared__ __align__(16) DecTile T;
    __shared__ uint32_t s_next[2];
    __shared__ uint8_t s_entry[TSQ_DEC_NSB];
    __shared__ uint32_t s_base[TSQ_DEC_NSB];
    const unsigned long long limit = (unsigned long long)a.cap_rows * (unsigned long long)a.n_cols;  // values wanted
    const uint32_t ncols = (uint32_t)a.n_cols;
    const int64_t lo = (int64_t)blockIdx.x * a.tiles_per_wg;
    const int64_t hi = lo + a.tiles_per_wg < a.n_tiles ? lo + a.tiles_per_wg : a.n_tiles;
    uint32_t tile_entry = a.wg_entry[blockIdx.x];
    unsigned long long tile_base = a.wg_base[blockIdx.x];
    for (int64_t t = lo; t < hi && tile_base <= limit; t++) {  // block-uniform: tiles beyond cap_rows are not needed at all
        {
            const int64_t t0 = t * TSQ_DEC_TB;
            const uint32_t valid = dec_prepare_tile<true>(a, t, T);
            if (threadIdx.x < TSQ_DEC_NSEG) {
                // segment `seg` finds its own entry on the true path (the segments before it, <= 19 steps) and then walks its
                // 13 sub-blocks; the last segment also knows where the path leaves the tile
                const uint32_t seg = threadIdx.x;
                uint32_t state = tile_entry, base = 0;
                for (uint32_t sg = 0; sg < seg; sg++) {
                    base += T.seg_cnt[sg][state];
                    state = T.seg_exit[sg][state];
                }
                const uint32_t i1 = seg * TSQ_DEC_SEGLEN + TSQ_DEC_SEGLEN < TSQ_DEC_NSB ? seg * TSQ_DEC_SEGLEN + TSQ_DEC_SEGLEN : TSQ_DEC_NSB;
                for (uint32_t i = seg * TSQ_DEC_SEGLEN; i < i1; i++) {
                    s_entry[i] = (uint8_t)state;
                    s_base[i] = base;
                    base += dec_cnt_of(T.cnt[i][0], T.cnt[i][1], T.cnt[i][2], state);
                    state = (uint32_t)(T.map[i] >> (4 * state)) & 15u;
                }
                if (seg == TSQ_DEC_NSEG - 1) {
                    s_next[0] = state;  // the true path leaves the tile here, after `base` values
                    s_next[1] = base;
                }
            }
            __syncthreads();
            {
                const uint32_t sblo = threadIdx.x * TSQ_DEC_SB;
                const uint32_t lim = sblo + TSQ_DEC_SB <= valid ? (uint32_t)TSQ_DEC_SB : (valid > sblo ? valid - sblo : 0u);
                uint32_t pos = s_entry[threadIdx.x];
                unsigned long long ord = tile_base + s_base[threadIdx.x];
                // (row, column) of the first value by one division, of the following ones by counting
                unsigned long long row = ord / ncols;
                uint32_t col = (uint32_t)(ord - row * ncols);
                while (pos < lim) {
                    const uint32_t len = dec_len_at(T.bytes, sblo + pos);
                    if (ord == limit) a.result[1] = (unsigned long long)(t0 + sblo + pos);  // first byte that is not consumed
                    if (ord < limit) dec_value(a, T.bytes, sblo + pos, len, ord, row, col, t0);
                    pos += len;
                    ord++;
                    if (++col == ncols) { col = 0; row++; }
                }
            }
            tile_entry = s_next[0];
            tile_base += s_next[1];
            __syncthreads();
        }
    }
}

// ====================================================================== host side
// A schema with a var-len column (round 5): bytes datums have no bounded length, so the row boundaries come from one sequential
// host walk over flags and lengths (tsq_dec_walk_rows, tsq_decode_dp.h) and the values are decoded by the chunk-parallel kernels of
// tsq_decodec.hip, 64 rows per piece — the size the storage side cuts a response into (cop_handler_dag.go:510-519).  A stream that
// lives in HBM is copied to the host for the walk (the bytes are read there, nothing is decoded there).
extern "C" tsq_status tsq_rows_decode_chunks(tsq_ctx* ctx, const uint8_t* rows_data, int64_t n_bytes, const int64_t* chunk_offsets, int64_t n_chunks,
                                              uint32_t data_flags, int32_t n_cols, const int32_t* col_types, tsq_col* out_cols, int64_t cap_rows,
                                              int64_t* nrows_out);
static tsq_status tsq_rows_decode_varlen(tsq_ctx* ctx, const uint8_t* rows_data, int64_t n_bytes, uint32_t data_flags, int32_t n_cols, const int32_t* col_types,
                                         tsq_col* out_cols, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_consumed) {
    tsq_handle_hdr* h = &ctx->hdr;
    if (n_bytes == 0 || cap_rows == 0) return tsq_rows_decode_chunks(ctx, rows_data, 0, nullptr, 0, data_flags, n_cols, col_types, out_cols, cap_rows, nrows_out);
    const uint8_t* host = rows_data;
    std::vector<uint8_t> copy;
    if (data_flags & TSQ_COL_DEVICE) {
        TSQ_HIP(h, hipSetDevice(ctx->device));
        copy.resize((size_t)n_bytes);
        TSQ_HIP(h, hipMemcpy(copy.data(), rows_data, (size_t)n_bytes, hipMemcpyDeviceToHost));
        host = copy.data();
    }
    std::vector<int64_t> offs;
    int64_t end = 0;
    bool damaged = false;
    (void)tsq_dec_walk_rows(host, n_bytes, n_cols, cap_rows, 64, offs, &end, &damaged);
    const int64_t* po = offs.data();
    DevBuf doffs;  // (a stream in HBM: the piece boundaries live next to it — the chunk decoder reads both where the data is)
    if (data_flags & TSQ_COL_DEVICE) {
        TSQ_TRY(doffs.reserve(ctx, h, offs.size() * 8 + 64));
        const hipError_t e = hipMemcpy(doffs.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            doffs.release();
            return tsq_fail(h, TSQ_ERR_HIP, std::string("tsq_rows_decode(piece boundaries): ") + hipGetErrorString(e));
        }
        po = doffs.as<int64_t>();
    }
    const tsq_status s = tsq_rows_decode_chunks(ctx, rows_data, end, po, (int64_t)offs.size() - 1, data_flags, n_cols, col_types, out_cols, cap_rows, nrows_out);
    doffs.release();
    if (s == TSQ_OK) *bytes_consumed = end;
    return s;
}

TSQ_API tsq_status tsq_rows_decode(tsq_ctx* ctx, const uint8_t* rows_data, int64_t n_bytes, uint32_t data_flags, int32_t n_cols,
                                   const int32_t* col_types, tsq_col* out_cols, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_consumed) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx) return TSQ_ERR_INVALID;
    tsq_handle_hdr* h = &ctx->hdr;
    if (nrows_out) *nrows_out = 0;
    if (bytes_consumed) *bytes_consumed = 0;
    if (!nrows_out || !bytes_consumed || !col_types || !out_cols || n_bytes < 0 || cap_rows < 0 || (n_bytes > 0 && !rows_data))
        return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_decode: bad arguments");
    if (n_cols < 1 || n_cols > TSQ_MAX_COLS) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "1..16 columns supported");
    for (int c = 0; c < n_cols; c++)
        if (col_types[c] == TSQ_BYTES) return tsq_rows_decode_varlen(ctx, rows_data, n_bytes, data_flags, n_cols, col_types, out_cols, cap_rows, nrows_out, bytes_consumed);
    for (int c = 0; c < n_cols; c++) {
        if (col_types[c] < TSQ_I64 || col_types[c] > TSQ_F64) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_rows_decode: unknown column type");
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
    {   // workgroup b owns tiles [b*R, (b+1)*R): <= 1024 workgroups, 4 per CU
        const int64_t want = std::min<int64_t>(std::min<int64_t>(a.n_tiles, (int64_t)ctx->num_cus * 4), TSQ_DEC_MAXWG);
        a.tiles_per_wg = (a.n_tiles + want - 1) / want;
        a.n_wg = (int32_t)((a.n_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg);
    }
    a.n_cols = n_cols;
    a.cap_rows = cap_rows;
    DevBuf dbytes, dsbmap, dsbcnt, dwmap, dwcnt, dentry, dbase, dres, ddata[TSQ_MAX_COLS], dnn[TSQ_MAX_COLS], dbm[TSQ_MAX_COLS];
    auto release_all = [&]() {
        for (DevBuf* b : {&dbytes, &dsbmap, &dsbcnt, &dwmap, &dwcnt, &dentry, &dbase, &dres}) b->release();
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
    if (s == TSQ_OK) s = dsbmap.reserve(ctx, h, (size_t)a.n_tiles * TSQ_DEC_NSB * 8 + 64);
    if (s == TSQ_OK) s = dsbcnt.reserve(ctx, h, (size_t)a.n_tiles * TSQ_DEC_NSB * 12 + 64);
    if (s == TSQ_OK) s = dwmap.reserve(ctx, h, (size_t)a.n_wg * 8 + 64);
    if (s == TSQ_OK) s = dwcnt.reserve(ctx, h, (size_t)a.n_wg * 48 + 64);
    if (s == TSQ_OK) s = dentry.reserve(ctx, h, (size_t)a.n_wg * 4 + 64);
    if (s == TSQ_OK) s = dbase.reserve(ctx, h, ((size_t)a.n_wg + 1) * 8 + 64);
    if (s == TSQ_OK) s = dres.reserve(ctx, h, 64);
    // null bitmaps: written in place (preset to ones, NULLs clear their bit) when the destination is 4-byte aligned,
    // otherwise through one flag byte per value + a pack pass
    bool direct_bm = true;
    for (int c = 0; c < n_cols; c++) direct_bm = direct_bm && (!out_dev || (((uintptr_t)out_cols[c].null_bitmap) & 3) == 0);
    const size_t bm_bytes = (tsq_bitmap_bytes(max_rows) + 3) & ~(size_t)3;
    // the NULL-clearing atomics work on whole 32-bit words: a device destination (sized for cap_rows rows by contract) must
    // consist of whole words
    if (out_dev) direct_bm = direct_bm && (tsq_bitmap_bytes(cap_rows) & 3) == 0;
    for (int c = 0; c < n_cols && s == TSQ_OK; c++) {
        a.col_type[c] = col_types[c];
        if (!direct_bm) s = dnn[c].reserve(ctx, h, (size_t)max_rows + 64);
        if (s == TSQ_OK && !out_dev) s = ddata[c].reserve(ctx, h, (size_t)max_rows * tsq_elem_size(col_types[c]) + 64);
        if (s == TSQ_OK && !out_dev) s = dbm[c].reserve(ctx, h, bm_bytes + 64);
        a.out_data[c] = out_dev ? out_cols[c].data : ddata[c].p;
        a.out_notnull[c] = direct_bm ? nullptr : dnn[c].as<uint8_t>();
        a.out_bm32[c] = direct_bm ? (uint32_t*)(out_dev ? out_cols[c].null_bitmap : dbm[c].as<uint8_t>()) : nullptr;
        if (s == TSQ_OK && direct_bm) {
            // bytes of rows the caller asked for: a device destination holds cap_rows rows, so (cap_rows + 7) / 8 bytes exist;
            // the word-wise atomics may touch up to 3 bytes beyond the last row's byte only inside this rounded size
            const size_t dst_bytes = out_dev ? std::min<size_t>(tsq_bitmap_bytes(cap_rows), bm_bytes) : bm_bytes;
            hipError_t e0 = hipMemsetAsync(a.out_bm32[c], 0xff, dst_bytes, ctx->stream);
            if (e0 != hipSuccess) s = tsq_fail(h, TSQ_ERR_HIP, std::string("hipMemsetAsync(bitmap): ") + hipGetErrorString(e0));
        }
    }
    if (s != TSQ_OK) return fail(s);
    a.sb_map = dsbmap.as<unsigned long long>();
    a.sb_cnt = dsbcnt.as<uint32_t>();
    a.wg_map = dwmap.as<unsigned long long>();
    a.wg_cnt = dwcnt.as<uint32_t>();
    a.wg_entry = dentry.as<uint32_t>();
    a.wg_base = dbase.as<unsigned long long>();
    a.result = dres.as<unsigned long long>();
    ctx->pinned[0] = ~0ull;             // no error
    ctx->pinned[1] = (uint64_t)n_bytes; // everything consumed unless value number cap_rows * n_cols exists
    hipError_t e = hipMemcpyAsync(a.result, ctx->pinned, 16, hipMemcpyHostToDevice, ctx->stream);
    const int grid = a.n_wg;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_dec_map, dim3(grid), dim3(TSQ_DEC_NT), 0, ctx->stream, a);
        hipLaunchKernelGGL(k_dec_scan, dim3(1), dim3(1024), 0, ctx->stream, a);
        hipLaunchKernelGGL(k_dec_emit, dim3(grid), dim3(TSQ_DEC_NT), 0, ctx->stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 2, a.result, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned + 4, a.wg_base + a.n_wg, 8, hipMemcpyDeviceToHost, ctx->stream);
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
            if (!direct_bm) s = tsq_launch_pack_bitmap(ctx, h, a.out_notnull[c], bm, rows);
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
        case DEC_VARLEN: return tsq_fail(h, TSQ_ERR_INVALID, "datum kind does not match the column type");  // a bytes datum, a number column (as tsq_rows_decode_chunks)
        default: return tsq_fail(h, TSQ_ERR_INVALID, "invalid encoded key flag");                             // codec.go:683
    }
}
