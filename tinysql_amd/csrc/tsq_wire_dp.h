// tsq_wire_dp.h — the chunk wire format (util/chunk/codec.go:42-143), header walk shared by the device kernel (tsq_wire.hip),
// the host entry points (a host buffer is walked by the same function) and the CPU walk-through in tests/hostsim.
//
// Codec.encodeColumn (codec.go:50-76) writes, per column and with no alignment or padding between the pieces:
//     u32 length | u32 nullCount | [ (length + 7) / 8 bitmap bytes, only when nullCount > 0 ] |
//     [ (length + 1) x int64 offsets, only for a var-len column ] | data (length x elem size, or offsets[length] bytes)
// all little endian; Codec.decodeColumn (codec.go:96-143) reads them back and sets an all-ones bitmap when nullCount is 0.
#ifndef TSQ_WIRE_DP_H
#define TSQ_WIRE_DP_H

#include <stdint.h>

#ifndef TSQ_HD
#if defined(__HIPCC__)
#define TSQ_HD __host__ __device__ __forceinline__
#else
#define TSQ_HD inline
#endif
#endif

struct tsq_wire_col {
    int64_t rows;        // Column.length
    int64_t null_count;
    int64_t bitmap_pos;  // byte positions in the buffer; bitmap_pos < 0: no bitmap on the wire (all rows NOT NULL)
    int64_t offs_pos;    // < 0: fixed width
    int64_t data_pos;
    int64_t data_bytes;
    int64_t end_pos;     // first byte of the next column
};

TSQ_HD uint32_t tsq_wire_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
TSQ_HD int64_t tsq_wire_i64(const uint8_t* p) { return (int64_t)((uint64_t)tsq_wire_u32(p) | ((uint64_t)tsq_wire_u32(p + 4) << 32)); }

// One column starting at `pos`.  elem: 4 / 8, or -1 for a var-len column (getFixedLen, codec.go:169-181).  Returns 0, or 1 when the
// buffer ends inside the column, its offsets do not start at 0 or its last offset is negative / beyond the buffer.  The offsets IN
// BETWEEN are checked where they are used: tsq_chunk_decode verifies that the offsets of the window it appends do not decrease
// (k_wire_check_offs, before anything is written) and that the window's endpoints lie inside the data (wire_view) — a damaged chunk
// is an error (TSQ_ERR_INVALID) where the reference slices out of range and panics.
TSQ_HD int32_t tsq_wire_parse_col(const uint8_t* buf, int64_t n_bytes, int64_t pos, int32_t elem, tsq_wire_col* d) {
    if (pos < 0 || n_bytes - pos < 8) return 1;
    d->rows = (int64_t)tsq_wire_u32(buf + pos);
    d->null_count = (int64_t)tsq_wire_u32(buf + pos + 4);
    pos += 8;
    d->bitmap_pos = -1;
    if (d->null_count > 0) {
        const int64_t nb = (d->rows + 7) / 8;
        if (n_bytes - pos < nb) return 1;
        d->bitmap_pos = pos;
        pos += nb;
    }
    d->offs_pos = -1;
    if (elem < 0) {
        const int64_t nb = (d->rows + 1) * 8;
        if (n_bytes - pos < nb) return 1;
        d->offs_pos = pos;
        d->data_bytes = tsq_wire_i64(buf + pos + d->rows * 8);
        if (tsq_wire_i64(buf + pos) != 0 || d->data_bytes < 0) return 1;
        pos += nb;
    } else {
        d->data_bytes = d->rows * elem;
    }
    if (n_bytes - pos < d->data_bytes) return 1;
    d->data_pos = pos;
    d->end_pos = pos + d->data_bytes;
    return 0;
}

// What the device walk reports back per column (four words: tsq_ctx::pinned holds 16 x 4)
//   [0] rows | null_count << 32, or ~0 when the walk failed at this column      [1] data bytes of the column
//   [2] offsets[first]                                                            [3] offsets[first + take]   (var-len only)
TSQ_HD void tsq_wire_walk(const uint8_t* buf, int64_t n_bytes, const int32_t* elem, int32_t n_cols, int64_t first, int64_t max_rows, uint64_t* out) {
    int64_t pos = 0;
    for (int32_t c = 0; c < n_cols; c++) {
        tsq_wire_col d;
        if (tsq_wire_parse_col(buf, n_bytes, pos, elem[c], &d) != 0) {
            for (int32_t k = c; k < n_cols; k++) out[4 * k] = ~0ull;
            return;
        }
        out[4 * c + 0] = (uint64_t)d.rows | ((uint64_t)d.null_count << 32);
        out[4 * c + 1] = (uint64_t)d.data_bytes;
        out[4 * c + 2] = out[4 * c + 3] = 0;
        if (d.offs_pos >= 0) {
            const int64_t f = first < d.rows ? first : d.rows;
            const int64_t take = max_rows < d.rows - f ? max_rows : d.rows - f;
            out[4 * c + 2] = (uint64_t)tsq_wire_i64(buf + d.offs_pos + f * 8);
            out[4 * c + 3] = (uint64_t)tsq_wire_i64(buf + d.offs_pos + (f + take) * 8);
        }
        pos = d.end_pos;
    }
}

// ---- moving the pieces (k_wire_move, tsq_wire.hip): what lane t of workgroup `blk` of one piece does
enum { WM_COPY = 0, WM_HDR = 1, WM_BITS = 2, WM_OFFS = 3 };
#define TSQ_WIRE_BLOCK_BYTES 16384  // bytes of one workgroup (COPY; OFFS: 2048 offsets; BITS: 4096 bitmap bytes)

struct tsq_wire_v16 { uint64_t lo, hi; };
struct __attribute__((packed, aligned(1))) tsq_wire_v16u { uint64_t lo, hi; };  // the same sixteen bytes at any address
struct __attribute__((packed, aligned(1))) tsq_wire_i64u { int64_t v; };
TSQ_HD tsq_wire_v16 tsq_wire_ld16(const uint8_t* p) {
    const tsq_wire_v16u* q = (const tsq_wire_v16u*)p;
    tsq_wire_v16 r;
    r.lo = q->lo;
    r.hi = q->hi;
    return r;
}
#if defined(__HIPCC__)
#define TSQ_WIRE_UNROLL _Pragma("unroll")
#else
#define TSQ_WIRE_UNROLL
#endif

//   WM_COPY  n bytes src -> dst, both at any byte position: 16-byte destination-aligned vectors loaded with unaligned 16-byte loads
//            (global loads take any byte address); the bytes before the first and after the last vector go one per lane in workgroup 0
//   WM_HDR   the eight bytes of imm (u32 length | u32 nullCount)
//   WM_BITS  rows [imm, imm + n) of the destination bitmap <- source bits [0, n): byte k of the source lands b = imm % 8 bits up in
//            destination byte k and spills its high bits into byte k + 1; the bits below the first appended row are kept, the bits at
//            and above row imm + n of the last byte are cleared (Decoder.decodeColumn, codec.go:325-343).  src null: every bit set
//            (a column that travelled without its bitmap, setAllNotNull codec.go:147-155)
//   WM_OFFS  n offsets: aligned int64 destination <- source at any byte position, + imm (the rebase of codec.go:314-320)
TSQ_HD void tsq_wire_move_lane(int32_t mode, const uint8_t* src, uint8_t* dst, int64_t n, int64_t imm, int64_t blk, int t) {
    if (mode == WM_HDR) {
        if (t < 8) dst[t] = (uint8_t)((uint64_t)imm >> (8 * t));
    } else if (mode == WM_COPY) {
        int64_t head = (int64_t)((16 - ((uintptr_t)dst & 15)) & 15);
        if (head > n) head = n;
        const int64_t nvec = (n - head) >> 4;
        if (blk == 0) {
            if (t < head) dst[t] = src[t];
            const int64_t tail0 = head + (nvec << 4);
            if (t >= 32 && t - 32 < n - tail0) dst[tail0 + t - 32] = src[tail0 + t - 32];
        }
        const int64_t v0 = blk * (TSQ_WIRE_BLOCK_BYTES / 16);
        if (v0 >= nvec) return;
        // all four loads are issued before the first store (clamped to the last vector instead of branching: a load under a branch
        // makes the compiler wait for it on the spot)
        const uint8_t* sp = src + head;
        uint8_t* dp = dst + head;
        const int64_t last = nvec - 1, va = v0 + t, vb = va + 256, vc = va + 512, vd = va + 768;
        const tsq_wire_v16 ra = tsq_wire_ld16(sp + ((va < nvec ? va : last) << 4));
        const tsq_wire_v16 rb = tsq_wire_ld16(sp + ((vb < nvec ? vb : last) << 4));
        const tsq_wire_v16 rc = tsq_wire_ld16(sp + ((vc < nvec ? vc : last) << 4));
        const tsq_wire_v16 rd = tsq_wire_ld16(sp + ((vd < nvec ? vd : last) << 4));
        if (va < nvec) *(tsq_wire_v16*)(dp + (va << 4)) = ra;
        if (vb < nvec) *(tsq_wire_v16*)(dp + (vb << 4)) = rb;
        if (vc < nvec) *(tsq_wire_v16*)(dp + (vc << 4)) = rc;
        if (vd < nvec) *(tsq_wire_v16*)(dp + (vd << 4)) = rd;
    } else if (mode == WM_BITS) {
        const int64_t rows0 = imm, total = rows0 + n;
        const int b = (int)(rows0 & 7);
        const int64_t first = rows0 >> 3, nb_dst = ((total + 7) >> 3) - first, nb_src = (n + 7) >> 3;
        for (int i = 0; i < 16; i++) {
            const int64_t k = blk * 4096 + i * 256 + t;
            if (k >= nb_dst) break;
            const uint32_t cur = k < nb_src ? (src ? src[k] : 0xffu) : 0u;
            const uint32_t prev = (k >= 1 && k - 1 < nb_src) ? (src ? src[k - 1] : 0xffu) : 0u;
            uint32_t v = b ? (((cur << b) | (prev >> (8 - b))) & 0xffu) : cur;
            if (k == 0 && b) v |= dst[first] & ((1u << b) - 1u);
            if (k == nb_dst - 1 && (total & 7)) v &= (1u << (total & 7)) - 1u;
            dst[first + k] = (uint8_t)v;
        }
    } else {
        int64_t* d = (int64_t*)dst;
        if (blk * 2048 >= n) return;
        int64_t x[8];
        TSQ_WIRE_UNROLL
        for (int i = 0; i < 8; i++) {
            const int64_t k = blk * 2048 + i * 256 + t;
            x[i] = ((const tsq_wire_i64u*)(src + (k < n ? k : n - 1) * 8))->v;
        }
        TSQ_WIRE_UNROLL
        for (int i = 0; i < 8; i++) {
            const int64_t k = blk * 2048 + i * 256 + t;
            if (k < n) d[k] = x[i] + imm;
        }
    }
}


#endif
