// tsq_rowcodec_dp.h — the scalar core of tsq_rowcodec_decode (tsq_rowcodec.hip): one stored row (rowcodec v2) -> the values of
// the requested columns.  TSQ_HD and templated on the byte reader, so the kernel runs it on a tile staged in LDS (or on global
// memory for very wide rows) and the CPU test-suite runs the very same code through tests/hostsim against the oracle.
// Reference: util/rowcodec/row.go:37-78 (fromBytes), :101-150 (findColID), :37-52 (getData); util/rowcodec/decoder.go:158-238
// (ChunkDecoder.DecodeToChunk / decodeColToChunk); util/rowcodec/common.go:103-114,199-210 (decodeInt / decodeUint);
// util/codec/float.go:32-46 (DecodeFloat).
#ifndef TSQ_ROWCODEC_DP_H
#define TSQ_ROWCODEC_DP_H

#include "tsq_device.h"

enum { RC_OK = 0, RC_BAD_VERSION = 1, RC_MALFORMED = 2, RC_SHORT_FLOAT = 3 };
#define TSQ_RC_CODEC_VER 128u

// How a workgroup brings the bytes [tile_lo, tile_hi) of `values` (device address base_addr) into LDS: whole 16-byte vectors
// starting at the 16-byte boundary at or below the first byte, so that every lane issues aligned 16-byte loads.  `skew` = where
// byte tile_lo lands in the staged copy.  Not staged: the offsets are inconsistent, or the span exceeds the LDS budget (the rows
// are then parsed from global memory).
struct tsq_rc_plan {
    uint32_t staged;
    uint32_t skew;
    uint32_t n_vec;     // 16-byte vectors to copy
    int64_t copy_from;  // byte offset into `values` of the first vector (down to -15 when `values` itself is not 16-byte aligned)
};
TSQ_HD tsq_rc_plan tsq_rc_tile_plan(uint64_t base_addr, int64_t tile_lo, int64_t tile_hi, int64_t n_bytes, uint32_t lds_bytes) {
    tsq_rc_plan p;
    p.staged = 0;
    p.skew = 0;
    p.n_vec = 0;
    p.copy_from = 0;
    if (tile_lo < 0 || tile_hi < tile_lo || tile_hi > n_bytes) return p;
    const uint32_t skew = (uint32_t)((base_addr + (uint64_t)tile_lo) & 15u);
    const int64_t span = tile_hi - tile_lo;
    if (span + skew > (int64_t)lds_bytes) return p;
    p.staged = 1;
    p.skew = skew;
    p.n_vec = (uint32_t)((span + skew + 15) >> 4);
    p.copy_from = tile_lo - (int64_t)skew;
    return p;
}

// A byte reader R gives the row's bytes: R::operator()(p) = byte p, R::le(p, n) = the n <= 8 bytes at p as a little-endian
// number whose bytes ABOVE n are unspecified (a reader over a staged tile returns whatever follows: it always fetches 8 bytes
// from aligned words; a reader over plain memory touches exactly n bytes).  Callers mask or shift the excess away.
// low 64 bits of the 96-bit little-endian number w2:w1:w0 shifted right by 8 * (byte offset 0..3): how the staged-tile reader
// turns three aligned 32-bit words into the 8 bytes that start at an arbitrary byte
TSQ_HD uint64_t tsq_rc_funnel(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t byte_off) {
    const uint32_t sh = 8u * (byte_off & 3u);
    const uint64_t lo = (uint64_t)w0 | ((uint64_t)w1 << 32);
    return sh ? (lo >> sh) | ((uint64_t)w2 << (64u - sh)) : lo;
}
// exact n-byte little-endian read through byte accesses only (the reader over plain memory, and the host-side readers)
template <class R>
TSQ_HD uint64_t tsq_rc_le_bytes(const R& b, uint32_t p, uint32_t n) {
    uint64_t v = 0;
    for (uint32_t i = 0; i < n; i++) v |= (uint64_t)b(p + i) << (8 * i);
    return v;
}
template <class R>
TSQ_HD uint32_t rc_u16(const R& b, uint32_t p) { return (uint32_t)b.le(p, 2) & 0xffffu; }
template <class R>
TSQ_HD uint32_t rc_u32(const R& b, uint32_t p) { return (uint32_t)b.le(p, 4); }

// the parsed header of one row (row.fromBytes): where the id / offset / value arrays start
struct tsq_rc_row {
    uint32_t len;       // bytes of the row
    uint32_t large;     // ids and offsets are 4 bytes wide instead of 1 / 2
    uint32_t n_notnull, n_null;
    uint32_t ids_at, offs_at, data_at;
};

template <class R>
TSQ_HD int tsq_rc_parse(const R& b, uint32_t len, tsq_rc_row* r) {
    r->len = len;
    if (len < 1) return RC_MALFORMED;                   // rowData[0] on an empty value: index out of range in the reference
    if (b(0) != TSQ_RC_CODEC_VER) return RC_BAD_VERSION;  // row.go:54-56
    if (len < 6) return RC_MALFORMED;
    const uint64_t hdr = b.le(0, 6);  // [ver][flag][numNotNull u16][numNull u16]
    r->large = (uint32_t)(hdr >> 8) & 1u;
    r->n_notnull = (uint32_t)(hdr >> 16) & 0xffffu;
    r->n_null = (uint32_t)(hdr >> 32) & 0xffffu;
    r->ids_at = 6;
    r->offs_at = 6 + (r->n_notnull + r->n_null) * (r->large ? 4u : 1u);
    r->data_at = r->offs_at + r->n_notnull * (r->large ? 4u : 2u);
    return r->data_at <= len ? RC_OK : RC_MALFORMED;  // the reference slices rowData[cursor : cursor+len]: out of range panics
}

template <class R>
TSQ_HD uint32_t tsq_rc_id(const R& b, const tsq_rc_row& r, uint32_t i) {
    return r.large ? rc_u32(b, r.ids_at + 4 * i) : b(r.ids_at + i);
}

// row.findColID (row.go:101-150): binary search in the not-null ids, then in the null ids.
// returns 0 = found (idx_out), 1 = the column is NULL in this row, 2 = not in the row
template <class R>
TSQ_HD int tsq_rc_find(const R& b, const tsq_rc_row& r, int64_t col_id64, uint32_t* idx_out) {
    // stored ids are uint8 / uint32: an id outside [0, 2^32) compares the same way against every stored id, both searches
    // run off one end and the column is not in the row; inside that range the comparisons are the reference's, in 32 bits
    if (col_id64 < 0 || col_id64 > 0xffffffffLL) return 2;
    const uint32_t col_id = (uint32_t)col_id64;
    uint32_t i = 0, j = r.n_notnull;
    while (i < j) {
        const uint32_t h = (i + j) >> 1;
        const uint32_t v = tsq_rc_id(b, r, h);
        if (v < col_id) i = h + 1;
        else if (v > col_id) j = h;
        else { *idx_out = h; return 0; }
    }
    i = r.n_notnull;
    j = r.n_notnull + r.n_null;
    while (i < j) {
        const uint32_t h = (i + j) >> 1;
        const uint32_t v = tsq_rc_id(b, r, h);
        if (v < col_id) i = h + 1;
        else if (v > col_id) j = h;
        else return 1;
    }
    return 2;
}

// the n bytes at p as the column stores them (decodeColToChunk, decoder.go:200-238)
template <class R>
TSQ_HD int tsq_rc_value(const R& b, uint32_t p, uint32_t n, int32_t type, uint64_t* bits_out) {
    uint64_t bits;
    if (type == TSQ_BYTES) {
        // chk.AppendBytes(colIdx, colData) (decoder.go:226-228): the cell is the value's bytes as they are.  What the column stores
        // for now is a REFERENCE — where the bytes start inside the row (high dword) and how many there are (low dword); the
        // bytes are copied once every row's length is known (tsq_rowcodec.hip: lengths -> scan -> copy)
        bits = ((uint64_t)p << 32) | (uint64_t)n;
    } else if (type == TSQ_I64 || type == TSQ_U64) {
        // decodeInt / decodeUint (common.go:103-114,199-210): 1, 2, 4 bytes, or LittleEndian.Uint64 of the first 8 (which needs
        // 8): read min(n, 8) bytes, then sign- / zero-extend from n bytes with one pair of shifts
        if (!(n == 1 || n == 2 || n == 4 || n >= 8)) return RC_MALFORMED;
        const uint32_t nb = n < 8 ? n : 8u;
        const uint64_t raw = b.le(p, nb);
        const uint32_t sh = 64u - 8u * nb;
        bits = type == TSQ_I64 ? (uint64_t)((int64_t)(raw << sh) >> sh) : ((raw << sh) >> sh);
    } else {
        if (n < 8) return RC_SHORT_FLOAT;  // DecodeFloat -> DecodeUint: "insufficient bytes to decode value"
        const uint64_t le8 = b.le(p, 8);
        const uint64_t u = ((uint64_t)__builtin_bswap32((uint32_t)le8) << 32) | (uint64_t)__builtin_bswap32((uint32_t)(le8 >> 32));  // big endian
        bits = (u & 0x8000000000000000ULL) ? (u & ~0x8000000000000000ULL) : ~u;  // decodeCmpUintToFloat (float.go:32-40)
        if (type == TSQ_F32) {  // chk.AppendFloat32(colIdx, float32(fVal))
            double d;
            memcpy(&d, &bits, 8);
            const float f32 = (float)d;
            uint32_t w32;
            memcpy(&w32, &f32, 4);
            bits = w32;
        }
    }
    *bits_out = bits;
    return RC_OK;
}

// one output column of one row (DecodeToChunk's loop body, decoder.go:164-196): *bits_out is what the column stores (a float32
// in the low 4 bytes; a TSQ_BYTES cell as a (start in the row, length) reference), *notnull_out its bitmap bit.
template <class R>
TSQ_HD int tsq_rc_column(const R& b, const tsq_rc_row& r, int64_t col_id, int32_t type, uint32_t flags, uint64_t def_bits, int64_t handle,
                         uint64_t* bits_out, bool* notnull_out) {
    *bits_out = 0;  // a NULL slot holds zero bytes (column.go:150-158)
    *notnull_out = false;
    if (flags & TSQ_RC_HANDLE) {  // chk.AppendInt64(colIdx, handle)
        *bits_out = (uint64_t)handle;
        *notnull_out = true;
        return RC_OK;
    }
    uint32_t idx = 0;
    const int f = tsq_rc_find(b, r, col_id, &idx);
    if (f == 1) return RC_OK;  // AppendNull
    if (f == 2) {              // not in the row: the default, if there is one (decoder.go:186-194)
        if (flags & TSQ_RC_HAS_DEFAULT) { *bits_out = def_bits; *notnull_out = true; }
        return RC_OK;
    }
    // getData (row.go:37-52): data[offsets[idx-1] : offsets[idx]] — both offsets with one read: the w bytes before entry idx
    // are entry idx-1, or (idx = 0) the tail of the id array, which is inside the row and ignored
    const uint32_t w = r.large ? 4u : 2u;
    const uint64_t oo = b.le(r.offs_at + w * idx - w, 2 * w);
    const uint32_t start = idx > 0 ? (uint32_t)(r.large ? oo : (oo & 0xffffu)) : 0u;
    const uint32_t end = r.large ? (uint32_t)(oo >> 32) : (uint32_t)(oo >> 16) & 0xffffu;
    if (start > end || end > r.len - r.data_at) return RC_MALFORMED;  // slice bounds out of range in the reference
    const int vc = tsq_rc_value(b, r.data_at + start, end - start, type, bits_out);
    if (vc != RC_OK) return vc;
    *notnull_out = true;
    return RC_OK;
}

// ---------------------------------------------------------------- rows that share their layout (the fast path of a wave)
// The rows of one table usually carry the same columns: same header and same id array, only the values differ.  When the 64
// rows of a wave agree on (header, ids) — small rows with at most 8 ids — the column -> value-index search is the same for all
// of them: it is done once on the shared signature (uniform values, scalar unit) and every lane only reads its offsets and value.
// The functions below are what a lane runs on that path; which path a wave takes is a vote (ballot) in the kernel.

// signature of a small row: hdr = its 6 header bytes, ids8 = its <= 8 id bytes (zero padded).  false: not a candidate
// (not the new format, large, more than 8 ids, or the header arrays do not fit the row) — such rows take the general path
template <class R>
TSQ_HD bool tsq_rc_signature(const R& b, uint32_t len, uint64_t* hdr_out, uint64_t* ids8_out) {
    *hdr_out = 0;
    *ids8_out = 0;
    if (len < 6) return false;
    const uint64_t hdr = b.le(0, 6) & 0xffffffffffffULL;
    const uint32_t nn = (uint32_t)(hdr >> 16) & 0xffffu, nl = (uint32_t)(hdr >> 32) & 0xffffu;
    if ((hdr & 0xffu) != TSQ_RC_CODEC_VER || ((hdr >> 8) & 1u) || nn + nl > 8 || 6 + nn + nl + 2 * nn > len) return false;
    const uint32_t k = nn + nl;
    const uint64_t ids = k ? b.le(6, k) : 0;
    *hdr_out = hdr;
    *ids8_out = k >= 8 ? ids : ids & ((1ULL << (8 * k)) - 1);
    return true;
}

// row.findColID over the ids of a signature (one byte each): the same two binary searches as tsq_rc_find
TSQ_HD int tsq_rc_find_small(uint64_t ids8, uint32_t nn, uint32_t nl, int64_t col_id64, uint32_t* idx_out) {
    if (col_id64 < 0 || col_id64 > 0xffffffffLL) return 2;
    const uint32_t col_id = (uint32_t)col_id64;
    uint32_t i = 0, j = nn;
    while (i < j) {
        const uint32_t h = (i + j) >> 1;
        const uint32_t v = (uint32_t)(ids8 >> (8 * h)) & 255u;
        if (v < col_id) i = h + 1;
        else if (v > col_id) j = h;
        else { *idx_out = h; return 0; }
    }
    i = nn;
    j = nn + nl;
    while (i < j) {
        const uint32_t h = (i + j) >> 1;
        const uint32_t v = (uint32_t)(ids8 >> (8 * h)) & 255u;
        if (v < col_id) i = h + 1;
        else if (v > col_id) j = h;
        else return 1;
    }
    return 2;
}

// the row's <= 8 end offsets (2 bytes each) in two words
template <class R>
TSQ_HD void tsq_rc_fast_offsets(const R& b, uint32_t nn, uint32_t offs_at, uint64_t* o_lo, uint64_t* o_hi) {
    *o_lo = nn ? b.le(offs_at, nn >= 4 ? 8 : 2 * nn) : 0;
    *o_hi = nn > 4 ? b.le(offs_at + 8, 2 * (nn - 4)) : 0;
}

// one output column of a row on the shared-layout path: hdr0 / ids0 = the signature every row of the wave has
template <class R>
TSQ_HD int tsq_rc_fast_column(const R& b, uint32_t len, uint64_t hdr0, uint64_t ids0, uint64_t o_lo, uint64_t o_hi, int64_t col_id, int32_t type,
                              uint32_t flags, uint64_t def_bits, int64_t handle, uint64_t* bits_out, bool* notnull_out) {
    *bits_out = 0;
    *notnull_out = false;
    if (flags & TSQ_RC_HANDLE) {
        *bits_out = (uint64_t)handle;
        *notnull_out = true;
        return RC_OK;
    }
    const uint32_t nn = (uint32_t)(hdr0 >> 16) & 0xffffu, nl = (uint32_t)(hdr0 >> 32) & 0xffffu;
    uint32_t idx = 0;
    const int f = tsq_rc_find_small(ids0, nn, nl, col_id, &idx);  // the same for every row of the wave
    if (f == 1) return RC_OK;
    if (f == 2) {
        if (flags & TSQ_RC_HAS_DEFAULT) { *bits_out = def_bits; *notnull_out = true; }
        return RC_OK;
    }
    const uint32_t data_at = 6 + nn + nl + 2 * nn;
    const uint32_t end = (uint32_t)((idx < 4 ? o_lo : o_hi) >> (16 * (idx & 3))) & 0xffffu;
    const uint32_t start = idx > 0 ? (uint32_t)((idx - 1 < 4 ? o_lo : o_hi) >> (16 * ((idx - 1) & 3))) & 0xffffu : 0u;
    if (start > end || end > len - data_at) return RC_MALFORMED;
    const int vc = tsq_rc_value(b, data_at + start, end - start, type, bits_out);
    if (vc != RC_OK) return vc;
    *notnull_out = true;
    return RC_OK;
}

#endif
