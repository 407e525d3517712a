// tsq_decode_dp.h — the scalar core of tsq_rows_decode (tsq_decode.hip), TSQ_HD so that the CPU test-suite can run the very
// same code through tests/hostsim and compare it with the oracle without a GPU:
//   tsq_dec_subblock : exit map + counts of one 32-byte sub-block for the 11 possible entry offsets (backward pass)
//   tsq_dec_value    : one value (flag byte + payload) from the 12 bytes that start at its position
// Reference: codec.Decoder.DecodeOne (util/codec/codec.go:623-690), number.go:24-130, float.go:22-46, Go encoding/binary.
#ifndef TSQ_DECODE_DP_H
#define TSQ_DECODE_DP_H

#include <vector>

#include "tsq_device.h"

enum { DEC_OK = 0, DEC_ROW_CUT = 1, DEC_INSUFFICIENT = 2, DEC_OVERFLOW = 3, DEC_BAD_FLAG = 4, DEC_VARLEN = 5 };

// the continuation bits (bit 7) of the four bytes of w as a 4-bit number, byte 0 first
TSQ_HD uint32_t dec_msb4(uint32_t w) { return ((((w >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 15u; }
// low 32 bits of (hi:lo) >> sh, 0 <= sh < 32
TSQ_HD uint32_t tsq_funnelshift_r(uint32_t lo, uint32_t hi, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __funnelshift_r(lo, hi, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}

// w[0..7] = the 32 bytes of the sub-block (little-endian words), w[8..10] = the 12 bytes that follow it (zero past the end
// of the stream); lim = number of its bytes that belong to the stream (32 except at the very end).
// *map_out: 11 x 4 bits, exit offset into the next sub-block when entered at offset e; cnt_out[3]: 11 x 8 bits, values
// that start inside when entered at e.
TSQ_HD void tsq_dec_subblock(const uint32_t* w, uint32_t lim, unsigned long long* map_out, uint32_t* cnt_out) {
    uint32_t m_lo = 0, m_hi = 0;  // continuation bits of bytes 0..31 / 32..43
#pragma unroll
    for (int i = 0; i < 8; i++) m_lo |= dec_msb4(w[i]) << (4 * i);
#pragma unroll
    for (int i = 0; i < 3; i++) m_hi |= dec_msb4(w[8 + i]) << (4 * i);
    // (exit | count << 4) of the 11 positions after the current one, 10 bits each, position o + j in field j - 1 of a
    // 128-bit window: the successor o + len is picked with one variable shift instead of ten compare-selects.
    // Positions 32..42 lie in the next sub-block: exit = p - 32, count 0.
    unsigned long long w_lo = 0, w_hi = 0;
#pragma unroll
    for (int j = 1; j <= 11; j++) {
        const unsigned long long f = (unsigned long long)(j - 1);
        if (10 * (j - 1) < 64) w_lo |= f << (10 * (j - 1));
        if (10 * (j - 1) + 10 > 64) w_hi |= 10 * (j - 1) >= 64 ? f << (10 * (j - 1) - 64) : f >> (64 - 10 * (j - 1));
    }
#pragma unroll
    for (int o = 31; o >= 0; o--) {
        const uint32_t f = (w[o >> 2] >> (8 * (o & 3))) & 255u;
        // a varint has at most 10 bytes; one whose 10th byte still has the continuation bit is an overflow
        // (binary.Uvarint) and gets the maximal length 11 as well
        const uint32_t after = o + 1 < 32 ? tsq_funnelshift_r(m_lo, m_hi, o + 1) : m_hi >> (o + 1 - 32);  // continuation bits from o + 1 on
        const uint32_t run = (uint32_t)__builtin_ctz(~after | (1u << 9));  // continuation bytes after the flag, <= 9
        uint32_t len = 1;  // NULL, or a flag that is an error if this position is ever reached on the true path
        len = (f == 8 || f == 9) ? run + 2 : len;
        len = (f == 3 || f == 4 || f == 5) ? 9u : len;
        const uint32_t sh = 10u * (len - 1);
        const unsigned long long pick = sh < 64 ? ((w_lo >> sh) | (sh ? w_hi << (64 - sh) : 0ull)) : (w_hi >> (sh - 64));
        const uint32_t e = (uint32_t)o < lim ? ((uint32_t)pick & 0x3ffu) + (1u << 4) : 0u;  // past the end of the stream: not a value
        w_hi = (w_hi << 10) | (w_lo >> 54);
        w_lo = (w_lo << 10) | e;
    }
    // the window now holds positions 0..10 in fields 0..10
    unsigned long long m = 0;
    uint32_t cw[3] = {0, 0, 0};
#pragma unroll
    for (int e = 0; e < 11; e++) {
        const uint32_t v = (uint32_t)(10 * e < 64 ? ((w_lo >> (10 * e)) | (10 * e + 10 > 64 ? w_hi << (64 - 10 * e) : 0ull)) : (w_hi >> (10 * e - 64))) & 0x3ffu;
        m |= (unsigned long long)(v & 15u) << (4 * e);
        cw[e >> 2] |= (v >> 4) << (8 * (e & 3));
    }
    *map_out = m;
    cnt_out[0] = cw[0];
    cnt_out[1] = cw[1];
    cnt_out[2] = cw[2];
}

// The value whose 12 first bytes are b0 | b1 << 32 | b2 << 64 (little endian; its real length is len = flag + payload, as
// computed by the length rule above).  Big-endian payloads are two byte swaps, a varint is eight shift-and-mask terms cut
// to its length — no per-byte loop.  Returns DEC_OK or the reference's error for this value.
TSQ_HD int tsq_dec_value(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t len, uint64_t* bits_out, bool* isnull_out, bool* real_out) {
    const uint32_t f = b0 & 255u;
    const uint32_t p_lo = (b0 >> 8) | (b1 << 24), p_hi = (b1 >> 8) | (b2 << 24);  // payload bytes 1..8, little endian
    uint64_t bits = 0;
    bool isnull = false, real = false;
    int err = DEC_OK;
    if (f == 3 || f == 4 || f == 5) {
        const uint64_t u = ((uint64_t)__builtin_bswap32(p_lo) << 32) | __builtin_bswap32(p_hi);  // binary.BigEndian.Uint64
        if (f == 3) bits = u ^ 0x8000000000000000ULL;  // DecodeCmpUintToInt (number.go:29-31)
        else if (f == 4) bits = u;
        else {  // decodeCmpUintToFloat (float.go:32-40)
            bits = (u & 0x8000000000000000ULL) ? (u & ~0x8000000000000000ULL) : ~u;
            real = true;
        }
    } else if (f == 8 || f == 9) {
        const uint32_t byte9 = (b2 >> 8) & 255u, byte10 = (b2 >> 16) & 255u;
        // binary.Uvarint: a 10th byte with the continuation bit (an 11th byte would be needed) or above 1 is an overflow
        // ("value larger than 64 bits", number.go:119-121)
        if (len == 11 && byte10 > 1) err = DEC_OVERFLOW;
        else {
            const uint64_t P = (uint64_t)p_lo | ((uint64_t)p_hi << 32);
            uint64_t x = (P & 0x7full) | ((P >> 1) & (0x7full << 7)) | ((P >> 2) & (0x7full << 14)) | ((P >> 3) & (0x7full << 21)) |
                         ((P >> 4) & (0x7full << 28)) | ((P >> 5) & (0x7full << 35)) | ((P >> 6) & (0x7full << 42)) | ((P >> 7) & (0x7full << 49)) |
                         ((uint64_t)(byte9 & 0x7fu) << 56) | ((uint64_t)(byte10 & 1u) << 63);
            const uint32_t nb = len - 1;  // bytes of the varint, 1..10
            if (nb < 10) x &= (1ull << (7 * nb)) - 1;
            bits = f == 8 ? ((x >> 1) ^ (0 - (x & 1))) : x;  // zig-zag (binary.Varint)
        }
    } else if (f == 0) {
        isnull = true;
    } else {
        err = (f == 1 || f == 2) ? DEC_VARLEN : DEC_BAD_FLAG;
    }
    *bits_out = bits;
    *isnull_out = isnull;
    *real_out = real;
    return err;
}

// ---------------------------------------------------------------- a response CHUNK walked value by value (tsq_rows_decode_chunks)
// tipb.SelectResponse.Chunks cut the response every 64 rows (cop_handler_dag.go:510-519): the chunks are independent byte strings,
// so one lane walks one chunk, and a value may be of any length — a compact-bytes datum (flag 2: varint length + the bytes,
// util/codec/bytes.go:141-160) is what a varchar / blob column arrives as.
enum { DEC_KIND_MISMATCH = 6, DEC_BAD_MARKER = 7, DEC_BAD_PADDING = 8, DEC_NO_HANDLE = 9 };
enum { DECV_NULL = 0, DECV_INT = 1, DECV_UINT = 2, DECV_REAL = 3, DECV_BYTES = 4 };

TSQ_HD uint32_t dec_byte12(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t k) {  // byte k (0..11) of b0 | b1 << 32 | b2 << 64
    const uint32_t w = k < 4 ? b0 : (k < 8 ? b1 : b2);
    return (w >> (8 * (k & 3))) & 255u;
}
// binary.Uvarint (Go) over the bytes that follow the flag byte; `avail` of them exist.  n == 0 (the bytes run out) is
// "insufficient bytes to decode value", n < 0 "value larger than 64 bits" (number.go:113-123)
TSQ_HD int tsq_decc_uvarint(uint32_t b0, uint32_t b1, uint32_t b2, uint64_t avail, uint64_t* x_out, uint32_t* nb_out) {
    uint64_t x = 0;
    for (uint32_t i = 0;; i++) {
        if ((uint64_t)i >= avail) return DEC_INSUFFICIENT;  // `for i, b := range buf` ends: return 0, 0
        if (i == 10) return DEC_OVERFLOW;                   // i == MaxVarintLen64
        const uint32_t c = dec_byte12(b0, b1, b2, 1 + i);
        if (c < 0x80) {
            if (i == 9 && c > 1) return DEC_OVERFLOW;
            *x_out = x | ((uint64_t)c << (7 * i));
            *nb_out = i + 1;
            return DEC_OK;
        }
        x |= (uint64_t)(c & 0x7fu) << (7 * i);
    }
}
struct tsq_decc_val {
    uint64_t len;       // bytes of the whole value (flag included)
    uint32_t kind;      // DECV_*
    uint32_t data_at;   // DECV_BYTES: where the string's bytes start inside the value
    uint64_t bits;      // the decoded number (a real: its double image); DECV_BYTES: the string's length
};
// Decoder.DecodeOne (util/codec/codec.go:623-690) of the value whose first 12 bytes are b0 | b1 << 32 | b2 << 64 (zero past the
// chunk); avail = bytes from its first byte to the end of the chunk (>= 1)
TSQ_HD int tsq_decc_value(uint32_t b0, uint32_t b1, uint32_t b2, uint64_t avail, tsq_decc_val* v) {
    const uint32_t f = b0 & 255u;
    v->len = 1;
    v->kind = DECV_NULL;
    v->data_at = 0;
    v->bits = 0;
    if (f == 0) return DEC_OK;  // NilFlag
    if (f == 3 || f == 4 || f == 5) {
        if (avail < 9) return DEC_INSUFFICIENT;  // DecodeInt / DecodeUint / DecodeFloat need 8 bytes (number.go:44-53,82-90)
        bool isnull, real;
        (void)tsq_dec_value(b0, b1, b2, 9, &v->bits, &isnull, &real);
        v->len = 9;
        v->kind = f == 3 ? DECV_INT : (f == 4 ? DECV_UINT : DECV_REAL);
        return DEC_OK;
    }
    if (f == 8 || f == 9 || f == 2) {
        uint64_t x = 0;
        uint32_t nb = 0;
        const int st = tsq_decc_uvarint(b0, b1, b2, avail - 1, &x, &nb);
        if (st != DEC_OK) return st;
        const uint64_t zz = (x >> 1) ^ (0 - (x & 1));  // binary.Varint: zig-zag
        if (f == 9) { v->bits = x; v->kind = DECV_UINT; v->len = 1 + nb; return DEC_OK; }
        if (f == 8) { v->bits = zz; v->kind = DECV_INT; v->len = 1 + nb; return DEC_OK; }
        // compactBytesFlag: DecodeCompactBytes (bytes.go:150-160): `if int64(len(b)) < n` -> insufficient bytes; a negative length
        // makes the reference slice out of range (panic): reported the same way
        const int64_t n = (int64_t)zz;
        if (n < 0 || (uint64_t)n > avail - 1 - nb) return DEC_INSUFFICIENT;
        v->kind = DECV_BYTES;
        v->data_at = 1 + nb;
        v->bits = (uint64_t)n;
        v->len = 1 + (uint64_t)nb + (uint64_t)n;
        return DEC_OK;
    }
    return f == 1 ? DEC_VARLEN : DEC_BAD_FLAG;  // bytesFlag: the caller walks the groups in memory (tsq_decc_membytes)
}
// A bytesFlag datum — the memcomparable form an index key holds (EncodeBytes, util/codec/bytes.go:35-67): after the flag, groups
// of 8 data bytes + 1 marker byte; marker 0xFF = a full group, another follows; marker 0xFF - pad = the last group, whose final
// `pad` bytes are zero padding.  decodeBytes (bytes.go:69-112) step by step, over the value's bytes in memory (p = its flag byte,
// avail = bytes from there to the end of the chunk / key): v->bits = the string's length, v->len = 1 + 9 x groups; the string's
// byte i sits at p[1 + i + i / 8] (tsq_decc_grouped_at).
TSQ_HD int tsq_decc_membytes(const uint8_t* p, uint64_t avail, tsq_decc_val* v) {
    uint64_t at = 1, n = 0;
    for (;;) {
        if (avail < at + 9) return DEC_INSUFFICIENT;       // len(b) < encGroupSize + 1
        const uint32_t pad = 0xffu - (uint32_t)p[at + 8];  // encMarker - marker
        if (pad > 8) return DEC_BAD_MARKER;
        n += 8 - pad;
        if (pad != 0) {
            for (uint32_t k = 8 - pad; k < 8; k++)
                if (p[at + k] != 0) return DEC_BAD_PADDING;
            at += 9;
            break;
        }
        at += 9;
    }
    v->kind = DECV_BYTES;
    v->data_at = 1;
    v->bits = n;
    v->len = at;
    return DEC_OK;
}
// HOST: the row boundaries of a RowsData stream that holds bytes datums (round 5; tsq_rows_decode with a TSQ_BYTES column).  A
// compact-bytes datum is a varint length + that many raw bytes, so where a value starts depends on every value before it and no
// bounded window can be entered speculatively (what tsq_decode.hip does for the <= 11-byte number datums): the boundaries are found
// by ONE sequential walk that reads flags and lengths only — selectResult.readRowsData's own loop (distsql/select_result.go:139-155)
// minus the decoding — and the rows between them are decoded by the chunk-parallel kernels (tsq_decodec.hip), every `per` rows one
// piece.  offs receives the piece boundaries (first 0).  Stops after cap_rows rows or at the first value DecodeOne would reject; a
// stream with a damaged value keeps its remainder as the LAST piece, so the kernels meet the damage and report it with the
// reference's message and the rows before it.  Returns the rows walked; *end = bytes of the walked prefix (bytes_consumed on success).
inline int64_t tsq_dec_walk_rows(const uint8_t* data, int64_t n_bytes, int32_t n_cols, int64_t cap_rows, int64_t per, std::vector<int64_t>& offs, int64_t* end,
                                 bool* damaged) {
    int64_t pos = 0, rows = 0;
    offs.clear();
    offs.push_back(0);
    *damaged = false;
    while (pos < n_bytes && rows < cap_rows) {
        int64_t p = pos;
        bool ok = true;
        for (int32_t c = 0; c < n_cols && ok; c++) {
            if (p >= n_bytes) { ok = false; break; }  // the row ends early (codec.go:625)
            const uint64_t avail = (uint64_t)(n_bytes - p);
            uint32_t w[3] = {0, 0, 0};
            for (uint32_t k = 0; k < 12 && k < avail; k++) w[k >> 2] |= (uint32_t)data[p + k] << (8 * (k & 3));
            tsq_decc_val v;
            int st = tsq_decc_value(w[0], w[1], w[2], avail, &v);
            if (st == DEC_VARLEN) st = tsq_decc_membytes(data + p, avail, &v);
            if (st != DEC_OK) { ok = false; break; }
            p += (int64_t)v.len;
        }
        if (!ok) {
            *damaged = true;
            break;
        }
        pos = p;
        rows++;
        if (rows % per == 0) offs.push_back(pos);
    }
    if (*damaged) {  // the remainder travels as one piece (or extends the open one): the kernels find the first offending value
        offs.push_back(n_bytes);
        *end = n_bytes;
    } else {
        if (offs.back() != pos) offs.push_back(pos);
        *end = pos;
    }
    return rows;
}
#define TSQ_DECC_GROUPED (1ll << 62)  // flag in a cell reference: the bytes are grouped (skip one marker byte after every 8)
// what column type `type` stores for a datum (appendIntToChunk / appendUintToChunk / appendFloatToChunk / AppendBytes,
// codec.go:692-707): false = the datum's kind cannot go into that column (a string into a number column or the reverse)
TSQ_HD bool tsq_decc_store(int32_t type, const tsq_decc_val& v, uint64_t* bits_out) {
    *bits_out = 0;
    if (v.kind == DECV_NULL) return true;
    if ((type == TSQ_BYTES) != (v.kind == DECV_BYTES)) return false;
    if (type == TSQ_BYTES) return true;
    if (type == TSQ_F32) {
        if (v.kind == DECV_REAL) {  // appendFloatToChunk narrows for TypeFloat
            double d;
            memcpy(&d, &v.bits, 8);
            const float f32 = (float)d;
            uint32_t w;
            memcpy(&w, &f32, 4);
            *bits_out = w;
        } else {
            *bits_out = (uint32_t)v.bits;  // an integer datum in a float column: the raw low bytes (not meaningful in the reference either)
        }
        return true;
    }
    *bits_out = v.bits;
    return true;
}

#endif
