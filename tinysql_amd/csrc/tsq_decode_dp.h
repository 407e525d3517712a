// tsq_decode_dp.h — the scalar core of tsq_rows_decode (tsq_decode.hip), TSQ_HD so that the CPU test-suite can run the very
// same code through tests/hostsim and compare it with the oracle without a GPU:
//   tsq_dec_subblock : exit map + counts of one 32-byte sub-block for the 11 possible entry offsets (backward pass)
//   tsq_dec_value    : one value (flag byte + payload) from the 12 bytes that start at its position
// Reference: codec.Decoder.DecodeOne (util/codec/codec.go:623-690), number.go:24-130, float.go:22-46, Go encoding/binary.
#ifndef TSQ_DECODE_DP_H
#define TSQ_DECODE_DP_H

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

#endif
