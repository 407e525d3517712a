// tsq_encode_dp.h — the scalar core of tsq_rows_encode (tsq_encode.hip): one value -> its datum bytes, and how a workgroup copies
// the bytes of a tile from LDS to their (arbitrarily aligned) place in the output.  TSQ_HD so that the CPU test-suite runs the
// very same code through tests/hostsim against the CPU restatement that is pinned on codec_test.go.
// Reference: codec.encode (util/codec/codec.go:74-99) with comparable = false (EncodeValue, :205-209: varint forms) or true
// (EncodeKey, :199-203: the handle column of a table scan, util/rowcodec/decoder.go:263-273); encodeSignedInt / encodeUnsignedInt
// (codec.go:145-154, 167-176); EncodeInt / EncodeUint / EncodeVarint / EncodeUvarint (util/codec/number.go:24-111);
// EncodeFloat (util/codec/float.go:22-30); Go encoding/binary.PutUvarint / PutVarint.
#ifndef TSQ_ENCODE_DP_H
#define TSQ_ENCODE_DP_H

#include "tsq_device.h"

#define TSQ_ENC_MAX_VALUE 11u  // flag + 10 varint bytes

// bytes of a value's datum: NULL 1 (NilFlag); real 9; comparable int 9; varint 1 + ceil(bits / 7)
// bits: the column's 8 bytes (a float column: the DOUBLE image of the value, the caller widens float32 first)
TSQ_HD uint32_t tsq_enc_zigzag_or_plain_len(uint64_t x) {
    // PutUvarint writes one byte per started 7-bit group, at least one
    const uint32_t nbits = x ? 64u - (uint32_t)__builtin_clzll(x) : 1u;
    return (nbits + 6u) / 7u;
}
TSQ_HD uint32_t tsq_enc_len(int32_t type, bool comparable, uint64_t bits, bool notnull) {
    if (!notnull) return 1u;
    if (type == TSQ_F32 || type == TSQ_F64 || comparable) return 9u;
    const uint64_t x = type == TSQ_I64 ? ((bits << 1) ^ (uint64_t)((int64_t)bits >> 63)) : bits;  // PutVarint: zig-zag
    return 1u + tsq_enc_zigzag_or_plain_len(x);
}

// the datum bytes of a value, little endian in (lo: bytes 0..7, hi: bytes 8..10); byte 0 is the flag.  Returns the length.
TSQ_HD uint32_t tsq_enc_bytes(int32_t type, bool comparable, uint64_t bits, bool notnull, uint64_t* lo, uint32_t* hi) {
    *lo = 0;  // NilFlag
    *hi = 0;
    if (!notnull) return 1u;
    if (type == TSQ_F32 || type == TSQ_F64 || comparable) {
        uint64_t u;
        uint32_t flag;
        if (type == TSQ_F32 || type == TSQ_F64) {
            flag = 5;  // floatFlag; encodeFloatToCmpUint64 (float.go:22-30) decides on the VALUE: -0.0 is ">= 0", a NaN is not
            double f;
            memcpy(&f, &bits, 8);
            u = (f >= 0) ? (bits | 0x8000000000000000ULL) : ~bits;
        } else if (type == TSQ_I64) {
            flag = 3;                                                             // intFlag, EncodeIntToCmpUint
            u = bits ^ 0x8000000000000000ULL;
        } else {
            flag = 4;                                                             // uintFlag
            u = bits;
        }
        // 8 big-endian bytes after the flag: byte k (1..8) = u >> (64 - 8k)
        const uint64_t be = ((uint64_t)__builtin_bswap32((uint32_t)(u >> 32))) | ((uint64_t)__builtin_bswap32((uint32_t)u) << 32);  // bytes of u, most significant first
        *lo = flag | (be << 8);
        *hi = (uint32_t)(be >> 56);
        return 9u;
    }
    const uint64_t x = type == TSQ_I64 ? ((bits << 1) ^ (uint64_t)((int64_t)bits >> 63)) : bits;
    const uint32_t nb = tsq_enc_zigzag_or_plain_len(x);  // 1..10
    // byte k (0-based) of the varint = 7 bits of x from bit 7k, continuation bit on all but the last
    uint64_t v_lo = 0;  // varint bytes 0..7
    uint32_t v_hi = 0;  // varint bytes 8..9
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) v_lo |= (uint64_t)(((x >> (7 * k)) & 0x7fu) | (k + 1 < nb ? 0x80u : 0u)) << (8 * k);
    v_hi = (uint32_t)(((x >> 56) & 0x7fu) | (9 < nb ? 0x80u : 0u)) | ((uint32_t)((x >> 63) & 0x7fu) << 8);
    const uint64_t keep = nb >= 8 ? ~0ULL : ((1ULL << (8 * nb)) - 1);
    v_lo &= keep;
    v_hi &= nb >= 10 ? 0xffffu : (nb == 9 ? 0xffu : 0u);
    *lo = (type == TSQ_I64 ? 8u : 9u) | (v_lo << 8);  // varintFlag / uvarintFlag
    *hi = (uint32_t)(v_lo >> 56) | (v_hi << 8);
    return 1u + nb;
}

// a var-len cell of `n` bytes: compactBytesFlag + EncodeCompactBytes = [2][varint(n)][the bytes] (codec.go:101-109, bytes.go:141-148).
// The header (flag + varint, <= 11 bytes) is the varint form of the int64 n with the flag byte replaced
TSQ_HD uint32_t tsq_enc_str_hdr(uint64_t n, uint64_t* lo, uint32_t* hi) {
    const uint32_t h = tsq_enc_bytes(TSQ_I64, false, n, true, lo, hi);
    *lo = (*lo & ~0xffull) | 2ull;  // compactBytesFlag
    return h;
}
TSQ_HD uint32_t tsq_enc_str_hdr_len(uint64_t n) { return tsq_enc_len(TSQ_I64, false, n, true); }

// the EncodeKey form of a var-len cell of `n` bytes: bytesFlag + EncodeBytes = [1] then n / 8 + 1 groups of 8 data bytes (the last one
// padded with zeros) each followed by a marker byte 0xFF - its pad count (codec.go:86-91, bytes.go:35-67): a cell whose length is a
// multiple of 8 (the empty one too) ends with a group of 8 pad bytes and the marker 0xF7
TSQ_HD uint64_t tsq_enc_membytes_len(uint64_t n) { return 1u + (n / 8u + 1u) * 9u; }
// byte i (0-based, after the flag) of that encoding
TSQ_HD uint8_t tsq_enc_membytes_at(const uint8_t* src, uint64_t n, uint64_t i) {
    const uint64_t g = i / 9u, k = i - g * 9u;
    if (k < 8u) { const uint64_t at = g * 8u + k; return at < n ? src[at] : (uint8_t)0; }
    const uint64_t remain = n - g * 8u;  // bytes of the cell from this group on (g <= n / 8)
    return remain >= 8u ? (uint8_t)0xFF : (uint8_t)(0xFFu - (8u - remain));
}

// How the T bytes of a tile, assembled in LDS at [skew, skew + T), reach out[base, base + T): LDS byte i <-> global byte
// (out + base - skew) + i with skew = (address of out[base]) & 15, so whole 16-byte vectors are stored aligned; the bytes before the
// first / after the last whole vector are shared with the neighbouring tiles' vectors and are stored one by one.
struct tsq_enc_copy {
    uint32_t skew;
    uint32_t head_end;          // bytes [skew, head_end) one by one
    uint32_t body_lo, body_hi;  // vectors [body_lo, body_hi) whole
    uint32_t tail_lo, tail_end; // bytes [tail_lo, tail_end) one by one
};
TSQ_HD tsq_enc_copy tsq_enc_copy_plan(uint64_t out_addr, int64_t base, uint32_t T) {
    tsq_enc_copy p;
    p.skew = (uint32_t)((out_addr + (uint64_t)base) & 15u);
    const uint32_t end = p.skew + T;
    const uint32_t first_vec = (p.skew + 15u) >> 4;  // first vector that starts at or after skew
    const uint32_t last_vec = end >> 4;              // vectors below this index end at or before `end`
    // [skew, head_end): up to 15 bytes before the first whole vector; [tail_lo, end): up to 15 bytes after the last one
    p.head_end = end < (first_vec << 4) ? end : (first_vec << 4);
    p.body_lo = first_vec;
    p.body_hi = last_vec > first_vec ? last_vec : first_vec;
    p.tail_lo = (last_vec << 4) > p.head_end ? (last_vec << 4) : p.head_end;
    p.tail_end = end;
    return p;
}

#endif
