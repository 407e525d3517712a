// tsq_keyrec_dp.h — the KEY RECORD of a row (tsq_keyrec.h, tsq_keydict.h): host/device-portable (TSQ_HD), so that the CPU suite walks the
// same code through tests/hostsim.  Per key column: a flag byte — 8 / 9 / 5, the classes of tsq_key_word; 2 for a string, as
// util/codec/codec.go:233-235; 0 (NilFlag) for a NULL cell of a GROUP BY key — followed by the 8-byte word, or by a length byte and the
// string's bytes (a NULL cell: zero bytes in their place); zero padding to 32 bytes.  Two rows have equal keys (codec.EqualChunkRow, codec.go:363-382 / the encoded group key,
// codec.go:700-760) iff their records are equal byte for byte.
#ifndef TSQ_KEYREC_DP_H
#define TSQ_KEYREC_DP_H

#include "tsq_device.h"

#define TSQ_KR_BYTES 32

struct KrSrc {
    tsq_colset cs;
    int32_t n_keys;
    int32_t col[TSQ_MAX_KEYS];
    int32_t keep_nulls;      // GROUP BY: a NULL cell is a key like any other (NilFlag + zero bytes, codec.go:718-719) — a join drops the row
    int32_t layout;          // kr_layout_of(): 0 = any key columns; else the cells' byte positions are compile-time constants (kr_record_fixed)
    const uint8_t* selected; // nullptr or one byte per row: 0 = the row has no key (an outer-side filter said no, join.go:344)
    int64_t nrows;
    // round 6, LONG string keys (the reference's own benchmark joins on a 5 KiB varstring, executor/benchmark_test.go:328-360): digest[k] !=
    // nullptr -> the cell of string key column k enters the record as flag 3 | length (4 bytes) | 64-bit digest of its bytes (digest[k][row],
    // k_kr_digest) instead of its bytes — 13 bytes whatever the length.  Equal records then mean "equal lengths and digests": the probe
    // kernel compares the bytes themselves before it counts a match (KrProbeArgs.n_verify), so the result is exact.
    const uint64_t* digest[TSQ_MAX_KEYS];
};
#define TSQ_KR_DIGEST_CELL 13u
// the digest of a cell's bytes: the cell as little-endian 8-byte words (the last one zero-padded), word i mixed with its position, the
// mixes SUMMED (any order: a wave takes a long cell's words lane by lane), the length folded in at the end
TSQ_HD uint64_t kr_digest_word(uint64_t w, uint32_t i) {
    uint64_t x = (w ^ ((uint64_t)(i + 1u) * 0x9E3779B97F4A7C15ULL)) * 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 29;
    return x * 0x94D049BB133111EBULL;
}
TSQ_HD uint64_t kr_digest_finish(uint64_t sum, uint64_t len) {
    uint64_t h = (sum ^ (len * 0xD6E8FEB86659FD93ULL)) * 0xFF51AFD7ED558CCDULL;
    return h ^ (h >> 32);
}
#define TSQ_KR_MAXPAY 4
// Cell sizes: an 8-byte cell takes 9 bytes (flag + word), a string 2 + its bytes (flag 2, length, bytes).  A NULL cell of a GROUP BY
// key takes the same room as a value's fixed part — 9 zero bytes, or 2 — so that the cells of a key with at most one string column,
// in last place, start at compile-time positions (kr_layout_of): the run-time positions of kr_put cost ~50 vector instructions per
// piece, and building the record was the larger half of the 350 instructions k_kr_hist issued per 64 rows (profiles/r05_keyrec_sq.txt).
// layout = 1 + (8-byte cells before the string) + 4 * (the key ends with a string): 2..4 = one to three 8-byte cells, 5..7 = a string
// after none / one / two 8-byte cells; 0 = everything else (strings elsewhere, four cells)
inline int32_t kr_layout_of(const int32_t* types, int32_t n_keys) {
    int ni = 0;
    while (ni < n_keys && types[ni] != TSQ_BYTES) ni++;
    const bool str_last = ni == n_keys - 1 && types[ni] == TSQ_BYTES;
    if (ni == n_keys && ni >= 1 && ni <= 3) return 1 + ni;
    if (str_last && ni <= 2) return 1 + ni + 4;
    return 0;
}

// `nb` (1..8) low bytes of x at byte position `at` of the record words (no array indexed by a run-time value: those live in scratch
// memory; a piece may straddle two words)
TSQ_HD void kr_put(uint64_t (&w)[4], uint32_t at, uint64_t x, uint32_t nb) {
    if (nb < 8) x &= (1ull << (8u * nb)) - 1ull;
    const uint32_t q = at >> 3, sh = 8u * (at & 7u);
    const uint64_t lo = x << sh, hi = sh ? x >> (64u - sh) : 0ull;
    w[0] |= q == 0 ? lo : 0ull;
    w[1] |= q == 1 ? lo : (q == 0 ? hi : 0ull);
    w[2] |= q == 2 ? lo : (q == 1 ? hi : 0ull);
    w[3] |= q == 3 ? lo : (q == 2 ? hi : 0ull);
}
// the same at a compile-time position (x holds nothing above its bytes; bytes beyond the record are the caller's business)
template <int AT>
TSQ_HD void kr_put_at(uint64_t (&w)[4], uint64_t x) {
    constexpr int Q = AT >> 3, SH = 8 * (AT & 7);
    if constexpr (Q < 4) w[Q] |= x << SH;
    if constexpr (SH != 0 && Q + 1 < 4) w[Q + 1] |= x >> (64 - SH);
}
// 8 bytes from any address: the two aligned words around it, funnel-shifted (the second word is read only when the bytes reach into it)
TSQ_HD uint64_t kr_load8(const uint8_t* p, uint32_t need) {
    const uintptr_t a = (uintptr_t)p;
    const uint64_t* q = (const uint64_t*)(a & ~(uintptr_t)7);
    const uint32_t sh = 8u * (uint32_t)(a & 7u);
    uint64_t v = q[0] >> sh;
    if (sh && (a & 7u) + need > 8u) v |= q[1] << (64u - sh);
    return v;
}
// the key record of row `row`: false = the row has no key (a NULL cell of a join key, selected == 0) or its cells do not fit 32 bytes
// (*toolong).  Any key columns, run-time positions:
TSQ_HD bool kr_record_any(const KrSrc& s, int64_t row, uint64_t (&w)[4], bool* toolong) {
    uint32_t at = 0;
    for (int k = 0; k < s.n_keys; k++) {
        const int c = s.col[k];
        const bool str = s.cs.type[c] == TSQ_BYTES;
        if (tsq_is_null(s.cs.nulls[c], row)) {
            if (!s.keep_nulls) return false;
            const uint32_t room = str ? (s.digest[k] ? TSQ_KR_DIGEST_CELL : 2u) : 9u;  // NilFlag = 0 and zero bytes: the record's bytes are zero already (every value starts with a non-zero flag)
            if (at + room > TSQ_KR_BYTES) { *toolong = true; return false; }
            at += room;
            continue;
        }
        if (str && s.digest[k]) {  // a long string: flag 3, its length, the digest of its bytes
            if (at + TSQ_KR_DIGEST_CELL > TSQ_KR_BYTES) { *toolong = true; return false; }
            const uint64_t n = (uint64_t)(s.cs.offs[c][row + 1] - s.cs.offs[c][row]);
            if (n >> 32) { *toolong = true; return false; }
            kr_put(w, at, 3ull | (n << 8), 5);
            kr_put(w, at + 5, s.digest[k][row], 8);
            at += TSQ_KR_DIGEST_CELL;
            continue;
        }
        if (str) {
            const int64_t o = s.cs.offs[c][row], n = s.cs.offs[c][row + 1] - o;
            if (n > 255 || at + 2 + (uint32_t)n > TSQ_KR_BYTES) { *toolong = true; return false; }
            kr_put(w, at, 2ull | ((uint64_t)n << 8), 2);  // compactBytesFlag (codec.go:233-235), then the length
            at += 2;
            const uint8_t* p = (const uint8_t*)s.cs.data[c] + o;
            for (uint32_t i = 0; i < (uint32_t)n; i += 8) {
                const uint32_t m = (uint32_t)n - i < 8u ? (uint32_t)n - i : 8u;
                kr_put(w, at + i, kr_load8(p + i, m), m);
            }
            at += (uint32_t)n;
        } else {
            if (at + 9 > TSQ_KR_BYTES) { *toolong = true; return false; }
            uint32_t flag;
            const uint64_t x = tsq_key_word(s.cs.data[c], s.cs.type[c], row, &flag);
            kr_put(w, at, flag, 1);
            kr_put(w, at + 1, x, 8);
            at += 9;
        }
    }
    return true;
}
// ... NI 8-byte cells, then a string when STR: every position is a constant
template <int AT>
TSQ_HD bool kr_cell8_at(const KrSrc& s, int k, int64_t row, uint64_t (&w)[4]) {
    const int c = s.col[k];
    if (tsq_is_null(s.cs.nulls[c], row)) return s.keep_nulls != 0;
    uint32_t flag;
    const uint64_t x = tsq_key_word(s.cs.data[c], s.cs.type[c], row, &flag);
    kr_put_at<AT>(w, (uint64_t)flag);
    kr_put_at<AT + 1>(w, x);
    return true;
}
template <int AT, int I>
TSQ_HD void kr_str_chunk_at(const uint8_t* p, uint32_t n, uint64_t (&w)[4]) {
    if constexpr (AT + 2 + I < TSQ_KR_BYTES) {
        if (n > (uint32_t)I) {
            const uint32_t m = n - (uint32_t)I;
            uint64_t v = kr_load8(p + I, m < 8u ? m : 8u);
            if (m < 8u) v &= (1ull << (8u * m)) - 1ull;
            kr_put_at<AT + 2 + I>(w, v);
        }
    }
}
template <int NI, bool STR>
TSQ_HD bool kr_record_fixed(const KrSrc& s, int64_t row, uint64_t (&w)[4], bool* toolong) {
    if constexpr (NI > 0) { if (!kr_cell8_at<0>(s, 0, row, w)) return false; }
    if constexpr (NI > 1) { if (!kr_cell8_at<9>(s, 1, row, w)) return false; }
    if constexpr (NI > 2) { if (!kr_cell8_at<18>(s, 2, row, w)) return false; }
    if constexpr (STR) {
        constexpr int AT = 9 * NI;
        const int c = s.col[NI];
        if (tsq_is_null(s.cs.nulls[c], row)) return s.keep_nulls != 0;
        const int64_t o = s.cs.offs[c][row], n64 = s.cs.offs[c][row + 1] - o;
        if (n64 > 255 || AT + 2 + n64 > TSQ_KR_BYTES) { *toolong = true; return false; }
        const uint32_t n = (uint32_t)n64;
        kr_put_at<AT>(w, 2ull | ((uint64_t)n << 8));
        const uint8_t* p = (const uint8_t*)s.cs.data[c] + o;
        kr_str_chunk_at<AT, 0>(p, n, w);
        kr_str_chunk_at<AT, 8>(p, n, w);
        kr_str_chunk_at<AT, 16>(p, n, w);
        kr_str_chunk_at<AT, 24>(p, n, w);
    }
    return true;
}
TSQ_HD bool kr_record(const KrSrc& s, int64_t row, uint64_t (&w)[4], bool* toolong) {
    w[0] = w[1] = w[2] = w[3] = 0;
    *toolong = false;
    if (s.selected && !s.selected[row]) return false;
    switch (s.layout) {
        case 2: return kr_record_fixed<1, false>(s, row, w, toolong);
        case 3: return kr_record_fixed<2, false>(s, row, w, toolong);
        case 4: return kr_record_fixed<3, false>(s, row, w, toolong);
        case 5: return kr_record_fixed<0, true>(s, row, w, toolong);
        case 6: return kr_record_fixed<1, true>(s, row, w, toolong);
        case 7: return kr_record_fixed<2, true>(s, row, w, toolong);
        default: return kr_record_any(s, row, w, toolong);
    }
}
// the way back (the aggregate's group keys leave the dictionary, k_kd_decode): cell k of a record starts at *at; returns its flag (0: a NULL
// cell) and moves *at behind the cell.  An 8-byte cell: *word; a string: its bytes are rec[*off .. *off + *len)
TSQ_HD uint32_t kr_parse_cell(const uint8_t* rec, uint32_t* at, bool is_str, uint64_t* word, uint32_t* off, uint32_t* len) {
    const uint32_t a = *at, flag = rec[a];
    *word = 0;
    *off = *len = 0;
    if (flag == 0) {  // NilFlag: the cell's fixed part is zero bytes
        *at = a + (is_str ? 2u : 9u);
    } else if (is_str) {
        *len = rec[a + 1];
        *off = a + 2;
        *at = a + 2 + *len;
    } else {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v |= (uint64_t)rec[a + 1 + b] << (8 * b);
        *word = v;
        *at = a + 9;
    }
    return flag;
}
// the 64-bit mix of a record: one multiply per word (the rotation hands the well-mixed high half of a product to the next multiply's
// low bits) and a multiply-xorshift finish — 5 multiplies where four rounds of splitmix64 took 8; the partition comes from the top bits,
// the LDS slot from the low 14, the tag from bits 14..31
TSQ_HD uint64_t kr_rotl32(uint64_t x) { return (x << 32) | (x >> 32); }
TSQ_HD uint64_t kr_hash(const uint64_t (&w)[4]) {
    uint64_t h = (w[0] ^ 0x6A09E667F3BCC908ULL) * 0x9E3779B97F4A7C15ULL;
    h = (w[1] ^ kr_rotl32(h)) * 0xBF58476D1CE4E5B9ULL;
    h = (w[2] ^ kr_rotl32(h)) * 0x94D049BB133111EBULL;
    h = (w[3] ^ kr_rotl32(h)) * 0xD6E8FEB86659FD93ULL;
    h = (h ^ (h >> 32)) * 0xFF51AFD7ED558CCDULL;
    return h ^ (h >> 29);
}

#endif
