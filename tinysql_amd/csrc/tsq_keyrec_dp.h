// tsq_keyrec_dp.h — the KEY RECORD of a row (tsq_keyrec.h, tsq_keydict.h): host/device-portable (TSQ_HD), so that the CPU suite walks the
// same code through tests/hostsim.  Per key column: a flag byte — 8 / 9 / 5, the classes of tsq_key_word; 2 for a string, as
// util/codec/codec.go:233-235; 0 (NilFlag) for a NULL cell of a GROUP BY key — followed by the 8-byte word, or by a length byte and the
// string's bytes; zero padding to 32 bytes.  Two rows have equal keys (codec.EqualChunkRow, codec.go:363-382 / the encoded group key,
// codec.go:700-760) iff their records are equal byte for byte.
#ifndef TSQ_KEYREC_DP_H
#define TSQ_KEYREC_DP_H

#include "tsq_device.h"

#define TSQ_KR_BYTES 32

struct KrSrc {
    tsq_colset cs;
    int32_t n_keys;
    int32_t col[TSQ_MAX_KEYS];
    int32_t keep_nulls;      // GROUP BY: a NULL cell is a key like any other (one NilFlag byte, codec.go:718-719) — a join drops the row
    const uint8_t* selected; // nullptr or one byte per row: 0 = the row has no key (an outer-side filter said no, join.go:344)
    int64_t nrows;
};
#define TSQ_KR_MAXPAY 4
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
// 8 bytes from any address: the two aligned words around it, funnel-shifted (the second word is read only when the bytes reach into it)
TSQ_HD uint64_t kr_load8(const uint8_t* p, uint32_t need) {
    const uintptr_t a = (uintptr_t)p;
    const uint64_t* q = (const uint64_t*)(a & ~(uintptr_t)7);
    const uint32_t sh = 8u * (uint32_t)(a & 7u);
    uint64_t v = q[0] >> sh;
    if (sh && (a & 7u) + need > 8u) v |= q[1] << (64u - sh);
    return v;
}
// the key record of row `row`: false = the row has no key (a NULL cell) or its cells do not fit 32 bytes (*toolong)
TSQ_HD bool kr_record(const KrSrc& s, int64_t row, uint64_t (&w)[4], bool* toolong) {
    w[0] = w[1] = w[2] = w[3] = 0;
    uint32_t at = 0;
    *toolong = false;
    if (s.selected && !s.selected[row]) return false;
    for (int k = 0; k < s.n_keys; k++) {
        const int c = s.col[k];
        if (tsq_is_null(s.cs.nulls[c], row)) {
            if (!s.keep_nulls) return false;
            if (at + 1 > TSQ_KR_BYTES) { *toolong = true; return false; }
            at += 1;  // NilFlag = 0: the record's bytes are zero already (every other cell starts with a non-zero flag)
            continue;
        }
        if (s.cs.type[c] == TSQ_BYTES) {
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
TSQ_HD uint64_t kr_hash(const uint64_t (&w)[4]) {
    uint64_t h = tsq_splitmix64(w[0] ^ 0x6A09E667F3BCC908ULL);
    h = tsq_splitmix64(h ^ w[1]);
    h = tsq_splitmix64(h ^ w[2]);
    return tsq_splitmix64(h ^ w[3]);
}


#endif
