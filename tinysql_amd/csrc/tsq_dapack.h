// tsq_dapack.h — the ARITHMETIC of packed keys (tsq_dajoin.h, tsq_daagg.h), usable on the host: no HIP in here.
//
// A key travels as a bijective mix of (key - kmin) on b bits; several key columns are composed into one key first.  Both facts are
// what the packed routes rest on — equal words <=> equal keys — so tests/hostsim walks them on the CPU: the mix is a bijection of
// [0, 2^b) for every b the routes use (13..31), unmix inverts it, and two rows get the same composite exactly when all their key
// cells are equal and inside the fields.
#ifndef TSQ_DAPACK_H
#define TSQ_DAPACK_H

#include <cstdint>

#ifndef TSQ_HD
#if defined(__HIPCC__)
#define TSQ_HD __host__ __device__ inline
#else
#define TSQ_HD inline
#endif
#endif

struct DaDomain {
    uint64_t kmin;      // smallest usable build key (the 64-bit cell; order: signed for BIGINT, unsigned for BIGINT UNSIGNED)
    uint64_t range;     // kmax - kmin, wrapping subtraction (exact in both orders)
    uint32_t b;         // domain bits: range < 2^b
    uint32_t s;         // xorshift distance
    uint32_t mask;      // 2^b - 1
    int32_t skip_high;  // BIGINT against BIGINT UNSIGNED: a cell >= 2^63 never matches (flag 8 vs 9, codec.go:219-224)
};
// bijection on [0, 2^b): xorshift, odd multiplier mod 2^b, xorshift
TSQ_HD uint32_t tsq_da_mix(uint32_t d, uint32_t s, uint32_t mask) {
    d ^= d >> s;
    d = (d * 0x9E3779B1u) & mask;
    d ^= d >> s;
    return d;
}
TSQ_HD uint32_t tsq_da_unmix(uint32_t u, uint32_t s, uint32_t mask) {
    for (uint32_t t = u >> s; t; t >>= s) u ^= t;  // inverse of x ^= x >> s
    u = (u * 0x0E8B2F51u) & mask;                  // 0x9E3779B1 * 0x0E8B2F51 = 1 mod 2^32
    for (uint32_t t = u >> s; t; t >>= s) u ^= t;
    return u;
}

// ---- several key columns -> one composite key (k_da_compose, tsq_dajoin.h)
#define TSQ_DA_MAXKEYS 4
struct DaFields {
    int32_t n;
    uint64_t kmin[TSQ_DA_MAXKEYS], maxd[TSQ_DA_MAXKEYS];
    uint32_t shift[TSQ_DA_MAXKEYS];
    int32_t skip_high[TSQ_DA_MAXKEYS];
};
// the composite of one row's key cells (none of them NULL): sum (k_i - kmin_i) << shift_i, or ~0 when a cell lies outside its field
// or is a cell >= 2^63 of a column compared across signedness — the row cannot match any build row
TSQ_HD uint64_t tsq_da_compose_cells(const DaFields& f, const uint64_t* key) {
    uint64_t w = 0;
    bool ok = true;
    for (int k = 0; k < TSQ_DA_MAXKEYS; k++) {
        if (k < f.n) {
            const uint64_t d = key[k] - f.kmin[k];
            ok = ok && !(f.skip_high[k] && (key[k] >> 63)) && d <= f.maxd[k];
            w |= d << f.shift[k];
        }
    }
    return ok ? w : ~0ull;
}

#endif
