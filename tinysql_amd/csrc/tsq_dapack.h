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

// ---- the geometry of a packed build side, from the one fact the routes need: the key range of its usable rows.
// Host arithmetic shared by da_prepare (tsq_join.hip: one GPU, or the GLOBAL range of a build side sharded over several ranks —
// tsq_join_build_finish_shared) and tests/hostsim (walked for world sizes 1..8 without a GPU).
#define TSQ_DA_PLAN_MIN_BITS 13
#define TSQ_DA_PLAN_MAX_BITS 28       /* byte cells: one multiplicity byte per key of the range */
#define TSQ_DA_PLAN_MAX_BITS_UNIQ 31  /* bit cells: a build side without duplicate keys, COUNT(*) only */
#define TSQ_DA_PLAN_MAX_EBITS 17
#define TSQ_DA_PLAN_MAX_EBITS_UNIQ 20
#define TSQ_DA_PLAN_MIN_PBITS 3
#define TSQ_DA_PLAN_MAX_PBITS 11
struct DaPlan {
    int32_t ok;         // 0: this build side keeps 64-bit words
    int32_t bit_cells;  // one BIT per cell (unique build sides whose range needs 29..31 bits)
    uint32_t pbits, ebits;
    DaDomain dm;
};
// kmin / kmax: smallest / largest usable key in the order of the join (already un-flipped 64-bit cells); usable: rows with a usable
// key (over ALL ranks for a shared build side); count_only: bit cells serve COUNT(*) only; force: skip the density test
// (tsq_join_set_key_packing(FORCE): tests and measurements); pb_override < 0: the default partition count
inline DaPlan tsq_da_plan(uint64_t kmin, uint64_t kmax, uint64_t usable, int skip_high, bool count_only, bool force, int pb_override = -1) {
    DaPlan pl;
    pl.ok = 0;
    pl.bit_cells = 0;
    pl.pbits = pl.ebits = 0;
    pl.dm = DaDomain{0, 0, 0, 0, 0, 0};
    if (usable == 0) return pl;
    const uint64_t range = kmax - kmin;
    const bool bits_mode = (range >> TSQ_DA_PLAN_MAX_BITS) != 0;
    if (bits_mode && ((range >> TSQ_DA_PLAN_MAX_BITS_UNIQ) != 0 || !count_only)) return pl;
    uint32_t b = TSQ_DA_PLAN_MIN_BITS;
    while ((range >> b) != 0) b++;
    // a sparse domain: the images would be mostly zeros (byte cells: at most 32 B of image per build row; bit cells: the same 32 B)
    if (!force && (1ULL << b) > (bits_mode ? 256ULL : 32ULL) * usable) return pl;
    int pb = (int)b - 10 < TSQ_DA_PLAN_MAX_PBITS ? (int)b - 10 : TSQ_DA_PLAN_MAX_PBITS;
    if (pb_override >= 0) pb = pb_override;
    const int max_ebits = bits_mode ? TSQ_DA_PLAN_MAX_EBITS_UNIQ : TSQ_DA_PLAN_MAX_EBITS;
    if (pb < (int)b - max_ebits) pb = (int)b - max_ebits;
    if (pb < TSQ_DA_PLAN_MIN_PBITS) pb = TSQ_DA_PLAN_MIN_PBITS;
    if (pb > TSQ_DA_PLAN_MAX_PBITS) pb = TSQ_DA_PLAN_MAX_PBITS;
    if ((int)b - pb > max_ebits || (int)b - pb < 4) return pl;
    pl.ok = 1;
    pl.bit_cells = bits_mode ? 1 : 0;
    pl.pbits = (uint32_t)pb;
    pl.ebits = b - (uint32_t)pb;
    pl.dm.kmin = kmin;
    pl.dm.range = range;
    pl.dm.b = b;
    pl.dm.s = (b + 1) / 2;
    pl.dm.mask = (uint32_t)((1ULL << b) - 1);
    pl.dm.skip_high = skip_high;
    return pl;
}
// bytes of the images of a plan (what a shared build side all-reduces once per build: tsq_join_build_finish_shared)
inline uint64_t tsq_da_image_bytes(const DaPlan& pl) { return pl.bit_cells ? (1ULL << pl.dm.b) / 8 : (1ULL << pl.dm.b); }
// A build side sharded over several ranks: every rank assembles the images of ITS rows over the GLOBAL range and the images are
// SUMMED element-wise (ncclSum over bytes for byte cells, over 32-bit words for bit cells).  Sums can go wrong in exactly one way
// each, and both are visible in the population of the result:
//   byte cells: a cell that passes 255 wraps and loses 256 -> the bytes add up to fewer than the usable rows
//   bit cells : a key present on two ranks makes a carry; popcount(a + b) = popcount(a) + popcount(b) - carries -> fewer set bits than usable rows
// so `population == usable rows of all ranks` accepts exactly the build sides the single-GPU images kernels accept.
inline bool tsq_da_shared_images_ok(uint64_t population, uint64_t usable_all_ranks) { return population == usable_all_ranks; }

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
