// tsq_sort_image.h — the order-preserving 64-bit image of an ORDER BY key cell (tsq_sort.hip), TSQ_HD so that the CPU
// test-suite checks it against the oracle's comparators through tests/hostsim.
// image(a) < image(b)  <=>  chunk.GetCompareFunc's comparator says a < b (util/chunk/compare.go:58-103: cmpInt64,
// cmpUint64, cmpFloat32 widened to float64, cmpFloat64); equal images <=> the comparator says equal (-0.0 == +0.0).
// NaN (CompareFloat64 answers "greater" whenever one is involved, types/compare.go:104-112) sorts after +inf.
// DESC inverts the image (SortExec.lessRow negates the comparison, executor/sort.go:121-123).  NULL is not part of the
// image: it is its own, most significant digit.
#ifndef TSQ_SORT_IMAGE_H
#define TSQ_SORT_IMAGE_H

#include "tsq_device.h"

TSQ_HD uint64_t tsq_sort_image(const void* data, int32_t type, int32_t desc, uint64_t row) {
    uint64_t u;
    if (type == TSQ_I64) u = ((const uint64_t*)data)[row] ^ 0x8000000000000000ULL;
    else if (type == TSQ_U64) u = ((const uint64_t*)data)[row];
    else {
        const double f = type == TSQ_F32 ? (double)((const float*)data)[row] : ((const double*)data)[row];
        const uint64_t b = tsq_f64_bits(f);
        // -0.0 == +0.0 for CompareFloat64: one image, so that such rows stay in input order like every other tie
        u = f != f ? ~0ull : (f == 0.0 ? 0x8000000000000000ULL : ((b >> 63) ? ~b : (b | 0x8000000000000000ULL)));
    }
    return desc ? ~u : u;
}

// A STRING key (chunk.GetCompareFunc's cmpString: byte-wise, a prefix sorts before the longer string — Go's string order,
// util/chunk/compare.go:71-77, types.CompareString types/compare.go:115-123) has no 64-bit image; it is a SEQUENCE of images, most significant first:
//     chunk 0 = bytes [0, 8) big endian, zero padded;  chunk 1 = bytes [8, 16);  ...  chunk m - 1;  then the length
// Zero padding makes a string and the same string followed by zero bytes look alike chunk by chunk — they differ exactly in their
// lengths, and the shorter one is the smaller: the length is the last (least significant) image.  The stable least-significant-digit
// sort processes the sequence backwards (length first, chunk 0 last).  chunk < 0 selects the length image.
TSQ_HD uint64_t tsq_sort_image_str(const uint8_t* data, const int64_t* offs, int32_t chunk, int32_t desc, uint64_t row) {
    const int64_t lo = offs[row], len = offs[row + 1] - lo;
    uint64_t u = 0;
    if (chunk < 0) {
        u = (uint64_t)len;
    } else {
        const int64_t at = (int64_t)chunk * 8;
        for (int i = 0; i < 8; i++) u = (u << 8) | (at + i < len ? (uint64_t)data[lo + at + i] : 0ull);
    }
    return desc ? ~u : u;
}

#endif
