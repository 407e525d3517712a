/*
 * sort_rows.cpp — CPU restatement of SortExec / TopNExec row ordering (SURVEY.md §8 f, rank 3).
 * TEST INFRASTRUCTURE ONLY (see oracle.h): the product never links or calls this file.
 *
 * Follows:
 *   chunk.GetCompareFunc / cmpNull / cmpInt64 / cmpUint64 / cmpString / cmpFloat32 / cmpFloat64   util/chunk/compare.go:27-103
 *   types.CompareInt64 / CompareUint64 / CompareFloat64 / CompareString              types/compare.go:22-43,103-123
 *   SortExec.lessRow (ByItems, Desc negates the comparison)                           executor/sort.go:116-131
 *   SortExec.Next: sort.Slice(rowPtrs, keyColumnsLess)                                executor/sort.go:58-78
 *   TopNExec: rows [Offset, Offset+Count) of the sorted order                          executor/sort.go:213-238
 * sort.Slice is not stable and the TopN heap keeps an arbitrary row among equal ones, so the order of rows whose keys all
 * compare equal is unspecified in the reference; this restatement uses a stable sort (one legal outcome) and the parity
 * tests compare the key columns position by position and the rows as multisets inside runs of equal keys.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {
bool is_null(const tsq_col& c, int64_t r) { return c.null_bitmap && ((c.null_bitmap[r >> 3] >> (r & 7)) & 1) == 0; }
int cmp_null(bool l, bool r) { return (l && r) ? 0 : (l ? -1 : 1); }  // compare.go:48-56
template <class T>
int cmp3(T a, T b) { return a < b ? -1 : (a == b ? 0 : 1); }  // types/compare.go: x < y -> -1, x == y -> 0, else 1 (NaN: 1)

int cmp_cell(const tsq_col& c, int64_t i, int64_t j) {
    const bool ln = is_null(c, i), rn = is_null(c, j);
    if (ln || rn) return cmp_null(ln, rn);
    switch (c.type) {
        case TSQ_I64: return cmp3(((const int64_t*)c.data)[i], ((const int64_t*)c.data)[j]);
        case TSQ_U64: return cmp3(((const uint64_t*)c.data)[i], ((const uint64_t*)c.data)[j]);
        case TSQ_BYTES: {  // cmpString (compare.go:71-77): Go's string order = bytes, then the shorter first
            const int64_t li = c.offsets[i + 1] - c.offsets[i], lj = c.offsets[j + 1] - c.offsets[j];
            const int m = memcmp((const uint8_t*)c.data + c.offsets[i], (const uint8_t*)c.data + c.offsets[j], (size_t)std::min(li, lj));
            return m ? (m < 0 ? -1 : 1) : cmp3(li, lj);
        }
        case TSQ_F32: return cmp3((double)((const float*)c.data)[i], (double)((const float*)c.data)[j]);  // compare.go:86-92
        default: return cmp3(((const double*)c.data)[i], ((const double*)c.data)[j]);
    }
}
}  // namespace

extern "C" {

/* lessRow (sort.go:116-131) on rows i and j of the same chunk: -1 / 0 / +1 like the loop's cmp */
int32_t orc_row_compare(const tsq_col* cols, const int32_t* key_col, const int32_t* key_desc, int32_t n_keys, int64_t i, int64_t j) {
    for (int k = 0; k < n_keys; k++) {
        int c = cmp_cell(cols[key_col[k]], i, j);
        if (key_desc[k]) c = -c;
        if (c != 0) return c;
    }
    return 0;
}

/* SortExec: the permutation of row indices (a stable sort: one of the orders sort.Slice may produce) */
void orc_sort_rows(const tsq_col* cols, int64_t nrows, const int32_t* key_col, const int32_t* key_desc, int32_t n_keys, int64_t* perm_out) {
    std::vector<int64_t> p((size_t)nrows);
    for (int64_t i = 0; i < nrows; i++) p[(size_t)i] = i;
    std::stable_sort(p.begin(), p.end(), [&](int64_t a, int64_t b) { return orc_row_compare(cols, key_col, key_desc, n_keys, a, b) < 0; });
    memcpy(perm_out, p.data(), (size_t)nrows * 8);
}
}
