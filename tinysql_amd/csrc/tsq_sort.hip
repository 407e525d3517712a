// tsq_sort.hip — ORDER BY / TopN on the GPU (gfx950).  SURVEY.md §8 (f) rank 3.
//
// Replaces SortExec (executor/sort.go:27-144: fetchRowChunks + sort.Slice(rowPtrs, keyColumnsLess) over
// chunk.GetCompareFunc comparators, util/chunk/compare.go:27-103) and TopNExec (sort.go:146-318: rows
// [Offset, Offset + Count) of that order, kept in a heap there).
//
// The row order of `ORDER BY k1 [DESC], k2 [DESC], ...` is a lexicographic order, so it is produced by stable
// least-significant-digit radix passes over (key image, row id) pairs: key columns from the LAST to the first, and
// inside a column the 8 bytes of an order-preserving 64-bit image of the value (ints: sign bit flipped; unsigned:
// as is; reals: the memcomparable image of the double, float32 widened like cmpFloat32; DESC: image inverted), then
// one pass on the NULL flag (NULL is smaller than every value, cmpNull compare.go:48-56; DESC puts it last).
//   K14a k_sort_image    : image[i] = f(column[row_id[i]]) for the key column being processed
//                          + the OR and the AND of all images -> passes whose digit is the same for every row are
//                          skipped (day numbers, small ints: 2-3 passes instead of 8)
//   K14c k_sort_tilehist : per 4096-row tile digit histogram            (one pass = K14c + K14d + K14e)
//   K14d k_sort_scan     : digit-major exclusive scan (one workgroup per digit, then the 256 digit totals)
//   K14e k_sort_scatter  : stable ranks without atomics — every wave owns a contiguous quarter of the tile, matches equal
//                          digits with 8 ballots, and keeps its running digit counts in LDS; the tile is laid out sorted in
//                          LDS and every digit run is written with consecutive lanes on consecutive addresses
//   K14f k_gather_rows   : output columns = input columns gathered through the final row ids (data + packed null bitmap)
// Equal keys keep their input order (stable), one of the orders sort.Slice may produce (it is not stable; the TopN heap
// is not either): parity is checked on key columns position by position and on rows as multisets within equal-key runs.
// Algorithmic bytes per pass: 12 B read + 12 B written per row (the implementation reads the images twice: 32 B).
#include "tsq_radix.h"
#include "tsq_stage.h"
#include "tsq_sort_image.h"

#include <memory>

#define TSQ_SORT_NT 256
#define TSQ_SORT_K 16
#define TSQ_SORT_T (TSQ_SORT_NT * TSQ_SORT_K)  // rows per tile (measured: 2048 -> 13.4 ms, 4096 -> 8.5 ms, 6144 -> 9.6 ms per 1e8 x 8 passes)
#define TSQ_SORT_NW (TSQ_SORT_NT / 64)

struct SortKeySrc {
    const void* data;
    const uint8_t* nulls;  // bit 1 = NOT NULL, or null
    int32_t type;
    int32_t desc;
    const int64_t* offs;   // a var-len (string) key: its offsets; the image is then chunk `chunk` of the cell (tsq_sort_image.h)
    int32_t chunk;         // >= 0: bytes [8 chunk, 8 chunk + 8); -1: the length
};

__device__ __forceinline__ uint64_t sort_image(const SortKeySrc& k, uint32_t row) {
    if (k.type == TSQ_BYTES) return tsq_sort_image_str((const uint8_t*)k.data, k.offs, k.chunk, k.desc, row);
    return tsq_sort_image(k.data, k.type, k.desc, row);
}

// longest cell of a var-len key column (the number of 8-byte chunks its images need)
__global__ void __launch_bounds__(256) k_sort_str_maxlen(const int64_t* offs, int64_t n, unsigned long long* out) {
    unsigned long long m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long l = (unsigned long long)(offs[i + 1] - offs[i]);
        m = l > m ? l : m;
    }
    for (int o = 32; o; o >>= 1) {
        const unsigned long long y = __shfl_xor(m, o, 64);
        m = y > m ? y : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

struct SortArgs {
    SortKeySrc key;
    int64_t n;
    const uint64_t* img_in;
    const uint32_t* idx_in;  // null: identity
    uint64_t* img_out;
    uint32_t* idx_out;
    int32_t digit;           // 0..7: byte of the image; 8: the NULL flag of key.nulls (through idx_in)
    int64_t ntiles;
    uint32_t* hist;          // [256][ntiles]
    uint32_t* totals;        // [256] row totals, then exclusive bases
    unsigned long long* orand;   // [0] = OR of all images, [1] = AND
};

__device__ __forceinline__ uint32_t sort_digit(const SortArgs& a, uint64_t img, uint32_t row) {
    if (a.digit < 8) return (uint32_t)(img >> (8 * a.digit)) & 255u;
    const uint32_t notnull = (a.key.nulls[row >> 3] >> (row & 7)) & 1u;
    return a.key.desc ? 1u - notnull : notnull;  // ASC: NULL first; DESC: NULL last (sort.go:121-123 negates cmpNull)
}

// Also reduces the bitwise OR and AND of all images (orand[0], orand[1]): byte d is the same in every row exactly when
// the two agree on it, and then the pass on digit d would not move anything.  (Eight digit histograms for the same
// decision cost 2 ms per 1e8 rows in LDS atomics.)
__global__ void __launch_bounds__(256) k_sort_image(SortArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long vor = 0, vand = ~0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        const uint32_t row = a.idx_in ? a.idx_in[i] : (uint32_t)i;
        const uint64_t v = sort_image(a.key, row);
        a.img_out[i] = v;
        vor |= v;
        vand &= v;
        if (!a.idx_in) a.idx_out[i] = (uint32_t)i;
    }
    for (int o = 32; o; o >>= 1) {
        vor |= __shfl_xor(vor, o, 64);
        vand &= __shfl_xor(vand, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicOr(&a.orand[0], vor);
        atomicAnd(&a.orand[1], vand);
    }
}

__global__ void __launch_bounds__(TSQ_SORT_NT) k_sort_tilehist(SortArgs a) {
    __shared__ uint32_t s_h[256];
    for (int64_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        s_h[threadIdx.x] = 0;
        __syncthreads();
        const int64_t base = t * TSQ_SORT_T;
#pragma unroll 4
        for (int j = 0; j < TSQ_SORT_K; j++) {
            const int64_t i = base + (int64_t)j * TSQ_SORT_NT + threadIdx.x;
            if (i < a.n) atomicAdd(&s_h[sort_digit(a, a.digit < 8 ? a.img_in[i] : 0ull, a.digit < 8 ? 0u : a.idx_in[i])], 1u);
        }
        __syncthreads();
        a.hist[(int64_t)threadIdx.x * a.ntiles + t] = s_h[threadIdx.x];
        __syncthreads();
    }
}

// workgroup d: exclusive scan of hist[d][0..ntiles) in place, totals[d] = row sum
__global__ void __launch_bounds__(1024) k_sort_scan_rows(SortArgs a) {
    __shared__ uint32_t s_wsum[16];
    uint32_t* row = a.hist + (int64_t)blockIdx.x * a.ntiles;
    uint32_t carry = 0;
    for (int64_t c0 = 0; c0 < a.ntiles; c0 += 1024) {
        const int64_t i = c0 + threadIdx.x;
        const uint32_t v = i < a.ntiles ? row[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan<1024>(v, s_wsum, &total);
        if (i < a.ntiles) row[i] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.totals[blockIdx.x] = carry;
}
__global__ void __launch_bounds__(256) k_sort_scan_totals(SortArgs a) {
    __shared__ uint32_t s_wsum[4];
    uint32_t total;
    const uint32_t ex = block_excl_scan<256>(a.totals[threadIdx.x], s_wsum, &total);
    a.totals[threadIdx.x] = ex;
}

__global__ void __launch_bounds__(TSQ_SORT_NT) k_sort_scatter(SortArgs a) {
    __shared__ uint64_t s_img[TSQ_SORT_T];
    __shared__ uint32_t s_idx[TSQ_SORT_T];
    __shared__ uint32_t s_whist[TSQ_SORT_NW][256];  // running digit counts of each wave, then its exclusive base
    __shared__ uint32_t s_start[256];               // first position of digit d in the sorted tile
    __shared__ uint32_t s_gbase[256];               // global position of that first element
    __shared__ uint32_t s_wsum[TSQ_SORT_NT / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int64_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int64_t base = t * TSQ_SORT_T;
#pragma unroll
        for (int q = 0; q < TSQ_SORT_NW; q++) s_whist[q][tid] = 0;
        s_gbase[tid] = a.totals[tid] + a.hist[(int64_t)tid * a.ntiles + t];
        __syncthreads();
        uint64_t img[TSQ_SORT_K];
        uint32_t idx[TSQ_SORT_K], rk[TSQ_SORT_K];  // rk = digit << 16 | rank inside the wave's quarter
        // wave w owns rows [w*1024, (w+1)*1024) of the tile, 64 consecutive rows per step: memory order = rank order
#pragma unroll
        for (int j = 0; j < TSQ_SORT_K; j++) {
            const int64_t i = base + (int64_t)w * (64 * TSQ_SORT_K) + j * 64 + lane;
            const bool ok = i < a.n;
            img[j] = ok ? a.img_in[i] : 0ull;
            idx[j] = ok ? a.idx_in[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < TSQ_SORT_K; j++) {
            const int64_t i = base + (int64_t)w * (64 * TSQ_SORT_K) + j * 64 + lane;
            const bool ok = i < a.n;
            const uint32_t d = ok ? sort_digit(a, img[j], idx[j]) : 0u;
            unsigned long long peers = __ballot(ok);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const uint32_t before = (uint32_t)__popcll(peers & lt);
            const uint32_t prev = s_whist[w][d];  // every lane of the match group reads before its first lane adds
            if (ok && before == 0) s_whist[w][d] = prev + (uint32_t)__popcll(peers);
            rk[j] = ok ? ((d << 16) | (prev + before)) : 0xffffffffu;
        }
        __syncthreads();
        {   // digit totals of the tile -> exclusive start of every digit; per-wave exclusive bases
            uint32_t run = 0;
#pragma unroll
            for (int q = 0; q < TSQ_SORT_NW; q++) {
                const uint32_t c = s_whist[q][tid];
                s_whist[q][tid] = run;
                run += c;
            }
            uint32_t total;
            s_start[tid] = block_excl_scan<TSQ_SORT_NT>(run, s_wsum, &total);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TSQ_SORT_K; j++) {
            if (rk[j] != 0xffffffffu) {
                const uint32_t d = rk[j] >> 16;
                const uint32_t pos = s_start[d] + s_whist[w][d] + (rk[j] & 0xffffu);
                s_img[pos] = img[j];
                s_idx[pos] = idx[j];
            }
        }
        __syncthreads();
        const int64_t left = a.n - base;
        const uint32_t nt = left < TSQ_SORT_T ? (uint32_t)left : (uint32_t)TSQ_SORT_T;
        for (uint32_t i = tid; i < nt; i += TSQ_SORT_NT) {
            const uint64_t v = s_img[i];
            const uint32_t r = s_idx[i];
            const uint32_t d = sort_digit(a, v, r);
            const uint32_t dst = s_gbase[d] + (i - s_start[d]);
            a.img_out[dst] = v;
            a.idx_out[dst] = r;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- TopN: radix select before the sort
// TopNExec needs rows [Offset, Offset + Count) of the order only (sort.go:213-238; a heap there).  With K = Offset + Count
// much smaller than the input, the K-th smallest key of the FIRST ORDER BY item is found by a most-significant-digit radix
// select on (NULL flag, image) — one 256-bin histogram pass per digit over the rows that still match the decided prefix,
// no data movement — until the bucket that holds rank K is small; the rows whose key is <= that bucket (a superset of
// the answer, ties included) are compacted IN INPUT ORDER (so that ties stay stable) and only they are sorted.
struct SelArgs {
    SortKeySrc key;
    int64_t n;
    const uint64_t* img;        // images of the first ORDER BY item, identity order
    uint64_t mask, val;         // decided high digits of the threshold: candidates match (img & mask) == val at this level
    int32_t digit;              // 8: histogram of the NULL-flag level (2 bins); 0..7: histogram of that byte
    int32_t null_level;         // level of the threshold on the NULL flag (0 or 1), -1: the column has no NULLs
    unsigned long long* hist;   // [256]
    unsigned long long* block_cnt;  // per-workgroup candidate counts -> exclusive bases (compaction)
    int64_t rows_per_block;
    uint32_t* idx_out;
};
__device__ __forceinline__ uint32_t sel_null_level(const SelArgs& a, int64_t i) {
    const uint32_t notnull = (a.key.nulls[i >> 3] >> (i & 7)) & 1u;
    return a.key.desc ? 1u - notnull : notnull;
}
__global__ void __launch_bounds__(256) k_select_hist(SelArgs a) {
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        if (a.digit == 8) {
            atomicAdd(&s_h[sel_null_level(a, i)], 1u);
        } else {
            if (a.null_level >= 0 && sel_null_level(a, i) != (uint32_t)a.null_level) continue;
            const uint64_t v = a.img[i];
            if ((v & a.mask) != a.val) continue;
            atomicAdd(&s_h[(v >> (8 * a.digit)) & 255u], 1u);
        }
    }
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&a.hist[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}
// candidate: key <= threshold bucket, i.e. NULL level below the threshold's, or the same level and decided digits <= val
__device__ __forceinline__ bool sel_is_candidate(const SelArgs& a, int64_t i) {
    if (a.null_level >= 0) {
        const uint32_t lv = sel_null_level(a, i);
        if (lv != (uint32_t)a.null_level) return lv < (uint32_t)a.null_level;
    }
    return (a.img[i] & a.mask) <= a.val;
}
__global__ void __launch_bounds__(256) k_select_count(SelArgs a) {
    __shared__ unsigned int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
    unsigned int c = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) c += sel_is_candidate(a, i) ? 1u : 0u;
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0) a.block_cnt[blockIdx.x] = s_n;
}
__global__ void __launch_bounds__(1024) k_select_scan(unsigned long long* v, int n, unsigned long long* total) {
    __shared__ uint32_t s_wsum[16];
    const int per = (n + 1023) / 1024, lo = threadIdx.x * per;
    uint32_t sum = 0;
    for (int i = lo; i < lo + per && i < n; i++) sum += (uint32_t)v[i];
    uint32_t tot;
    uint32_t run = block_excl_scan<1024>(sum, s_wsum, &tot);
    for (int i = lo; i < lo + per && i < n; i++) {
        const uint32_t c = (uint32_t)v[i];
        v[i] = run;
        run += c;
    }
    if (threadIdx.x == 0) *total = tot;
}
// order-preserving: a workgroup walks its contiguous rows 256 at a time, positions by block scan of the flags
__global__ void __launch_bounds__(256) k_select_scatter(SelArgs a) {
    __shared__ uint32_t s_wsum[4];
    const int64_t lo = (int64_t)blockIdx.x * a.rows_per_block;
    const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
    uint32_t base = (uint32_t)a.block_cnt[blockIdx.x];
    for (int64_t i0 = lo; i0 < hi; i0 += 256) {
        const int64_t i = i0 + threadIdx.x;
        const uint32_t f = (i < hi && sel_is_candidate(a, i)) ? 1u : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan<256>(f, s_wsum, &tot);
        if (f) a.idx_out[base + ex] = (uint32_t)i;
        base += tot;
        __syncthreads();
    }
}

struct GatherRowsArgs {
    const uint32_t* idx;  // row ids, already offset to the first wanted row
    int64_t rows;
    const void* src[TSQ_MAX_COLS];
    const uint8_t* src_nulls[TSQ_MAX_COLS];
    void* dst[TSQ_MAX_COLS];
    uint8_t* dst_bitmap[TSQ_MAX_COLS];
    int32_t es[TSQ_MAX_COLS];
    // The first ORDER BY column, when it is an integer column, is not gathered at all: its sorted images are still there
    // (consecutive reads) and the value is the inverse of the image.  (Reals are gathered: -0.0 and NaN payloads are not
    // recoverable from the image.)
    const uint64_t* key_img;  // images in output order, already offset; null: gather every column
    int32_t key_col, key_desc;
    uint64_t key_flip;        // 0x8000... for a signed column, 0 for an unsigned one
};
// blockIdx.y = column; eight consecutive output rows per thread (independent gathers, one bitmap byte)
__global__ void __launch_bounds__(256) k_gather_rows(GatherRowsArgs a) {
    const int c = blockIdx.y;
    const int64_t groups = (a.rows + 7) >> 3;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r0 = g << 3;
        const int n = a.rows - r0 < 8 ? (int)(a.rows - r0) : 8;
        uint32_t id[8], nn = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) id[i] = i < n ? a.idx[r0 + i] : 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            bool ok = i < n;
            if (ok && a.src_nulls[c]) ok = (a.src_nulls[c][id[i] >> 3] >> (id[i] & 7)) & 1;
            nn |= ok ? (1u << i) : 0u;
        }
        if (a.key_img && c == a.key_col) {  // I64: image ^ sign bit, U64: the image itself (tsq_sort_image.h), DESC inverted
            const uint64_t flip = (a.key_desc ? ~0ull : 0ull) ^ a.key_flip;
            for (int i = 0; i < n; i++) ((uint64_t*)a.dst[c])[r0 + i] = (nn >> i) & 1 ? (a.key_img[r0 + i] ^ flip) : 0ull;
        } else if (a.es[c] == 0) {
            // a var-len column: its bytes are gathered by k_sort_var_* (lengths -> scan -> copy); only the NULL flags here
        } else if (a.es[c] == 8) {
            uint64_t v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (nn >> i) & 1 ? ((const uint64_t*)a.src[c])[id[i]] : 0ull;
            for (int i = 0; i < n; i++) ((uint64_t*)a.dst[c])[r0 + i] = v[i];
        } else {
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (nn >> i) & 1 ? ((const uint32_t*)a.src[c])[id[i]] : 0u;
            for (int i = 0; i < n; i++) ((uint32_t*)a.dst[c])[r0 + i] = v[i];
        }
        a.dst_bitmap[c][g] = (uint8_t)nn;
    }
}

// K14g — var-len (string) PAYLOAD columns of the ordered rows (Chunk.AppendRow of a var-len cell, util/chunk/chunk.go:334-356;
// ORDER BY keys are fixed width): the lengths of the cells in output order, their exclusive scan = the output offsets
// (tsq_launch_scan64), then the bytes — one cell per lane, or one per wave when the cells are long.
struct SortVarArgs {
    const uint32_t* idx;
    int64_t rows;
    const int64_t* src_offs;
    const uint8_t* src_data;
    const uint8_t* src_nulls;
    int64_t* out_offs;  // [rows + 1]
    uint8_t* out_data;
};
__global__ void __launch_bounds__(256) k_sort_var_len(SortVarArgs a) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.rows; r += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t id = a.idx[r];
        const bool ok = !a.src_nulls || ((a.src_nulls[id >> 3] >> (id & 7)) & 1);
        a.out_offs[r] = ok ? a.src_offs[id + 1] - a.src_offs[id] : 0;  // a NULL cell has no bytes
    }
}
template <bool WAVE>
__global__ void __launch_bounds__(256) k_sort_var_copy(SortVarArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = WAVE ? gtid >> 6 : gtid, step = WAVE ? nthr >> 6 : nthr;
    for (int64_t r = first; r < a.rows; r += step) {
        const int64_t n = a.out_offs[r + 1] - a.out_offs[r];
        if (n == 0) continue;
        const uint8_t* s = a.src_data + a.src_offs[a.idx[r]];
        uint8_t* d = a.out_data + a.out_offs[r];
        if (!WAVE) {
            tsq_copy_cell(d, s, n);
        } else {  // head up to an 8-byte boundary of the destination, then 8 bytes per lane, then the tail
            int64_t head = (8 - ((uintptr_t)d & 7)) & 7;
            head = head < n ? head : n;
            if (lane < head) d[lane] = s[lane];
            const int64_t words = (n - head) >> 3;
            for (int64_t w = lane; w < words; w += 64) {
                uint64_t x;
                memcpy(&x, s + head + w * 8, 8);  // the source is not aligned with the destination
                *reinterpret_cast<uint64_t*>(d + head + w * 8) = x;
            }
            const int64_t done = head + words * 8;
            if (done + lane < n) d[done + lane] = s[done + lane];
        }
    }
}

// ====================================================================== host side
#define TSQ_MAGIC_SORT 0x74737153u /* 'tsqS' */

struct tsq_sort {
    tsq_handle_hdr hdr;
    tsq_ctx* ctx = nullptr;
    tsq_sort_cfg cfg;
    std::atomic<int> cancelled{0};
    std::vector<ColStore> cols;
    HostStage stage;
    bool finished = false;
    int64_t n = 0, first = 0, last = 0, cursor = 0;  // output rows [first, last) of the sorted order
    DevBuf img[2], idx[2], hist, totals, count8;
    int cur = 0;                                       // which of idx[] holds the final row ids
    std::vector<DevBuf> odata, obm, ooffs;             // gather targets for host pulls (ooffs: offsets of var-len columns)
    std::vector<DevBuf> voffs;                         // var-len columns: the offsets tsq_sort_peek computed for the next pull
    std::vector<int64_t> vbytes;                       // ... and their data bytes
    int64_t peek_cursor = -1, peek_rows = 0;
    DevBuf scratch;
    int32_t passes = 0, passes_skipped = 0;
    bool key_img_kept = false;
    int64_t rows_sorted = 0;                           // rows that went through the radix passes (TopN: the selected candidates)
    double sort_ms = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
};

namespace {
tsq_status sort_cancelled(tsq_sort* s) {
    if (s->cancelled.load()) return tsq_fail(&s->hdr, TSQ_ERR_CANCELLED, "sort cancelled");
    return TSQ_OK;
}
tsq_status sort_flush(tsq_sort* s) {
    HostStage& sg = s->stage;
    if (sg.staged == 0) return TSQ_OK;
    DevBuf tmp, tmp_offs;
    for (size_t c = 0; c < s->cols.size(); c++) {
        tsq_status st = sg.append_to(s->ctx, &s->hdr, (int)c, s->cols[c], tmp, tmp_offs);
        if (st != TSQ_OK) { tmp.release(); tmp_offs.release(); return st; }
    }
    hipError_t e = hipStreamSynchronize(s->ctx->stream);  // staging memory is reused
    tmp.release();
    tmp_offs.release();
    sg.reset();
    if (e != hipSuccess) return tsq_fail(&s->hdr, TSQ_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return TSQ_OK;
}

// one stable pass on `digit` of the current images / row ids
tsq_status sort_pass(tsq_sort* s, SortArgs& a, int digit) {
    tsq_ctx* ctx = s->ctx;
    a.digit = digit;
    a.img_in = s->img[s->cur].as<uint64_t>();
    a.idx_in = s->idx[s->cur].as<uint32_t>();
    a.img_out = s->img[s->cur ^ 1].as<uint64_t>();
    a.idx_out = s->idx[s->cur ^ 1].as<uint32_t>();
    const int grid = (int)std::min<int64_t>(a.ntiles, (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(k_sort_tilehist, dim3(grid), dim3(TSQ_SORT_NT), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_sort_scan_rows, dim3(256), dim3(1024), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_sort_scan_totals, dim3(1), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(k_sort_scatter, dim3(grid), dim3(TSQ_SORT_NT), 0, ctx->stream, a);
    TSQ_HIP(&s->hdr, hipGetLastError());
    s->cur ^= 1;
    s->passes++;
    return TSQ_OK;
}
// offsets (and data bytes) of the var-len columns for the next n output rows, cached for the pull that follows a peek
tsq_status sort_var_offsets(tsq_sort* s, int64_t n) {
    if (s->peek_cursor == s->cursor && s->peek_rows == n) return TSQ_OK;
    tsq_ctx* ctx = s->ctx;
    tsq_handle_hdr* h = &s->hdr;
    const int n_cols = s->cfg.n_cols;
    s->voffs.resize(n_cols);
    s->vbytes.assign(n_cols, 0);
    int slot = 0;
    for (int c = 0; c < n_cols; c++) {
        if (s->cfg.col_types[c] != TSQ_BYTES) continue;
        TSQ_TRY(s->voffs[c].reserve(ctx, h, ((size_t)n + 1) * 8 + 64));
        SortVarArgs va;
        memset(&va, 0, sizeof va);
        va.idx = s->idx[s->cur].as<uint32_t>() + s->cursor;
        va.rows = n;
        va.src_offs = s->cols[c].offs.as<int64_t>();
        va.src_nulls = s->cols[c].has_nulls ? s->cols[c].nulls.as<uint8_t>() : nullptr;
        va.out_offs = s->voffs[c].as<int64_t>();
        hipLaunchKernelGGL(k_sort_var_len, dim3(tsq_grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, va);
        TSQ_HIP(h, hipGetLastError());
        TSQ_TRY(tsq_launch_scan64(ctx, h, va.out_offs, n, s->scratch));
        if (slot >= 8) return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "more than 8 var-len columns in one sort");
        TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 48 + slot, va.out_offs + n, 8, hipMemcpyDeviceToHost, ctx->stream));
        slot++;
    }
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    slot = 0;
    for (int c = 0; c < n_cols; c++)
        if (s->cfg.col_types[c] == TSQ_BYTES) s->vbytes[c] = (int64_t)ctx->pinned[48 + slot++];
    s->peek_cursor = s->cursor;
    s->peek_rows = n;
    return TSQ_OK;
}
}  // namespace

TSQ_API tsq_status tsq_sort_create(tsq_ctx* ctx, const tsq_sort_cfg* cfg, tsq_sort** out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !cfg || !out) return tsq_fail(nullptr, TSQ_ERR_INVALID, "tsq_sort_create: NULL argument");
    *out = nullptr;
    tsq_handle_hdr* ch = &ctx->hdr;
    if (cfg->n_cols < 1 || cfg->n_cols > TSQ_MAX_COLS) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "1..16 columns supported");
    if (cfg->n_keys < 1 || cfg->n_keys > TSQ_MAX_KEYS) return tsq_fail(ch, TSQ_ERR_UNSUPPORTED, "1..4 ORDER BY items supported");
    for (int c = 0; c < cfg->n_cols; c++)
        if (cfg->col_types[c] < TSQ_I64 || cfg->col_types[c] > TSQ_BYTES) return tsq_fail(ch, TSQ_ERR_INVALID, "unknown column type");
    for (int k = 0; k < cfg->n_keys; k++) {
        if (cfg->key_col[k] < 0 || cfg->key_col[k] >= cfg->n_cols) return tsq_fail(ch, TSQ_ERR_INVALID, "ORDER BY column index out of range");
    }
    if (cfg->limit_offset < 0) return tsq_fail(ch, TSQ_ERR_INVALID, "negative offset");
    std::unique_ptr<tsq_sort> s(new tsq_sort());
    s->hdr.magic = TSQ_MAGIC_SORT;
    s->ctx = ctx;
    s->cfg = *cfg;
    s->cols.resize(cfg->n_cols);
    for (int c = 0; c < cfg->n_cols; c++) s->cols[c].type = cfg->col_types[c];
    TSQ_HIP(ch, hipSetDevice(ctx->device));
    for (int i = 0; i < 2; i++) TSQ_HIP(ch, hipEventCreate(&s->ev[i]));
    *out = s.release();
    return TSQ_OK;
}

TSQ_API tsq_status tsq_sort_push(tsq_sort* s, const tsq_col* cols, int32_t n_cols, int64_t nrows) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(s, TSQ_MAGIC_SORT));
    if (!s || s->hdr.magic != TSQ_MAGIC_SORT) return TSQ_ERR_INVALID;
    TSQ_TRY(sort_cancelled(s));
    if (s->finished) return tsq_fail(&s->hdr, TSQ_ERR_INVALID, "push after finish");
    if (nrows < 0 || (!cols && nrows > 0)) return tsq_fail(&s->hdr, TSQ_ERR_INVALID, "bad arguments");
    if (nrows == 0) return TSQ_OK;
    bool dev = false;
    TSQ_TRY(tsq_validate_cols(&s->hdr, cols, n_cols, s->cfg.n_cols, s->cfg.col_types, nrows, &dev));
    TSQ_HIP(&s->hdr, hipSetDevice(s->ctx->device));
    if (s->cols[0].rows + s->stage.staged + nrows >= 0xfffffff0LL) return tsq_fail(&s->hdr, TSQ_ERR_UNSUPPORTED, "more than 2^32 rows per GPU");
    if (dev) {
        TSQ_TRY(sort_flush(s));
        DevBuf tmp, tmp_offs;
        for (int c = 0; c < n_cols; c++) {
            tsq_status st = cols[c].type == TSQ_BYTES
                                ? tsq_col_append_varlen(s->ctx, &s->hdr, s->cols[c], cols[c].data, cols[c].offsets, cols[c].null_bitmap, nrows, true, tmp, tmp_offs)
                                : tsq_col_append(s->ctx, &s->hdr, s->cols[c], cols[c].data, cols[c].null_bitmap, nrows, true, tmp);
            if (st != TSQ_OK) { tmp.release(); tmp_offs.release(); return st; }
        }
        tmp.release();
        tmp_offs.release();
        return TSQ_OK;
    }
    if (s->stage.cap == 0) TSQ_TRY(s->stage.init(&s->hdr, n_cols, s->cfg.col_types, 1 << 20));
    int64_t off = 0;
    while (off < nrows) {
        const int64_t n = std::min<int64_t>(nrows - off, s->stage.room());
        s->stage.add(cols, off, n, nullptr);
        off += n;
        if (s->stage.room() == 0) TSQ_TRY(sort_flush(s));
    }
    return TSQ_OK;
}

TSQ_API tsq_status tsq_sort_finish(tsq_sort* s) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(s, TSQ_MAGIC_SORT));
    if (!s || s->hdr.magic != TSQ_MAGIC_SORT) return TSQ_ERR_INVALID;
    TSQ_TRY(sort_cancelled(s));
    if (s->finished) return TSQ_OK;
    tsq_ctx* ctx = s->ctx;
    tsq_handle_hdr* h = &s->hdr;
    TSQ_HIP(h, hipSetDevice(ctx->device));
    TSQ_TRY(sort_flush(s));
    s->stage.release();
    const int64_t n = s->cols[0].rows;
    s->n = n;
    s->first = std::min<int64_t>(n, s->cfg.limit_offset);
    // TopNExec: rows [Offset, Offset + Count) (sort.go:215; uint64 there).  "LIMIT off, 9223372036854775807" must not overflow the sum.
    s->last = (s->cfg.limit_count < 0 || s->cfg.limit_count >= n - s->first) ? n : s->first + s->cfg.limit_count;
    if (s->last < s->first) s->last = s->first;
    s->cursor = s->first;
    s->finished = true;
    if (n == 0 || s->first == s->last) return TSQ_OK;
    SortArgs a;
    memset(&a, 0, sizeof a);
    a.n = n;
    a.ntiles = (n + TSQ_SORT_T - 1) / TSQ_SORT_T;
    for (int i = 0; i < 2; i++) {
        TSQ_TRY(s->img[i].reserve(ctx, h, (size_t)n * 8 + 64));
        TSQ_TRY(s->idx[i].reserve(ctx, h, (size_t)n * 4 + 64));
    }
    TSQ_TRY(s->hist.reserve(ctx, h, std::max<size_t>((size_t)a.ntiles * 256 * 4, 4096 + (size_t)ctx->num_cus * 8 * 8) + 64));
    TSQ_TRY(s->totals.reserve(ctx, h, 256 * 4));
    TSQ_TRY(s->count8.reserve(ctx, h, 64));
    a.hist = s->hist.as<uint32_t>();
    a.totals = s->totals.as<uint32_t>();
    a.orand = s->count8.as<unsigned long long>();
    TSQ_HIP(h, hipEventRecord(s->ev[0], ctx->stream));
    s->cur = 0;
    bool have_idx = false;
    PinnedBuf hcount;
    TSQ_TRY(hcount.reserve(h, 4096));
    int egrid = tsq_grid_for(ctx, n, 256, 4);
    s->rows_sorted = n;
    const int64_t K = s->last;  // rows of the order that are needed at all
    if (s->cfg.limit_count >= 0 && n >= (1 << 20) && K * 16 <= n) {
        // ---- TopN: radix select on the first ORDER BY item, then sort only the candidates (see k_select_hist)
        const int kc = s->cfg.key_col[0];
        a.key.data = s->cols[kc].data.p;
        a.key.nulls = s->cols[kc].has_nulls ? s->cols[kc].nulls.as<uint8_t>() : nullptr;
        a.key.type = s->cfg.col_types[kc];
        a.key.desc = s->cfg.key_desc[0] ? 1 : 0;
        // a string item is selected on its FIRST image (bytes [0, 8), the most significant one): a row whose first eight bytes are
        // beyond those of the K-th row cannot be among the first K; ties on the image stay candidates and the sort decides
        a.key.offs = a.key.type == TSQ_BYTES ? s->cols[kc].offs.as<int64_t>() : nullptr;
        a.key.chunk = 0;
        a.idx_in = nullptr;
        a.img_out = s->img[0].as<uint64_t>();
        a.idx_out = s->idx[1].as<uint32_t>();  // identity row ids: not needed here
        hipLaunchKernelGGL(k_sort_image, dim3(egrid), dim3(256), 0, ctx->stream, a);
        SelArgs sa;
        memset(&sa, 0, sizeof sa);
        sa.key = a.key;
        sa.n = n;
        sa.img = s->img[0].as<uint64_t>();
        sa.hist = (unsigned long long*)s->hist.p;
        sa.null_level = -1;
        const int sgrid = std::min(egrid, ctx->num_cus * 8);
        unsigned long long* hh = (unsigned long long*)hcount.p;
        auto hist_pass = [&](int digit) -> hipError_t {
            sa.digit = digit;
            hipError_t e = hipMemsetAsync(sa.hist, 0, 256 * 8, ctx->stream);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_select_hist, dim3(sgrid), dim3(256), 0, ctx->stream, sa);
            e = hipMemcpyAsync(hh, sa.hist, 256 * 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e != hipSuccess) return e;
            return hipStreamSynchronize(ctx->stream);
        };
        hipError_t e = hipSuccess;
        uint64_t rank = (uint64_t)K, bucket = (uint64_t)n;
        if (sa.key.nulls) {
            e = hist_pass(8);
            if (e == hipSuccess) {
                if (rank <= hh[0]) { sa.null_level = 0; bucket = hh[0]; }
                else { sa.null_level = 1; rank -= hh[0]; bucket = hh[1]; }
            }
        }
        const uint64_t small = std::max<uint64_t>((uint64_t)K * 4, 1u << 16);
        for (int d = 7; d >= 0 && e == hipSuccess && bucket > small; d--) {
            e = hist_pass(d);
            if (e != hipSuccess) break;
            uint64_t cum = 0;
            int b = 0;
            while (b < 255 && cum + hh[b] < rank) cum += hh[b++];
            rank -= cum;
            bucket = hh[b];
            sa.val |= (uint64_t)b << (8 * d);
            sa.mask |= 0xffull << (8 * d);
        }
        if (e == hipSuccess) {  // order-preserving compaction of the candidates into idx[0]
            const int cgrid = std::min<int>(ctx->num_cus * 8, (int)((n + 255) / 256));
            sa.rows_per_block = (((n + cgrid - 1) / cgrid) + 255) & ~(int64_t)255;
            sa.block_cnt = (unsigned long long*)((char*)s->hist.p + 4096);
            sa.idx_out = s->idx[0].as<uint32_t>();
            hipLaunchKernelGGL(k_select_count, dim3(cgrid), dim3(256), 0, ctx->stream, sa);
            hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(1024), 0, ctx->stream, sa.block_cnt, cgrid, sa.hist);
            hipLaunchKernelGGL(k_select_scatter, dim3(cgrid), dim3(256), 0, ctx->stream, sa);
            e = hipMemcpyAsync(hh, sa.hist, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        if (e != hipSuccess) { hcount.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("TopN select: ") + hipGetErrorString(e)); }
        const int64_t n_cand = (int64_t)hh[0];
        if (n_cand < K) { hcount.release(); return tsq_fail(h, TSQ_ERR_HIP, "internal: TopN select lost rows"); }
        a.n = n_cand;
        a.ntiles = (n_cand + TSQ_SORT_T - 1) / TSQ_SORT_T;
        egrid = tsq_grid_for(ctx, n_cand, 256, 4);
        s->rows_sorted = n_cand;
        s->cur = 0;
        have_idx = true;
    }
    for (int k = s->cfg.n_keys - 1; k >= 0; k--) {  // least significant ORDER BY item first
        tsq_status st = sort_cancelled(s);
        if (st != TSQ_OK) { hcount.release(); return st; }
        const int kc = s->cfg.key_col[k];
        a.key.data = s->cols[kc].data.p;
        a.key.nulls = s->cols[kc].has_nulls ? s->cols[kc].nulls.as<uint8_t>() : nullptr;
        a.key.type = s->cfg.col_types[kc];
        a.key.desc = s->cfg.key_desc[k] ? 1 : 0;
        a.key.offs = nullptr;
        a.key.chunk = 0;
        // a string key is a sequence of images, least significant first: its length, then its 8-byte chunks from the last to the
        // first (tsq_sort_image.h); a fixed-width key is one image
        int n_sub = 1;
        if (a.key.type == TSQ_BYTES) {
            a.key.offs = s->cols[kc].offs.as<int64_t>();
            ((unsigned long long*)hcount.p)[0] = 0;
            hipError_t e0 = hipMemcpyAsync(s->count8.p, hcount.p, 8, hipMemcpyHostToDevice, ctx->stream);
            if (e0 == hipSuccess) {
                hipLaunchKernelGGL(k_sort_str_maxlen, dim3(egrid), dim3(256), 0, ctx->stream, a.key.offs, n, (unsigned long long*)s->count8.p);
                e0 = hipMemcpyAsync((char*)hcount.p + 32, s->count8.p, 8, hipMemcpyDeviceToHost, ctx->stream);
            }
            if (e0 == hipSuccess) e0 = hipStreamSynchronize(ctx->stream);
            if (e0 != hipSuccess) { hcount.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("sort: ") + hipGetErrorString(e0)); }
            const unsigned long long maxlen = ((const unsigned long long*)hcount.p)[4];
            n_sub = 1 + (int)((maxlen + 7) / 8);
        }
        for (int sub = 0; sub < n_sub; sub++) {
        if (a.key.type == TSQ_BYTES) a.key.chunk = sub == 0 ? -1 : n_sub - 1 - sub;  // length, then chunk m - 1 .. 0
        // images of this key column in the current row order (first key: identity order, row ids are initialised here)
        a.idx_in = have_idx ? s->idx[s->cur].as<uint32_t>() : nullptr;
        a.img_out = s->img[s->cur].as<uint64_t>();
        a.idx_out = s->idx[s->cur].as<uint32_t>();
        ((unsigned long long*)hcount.p)[0] = 0;
        ((unsigned long long*)hcount.p)[1] = ~0ull;
        hipError_t e = hipMemcpyAsync(s->count8.p, hcount.p, 16, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_sort_image, dim3(egrid), dim3(256), 0, ctx->stream, a);
            e = hipMemcpyAsync((char*)hcount.p + 16, s->count8.p, 16, hipMemcpyDeviceToHost, ctx->stream);
        }
        have_idx = true;
        a.img_in = s->img[s->cur].as<uint64_t>();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { hcount.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("sort: ") + hipGetErrorString(e)); }
        const unsigned long long differ = ((const unsigned long long*)hcount.p)[2] ^ ((const unsigned long long*)hcount.p)[3];  // OR ^ AND
        for (int d = 0; d < 8; d++) {
            if (((differ >> (8 * d)) & 255ull) == 0) { s->passes_skipped++; continue; }  // every row has the same digit
            st = sort_pass(s, a, d);
            if (st != TSQ_OK) { hcount.release(); return st; }
        }
        }  // sub-keys
        if (a.key.nulls) {
            st = sort_pass(s, a, 8);
            if (st != TSQ_OK) { hcount.release(); return st; }
        }
    }
    TSQ_HIP(h, hipEventRecord(s->ev[1], ctx->stream));
    hipError_t e = hipStreamSynchronize(ctx->stream);
    hcount.release();
    if (e != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("sort: ") + hipGetErrorString(e));
    float ms = 0;
    if (hipEventElapsedTime(&ms, s->ev[0], s->ev[1]) == hipSuccess) s->sort_ms = ms;
    {   // the sorted images of the first ORDER BY column stay for the pulls when it is an integer column (see k_gather_rows)
        const int32_t t0 = s->cfg.col_types[s->cfg.key_col[0]];
        s->key_img_kept = t0 == TSQ_I64 || t0 == TSQ_U64;
        s->img[s->cur ^ 1].release();
        if (!s->key_img_kept) s->img[s->cur].release();
    }
    s->idx[s->cur ^ 1].release();
    s->hist.release();
    return TSQ_OK;
}

TSQ_API tsq_status tsq_sort_pull(tsq_sort* s, tsq_col* out_cols, int32_t n_cols, int64_t cap_rows, int64_t* nrows_out, int32_t* eos) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(s, TSQ_MAGIC_SORT));
    if (!s || s->hdr.magic != TSQ_MAGIC_SORT) return TSQ_ERR_INVALID;
    if (!nrows_out || !eos) return tsq_fail(&s->hdr, TSQ_ERR_INVALID, "NULL out pointer");
    *nrows_out = 0;
    *eos = 0;
    TSQ_TRY(sort_cancelled(s));
    if (!s->finished) return tsq_fail(&s->hdr, TSQ_ERR_INVALID, "pull before finish");
    if (n_cols != s->cfg.n_cols) return tsq_fail(&s->hdr, TSQ_ERR_INVALID, "pull: column count must equal the input schema");
    tsq_ctx* ctx = s->ctx;
    tsq_handle_hdr* h = &s->hdr;
    const int64_t n = std::min<int64_t>(cap_rows, s->last - s->cursor);
    if (n <= 0) { *eos = 1; return TSQ_OK; }
    TSQ_HIP(h, hipSetDevice(ctx->device));
    const bool odev = out_cols[0].flags & TSQ_COL_DEVICE;
    GatherRowsArgs g;
    memset(&g, 0, sizeof g);
    g.idx = s->idx[s->cur].as<uint32_t>() + s->cursor;
    g.rows = n;
    if (s->key_img_kept) {
        g.key_img = s->img[s->cur].as<uint64_t>() + s->cursor;
        g.key_col = s->cfg.key_col[0];
        g.key_desc = s->cfg.key_desc[0] ? 1 : 0;
        g.key_flip = s->cfg.col_types[g.key_col] == TSQ_I64 ? 0x8000000000000000ULL : 0ull;
    }
    if (!odev) { s->odata.resize(n_cols); s->obm.resize(n_cols); s->ooffs.resize(n_cols); }
    bool any_var = false;
    for (int c = 0; c < n_cols; c++) {
        const bool var = s->cfg.col_types[c] == TSQ_BYTES;
        any_var = any_var || var;
        if (!out_cols[c].null_bitmap || (var ? !out_cols[c].offsets : !out_cols[c].data))
            return tsq_fail(h, TSQ_ERR_INVALID, "pull: out columns need data and null_bitmap buffers (a var-len column: offsets, and data for the bytes tsq_sort_peek announced)");
        if (((out_cols[c].flags & TSQ_COL_DEVICE) != 0) != odev) return tsq_fail(h, TSQ_ERR_INVALID, "pull: mixed host/device outputs");
        g.es[c] = var ? 0 : tsq_elem_size(s->cfg.col_types[c]);
        g.src[c] = s->cols[c].data.p;
        g.src_nulls[c] = s->cols[c].has_nulls ? s->cols[c].nulls.as<uint8_t>() : nullptr;
        if (odev) {
            g.dst[c] = out_cols[c].data;
            g.dst_bitmap[c] = out_cols[c].null_bitmap;
        } else {
            if (!var) TSQ_TRY(s->odata[c].reserve(ctx, h, (size_t)n * g.es[c] + 64));
            TSQ_TRY(s->obm[c].reserve(ctx, h, tsq_bitmap_bytes(n) + 64));
            g.dst[c] = var ? nullptr : s->odata[c].p;
            g.dst_bitmap[c] = s->obm[c].as<uint8_t>();
        }
    }
    const int64_t groups = (n + 7) / 8;
    const int gx = (int)std::min<int64_t>((groups + 255) / 256, (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(k_gather_rows, dim3(gx, n_cols), dim3(256), 0, ctx->stream, g);
    TSQ_HIP(h, hipGetLastError());
    // var-len payload columns: the offsets of these n rows (computed now, or by the tsq_sort_peek that sized the caller's buffers)
    if (any_var) TSQ_TRY(sort_var_offsets(s, n));
    for (int c = 0; c < n_cols && any_var; c++) {
        if (s->cfg.col_types[c] != TSQ_BYTES) continue;
        SortVarArgs va;
        va.idx = g.idx;
        va.rows = n;
        va.src_offs = s->cols[c].offs.as<int64_t>();
        va.src_data = s->cols[c].data.as<uint8_t>();
        va.src_nulls = g.src_nulls[c];
        va.out_offs = s->voffs[c].as<int64_t>();
        const int64_t nbytes = s->vbytes[c];
        if (nbytes > 0 && !out_cols[c].data) return tsq_fail(h, TSQ_ERR_INVALID, "pull: a var-len column needs a data buffer (tsq_sort_peek tells its size)");
        if (odev) {
            va.out_data = (uint8_t*)out_cols[c].data;
            TSQ_HIP(h, hipMemcpyAsync(out_cols[c].offsets, va.out_offs, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            TSQ_TRY(s->odata[c].reserve(ctx, h, (size_t)nbytes + 64));
            va.out_data = s->odata[c].as<uint8_t>();
        }
        if (nbytes > 0) {
            if (nbytes / n > 32) hipLaunchKernelGGL(k_sort_var_copy<true>, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, va);
            else hipLaunchKernelGGL(k_sort_var_copy<false>, dim3(tsq_grid_for(ctx, n, 256)), dim3(256), 0, ctx->stream, va);
            TSQ_HIP(h, hipGetLastError());
        }
        if (!odev) {
            TSQ_HIP(h, hipMemcpyAsync(out_cols[c].offsets, va.out_offs, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
            if (nbytes > 0) TSQ_HIP(h, hipMemcpyAsync(out_cols[c].data, va.out_data, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    if (!odev)
        for (int c = 0; c < n_cols; c++) {
            if (g.es[c]) TSQ_HIP(h, hipMemcpyAsync(out_cols[c].data, g.dst[c], (size_t)n * g.es[c], hipMemcpyDeviceToHost, ctx->stream));
            TSQ_HIP(h, hipMemcpyAsync(out_cols[c].null_bitmap, g.dst_bitmap[c], tsq_bitmap_bytes(n), hipMemcpyDeviceToHost, ctx->stream));
        }
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    s->peek_cursor = -1;
    for (int c = 0; c < n_cols; c++) {
        out_cols[c].length = n;
        out_cols[c].type = s->cfg.col_types[c];
        out_cols[c].elem_size = g.es[c] ? g.es[c] : -1;
    }
    s->cursor += n;
    *nrows_out = n;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_sort_peek(tsq_sort* s, int64_t cap_rows, int64_t* nrows_out, int64_t* bytes_out, int32_t n_cols) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(s, TSQ_MAGIC_SORT));
    if (!s || s->hdr.magic != TSQ_MAGIC_SORT) return TSQ_ERR_INVALID;
    if (!nrows_out || !bytes_out) return tsq_fail(&s->hdr, TSQ_ERR_INVALID, "NULL out pointer");
    *nrows_out = 0;
    TSQ_TRY(sort_cancelled(s));
    if (!s->finished) return tsq_fail(&s->hdr, TSQ_ERR_INVALID, "peek before finish");
    if (n_cols != s->cfg.n_cols) return tsq_fail(&s->hdr, TSQ_ERR_INVALID, "peek: column count must equal the input schema");
    for (int c = 0; c < n_cols; c++) bytes_out[c] = 0;
    const int64_t n = std::min<int64_t>(cap_rows, s->last - s->cursor);
    if (n <= 0) return TSQ_OK;
    bool any_var = false;
    for (int c = 0; c < n_cols; c++) any_var = any_var || s->cfg.col_types[c] == TSQ_BYTES;
    if (any_var) {
        TSQ_HIP(&s->hdr, hipSetDevice(s->ctx->device));
        TSQ_TRY(sort_var_offsets(s, n));
        for (int c = 0; c < n_cols; c++) bytes_out[c] = s->vbytes[c];
    }
    *nrows_out = n;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_sort_stats(tsq_sort* s, int64_t* rows, int32_t* passes, int32_t* passes_skipped, double* sort_kernel_ms) {
    if (!s || s->hdr.magic != TSQ_MAGIC_SORT) return TSQ_ERR_INVALID;
    if (rows) *rows = s->rows_sorted;
    if (passes) *passes = s->passes;
    if (passes_skipped) *passes_skipped = s->passes_skipped;
    if (sort_kernel_ms) *sort_kernel_ms = s->sort_ms;
    return TSQ_OK;
}

TSQ_API tsq_status tsq_sort_cancel(tsq_sort* s) {
    if (!s || s->hdr.magic != TSQ_MAGIC_SORT) return TSQ_ERR_INVALID;
    s->cancelled.store(1);
    return TSQ_OK;
}

TSQ_API void tsq_sort_destroy(tsq_sort* s) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(s, TSQ_MAGIC_SORT));
    if (!s || s->hdr.magic != TSQ_MAGIC_SORT) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    for (auto& c : s->cols) c.release();
    s->stage.release();
    for (int i = 0; i < 2; i++) { s->img[i].release(); s->idx[i].release(); if (s->ev[i]) (void)hipEventDestroy(s->ev[i]); }
    s->hist.release();
    s->totals.release();
    s->count8.release();
    for (auto& b : s->odata) b.release();
    for (auto& b : s->obm) b.release();
    for (auto& b : s->ooffs) b.release();
    for (auto& b : s->voffs) b.release();
    s->scratch.release();
    s->hdr.magic = 0;
    delete s;
}
