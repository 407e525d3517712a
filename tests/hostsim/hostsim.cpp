// hostsim.cpp — TEST-ONLY host build of tinysql_amd/csrc/tsq_device.h.
//
// The per-row scalar semantics the HIP kernels execute (expression interpreter, filter rows,
// generator, hashes, key words) are TSQ_HD functions; compiling the same header with g++ lets the
// CPU test-suite diff them against the oracle without a GPU.  This object is never loaded by the
// product: libtsq's entry points launch HIP kernels or fail.
#include "../../tinysql_amd/csrc/tsq_device.h"

static void fill(tsq_colset& cs, const tsq_col* cols, int n) {
    memset(&cs, 0, sizeof cs);
    cs.n = n;
    for (int c = 0; c < n; c++) {
        cs.data[c] = cols[c].data;
        cs.nulls[c] = cols[c].null_bitmap;
        cs.offs[c] = cols[c].type == TSQ_BYTES ? cols[c].offsets : nullptr;
        cs.type[c] = cols[c].type;
    }
}

extern "C" {

// mirrors k_expr_eval (tsq_expr.hip) row by row, including the error-word ordering
int32_t sim_expr_eval(const tsq_expr_prog* prog, const tsq_col* cols, int32_t n_cols, int64_t nrows, const int32_t* sel,
                      uint64_t* out, uint8_t* notnull, int64_t* div0) {
    tsq_colset cs;
    fill(cs, cols, n_cols);
    uint64_t errw = TSQ_ERRWORD_NONE;
    int64_t d = 0;
    for (int64_t i = 0; i < nrows; i++) {
        tsq_chunk_src src{&cs, sel ? (int64_t)sel[i] : i};
        tsq_val v;
        int node = 0, d0 = 0;
        tsq_status s = tsq_eval_row(*prog, src, &v, &node, &d0);
        d += d0;
        if (s != TSQ_OK) {
            uint64_t w = tsq_errword(0, node, (uint64_t)i, s);
            if (w < errw) errw = w;
            continue;
        }
        out[i] = (uint64_t)v.v;
        notnull[i] = v.null ? 0 : 1;
    }
    if (div0) *div0 = d;
    return errw == TSQ_ERRWORD_NONE ? TSQ_OK : (int32_t)(errw & 15);
}

int32_t sim_filter_eval(const tsq_expr_prog* progs, int32_t n_progs, const tsq_col* cols, int32_t n_cols, int64_t nrows,
                        const int32_t* sel, uint8_t* selected, uint8_t* isnull, int64_t* div0) {
    tsq_colset cs;
    fill(cs, cols, n_cols);
    uint64_t errw = TSQ_ERRWORD_NONE;
    int64_t d = 0;
    for (int64_t i = 0; i < nrows; i++) {
        tsq_chunk_src src{&cs, sel ? (int64_t)sel[i] : i};
        bool s1 = false, n1 = false;
        int conj = 0, node = 0, d0 = 0;
        tsq_status s = tsq_filter_row(progs, n_progs, src, &s1, &n1, &conj, &node, &d0);
        d += d0;
        if (s != TSQ_OK) {
            uint64_t w = tsq_errword(conj, node, (uint64_t)i, s);
            if (w < errw) errw = w;
            continue;
        }
        selected[i] = s1;
        isnull[i] = n1;
    }
    if (div0) *div0 = d;
    return errw == TSQ_ERRWORD_NONE ? TSQ_OK : (int32_t)(errw & 15);
}

void sim_gen_column(const tsq_gen_spec* spec, int64_t nrows, uint64_t* dst, uint8_t* notnull, const uint64_t* src) {
    for (int64_t k = 0; k < nrows; k++) {
        uint64_t i = (uint64_t)(spec->start + k);
        bool isnull = tsq_gen_is_null(*spec, i);
        dst[k] = isnull ? 0 : tsq_gen_value(*spec, i, spec->kind == TSQ_GEN_HASH_OF_COL ? src[k] : 0);
        if (notnull) notnull[k] = isnull ? 0 : 1;
    }
}

int32_t sim_validate(const tsq_expr_prog* p, int32_t n_cols) {
    const char* why = "";
    return tsq_validate_prog(*p, n_cols, &why);
}
int32_t sim_validate_typed(const tsq_expr_prog* p, int32_t n_cols, const int32_t* col_types) {
    const char* why = "";
    return tsq_validate_prog(*p, n_cols, &why, col_types);
}

uint64_t sim_rowhash(const uint64_t* vals, const uint8_t* notnull, int32_t n) {
    uint64_t h = TSQ_ROWHASH_SEED;
    for (int32_t c = 0; c < n; c++) h = tsq_rowhash_step(h, notnull[c] ? vals[c] : TSQ_ROWHASH_NULL, (uint32_t)c);
    return h;
}

uint32_t sim_key_rank(uint64_t kw, uint32_t n_parts) { return tsq_key_rank(kw, n_parts); }
uint64_t sim_mix64(uint64_t k) { return tsq_mix64(k); }
uint64_t sim_mulhi64(uint64_t a, uint64_t b) { return tsq_mulhi64(a, b); }
}

// ---- tsq_rows_decode's scalar core (tsq_decode_dp.h), driven over a whole byte stream the way the kernels do it
#include "../../tinysql_amd/csrc/tsq_decode_dp.h"

extern "C" {

// exit map + packed counts of every 32-byte sub-block of `bytes` (k_dec_map's per-thread work)
void sim_dec_maps(const uint8_t* bytes, int64_t n, unsigned long long* maps, uint32_t* cnts) {
    const int64_t nsb = (n + 31) / 32;
    for (int64_t sb = 0; sb < nsb; sb++) {
        uint8_t buf[44];
        for (int i = 0; i < 44; i++) buf[i] = sb * 32 + i < n ? bytes[sb * 32 + i] : 0;  // zero padded past the end, like the LDS tile
        uint32_t w[11];
        memcpy(w, buf, 44);
        const uint32_t lim = n - sb * 32 >= 32 ? 32u : (uint32_t)(n - sb * 32);
        tsq_dec_subblock(w, lim, &maps[sb], &cnts[sb * 3]);
    }
}

// the value at byte position pos with length len (k_dec_emit's dec_value without the store): returns the status code
int32_t sim_dec_value(const uint8_t* bytes, int64_t n, int64_t pos, uint32_t len, uint64_t* bits, uint8_t* isnull, uint8_t* real) {
    uint8_t buf[12];
    for (int i = 0; i < 12; i++) buf[i] = pos + i < n ? bytes[pos + i] : 0;
    uint32_t b[3];
    memcpy(b, buf, 12);
    bool nl = false, re = false;
    if (pos + len > n) return DEC_INSUFFICIENT;
    const int e = tsq_dec_value(b[0], b[1], b[2], len, bits, &nl, &re);
    *isnull = nl;
    *real = re;
    return e;
}
}

// ---- ORDER BY key images (tsq_sort_image.h)
#include "../../tinysql_amd/csrc/tsq_sort_image.h"
extern "C" void sim_sort_images(const void* data, int32_t type, int32_t desc, int64_t n, uint64_t* out) {
    for (int64_t i = 0; i < n; i++) out[i] = tsq_sort_image(data, type, desc, (uint64_t)i);
}

#include <algorithm>
#include <vector>
// ORDER BY a string column the way tsq_sort_finish walks it (tsq_sort.hip): one stable sort per image, least significant first —
// the length, then the 8-byte chunks from the last to the first — and a last stable pass that puts the NULL rows first
// (ascending) or last (descending).  perm_io holds the current row order (identity, or the order of the less significant keys).
extern "C" void sim_sort_str_key(const uint8_t* data, const int64_t* offs, const uint8_t* notnull, int32_t desc, int64_t n, int64_t* perm_io) {
    int64_t maxlen = 0;
    for (int64_t i = 0; i < n; i++) maxlen = std::max(maxlen, offs[i + 1] - offs[i]);
    const int n_sub = 1 + (int)((maxlen + 7) / 8);
    std::vector<uint64_t> img((size_t)n);
    for (int sub = 0; sub < n_sub; sub++) {
        const int32_t chunk = sub == 0 ? -1 : n_sub - 1 - sub;
        for (int64_t i = 0; i < n; i++) img[(size_t)i] = tsq_sort_image_str(data, offs, chunk, desc, (uint64_t)i);
        std::stable_sort(perm_io, perm_io + n, [&](int64_t a, int64_t b) { return img[(size_t)a] < img[(size_t)b]; });
    }
    if (notnull) {
        auto level = [&](int64_t r) { const int nn = (notnull[r >> 3] >> (r & 7)) & 1; return desc ? !nn : nn; };  // NULL first asc, last desc
        std::stable_sort(perm_io, perm_io + n, [&](int64_t a, int64_t b) { return level(a) < level(b); });
    }
}

#include <stdlib.h>
// ---- stored rows -> columns (tsq_rowcodec_dp.h): a CPU walk-through of k_rowcodec_decode (tsq_rowcodec.hip) with the same tile
// plan, the same staged copy (aligned 16-byte vectors into the tile; bytes outside `values` and stale tile bytes are garbage here), the same
// per-lane row code and the same bitmap bytes (one ballot per 64 rows, lanes 0..7 store one byte each when it exists).
#include "../../tinysql_amd/csrc/tsq_rowcodec_dp.h"
namespace {
struct SimBytes {  // RcGlobal of the kernel
    const uint8_t* p;
    uint32_t operator()(uint32_t i) const { return p[i]; }
    uint64_t le(uint32_t q, uint32_t n) const { return tsq_rc_le_bytes(*this, q, n); }
};
struct SimWords {  // RcLds of the kernel: aligned 32-bit words of the staged tile + funnel shift
    const uint32_t* w;
    uint32_t base;
    uint32_t operator()(uint32_t i) const { return ((const uint8_t*)w)[base + i]; }
    uint64_t le(uint32_t p, uint32_t) const {
        const uint32_t q = base + p, i = q >> 2;
        return tsq_rc_funnel(w[i], w[i + 1], w[i + 2], q);
    }
};
template <class R>
int sim_row(const R& rd, uint32_t len, bool bad_offsets, const tsq_rowcodec_col* cols, int c, int64_t handle, uint64_t* bits, bool* notnull) {
    // the kernel carries `code` across the column loop; replaying columns 0..c gives the same state
    tsq_rc_row row = {0, 0, 0, 0, 0, 0, 0};
    int code = bad_offsets ? RC_MALFORMED : tsq_rc_parse(rd, len, &row);
    *bits = 0;
    *notnull = false;
    for (int k = 0; k <= c; k++) {
        *bits = 0;
        *notnull = false;
        if (code == RC_OK) code = tsq_rc_column(rd, row, cols[k].col_id, cols[k].type, cols[k].flags, cols[k].def_bits, handle, bits, notnull);
    }
    return code;
}
}  // namespace
extern "C" uint64_t sim_rowcodec_decode(const uint8_t* values, int64_t n_bytes, uint64_t base_addr, const int64_t* offsets, const int64_t* handles,
                                        int64_t nrows, const tsq_rowcodec_col* cols, int32_t n_cols, void** out, uint8_t** out_bm, uint32_t lds_bytes,
                                        int32_t fast_layout, int64_t* staged_tiles, int64_t* fast_waves) {
    const int NT = 256;
    uint64_t err = ~0ull;
    const int64_t n_tiles = (nrows + NT - 1) / NT, bm_bytes = (nrows + 7) / 8;
    uint8_t* tile = (uint8_t*)calloc((size_t)lds_bytes + 32, 1);  // + the slack the word reads may touch
    *staged_tiles = 0;
    *fast_waves = 0;
    for (int64_t t = 0; t < n_tiles; t++) {
        const int64_t r0 = t * NT, r1 = r0 + NT < nrows ? r0 + NT : nrows;
        const int64_t tile_lo = offsets[r0], tile_hi = offsets[r1];
        const tsq_rc_plan plan = tsq_rc_tile_plan(base_addr, tile_lo, tile_hi, n_bytes, lds_bytes);
        if (plan.staged) {
            (*staged_tiles)++;
            // whatever the previous tile left behind (and whatever surrounds `values` in memory) must not matter: garbage
            memset(tile, 0xA5, (size_t)lds_bytes + 32);
            for (uint32_t i = 0; i < plan.n_vec * 16u; i++) {
                const int64_t at = plan.copy_from + (int64_t)i;
                tile[i] = (at >= 0 && at < n_bytes) ? values[at] : (uint8_t)0x5A;
            }
        }
        for (int c = 0; c < n_cols; c++)
            for (int w = 0; w < NT / 64; w++) {  // one wave
                uint64_t ballot = 0;
                // the vote of rc_rows_lds: every live row of the wave has the signature of the wave's first lane
                bool fast = false;
                uint64_t hdr0 = 0, ids0 = 0;
                if (plan.staged && fast_layout) {
                    int n_live = 0, n_same = 0;
                    for (int lane = 0; lane < 64; lane++) {
                        const int64_t r = r0 + w * 64 + lane;
                        if (r >= r1) continue;
                        n_live++;
                        const int64_t lo = offsets[r], hi = offsets[r + 1];
                        const bool bad_offsets = lo < tile_lo || hi < lo || hi > tile_hi || tile_hi > n_bytes || hi - lo > 0x7fffffffLL;
                        SimWords rd;
                        rd.w = (const uint32_t*)tile;
                        rd.base = bad_offsets ? 0u : plan.skew + (uint32_t)(lo - tile_lo);
                        uint64_t hdr = 0, ids8 = 0;
                        const bool cand = !bad_offsets && tsq_rc_signature(rd, (uint32_t)(hi - lo), &hdr, &ids8);
                        if (lane == 0) { hdr0 = cand ? hdr : 0; ids0 = cand ? ids8 : 0; }
                        if (cand && hdr == hdr0 && ids8 == ids0) n_same++;
                    }
                    fast = (hdr0 & 0xffu) == TSQ_RC_CODEC_VER && n_same == n_live;
                    if (fast && c == 0) (*fast_waves)++;
                }
                for (int lane = 0; lane < 64; lane++) {
                    const int tid = w * 64 + lane;
                    const int64_t r = r0 + tid;
                    const bool live = r < r1;
                    if (!live) continue;
                    const int64_t lo = offsets[r], hi = offsets[r + 1];
                    const bool bad_offsets = lo < tile_lo || hi < lo || hi > tile_hi || tile_hi > n_bytes || hi - lo > 0x7fffffffLL;
                    const uint32_t len = bad_offsets ? 0u : (uint32_t)(hi - lo);
                    uint64_t bits = 0;
                    bool notnull = false;
                    int code;
                    if (plan.staged && fast) {
                        SimWords rd;
                        rd.w = (const uint32_t*)tile;
                        rd.base = plan.skew + (uint32_t)(lo - tile_lo);
                        const uint32_t nn = (uint32_t)(hdr0 >> 16) & 0xffffu, nl = (uint32_t)(hdr0 >> 32) & 0xffffu;
                        uint64_t o_lo = 0, o_hi = 0;
                        tsq_rc_fast_offsets(rd, nn, 6 + nn + nl, &o_lo, &o_hi);
                        code = RC_OK;
                        for (int k = 0; k <= c; k++) {
                            bits = 0;
                            notnull = false;
                            if (code == RC_OK)
                                code = tsq_rc_fast_column(rd, len, hdr0, ids0, o_lo, o_hi, cols[k].col_id, cols[k].type, cols[k].flags, cols[k].def_bits,
                                                          handles ? handles[r] : 0, &bits, &notnull);
                        }
                    } else if (plan.staged) {
                        SimWords rd;
                        rd.w = (const uint32_t*)tile;
                        rd.base = bad_offsets ? 0u : plan.skew + (uint32_t)(lo - tile_lo);
                        code = sim_row(rd, len, bad_offsets, cols, c, handles ? handles[r] : 0, &bits, &notnull);
                    } else {
                        SimBytes rd;
                        rd.p = values + (bad_offsets ? 0 : lo);
                        code = sim_row(rd, len, bad_offsets, cols, c, handles ? handles[r] : 0, &bits, &notnull);
                    }
                    if (cols[c].type == TSQ_F32) ((uint32_t*)out[c])[r] = (uint32_t)bits;
                    else ((uint64_t*)out[c])[r] = bits;
                    if (notnull) ballot |= 1ull << lane;
                    if (c == n_cols - 1 && code != RC_OK) {
                        const uint64_t e = ((uint64_t)r << 4) | (uint64_t)code;
                        if (e < err) err = e;
                    }
                }
                for (int lane = 0; lane < 8; lane++) {
                    const int64_t byte_at = ((r0 + w * 64) >> 3) + lane;
                    if (byte_at < bm_bytes) out_bm[c][byte_at] = (uint8_t)(ballot >> (8 * lane));
                }
            }
    }
    free(tile);
    return err;
}

// ---- chunk rows -> response bytes (tsq_encode_dp.h): a CPU walk-through of k_enc_size / k_enc_scan / k_enc_emit (tsq_encode.hip):
// the same value lengths and datum bytes, the same ownership of contiguous tile ranges by workgroups, the per-tile LDS image at
// the same skew, and the same copy-out plan (head bytes by lanes 0..15, whole vectors, tail bytes by lanes 16..31) into an output
// whose address alignment is `out_phase` — bytes outside [0, total) must stay untouched.
#include "../../tinysql_amd/csrc/tsq_encode_dp.h"
namespace {
uint64_t sim_enc_load(const tsq_col& col, int64_t r, bool* notnull) {
    const uint8_t* bm = col.null_bitmap;
    *notnull = bm ? ((bm[r >> 3] >> (r & 7)) & 1) != 0 : true;
    if (col.type == TSQ_F32) {
        const double d = (double)((const float*)col.data)[r];
        uint64_t b;
        memcpy(&b, &d, 8);
        return b;
    }
    return ((const uint64_t*)col.data)[r];
}
uint32_t sim_enc_row_len(const tsq_col* cols, int n_cols, uint32_t comparable, int64_t r) {
    uint32_t len = 0;
    for (int c = 0; c < n_cols; c++) {
        if (cols[c].type == TSQ_BYTES) {
            const uint8_t* bm = cols[c].null_bitmap;
            const bool nn = bm ? ((bm[r >> 3] >> (r & 7)) & 1) != 0 : true;
            const uint64_t n = (uint64_t)(cols[c].offsets[r + 1] - cols[c].offsets[r]);
            if ((comparable >> c) & 1u) len += nn ? (uint32_t)tsq_enc_membytes_len(n) : 1u;
            else len += nn ? tsq_enc_str_hdr_len(n) + (uint32_t)n : 1u;
            continue;
        }
        bool nn;
        const uint64_t bits = sim_enc_load(cols[c], r, &nn);
        len += tsq_enc_len(cols[c].type, (comparable >> c) & 1u, bits, nn);
    }
    return len;
}
}  // namespace
// out: a buffer with 64 guard bytes on both sides of the region the caller checks; out_phase = (address of out[0]) & 15 to simulate.
// Returns the total; row_offsets gets nrows + 1 entries.
extern "C" int64_t sim_rows_encode(const tsq_col* cols, int32_t n_cols, uint32_t comparable, int64_t nrows, int32_t n_wg_want, uint8_t* out, int64_t cap,
                                   uint32_t out_phase, int64_t* row_offsets) {
    const int NT = 256;
    const int64_t n_tiles = (nrows + NT - 1) / NT;
    if (n_tiles == 0) { row_offsets[0] = 0; return 0; }
    const int64_t want = n_wg_want < n_tiles ? n_wg_want : n_tiles;
    const int64_t tiles_per_wg = (n_tiles + want - 1) / want;
    const int n_wg = (int)((n_tiles + tiles_per_wg - 1) / tiles_per_wg);
    std::vector<unsigned long long> wg(n_wg + 1, 0);
    for (int b = 0; b < n_wg; b++) {  // k_enc_size
        const int64_t t0 = b * tiles_per_wg, t1 = t0 + tiles_per_wg < n_tiles ? t0 + tiles_per_wg : n_tiles;
        for (int64_t t = t0; t < t1; t++)
            for (int tid = 0; tid < NT; tid++) {
                const int64_t r = t * NT + tid;
                if (r < nrows) wg[b] += sim_enc_row_len(cols, n_cols, comparable, r);
            }
    }
    unsigned long long run = 0;  // k_enc_scan
    for (int b = 0; b < n_wg; b++) { const unsigned long long v = wg[b]; wg[b] = run; run += v; }
    wg[n_wg] = run;
    if ((int64_t)run > cap) return (int64_t)run;
    uint32_t row_max = 0;
    bool any_var = false;
    for (int c = 0; c < n_cols; c++) {
        any_var = any_var || cols[c].type == TSQ_BYTES;
        row_max += cols[c].type == TSQ_BYTES ? 32u : ((cols[c].type == TSQ_F32 || cols[c].type == TSQ_F64 || ((comparable >> c) & 1u)) ? 9u : TSQ_ENC_MAX_VALUE);
    }
    size_t lds = (((size_t)NT * row_max + 15 + 16 + 15) / 16) * 16;
    if (any_var) lds = lds < 32 * 1024 ? 32 * 1024 : (lds > 64 * 1024 ? 64 * 1024 : lds);
    std::vector<uint8_t> img(lds);
    const uint64_t out_addr = 0x7f0000002000ULL + out_phase;  // only its low bits matter
    for (int b = 0; b < n_wg; b++) {  // k_enc_emit
        const int64_t t0 = b * tiles_per_wg, t1 = t0 + tiles_per_wg < n_tiles ? t0 + tiles_per_wg : n_tiles;
        int64_t base = (int64_t)wg[b];
        for (int64_t t = t0; t < t1; t++) {
            uint32_t len[NT], ex[NT], T = 0;
            for (int tid = 0; tid < NT; tid++) {
                const int64_t r = t * NT + tid;
                len[tid] = r < nrows ? sim_enc_row_len(cols, n_cols, comparable, r) : 0u;
                ex[tid] = T;  // block_excl_scan
                T += len[tid];
            }
            const tsq_enc_copy plan = tsq_enc_copy_plan(out_addr, base, T);
            const bool staged = (size_t)plan.skew + T + 16 <= lds;  // otherwise the rows go to their place directly
            memset(img.data(), 0xA5, lds);  // stale bytes of the previous tile must not leak
            for (int tid = 0; tid < NT; tid++) {
                const int64_t r = t * NT + tid;
                if (r >= nrows) continue;
                row_offsets[r] = base + (int64_t)ex[tid];
                uint8_t* dst = staged ? img.data() + plan.skew + ex[tid] : out + base + ex[tid];
                uint32_t pos = 0;
                for (int c = 0; c < n_cols; c++) {
                    uint64_t lo;
                    uint32_t hi;
                    if (cols[c].type == TSQ_BYTES) {
                        const uint8_t* bm = cols[c].null_bitmap;
                        const bool nn = bm ? ((bm[r >> 3] >> (r & 7)) & 1) != 0 : true;
                        if (!nn) { dst[pos++] = 0; continue; }
                        const int64_t s0 = cols[c].offsets[r];
                        const uint64_t n = (uint64_t)(cols[c].offsets[r + 1] - s0);
                        if ((comparable >> c) & 1u) {
                            const uint8_t* src = (const uint8_t*)cols[c].data + s0;
                            const uint32_t m = (uint32_t)tsq_enc_membytes_len(n) - 1u;
                            dst[pos++] = 1;
                            for (uint32_t i = 0; i < m; i++) dst[pos + i] = tsq_enc_membytes_at(src, n, i);
                            pos += m;
                            continue;
                        }
                        const uint32_t hn = tsq_enc_str_hdr(n, &lo, &hi);
                        for (uint32_t i = 0; i < hn; i++) dst[pos + i] = (uint8_t)(i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8)));
                        pos += hn;
                        memcpy(dst + pos, (const uint8_t*)cols[c].data + s0, (size_t)n);
                        pos += (uint32_t)n;
                        continue;
                    }
                    bool nn;
                    const uint64_t bits = sim_enc_load(cols[c], r, &nn);
                    const uint32_t n = tsq_enc_bytes(cols[c].type, (comparable >> c) & 1u, bits, nn, &lo, &hi);
                    for (uint32_t i = 0; i < n; i++) dst[pos + i] = (uint8_t)(i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8)));
                    pos += n;
                }
            }
            if (!staged) { base += T; continue; }
            // copy-out: global byte (base - skew) + i <-> image byte i.  `out` here is indexed from the region start.
            uint8_t* g = out + base - plan.skew;
            for (uint32_t tid = 0; tid < 16; tid++)
                if (plan.skew + tid < plan.head_end) g[plan.skew + tid] = img[plan.skew + tid];
            for (uint32_t tid = 16; tid < 32; tid++)
                if (plan.tail_lo + (tid - 16) < plan.tail_end) g[plan.tail_lo + (tid - 16)] = img[plan.tail_lo + (tid - 16)];
            if (((out_addr + (uint64_t)base - plan.skew) & 15u) != 0) return -2;  // the vector stores would be misaligned
            for (uint32_t i = plan.body_lo; i < plan.body_hi; i++) memcpy(g + 16 * (size_t)i, img.data() + 16 * (size_t)i, 16);
            base += T;
        }
        if (t1 == n_tiles && t1 > t0) row_offsets[nrows] = base;
    }
    return (int64_t)run;
}

// ---- record keys <-> handles (tsq_tablecodec_dp.h): a CPU walk-through of k_rowkeys_decode / k_rowkeys_encode (tsq_tablecodec.hip) with the
// same tile plans: the staged copy of 1024 keys as aligned 16-byte vectors (bytes outside `keys` and stale tile bytes are garbage), the word +
// funnel reader, and the encoder's LDS image at the skew of its destination with head / body / tail copy-out.
#include "../../tinysql_amd/csrc/tsq_tablecodec_dp.h"
extern "C" uint64_t sim_rowkeys_decode(const uint8_t* keys, int64_t n_bytes, uint64_t base_addr, const int64_t* offsets, int64_t n, int64_t* handles,
                                       int64_t* table_ids, int64_t* staged_tiles) {
    const int NT = 256, KPL = 4, TILE = NT * KPL;
    const uint32_t LDS = TILE * 19 + 64;
    uint64_t err = ~0ull;
    std::vector<uint8_t> tile((LDS / 16 + 1) * 16);
    *staged_tiles = 0;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t t = 0; t < n_tiles; t++) {
        const int64_t r0 = t * TILE, r1 = r0 + TILE < n ? r0 + TILE : n;
        const int64_t tile_lo = offsets ? offsets[r0] : r0 * 19, tile_hi = offsets ? offsets[r1] : r1 * 19;
        const tsq_rc_plan plan = tsq_rc_tile_plan(base_addr, tile_lo, tile_hi, n_bytes, LDS - 32);
        memset(tile.data(), 0xA5, tile.size());
        if (plan.staged) {
            (*staged_tiles)++;
            if ((size_t)plan.n_vec * 16 > tile.size()) return 0xBAD0;  // the kernel's LDS array would be overrun
            for (uint32_t i = 0; i < plan.n_vec * 16u; i++) {
                const int64_t at = plan.copy_from + (int64_t)i;
                tile[i] = (at >= 0 && at < n_bytes) ? keys[at] : (uint8_t)0x5A;
            }
        }
        for (int k = 0; k < KPL; k++)
            for (int tid = 0; tid < NT; tid++) {
                const int64_t r = r0 + (int64_t)k * NT + tid;
                if (r >= r1) continue;
                const int64_t lo = offsets ? offsets[r] : r * 19, hi = offsets ? offsets[r + 1] : lo + 19;
                int code;
                int64_t table_id = 0, handle = 0;
                if (lo < tile_lo || hi < lo || hi > tile_hi || lo < 0 || hi > n_bytes) {
                    code = TC_INVALID_KEY;
                } else if (plan.staged) {
                    SimWords rd;
                    rd.w = (const uint32_t*)tile.data();
                    rd.base = plan.skew + (uint32_t)(lo - tile_lo);
                    if ((size_t)rd.base + 19 + 12 > tile.size()) return 0xBAD1;  // a word read past the LDS array
                    code = tsq_tc_decode_row_key(rd, (uint32_t)(hi - lo > 0xffff ? 0xffff : hi - lo), &table_id, &handle);
                } else {
                    SimBytes rd;
                    rd.p = keys + lo;
                    code = tsq_tc_decode_row_key(rd, (uint32_t)(hi - lo > 0xffff ? 0xffff : hi - lo), &table_id, &handle);
                }
                handles[r] = handle;
                if (table_ids) table_ids[r] = table_id;
                if (code != TC_OK) {
                    const uint64_t e = ((uint64_t)r << 4) | (uint64_t)code;
                    if (e < err) err = e;
                }
            }
    }
    return err;
}
// `out` is indexed from the start of the key array; out_phase = the low bits of its device address
extern "C" int32_t sim_rowkeys_encode(int64_t table_id, const int64_t* handles, int64_t n, uint8_t* out, uint32_t out_phase) {
    const int NT = 256, KPL = 4, TILE = NT * KPL;
    const uint32_t LDS = TILE * 19 + 64;
    std::vector<uint8_t> img((LDS / 16 + 1) * 16);
    const uint64_t out_addr = 0x7f0000004000ULL + out_phase;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    for (int64_t t = 0; t < n_tiles; t++) {
        const int64_t r0 = t * TILE, r1 = r0 + TILE < n ? r0 + TILE : n;
        const int64_t base = r0 * 19;
        const uint32_t T = (uint32_t)(r1 - r0) * 19u;
        const tsq_enc_copy plan = tsq_enc_copy_plan(out_addr, base, T);
        if ((size_t)plan.skew + T > img.size()) return -1;
        memset(img.data(), 0xA5, img.size());
        for (int k = 0; k < KPL; k++)
            for (int tid = 0; tid < NT; tid++) {
                const int64_t r = r0 + (int64_t)k * NT + tid;
                if (r >= r1) continue;
                uint64_t p0, p1;
                uint32_t p2;
                tsq_tc_encode_row_key(table_id, handles[r], &p0, &p1, &p2);
                const uint32_t pos = plan.skew + (uint32_t)(r - r0) * 19u;
                for (uint32_t i = 0; i < 8; i++) img[pos + i] = (uint8_t)(p0 >> (8 * i));
                for (uint32_t i = 0; i < 8; i++) img[pos + 8 + i] = (uint8_t)(p1 >> (8 * i));
                for (uint32_t i = 0; i < 3; i++) img[pos + 16 + i] = (uint8_t)(p2 >> (8 * i));
            }
        uint8_t* g = out + base - plan.skew;
        for (uint32_t tid = 0; tid < 16; tid++)
            if (plan.skew + tid < plan.head_end) g[plan.skew + tid] = img[plan.skew + tid];
        for (uint32_t tid = 16; tid < 32; tid++)
            if (plan.tail_lo + (tid - 16) < plan.tail_end) g[plan.tail_lo + (tid - 16)] = img[plan.tail_lo + (tid - 16)];
        for (uint32_t i = plan.body_lo; i < plan.body_hi; i++) memcpy(g + 16 * (size_t)i, img.data() + 16 * (size_t)i, 16);
    }
    return 0;
}

// ---- response chunks -> columns, var-len included (tsq_decode_dp.h: tsq_decc_*): a CPU walk-through of k_decc_count / k_decc_emit
// (tsq_decodec.hip): one "lane" per chunk, the 12-byte fetch built from aligned 8-byte words of a buffer whose surroundings are
// garbage (bytes at or past the chunk end must read as zero, words that hold no byte of the response are never touched), the error
// word, the rows-per-chunk scan, and the (position, length) references of string cells.
namespace {
struct SimFetch {
    const uint8_t* buf;   // the response at buf[guard + phase ...], garbage around it
    int64_t guard, phase, n_bytes;
    int64_t bad_touch = 0;
    uint64_t word(int64_t abs_at) {  // aligned 8-byte word at absolute buffer offset abs_at
        const int64_t lo = guard + phase, hi = lo + n_bytes;
        if (abs_at + 8 <= lo || abs_at >= hi) bad_touch++;  // a word without any response byte
        uint64_t w;
        memcpy(&w, buf + abs_at, 8);
        return w;
    }
    void fetch12(int64_t p, int64_t end, uint32_t* b0, uint32_t* b1, uint32_t* b2) {
        const int64_t addr = guard + phase + p, a = addr & ~(int64_t)7, lim = guard + phase + n_bytes;
        const uint32_t sh = (uint32_t)(addr & 7) * 8u;
        const uint64_t w0 = word(a);
        const uint64_t w1 = a + 8 < lim ? word(a + 8) : 0ull;
        const uint64_t w2 = a + 16 < lim ? word(a + 16) : 0ull;
        uint64_t lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
        uint64_t hi = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;
        const int64_t valid = end - p;
        if (valid < 8) { lo &= (1ull << (8 * valid)) - 1; hi = 0; }
        else if (valid < 12) hi &= (1ull << (8 * (valid - 8))) - 1;
        *b0 = (uint32_t)lo;
        *b1 = (uint32_t)(lo >> 32);
        *b2 = (uint32_t)hi;
    }
};
}  // namespace
// buf: `guard` garbage bytes, `phase` more (the alignment of the response pointer: guard is a multiple of 8), the response, garbage.
// out_bits[c][row] (8 bytes per row: the stored value, or for a TSQ_BYTES column the cell's absolute position in the response),
// out_len[c][row] (TSQ_BYTES: the cell's length), out_nn[c][row].  Returns the error word (~0: none); *rows_out = rows handed over.
extern "C" uint64_t sim_rows_decode_chunks(const uint8_t* buf, int64_t guard, int64_t phase, int64_t n_bytes, const int64_t* chunk_offs, int64_t n_chunks,
                                           int32_t n_cols, const int32_t* types, uint64_t** out_bits, int64_t** out_len, uint8_t** out_nn, int64_t cap_rows,
                                           int64_t* rows_out, int64_t* bad_touch_out) {
    SimFetch F{buf, guard, phase, n_bytes};
    std::vector<int64_t> rows((size_t)n_chunks + 1, 0);
    uint64_t err = ~0ull;
    auto range_ok = [&](int64_t k, int64_t* lo, int64_t* hi) { *lo = chunk_offs[k]; *hi = chunk_offs[k + 1]; return *lo >= 0 && *hi >= *lo && *hi <= n_bytes; };
    for (int64_t k = 0; k < n_chunks; k++) {  // k_decc_count
        int64_t lo, hi, vals = 0;
        int code = DEC_OK;
        if (!range_ok(k, &lo, &hi)) code = DEC_ROW_CUT;
        else {
            int32_t col = 0;
            for (int64_t p = lo; p < hi;) {
                uint32_t b0, b1, b2;
                F.fetch12(p, hi, &b0, &b1, &b2);
                tsq_decc_val v;
                code = tsq_decc_value(b0, b1, b2, (uint64_t)(hi - p), &v);
                if (code == DEC_VARLEN) code = tsq_decc_membytes(buf + guard + phase + p, (uint64_t)(hi - p), &v);
                uint64_t bits;
                if (code == DEC_OK && !tsq_decc_store(types[col], v, &bits)) code = DEC_KIND_MISMATCH;
                if (code != DEC_OK) break;
                p += (int64_t)v.len;
                vals++;
                col = col + 1 == n_cols ? 0 : col + 1;
            }
            if (code == DEC_OK && vals % n_cols != 0) code = DEC_ROW_CUT;
        }
        rows[(size_t)k] = vals / n_cols;
        if (code != DEC_OK) {
            const uint64_t ord = (uint64_t)(vals < (1 << 28) - 1 ? vals : (1 << 28) - 1);
            const uint64_t e = ((uint64_t)k << 32) | (ord << 4) | (uint64_t)code;
            if (e < err) err = e;
        }
    }
    int64_t run = 0;  // tsq_launch_scan64
    for (int64_t k = 0; k < n_chunks; k++) { const int64_t c = rows[(size_t)k]; rows[(size_t)k] = run; run += c; }
    rows[(size_t)n_chunks] = run;
    int64_t total = run, err_chunk = n_chunks, err_rows = 0;
    if (err != ~0ull) {
        err_chunk = (int64_t)(err >> 32);
        err_rows = rows[(size_t)err_chunk + 1] - rows[(size_t)err_chunk];
        total = rows[(size_t)err_chunk] + err_rows;
    }
    *rows_out = total;
    *bad_touch_out = F.bad_touch;
    if (total > cap_rows) return err;
    for (int64_t k = 0; k < n_chunks && k <= err_chunk; k++) {  // k_decc_emit
        int64_t lo, hi;
        if (!range_ok(k, &lo, &hi)) continue;
        const int64_t base = rows[(size_t)k], want = k == err_chunk ? err_rows : rows[(size_t)k + 1] - base;
        int64_t r = 0;
        int32_t col = 0;
        for (int64_t p = lo; p < hi && r < want;) {
            uint32_t b0, b1, b2;
            F.fetch12(p, hi, &b0, &b1, &b2);
            tsq_decc_val v;
            int code = tsq_decc_value(b0, b1, b2, (uint64_t)(hi - p), &v);
            const bool grouped = code == DEC_VARLEN;
            if (grouped) code = tsq_decc_membytes(buf + guard + phase + p, (uint64_t)(hi - p), &v);
            if (code != DEC_OK) break;
            uint64_t bits;
            (void)tsq_decc_store(types[col], v, &bits);
            const int64_t row = base + r;
            if (types[col] == TSQ_BYTES) {
                out_bits[col][row] = (uint64_t)((p + (int64_t)v.data_at) | (grouped ? TSQ_DECC_GROUPED : 0));
                out_len[col][row] = v.kind == DECV_BYTES ? (int64_t)v.bits : 0;
            } else {
                out_bits[col][row] = bits;
            }
            out_nn[col][row] = v.kind != DECV_NULL ? 1 : 0;
            p += (int64_t)v.len;
            col++;
            if (col == n_cols) { col = 0; r++; }
        }
    }
    *bad_touch_out = F.bad_touch;
    return err;
}

// ---- the chunk wire format (tsq_wire_dp.h): the header walk, and every lane of every workgroup of one piece of k_wire_move
#include "../../tinysql_amd/csrc/tsq_wire_dp.h"
extern "C" void sim_wire_walk(const uint8_t* buf, int64_t n_bytes, const int32_t* elem, int32_t n_cols, int64_t first, int64_t max_rows, uint64_t* out) {
    tsq_wire_walk(buf, n_bytes, elem, n_cols, first, max_rows, out);
}
extern "C" void sim_wire_move(int32_t mode, const uint8_t* src, uint8_t* dst, int64_t n, int64_t imm) {
    int64_t blocks = 1;  // MovePlan::add (tsq_wire.hip)
    if (mode == WM_COPY) blocks = std::max<int64_t>(1, (n + TSQ_WIRE_BLOCK_BYTES - 1) / TSQ_WIRE_BLOCK_BYTES);
    else if (mode == WM_BITS) blocks = std::max<int64_t>(1, (((imm + n + 7) >> 3) - (imm >> 3) + 4095) / 4096);
    else if (mode == WM_OFFS) blocks = std::max<int64_t>(1, (n + 2047) / 2048);
    if (mode != WM_HDR && n <= 0) return;
    // WM_BITS reads the destination's first byte (the rows already there) in lane 0 of workgroup 0 only, before it writes it
    for (int64_t b = 0; b < blocks; b++)
        for (int t = 0; t < 256; t++) tsq_wire_move_lane(mode, src, dst, n, imm, b, t);
}

// ---- tsq_rows_decode with a var-len column (round 5): the host walk that finds the row boundaries of a stream with bytes datums
// (tsq_decode_dp.h: tsq_dec_walk_rows) — the product's own function, called here without a GPU
extern "C" int64_t sim_dec_walk_rows(const uint8_t* data, int64_t n_bytes, int32_t n_cols, int64_t cap_rows, int64_t per, int64_t* offs_out, int64_t offs_cap,
                                     int64_t* n_offs_out, int64_t* end_out, int32_t* damaged_out) {
    std::vector<int64_t> offs;
    bool damaged = false;
    const int64_t rows = tsq_dec_walk_rows(data, n_bytes, n_cols, cap_rows, per, offs, end_out, &damaged);
    *damaged_out = damaged ? 1 : 0;
    *n_offs_out = (int64_t)offs.size();
    for (size_t i = 0; i < offs.size() && (int64_t)i < offs_cap; i++) offs_out[i] = offs[i];
    return rows;
}

// ---- key records (tsq_keyrec_dp.h): the record k_kr_hist / k_kr_scatter / k_kd_assign build for every row, and its 64-bit mix
#include "../../tinysql_amd/csrc/tsq_keyrec_dp.h"
// status[r]: 0 = a record, 1 = the row has no key (NULL cell of a join key, selected == 0), 2 = the cells do not fit 32 bytes
// generic != 0: run-time positions (kr_record_any) whatever the key columns are — the fixed-position builders must agree with it
extern "C" void sim_kr_records(const tsq_col* cols, int32_t n_cols, const int32_t* key_cols, int32_t n_keys, int32_t keep_nulls, const uint8_t* selected, int64_t nrows,
                               uint64_t* rec, uint8_t* status, uint64_t* hash, int32_t generic) {
    KrSrc s;
    memset(&s, 0, sizeof s);
    fill(s.cs, cols, n_cols);
    s.n_keys = n_keys;
    for (int k = 0; k < n_keys; k++) s.col[k] = key_cols[k];
    s.keep_nulls = keep_nulls;
    s.selected = selected;
    s.nrows = nrows;
    {
        int32_t kt[TSQ_MAX_KEYS];
        for (int k = 0; k < n_keys; k++) kt[k] = cols[key_cols[k]].type;
        s.layout = generic ? 0 : kr_layout_of(kt, n_keys);
    }
    for (int64_t r = 0; r < nrows; r++) {
        uint64_t w[4];
        bool toolong = false;
        const bool ok = kr_record(s, r, w, &toolong);
        status[r] = ok ? 0 : (toolong ? 2 : 1);
        for (int q = 0; q < 4; q++) rec[r * 4 + q] = ok ? w[q] : 0;
        hash[r] = ok ? kr_hash(w) : 0;
    }
}

// the cells of record `rec` (32 bytes) read back (kr_parse_cell, what k_kd_decode does for the aggregate's output key columns)
extern "C" void sim_kr_parse(const uint8_t* rec, int32_t n_keys, const int32_t* is_str, uint32_t* flag, uint64_t* word, uint32_t* off, uint32_t* len) {
    uint32_t at = 0;
    for (int k = 0; k < n_keys; k++) flag[k] = kr_parse_cell(rec, &at, is_str[k] != 0, &word[k], &off[k], &len[k]);
}
