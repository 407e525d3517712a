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
