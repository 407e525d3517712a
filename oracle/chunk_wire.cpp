/*
 * chunk_wire.cpp — CPU restatement of the chunk wire format: chunk.Codec and chunk.Decoder (SURVEY.md §8 a/A "wire Codec").
 * TEST INFRASTRUCTURE ONLY (see oracle.h): the product never links or calls this file.
 *
 * Follows, statement by statement (a Go slice is a std::vector plus a start index where the reference re-slices it):
 *   Column{length, nullBitmap, offsets, data, elemBuf}                      util/chunk/column.go:28-34
 *   newFixedLenColumn / newVarLenColumn (offsets = [0])                      util/chunk/chunk.go:126-148
 *   Column.nullCount / appendMultiSameNullBitmap                             util/chunk/column.go:316-328, 127-147
 *   Codec.Encode / encodeColumn                                              util/chunk/codec.go:42-76
 *   Codec.DecodeToChunk / decodeColumn / setAllNotNull                       util/chunk/codec.go:88-155
 *   Decoder.{Reset, Decode, IsFinished, RemainedRows, ReuseIntermChk, decodeColumn}   util/chunk/codec.go:246-353
 * The reference slices out of range (and panics) on a damaged buffer; here those accesses return -1.
 * Pinned on the reference's own TestCodec (util/chunk/codec_test.go:29-71) in
 * tests/test_oracle_codec_golden.py::test_storage_boundary_codecs_against_the_transcribed_vectors (the "chunk_codec" vector of
 * tests/golden/codec_cases.json: the wire length the reference asserts, the per-column lengths and headers, the rows after
 * DecodeToChunk).  The reference's test holds no golden BYTES beyond those sizes and headers; the byte layout itself is pinned by
 * the statement-level restatement of codec.go:50-76 below and by tests/test_hostsim_wire.py (walks every alignment).
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {
struct WCol {
    int64_t length = 0;
    std::vector<uint8_t> null_bitmap;
    std::vector<int64_t> offsets;
    std::vector<uint8_t> data;
    int elem = 8;  // len(elemBuf), or -1: var-len
    // what the Decoder's re-slicing of its intermediate chunk has consumed (codec.go:321, 345, 349)
    size_t bm0 = 0, off0 = 0, data0 = 0;
    bool fixed() const { return elem > 0; }
};
}  // namespace

struct orc_wire_chunk {
    std::vector<WCol> cols;
    int64_t remained_rows = 0;  // Decoder.remainedRows when this chunk is a Decoder's intermChk
};

namespace {
int64_t null_count(const WCol& c) {  // column.go:316-328
    int64_t cnt = 0, i = 0;
    for (; i + 8 <= c.length; i += 8) cnt += 8 - __builtin_popcount(c.null_bitmap[c.bm0 + (size_t)(i >> 3)]);
    for (; i < c.length; i++)
        if (((c.null_bitmap[c.bm0 + (size_t)(i >> 3)] >> (i & 7)) & 1) == 0) cnt++;
    return cnt;
}
void put_u32(std::vector<uint8_t>& b, uint32_t v) {
    for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i)));
}
void encode_column(std::vector<uint8_t>& buffer, const WCol& col) {  // codec.go:50-76
    put_u32(buffer, (uint32_t)col.length);
    const int64_t nc = null_count(col);
    put_u32(buffer, (uint32_t)nc);
    if (nc > 0) {
        const size_t n = (size_t)((col.length + 7) / 8);
        buffer.insert(buffer.end(), col.null_bitmap.begin() + col.bm0, col.null_bitmap.begin() + col.bm0 + n);
    }
    if (!col.fixed()) {
        const size_t n = (size_t)(col.length + 1);
        const uint8_t* p = (const uint8_t*)(col.offsets.data() + col.off0);
        buffer.insert(buffer.end(), p, p + n * 8);
    }
    buffer.insert(buffer.end(), col.data.begin() + col.data0, col.data.end());
}
// codec.go:96-143; returns the position behind the column, or -1 where the reference would slice out of range
int64_t decode_column(const uint8_t* buffer, int64_t n, int64_t pos, WCol& col) {
    if (n - pos < 8) return -1;
    uint32_t len32, nc32;
    memcpy(&len32, buffer + pos, 4);
    memcpy(&nc32, buffer + pos + 4, 4);
    pos += 8;
    col.length = (int64_t)len32;
    col.bm0 = col.off0 = col.data0 = 0;
    const int64_t nbm = (col.length + 7) / 8;
    if (nc32 > 0) {
        if (n - pos < nbm) return -1;
        col.null_bitmap.assign(buffer + pos, buffer + pos + nbm);
        pos += nbm;
    } else {
        col.null_bitmap.assign((size_t)nbm, 0xFF);  // setAllNotNull (codec.go:147-155; allNotNullBitmap is all 0xFF, :227-231)
    }
    int64_t num_data = (int64_t)col.elem * col.length;
    if (!col.fixed()) {
        const int64_t nob = (col.length + 1) * 8;
        if (n - pos < nob) return -1;
        col.offsets.resize((size_t)col.length + 1);
        memcpy(col.offsets.data(), buffer + pos, (size_t)nob);
        pos += nob;
        num_data = col.offsets[(size_t)col.length];
        if (num_data < 0) return -1;
    }
    if (n - pos < num_data) return -1;
    col.data.assign(buffer + pos, buffer + pos + num_data);
    return pos + num_data;
}
}  // namespace

extern "C" {

/* chunk.New over fields with getFixedLen = elem[c] (4, 8, or -1) */
orc_wire_chunk* orc_wire_new(const int32_t* elem, int32_t n_cols) {
    orc_wire_chunk* k = new orc_wire_chunk();
    k->cols.resize((size_t)n_cols);
    for (int c = 0; c < n_cols; c++) {
        k->cols[(size_t)c].elem = elem[c];
        if (elem[c] < 0) k->cols[(size_t)c].offsets.assign(1, 0);
    }
    return k;
}
void orc_wire_free(orc_wire_chunk* k) { delete k; }

/* a chunk holding the rows of tsq_col columns, byte for byte what Append* calls would have produced given these bytes
 * (a NULL bitmap pointer = every row NOT NULL: bitmap bytes 0xFF with the bits beyond the last row cleared, column.go:113-125) */
orc_wire_chunk* orc_wire_from_cols(const tsq_col* cols, int32_t n_cols, int64_t nrows) {
    orc_wire_chunk* k = new orc_wire_chunk();
    k->cols.resize((size_t)n_cols);
    for (int c = 0; c < n_cols; c++) {
        WCol& w = k->cols[(size_t)c];
        const tsq_col& s = cols[c];
        w.length = nrows;
        w.elem = s.type == TSQ_BYTES ? -1 : (s.type == TSQ_F32 ? 4 : 8);
        const size_t nbm = (size_t)((nrows + 7) / 8);
        if (s.null_bitmap) {
            w.null_bitmap.assign(s.null_bitmap, s.null_bitmap + nbm);
            if (nrows & 7) w.null_bitmap[nbm - 1] &= (uint8_t)((1u << (nrows & 7)) - 1u);  // appendNullBitmap never sets a bit beyond the last row
        } else {
            w.null_bitmap.assign(nbm, 0xFF);
            if (nrows & 7) w.null_bitmap[nbm - 1] = (uint8_t)((1u << (nrows & 7)) - 1u);
        }
        if (w.fixed()) w.data.assign((const uint8_t*)s.data, (const uint8_t*)s.data + nrows * w.elem);
        else {
            w.offsets.assign(s.offsets, s.offsets + nrows + 1);
            w.data.assign((const uint8_t*)s.data, (const uint8_t*)s.data + s.offsets[nrows]);
        }
    }
    return k;
}

/* Codec.Encode (codec.go:42-48): bytes needed; written when they fit cap */
int64_t orc_wire_encode(const orc_wire_chunk* k, uint8_t* out, int64_t cap) {
    std::vector<uint8_t> buffer;
    for (const WCol& c : k->cols) encode_column(buffer, c);
    if ((int64_t)buffer.size() <= cap && !buffer.empty()) memcpy(out, buffer.data(), buffer.size());
    return (int64_t)buffer.size();
}

/* Codec.DecodeToChunk (codec.go:88-93): the bytes consumed (len(buffer) - len(remained)), or -1 */
int64_t orc_wire_decode_to_chunk(orc_wire_chunk* k, const uint8_t* buffer, int64_t n) {
    int64_t pos = 0;
    for (WCol& c : k->cols) {
        pos = decode_column(buffer, n, pos, c);
        if (pos < 0) return -1;
    }
    return pos;
}

/* Decoder.Reset (codec.go:272-275) on the intermediate chunk */
int64_t orc_wire_decoder_reset(orc_wire_chunk* interm, const uint8_t* data, int64_t n) {
    const int64_t used = orc_wire_decode_to_chunk(interm, data, n);
    if (used < 0) return -1;
    interm->remained_rows = interm->cols.empty() ? 0 : interm->cols[0].length;  // intermChk.NumRows()
    return used;
}
int64_t orc_wire_decoder_remained(const orc_wire_chunk* interm) { return interm->remained_rows; }

/* Decoder.Decode (codec.go:257-269, 298-353): required = chk.RequiredRows() - chk.NumRows(); returns the rows appended */
int64_t orc_wire_decoder_decode(orc_wire_chunk* interm, orc_wire_chunk* chk, int64_t required) {
    int64_t required_rows = (required + 7) >> 3 << 3;
    if (required_rows > interm->remained_rows) required_rows = interm->remained_rows;
    for (size_t i = 0; i < chk->cols.size(); i++) {
        WCol& src = interm->cols[i];
        WCol& dst = chk->cols[i];
        int64_t num_data = (int64_t)src.elem * required_rows;
        if (!src.fixed()) {
            num_data = src.offsets[src.off0 + (size_t)required_rows] - src.offsets[src.off0];
            const int64_t delta = dst.offsets[(size_t)dst.length] - src.offsets[src.off0];
            dst.offsets.resize((size_t)dst.length + 1);
            for (int64_t r = 1; r <= required_rows; r++) dst.offsets.push_back(src.offsets[src.off0 + (size_t)r] + delta);
            src.off0 += (size_t)required_rows;
        }
        const int64_t nbm = (required_rows + 7) >> 3;
        dst.null_bitmap.resize((size_t)((dst.length + 7) >> 3));
        if (dst.length % 8 == 0) {
            dst.null_bitmap.insert(dst.null_bitmap.end(), src.null_bitmap.begin() + src.bm0, src.null_bitmap.begin() + src.bm0 + nbm);
        } else {
            // appendMultiSameNullBitmap(false, requiredRows): zero bytes up to the new length (column.go:127-138)
            dst.null_bitmap.resize((size_t)((dst.length + required_rows + 7) >> 3), 0);
            const int64_t bitmap_len = (int64_t)dst.null_bitmap.size();
            const int bit_offset = (int)(dst.length % 8);
            const int64_t start = (dst.length - 1) >> 3;
            for (int64_t j = 0; j < nbm; j++) {
                const uint8_t s = src.null_bitmap[src.bm0 + (size_t)j];
                dst.null_bitmap[(size_t)(start + j)] |= (uint8_t)(s << bit_offset);
                if (start + j + 1 < bitmap_len) dst.null_bitmap[(size_t)(start + j + 1)] |= (uint8_t)(s >> (8 - bit_offset));
            }
        }
        if (!dst.null_bitmap.empty()) {
            const unsigned redundant = (unsigned)((int64_t)dst.null_bitmap.size() * 8 - dst.length - required_rows);
            dst.null_bitmap.back() &= (uint8_t)((1u << (8 - redundant)) - 1u);
        }
        src.bm0 += (size_t)nbm;
        dst.length += required_rows;
        dst.data.insert(dst.data.end(), src.data.begin() + src.data0, src.data.begin() + src.data0 + num_data);
        src.data0 += (size_t)num_data;
    }
    interm->remained_rows -= required_rows;
    return required_rows;
}

/* Decoder.ReuseIntermChk (codec.go:291-308): chk takes the remaining rows of the intermediate chunk, offsets rebased to 0 */
void orc_wire_decoder_reuse(orc_wire_chunk* interm, orc_wire_chunk* chk) {
    for (WCol& col : interm->cols) {
        col.length = interm->remained_rows;
        if (!col.fixed()) {
            const int64_t delta = col.offsets[col.off0];
            if (delta != 0)
                for (size_t j = col.off0; j < col.offsets.size(); j++) col.offsets[j] -= delta;
        }
    }
    chk->cols.swap(interm->cols);
    interm->remained_rows = 0;
}

/* the state of column c as the reference holds it (the slices as re-sliced so far) */
int64_t orc_wire_col_length(const orc_wire_chunk* k, int32_t c) { return k->cols[(size_t)c].length; }
int64_t orc_wire_col_bitmap(const orc_wire_chunk* k, int32_t c, uint8_t* out, int64_t cap) {
    const WCol& w = k->cols[(size_t)c];
    const int64_t n = (int64_t)(w.null_bitmap.size() - w.bm0);
    if (out && n <= cap && n > 0) memcpy(out, w.null_bitmap.data() + w.bm0, (size_t)n);
    return n;
}
int64_t orc_wire_col_offsets(const orc_wire_chunk* k, int32_t c, int64_t* out, int64_t cap) {
    const WCol& w = k->cols[(size_t)c];
    const int64_t n = (int64_t)(w.offsets.size() - w.off0);
    if (out && n <= cap && n > 0) memcpy(out, w.offsets.data() + w.off0, (size_t)n * 8);
    return n;
}
int64_t orc_wire_col_data(const orc_wire_chunk* k, int32_t c, uint8_t* out, int64_t cap) {
    const WCol& w = k->cols[(size_t)c];
    const int64_t n = (int64_t)(w.data.size() - w.data0);
    if (out && n <= cap && n > 0) memcpy(out, w.data.data() + w.data0, (size_t)n);
    return n;
}
}
