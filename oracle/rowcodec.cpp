/*
 * rowcodec.cpp — CPU restatement of the stored-row format (rowcodec v2) and its decoders (SURVEY.md §8 f, rank 4).
 * TEST INFRASTRUCTURE ONLY (see oracle.h): the product never links or calls this file.
 *
 * Follows, line by line:
 *   Encoder.Encode / appendColVal / reformatCols / encodeRowCols   util/rowcodec/encoder.go:34-166
 *   EncodeValueDatum                                               encoder.go:168-194
 *   encodeInt / encodeUint / decodeInt / decodeUint                util/rowcodec/common.go:84-114, 180-210
 *   row.toBytes / fromBytes / getData / findColID                  util/rowcodec/row.go:37-150
 *   ChunkDecoder.DecodeToChunk / decodeColToChunk                  util/rowcodec/decoder.go:158-238
 *   BytesDecoder.DecodeToBytes / encodeOldDatum / fieldType2Flag   decoder.go:252-355
 *   codec.EncodeFloat / DecodeFloat                                util/codec/float.go:22-46
 * Where the reference would panic (index or slice bounds out of range on a damaged row) the restatement returns status 2.
 *
 * Pinning (tests/test_oracle_rowcodec_golden.py): the format has no byte-level golden vectors in the reference, its tests are
 * round trips and cross-format equalities — rowcodec_test.go:49-163 (handle column), :165-328 (all types, small / large ids /
 * large data), :330-438 (NULL and defaults), :440-503 (DecodeToBytes == tablecodec.EncodeValue byte for byte: checked against
 * the codec restatement of codec_rows.cpp, itself pinned on codec_test.go), :505-556 (ColumnIsNull); the header layout is
 * pinned by hand-assembled rows that follow row.toBytes (row.go:80-99) field by field.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "orc_result_internal.h"

namespace {
const uint8_t CodecVer = 128;
const uint8_t NilFlag = 0, CompactBytesFlag = 2, IntFlag = 3, UintFlag = 4, FloatFlag = 5, VarintFlag = 8, VaruintFlag = 9;
const uint64_t signMask = 0x8000000000000000ULL;

bool is_null(const tsq_col& c, int64_t r) { return c.null_bitmap && ((c.null_bitmap[r >> 3] >> (r & 7)) & 1) == 0; }

void encodeInt(std::vector<uint8_t>& buf, int64_t v) {  // common.go:84-101
    if ((int64_t)(int8_t)v == v) buf.push_back((uint8_t)v);
    else if ((int64_t)(int16_t)v == v) for (int i = 0; i < 2; i++) buf.push_back((uint8_t)((uint64_t)v >> (8 * i)));
    else if ((int64_t)(int32_t)v == v) for (int i = 0; i < 4; i++) buf.push_back((uint8_t)((uint64_t)v >> (8 * i)));
    else for (int i = 0; i < 8; i++) buf.push_back((uint8_t)((uint64_t)v >> (8 * i)));
}
void encodeUint(std::vector<uint8_t>& buf, uint64_t v) {  // common.go:180-197
    if ((uint64_t)(uint8_t)v == v) buf.push_back((uint8_t)v);
    else if ((uint64_t)(uint16_t)v == v) for (int i = 0; i < 2; i++) buf.push_back((uint8_t)(v >> (8 * i)));
    else if ((uint64_t)(uint32_t)v == v) for (int i = 0; i < 4; i++) buf.push_back((uint8_t)(v >> (8 * i)));
    else for (int i = 0; i < 8; i++) buf.push_back((uint8_t)(v >> (8 * i)));
}
void encodeFloat(std::vector<uint8_t>& buf, double f) {  // float.go:22-30 + EncodeUint (big endian)
    uint64_t u;
    memcpy(&u, &f, 8);
    if (f >= 0) u |= signMask; else u = ~u;
    for (int i = 0; i < 8; i++) buf.push_back((uint8_t)(u >> (56 - 8 * i)));
}

// what the Go `row` struct holds after fromBytes (row.go:20-35)
struct Row {
    bool large = false;
    uint32_t numNotNullCols = 0, numNullCols = 0;
    const uint8_t* colIDs = nullptr;    // small: one byte each; large: four little-endian bytes each
    const uint8_t* offsets = nullptr;   // small: uint16 each; large: uint32 each
    const uint8_t* data = nullptr;
    int64_t data_len = 0;
    int64_t id(uint32_t i) const {
        if (!large) return colIDs[i];
        uint32_t v;
        memcpy(&v, colIDs + 4 * i, 4);
        return v;
    }
    uint32_t off(uint32_t i) const {
        if (!large) { uint16_t v; memcpy(&v, offsets + 2 * i, 2); return v; }
        uint32_t v;
        memcpy(&v, offsets + 4 * i, 4);
        return v;
    }
};

// row.fromBytes (row.go:53-78).  0 ok, 1 errInvalidCodecVer, 2 = the reference panics (slice bounds out of range)
int fromBytes(Row& r, const uint8_t* rowData, int64_t len) {
    if (len < 1) return 2;
    if (rowData[0] != CodecVer) return 1;
    if (len < 6) return 2;
    r.large = (rowData[1] & 1) > 0;
    r.numNotNullCols = (uint32_t)rowData[2] | ((uint32_t)rowData[3] << 8);
    r.numNullCols = (uint32_t)rowData[4] | ((uint32_t)rowData[5] << 8);
    int64_t cursor = 6;
    const int64_t colIDsLen = (int64_t)(r.numNotNullCols + r.numNullCols) * (r.large ? 4 : 1);
    if (cursor + colIDsLen > len) return 2;
    r.colIDs = rowData + cursor;
    cursor += colIDsLen;
    const int64_t offsetsLen = (int64_t)r.numNotNullCols * (r.large ? 4 : 2);
    if (cursor + offsetsLen > len) return 2;
    r.offsets = rowData + cursor;
    cursor += offsetsLen;
    r.data = rowData + cursor;
    r.data_len = len - cursor;
    return 0;
}

// row.findColID (row.go:101-150)
void findColID(const Row& r, int64_t colID, int* idx, bool* isNil, bool* notFound) {
    *idx = 0;
    *isNil = false;
    *notFound = false;
    int i = 0, j = (int)r.numNotNullCols;
    while (i < j) {
        const int h = (int)((unsigned)(i + j) >> 1);
        const int64_t v = r.id((uint32_t)h);
        if (v < colID) i = h + 1;
        else if (v > colID) j = h;
        else { *idx = h; return; }
    }
    i = (int)r.numNotNullCols;
    j = (int)(r.numNotNullCols + r.numNullCols);
    while (i < j) {
        const int h = (int)((unsigned)(i + j) >> 1);
        const int64_t v = r.id((uint32_t)h);
        if (v < colID) i = h + 1;
        else if (v > colID) j = h;
        else { *isNil = true; return; }
    }
    *notFound = true;
}

// row.getData (row.go:37-52); false = slice bounds out of range
bool getData(const Row& r, int i, const uint8_t** val, int64_t* n) {
    uint32_t start = 0;
    if (i > 0) start = r.off((uint32_t)i - 1);
    const uint32_t end = r.off((uint32_t)i);
    if (start > end || (int64_t)end > r.data_len) return false;
    *val = r.data + start;
    *n = (int64_t)end - start;
    return true;
}

// decodeInt / decodeUint (common.go:103-114, 199-210); false = LittleEndian.Uint64 on fewer than 8 bytes (panic)
bool decodeInt(const uint8_t* val, int64_t n, int64_t* out) {
    switch (n) {
        case 1: *out = (int64_t)(int8_t)val[0]; return true;
        case 2: { uint16_t v; memcpy(&v, val, 2); *out = (int64_t)(int16_t)v; return true; }
        case 4: { uint32_t v; memcpy(&v, val, 4); *out = (int64_t)(int32_t)v; return true; }
        default: {
            if (n < 8) return false;
            uint64_t v;
            memcpy(&v, val, 8);
            *out = (int64_t)v;
            return true;
        }
    }
}
bool decodeUint(const uint8_t* val, int64_t n, uint64_t* out) {
    switch (n) {
        case 1: *out = val[0]; return true;
        case 2: { uint16_t v; memcpy(&v, val, 2); *out = v; return true; }
        case 4: { uint32_t v; memcpy(&v, val, 4); *out = v; return true; }
        default: {
            if (n < 8) return false;
            memcpy(out, val, 8);
            return true;
        }
    }
}
// codec.DecodeFloat (float.go:42-46) over DecodeUint (number.go:82-90); false = "insufficient bytes to decode value"
bool decodeFloat(const uint8_t* val, int64_t n, double* out) {
    if (n < 8) return false;
    uint64_t u = 0;
    for (int i = 0; i < 8; i++) u = (u << 8) | val[i];
    if (u & signMask) u &= ~signMask; else u = ~u;
    memcpy(out, &u, 8);
    return true;
}

size_t put_uvarint(uint8_t* b, uint64_t x) {  // encoding/binary.PutUvarint
    size_t i = 0;
    while (x >= 0x80) { b[i++] = (uint8_t)x | 0x80; x >>= 7; }
    b[i] = (uint8_t)x;
    return i + 1;
}
}  // namespace

extern "C" {

/* Encoder.Encode (encoder.go:34-43) for every row of a fixed-width chunk; column c carries column id col_ids[c]; datum kinds:
 * TSQ_I64 -> KindInt64, TSQ_U64 -> KindUint64, TSQ_F32 -> KindFloat32, TSQ_F64 -> KindFloat64 (both EncodeFloat(GetFloat64())),
 * NULL cell -> KindNull.  pad_col_id >= 0 adds a KindBytes datum of pad_len[r] bytes 'a' under that id (a var-len column next
 * to the fixed-width ones: ">65535 bytes of data" makes the row large, encoder.go:129-141).  Rows are written back to back,
 * offsets_out[r] .. offsets_out[r+1] bound row r.  Returns the bytes written, -1 when cap is too small. */
int64_t orc_rowcodec_encode(const tsq_col* cols, const int64_t* col_ids, int32_t n_cols, int64_t nrows, int64_t pad_col_id, const int64_t* pad_len,
                            uint8_t* out, int64_t cap, int64_t* offsets_out) {
    struct Val { int64_t id; int kind; /* 0 null 1 int 2 uint 3 float 4 bytes */ uint64_t bits; double f; int64_t blen; const uint8_t* bptr; };
    int64_t n = 0;
    offsets_out[0] = 0;
    for (int64_t r = 0; r < nrows; r++) {
        // reset + appendColVals (encoder.go:45-71)
        bool large = false;
        uint32_t numNotNull = 0, numNull = 0;
        std::vector<Val> vals;
        auto append = [&](int64_t id, Val v) {
            if (id > 255) large = true;
            if (v.kind == 0) numNull++; else numNotNull++;
            v.id = id;
            vals.push_back(v);
        };
        for (int c = 0; c < n_cols; c++) {
            Val v{0, 0, 0, 0.0, 0, nullptr};
            if (!is_null(cols[c], r)) {
                switch (cols[c].type) {
                    case TSQ_I64: v.kind = 1; v.bits = ((const uint64_t*)cols[c].data)[r]; break;
                    case TSQ_U64: v.kind = 2; v.bits = ((const uint64_t*)cols[c].data)[r]; break;
                    case TSQ_F32: v.kind = 3; v.f = (double)((const float*)cols[c].data)[r]; break;  // SetFloat32 keeps float64(f)
                    case TSQ_BYTES:  // KindString / KindBytes: the bytes as they are (EncodeValueDatum, encoder.go:180-181)
                        v.kind = 4;
                        v.blen = cols[c].offsets[r + 1] - cols[c].offsets[r];
                        v.bptr = (const uint8_t*)cols[c].data + cols[c].offsets[r];
                        break;
                    default: v.kind = 3; v.f = ((const double*)cols[c].data)[r];
                }
            }
            append(col_ids[c], v);
        }
        if (pad_col_id >= 0) {
            Val v{0, 4, 0, 0.0, pad_len[r], nullptr};
            append(pad_col_id, v);
        }
        // reformatCols (encoder.go:73-119): not-null columns first, each part sorted by id
        std::vector<Val> notnull, nulls;
        for (const Val& v : vals) (v.kind == 0 ? nulls : notnull).push_back(v);
        // the sorters compare the STORED ids: byte(colID) in a small row, uint32(colID) in a large one (common.go:212-266)
        auto stored = [&](int64_t id) { return large ? (uint64_t)(uint32_t)id : (uint64_t)(uint8_t)id; };
        auto by_id = [&](const Val& x, const Val& y) { return stored(x.id) < stored(y.id); };
        std::stable_sort(notnull.begin(), notnull.end(), by_id);
        std::stable_sort(nulls.begin(), nulls.end(), by_id);
        // encodeRowCols (encoder.go:121-166)
        std::vector<uint8_t> data;
        std::vector<uint32_t> offsets(notnull.size());
        for (size_t i = 0; i < notnull.size(); i++) {
            const Val& v = notnull[i];
            switch (v.kind) {  // EncodeValueDatum (encoder.go:168-194)
                case 1: encodeInt(data, (int64_t)v.bits); break;
                case 2: encodeUint(data, v.bits); break;
                case 3: encodeFloat(data, v.f); break;
                default:
                    if (v.bptr) data.insert(data.end(), v.bptr, v.bptr + v.blen);
                    else data.insert(data.end(), (size_t)v.blen, (uint8_t)'a');
            }
            if (data.size() > 65535 && !large) large = true;  // "handle convert to large" (ids and the offsets so far are widened)
            offsets[i] = (uint32_t)data.size();
        }
        // encoder.go:151-157: a row whose data is EXACTLY 65535 bytes is switched to large there with only the ids widened (the
        // 32-bit offsets are left unset — a reference bug that yields an undecodable row); this restatement writes the
        // offsets it means.  Not reachable with fixed-width values; the tests keep pad lengths away from it.
        if (!large && data.size() >= 65535) large = true;
        // row.toBytes (row.go:80-99)
        const int64_t need = 6 + (int64_t)(notnull.size() + nulls.size()) * (large ? 4 : 1) + (int64_t)notnull.size() * (large ? 4 : 2) + (int64_t)data.size();
        if (n + need > cap) return -1;
        uint8_t* b = out + n;
        *b++ = CodecVer;
        *b++ = large ? 1 : 0;
        *b++ = (uint8_t)numNotNull; *b++ = (uint8_t)(numNotNull >> 8);
        *b++ = (uint8_t)numNull; *b++ = (uint8_t)(numNull >> 8);
        auto put_id = [&](int64_t id) {
            if (large) { const uint32_t v = (uint32_t)id; memcpy(b, &v, 4); b += 4; }
            else *b++ = (uint8_t)id;
        };
        for (const Val& v : notnull) put_id(v.id);
        for (const Val& v : nulls) put_id(v.id);
        for (uint32_t o : offsets) {
            if (large) { memcpy(b, &o, 4); b += 4; }
            else { const uint16_t v = (uint16_t)o; memcpy(b, &v, 2); b += 2; }
        }
        if (!data.empty()) memcpy(b, data.data(), data.size());
        n += need;
        offsets_out[r + 1] = n;
    }
    return n;
}

/* The scan loop around ChunkDecoder.DecodeToChunk (decoder.go:158-198): row r = values[offsets[r], offsets[r+1]), handle =
 * handles[r].  out_data[c]: 8 bytes per row (4 for TSQ_F32), out_notnull[c]: one byte per row.  Status: 0 ok; 1 "invalid codec
 * version"; 2 the reference panics (damaged row / an int value of 3, 5, 6, 7 or 0 bytes); 3 "insufficient bytes to decode
 * value" (a real shorter than 8 bytes).  *nrows_out = rows decoded before the offending one. */
int32_t orc_rowcodec_decode(const uint8_t* values, const int64_t* offsets, const int64_t* handles, int64_t nrows, const tsq_rowcodec_col* cols,
                            int32_t n_cols, void** out_data, uint8_t** out_notnull, int64_t* nrows_out) {
    *nrows_out = 0;
    for (int64_t r = 0; r < nrows; r++) {
        Row row;
        const int st = fromBytes(row, values + offsets[r], offsets[r + 1] - offsets[r]);
        if (st) return st;
        for (int c = 0; c < n_cols; c++) {
            const tsq_rowcodec_col& col = cols[c];
            uint64_t bits = 0;
            bool notnull = false;
            if (col.flags & TSQ_RC_HANDLE) {  // col.ID == decoder.handleColID: chk.AppendInt64(colIdx, handle)
                bits = (uint64_t)handles[r];
                notnull = true;
            } else {
                int idx;
                bool isNil, notFound;
                findColID(row, col.col_id, &idx, &isNil, &notFound);
                if (!notFound && !isNil) {
                    const uint8_t* val;
                    int64_t n;
                    if (!getData(row, idx, &val, &n)) return 2;
                    // decodeColToChunk (decoder.go:200-238)
                    if (col.type == TSQ_I64) { int64_t v; if (!decodeInt(val, n, &v)) return 2; bits = (uint64_t)v; }
                    else if (col.type == TSQ_U64) { if (!decodeUint(val, n, &bits)) return 2; }
                    else {
                        double f;
                        if (!decodeFloat(val, n, &f)) return 3;
                        if (col.type == TSQ_F32) { const float f32 = (float)f; uint32_t w; memcpy(&w, &f32, 4); bits = w; }
                        else memcpy(&bits, &f, 8);
                    }
                    notnull = true;
                } else if (isNil) {
                    // chk.AppendNull(colIdx)
                } else if (col.flags & TSQ_RC_HAS_DEFAULT) {  // defDatum(colIdx) -> chk.AppendDatum
                    bits = col.def_bits;
                    notnull = true;
                }
            }
            if (col.type == TSQ_F32) ((uint32_t*)out_data[c])[r] = (uint32_t)bits;
            else ((uint64_t*)out_data[c])[r] = bits;
            out_notnull[c][r] = notnull ? 1 : 0;
        }
        *nrows_out = r + 1;
    }
    return 0;
}

/* The same scan loop with var-len columns (TSQ_BYTES: chk.AppendBytes(colIdx, colData), decoder.go:226-228; a NULL or absent
 * cell: AppendNull) into a materialised result.  *status as above; the result holds the rows decoded before the offending one. */
orc_result* orc_rowcodec_decode_chunk(const uint8_t* values, const int64_t* offsets, const int64_t* handles, int64_t nrows, const tsq_rowcodec_col* cols,
                                      int32_t n_cols, int32_t* status) {
    orc_result* res = new orc_result();
    res->cols.resize((size_t)n_cols);
    for (int c = 0; c < n_cols; c++) res->cols[c].type = cols[c].type;
    *status = 0;
    for (int64_t r = 0; r < nrows; r++) {
        Row row;
        int st = fromBytes(row, values + offsets[r], offsets[r + 1] - offsets[r]);
        // a row is appended column by column in the reference; this restatement appends it whole or not at all (the rows BEFORE the
        // offending one are what the callers compare)
        std::vector<uint64_t> bits((size_t)n_cols, 0);
        std::vector<uint8_t> nn((size_t)n_cols, 0);
        std::vector<std::pair<const uint8_t*, int64_t>> cell((size_t)n_cols, {nullptr, 0});
        std::vector<std::string> bitcell((size_t)n_cols);
        for (int c = 0; c < n_cols && !st; c++) {
            const tsq_rowcodec_col& col = cols[c];
            if (col.flags & TSQ_RC_HANDLE) { bits[c] = (uint64_t)handles[r]; nn[c] = 1; continue; }
            int idx;
            bool isNil, notFound;
            findColID(row, col.col_id, &idx, &isNil, &notFound);
            if (!notFound && !isNil) {
                const uint8_t* val;
                int64_t n;
                if (!getData(row, idx, &val, &n)) { st = 2; break; }
                if (col.type == TSQ_BYTES && (col.flags & TSQ_RC_BIT)) {
                    // byteSize := (Flen + 7) >> 3; NewBinaryLiteralFromUint(decodeUint(colData), byteSize): buf[8 - byteSize:] of
                    // BigEndian.PutUint64 (decoder.go:229-231, types/binary_literal.go:57-69)
                    uint64_t u;
                    if (!decodeUint(val, n, &u)) { st = 2; break; }
                    const int bsz = (int)TSQ_RC_BIT_SIZE(col.flags);
                    bitcell[c].resize((size_t)bsz);
                    for (int q = 0; q < bsz; q++) bitcell[c][(size_t)q] = (char)(uint8_t)(u >> (8 * (bsz - 1 - q)));
                    cell[c] = {(const uint8_t*)bitcell[c].data(), bsz};
                } else if (col.type == TSQ_BYTES) cell[c] = {val, n};
                else if (col.type == TSQ_I64) { int64_t v; if (!decodeInt(val, n, &v)) { st = 2; break; } bits[c] = (uint64_t)v; }
                else if (col.type == TSQ_U64) { if (!decodeUint(val, n, &bits[c])) { st = 2; break; } }
                else {
                    double f;
                    if (!decodeFloat(val, n, &f)) { st = 3; break; }
                    if (col.type == TSQ_F32) { const float f32 = (float)f; uint32_t w; memcpy(&w, &f32, 4); bits[c] = w; }
                    else memcpy(&bits[c], &f, 8);
                }
                nn[c] = 1;
            } else if (!isNil && (col.flags & TSQ_RC_HAS_DEFAULT)) {  // defDatum(colIdx) -> chk.AppendDatum (decoder.go:186-194)
                if (col.type == TSQ_BYTES) cell[c] = {col.def_bytes, col.def_len};
                else bits[c] = col.def_bits;
                nn[c] = 1;
            }
        }
        if (st) { *status = st; break; }
        for (int c = 0; c < n_cols; c++) {
            if (cols[c].type == TSQ_BYTES && nn[c]) res->cols[c].append_bytes(cell[c].first, (size_t)cell[c].second);
            else res->cols[c].append_raw(bits[c], nn[c] != 0);
        }
        res->rows = r + 1;
    }
    return res;
}

/* BytesDecoder.DecodeToBytes (decoder.go:252-302) for ONE row with outputOffset = column order and no default bytes: the old
 * datum bytes of every requested column, concatenated — one row of the RowsData a coprocessor table scan returns
 * (mocktikv/executor.go:124-196).  is_pk_handle[c] != 0 -> IsPKHandle.  Returns the bytes written, -1 invalid codec version,
 * -2 damaged row, -3 cap too small. */
int64_t orc_rowcodec_to_old_bytes(const uint8_t* row_data, int64_t len, int64_t handle, const tsq_rowcodec_col* cols, int32_t n_cols, uint8_t* out,
                                  int64_t cap) {
    Row r;
    const int st = fromBytes(r, row_data, len);
    if (st) return -st;
    int64_t n = 0;
    for (int c = 0; c < n_cols; c++) {
        if (n + 12 > cap) return -3;
        const tsq_rowcodec_col& col = cols[c];
        const bool is_unsigned = col.type == TSQ_U64;
        // fieldType2Flag (decoder.go:324-355): ints -> IntFlag / UintFlag, float and double -> FloatFlag
        const uint8_t tp = (col.type == TSQ_I64) ? IntFlag : (col.type == TSQ_U64) ? UintFlag : FloatFlag;
        if (col.flags & TSQ_RC_HANDLE) {  // col.IsPKHandle || colID == model.ExtraHandleID: comparable int / uint (decoder.go:263-273)
            out[n++] = is_unsigned ? UintFlag : IntFlag;
            uint64_t u = (uint64_t)handle;
            if (!is_unsigned) u ^= signMask;  // codec.EncodeInt = EncodeIntToCmpUint, big endian
            for (int i = 0; i < 8; i++) out[n++] = (uint8_t)(u >> (56 - 8 * i));
            continue;
        }
        int idx;
        bool isNil, notFound;
        findColID(r, col.col_id, &idx, &isNil, &notFound);
        if (!notFound && !isNil) {
            const uint8_t* val;
            int64_t vn;
            if (!getData(r, idx, &val, &vn)) return -2;
            // encodeOldDatum (decoder.go:304-322)
            if (tp == IntFlag) {
                int64_t v;
                if (!decodeInt(val, vn, &v)) return -2;
                out[n++] = VarintFlag;
                uint64_t ux = (uint64_t)v << 1;  // binary.PutVarint: zig-zag
                if (v < 0) ux = ~ux;
                n += (int64_t)put_uvarint(out + n, ux);
            } else if (tp == UintFlag) {
                uint64_t v;
                if (!decodeUint(val, vn, &v)) return -2;
                out[n++] = VaruintFlag;
                n += (int64_t)put_uvarint(out + n, v);
            } else {
                if (n + 1 + vn > cap) return -3;
                out[n++] = tp;
                memcpy(out + n, val, (size_t)vn);
                n += vn;
            }
            continue;
        }
        out[n++] = NilFlag;  // isNil, or absent without default bytes
    }
    (void)CompactBytesFlag;
    return n;
}

/* row.ColumnIsNull (row.go:152-165): 1 NULL, 0 not NULL, < 0 error; has_default = (defaultVal != nil) */
int32_t orc_rowcodec_column_is_null(const uint8_t* row_data, int64_t len, int64_t col_id, int32_t has_default) {
    Row r;
    const int st = fromBytes(r, row_data, len);
    if (st) return -st;
    int idx;
    bool isNil, notFound;
    findColID(r, col_id, &idx, &isNil, &notFound);
    if (notFound) return has_default ? 0 : 1;
    return isNil ? 1 : 0;
}
}
