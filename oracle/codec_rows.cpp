/*
 * codec_rows.cpp — CPU restatement of the coprocessor-response row codec (SURVEY.md §8 f, rank 2).
 * TEST INFRASTRUCTURE ONLY (see oracle.h): the product never links or calls this file.
 *
 * Follows, line by line:
 *   encode(..., comparable)            util/codec/codec.go:74-99      (EncodeValue :205-209 / EncodeKey :199-203)
 *   encodeSignedInt / encodeUnsignedInt codec.go:145-154, 167-176
 *   valueSizeOfSignedInt / Unsigned    codec.go:156-165, 178-187      (closed form pinned by codec_test.go:771-809)
 *   EncodeInt/DecodeInt, EncodeUint/DecodeUint, EncodeVarint/DecodeVarint, EncodeUvarint/DecodeUvarint
 *                                      util/codec/number.go:24-130 (+ Go's encoding/binary PutVarint/Varint/Uvarint)
 *   EncodeFloat / DecodeFloat          util/codec/float.go:22-46
 *   Decoder.DecodeOne                  util/codec/codec.go:623-690
 *   appendFloatToChunk (TypeFloat -> float32)  codec.go:701-707
 *   selectResult.readRowsData          distsql/select_result.go:139-155 (rows are decoded one after the other until the
 *                                      chunk is full or the bytes are used up; the remainder is kept)
 * encoding/binary is Go standard library (not in /root/reference): Uvarint = little-endian base-128, MSB = continuation,
 * at most 10 bytes, the 10th byte at most 1 (else "overflow", n < 0); Varint = zig-zag over Uvarint.
 */
#include <cstdint>
#include <cstring>

#include <string>
#include <vector>

#include "orc_result_internal.h"

namespace {
const uint8_t NilFlag = 0, bytesFlag = 1, compactBytesFlag = 2, intFlag = 3, uintFlag = 4, floatFlag = 5, varintFlag = 8, uvarintFlag = 9;
const uint64_t signMask = 0x8000000000000000ULL;

size_t put_uvarint(uint8_t* b, uint64_t x) {  // encoding/binary.PutUvarint
    size_t i = 0;
    while (x >= 0x80) {
        b[i++] = (uint8_t)x | 0x80;
        x >>= 7;
    }
    b[i] = (uint8_t)x;
    return i + 1;
}
size_t put_varint(uint8_t* b, int64_t x) {  // encoding/binary.PutVarint: zig-zag
    uint64_t ux = (uint64_t)x << 1;
    if (x < 0) ux = ~ux;
    return put_uvarint(b, ux);
}
// encoding/binary.Uvarint: returns n > 0 bytes read, 0 = buffer too small, < 0 = overflow
int uvarint(const uint8_t* b, int64_t len, uint64_t* out) {
    uint64_t x = 0;
    unsigned s = 0;
    for (int64_t i = 0; i < len; i++) {
        if (i == 10) return -(int)(i + 1);  // MaxVarintLen64: overflow
        const uint8_t c = b[i];
        if (c < 0x80) {
            if (i == 9 && c > 1) return -(int)(i + 1);  // overflow
            *out = x | (uint64_t)c << s;
            return (int)i + 1;
        }
        x |= (uint64_t)(c & 0x7f) << s;
        s += 7;
    }
    *out = 0;
    return 0;
}
void put_be64(uint8_t* b, uint64_t u) {
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(u >> (56 - 8 * i));
}
uint64_t get_be64(const uint8_t* b) {
    uint64_t u = 0;
    for (int i = 0; i < 8; i++) u = (u << 8) | b[i];
    return u;
}
uint64_t encodeFloatToCmpUint64(double f) {  // float.go:22-30
    uint64_t u;
    memcpy(&u, &f, 8);
    if (f >= 0) u |= signMask; else u = ~u;
    return u;
}
double decodeCmpUintToFloat(uint64_t u) {  // float.go:32-40
    if (u & signMask) u &= ~signMask; else u = ~u;
    double f;
    memcpy(&f, &u, 8);
    return f;
}
bool is_null(const tsq_col& c, int64_t r) { return c.null_bitmap && ((c.null_bitmap[r >> 3] >> (r & 7)) & 1) == 0; }
}  // namespace

extern "C" {

/* valueSizeOfSignedInt (codec.go:156-165) / valueSizeOfUnsignedInt (:178-187) */
int32_t orc_value_size_signed(int64_t v) {
    if (v < 0) v = 0 - v - 1;
    int32_t size = 2;
    v = v >> 6;
    while (v > 0) { size++; v = v >> 7; }
    return size;
}
int32_t orc_value_size_unsigned(uint64_t v) {
    int32_t size = 2;
    v = v >> 7;
    while (v > 0) { size++; v = v >> 7; }
    return size;
}

/* encode (codec.go:74-99) of rows of fixed-width columns, row after row as the coprocessor writes them.
 * I64 -> KindInt64, U64 -> KindUint64, F32/F64 -> floatFlag + EncodeFloat(float64(v)), NULL -> NilFlag.
 * Returns the number of bytes written (the call fails with -1 if cap is too small: at most 11 bytes per value). */
static int64_t encode_rows_impl(const tsq_col* cols, int32_t n_cols, int64_t nrows, int32_t comparable, uint8_t* out, int64_t cap, int64_t* row_ends);
int64_t orc_encode_rows(const tsq_col* cols, int32_t n_cols, int64_t nrows, int32_t comparable, uint8_t* out, int64_t cap) {
    return encode_rows_impl(cols, n_cols, nrows, comparable, out, cap, nullptr);
}
static int64_t encode_rows_impl(const tsq_col* cols, int32_t n_cols, int64_t nrows, int32_t comparable, uint8_t* out, int64_t cap, int64_t* row_ends) {
    int64_t n = 0;
    for (int64_t r = 0; r < nrows; r++) {
        if (row_ends && r > 0) row_ends[r - 1] = n;
        for (int c = 0; c < n_cols; c++) {
            if (n + 11 > cap) return -1;
            const tsq_col& col = cols[c];
            if (is_null(col, r)) { out[n++] = NilFlag; continue; }
            if (col.type == TSQ_BYTES) {  // encodeBytes (codec.go:101-109): comparable -> bytesFlag + EncodeBytes (bytes.go:35-67);
                                          // otherwise compactBytesFlag + EncodeCompactBytes = varint(len) + the bytes (bytes.go:141-148)
                const int64_t lo = col.offsets[r], len = col.offsets[r + 1] - lo;
                if (comparable) {
                    if (n + 1 + (len / 8 + 1) * 9 > cap) return -1;
                    out[n++] = bytesFlag;
                    const uint8_t* d = (const uint8_t*)col.data + lo;
                    for (int64_t idx = 0; idx <= len; idx += 8) {  // [group1][marker1]...[groupN][markerN]
                        const int64_t remain = len - idx;
                        int pad = 0;
                        if (remain >= 8) { memcpy(out + n, d + idx, 8); n += 8; }
                        else {
                            pad = (int)(8 - remain);
                            memcpy(out + n, d + idx, (size_t)remain);
                            n += remain;
                            memset(out + n, 0, (size_t)pad);
                            n += pad;
                        }
                        out[n++] = (uint8_t)(0xFF - pad);
                    }
                    continue;
                }
                if (n + 11 + len > cap) return -1;
                out[n++] = compactBytesFlag;
                n += (int64_t)put_varint(out + n, len);
                memcpy(out + n, (const uint8_t*)col.data + lo, (size_t)len);
                n += len;
                continue;
            }
            switch (col.type) {
                case TSQ_I64: {
                    const int64_t v = ((const int64_t*)col.data)[r];
                    if (comparable) { out[n++] = intFlag; put_be64(out + n, (uint64_t)v ^ signMask); n += 8; }  // number.go:24-42
                    else { out[n++] = varintFlag; n += (int64_t)put_varint(out + n, v); }                       // number.go:107-111
                    break;
                }
                case TSQ_U64: {
                    const uint64_t v = ((const uint64_t*)col.data)[r];
                    if (comparable) { out[n++] = uintFlag; put_be64(out + n, v); n += 8; }
                    else { out[n++] = uvarintFlag; n += (int64_t)put_uvarint(out + n, v); }
                    break;
                }
                default: {
                    const double f = col.type == TSQ_F32 ? (double)((const float*)col.data)[r] : ((const double*)col.data)[r];
                    out[n++] = floatFlag;
                    put_be64(out + n, encodeFloatToCmpUint64(f));
                    n += 8;
                }
            }
        }
    }
    if (row_ends && nrows > 0) row_ends[nrows - 1] = n;
    return n;
}

/* readRowsData (select_result.go:139-155) over Decoder.DecodeOne (codec.go:623-690): decode rows until cap_rows rows are
 * out or the bytes are used up.  out_data[c]: 8 bytes per row (4 for TSQ_F32 columns), out_notnull[c]: one byte per row.
 * Status: 0 ok; 1 "invalid encoded key" (a row ends in the middle: DecodeOne called with no bytes left);
 * 2 "insufficient bytes to decode value"; 3 "value larger than 64 bits"; 4 "invalid encoded key flag"; 5 var-len flag
 * (bytes / compact bytes: not a fixed-width column).  On error *nrows_out holds the complete rows decoded before it. */
int32_t orc_decode_rows(const uint8_t* data, int64_t n_bytes, int32_t n_cols, const int32_t* types, int64_t cap_rows, void** out_data,
                        uint8_t** out_notnull, int64_t* nrows_out, int64_t* consumed) {
    int64_t pos = 0, rows = 0;
    *nrows_out = 0;
    *consumed = 0;
    while (rows < cap_rows && pos < n_bytes) {
        for (int c = 0; c < n_cols; c++) {
            if (n_bytes - pos < 1) return 1;  // codec.go:624-626
            const uint8_t flag = data[pos++];
            const uint8_t* b = data + pos;
            const int64_t left = n_bytes - pos;
            uint64_t bits = 0;
            bool isnull = false, real = false;
            switch (flag) {
                case intFlag:  // DecodeInt (number.go:44-53)
                    if (left < 8) return 2;
                    bits = get_be64(b) ^ signMask;
                    pos += 8;
                    break;
                case uintFlag:  // DecodeUint (number.go:82-90)
                    if (left < 8) return 2;
                    bits = get_be64(b);
                    pos += 8;
                    break;
                case varintFlag: {  // DecodeVarint (number.go:113-123) over binary.Varint
                    uint64_t ux;
                    const int k = uvarint(b, left, &ux);
                    if (k < 0) return 3;
                    if (k == 0) return 2;
                    int64_t x = (int64_t)(ux >> 1);
                    if (ux & 1) x = ~x;
                    bits = (uint64_t)x;
                    pos += k;
                    break;
                }
                case uvarintFlag: {
                    uint64_t ux;
                    const int k = uvarint(b, left, &ux);
                    if (k < 0) return 3;
                    if (k == 0) return 2;
                    bits = ux;
                    pos += k;
                    break;
                }
                case floatFlag: {  // DecodeFloat (float.go:42-46)
                    if (left < 8) return 2;
                    const double f = decodeCmpUintToFloat(get_be64(b));
                    memcpy(&bits, &f, 8);
                    real = true;
                    pos += 8;
                    break;
                }
                case NilFlag: isnull = true; break;
                case bytesFlag:
                case compactBytesFlag: return 5;
                default: return 4;  // codec.go:683
            }
            // appendIntToChunk / appendUintToChunk / appendFloatToChunk (codec.go:692-707): the value goes into column c as
            // decoded; only TypeFloat narrows to float32.  AppendNull writes a zero slot (column.go:150-158).
            if (types[c] == TSQ_F32) {
                float f32 = 0;
                if (!isnull) {
                    if (real) { double f; memcpy(&f, &bits, 8); f32 = (float)f; }
                    else memcpy(&f32, &bits, 4);  // an int datum in a float column: the raw bytes (AppendInt64 on a 4-byte column is not meaningful)
                }
                ((float*)out_data[c])[rows] = f32;
            } else {
                ((uint64_t*)out_data[c])[rows] = isnull ? 0 : bits;
            }
            out_notnull[c][rows] = isnull ? 0 : 1;
        }
        rows++;
        *nrows_out = rows;
        *consumed = pos;
    }
    return 0;
}

namespace {
// Decoder.DecodeOne (codec.go:623-690) of the value at data[pos ..end) for a column of type `type`: 0, or the status of
// orc_decode_rows_chunks.  A string cell is (pointer, length); a memcomparable one is decoded into `owned` first.
int decode_one(const uint8_t* data, int64_t& pos, int64_t end, int32_t type, uint64_t& bits, uint8_t& nn, std::pair<const uint8_t*, int64_t>& cell,
               std::string& owned) {
    if (end - pos < 1) return 1;  // codec.go:624-626
    const uint8_t flag = data[pos++];
    const uint8_t* b = data + pos;
    const int64_t left = end - pos;
    int kind = 0;  // 0 null 1 int 2 uint 3 real 4 bytes
    switch (flag) {
        case intFlag: if (left < 8) return 2; bits = get_be64(b) ^ signMask; pos += 8; kind = 1; break;
        case uintFlag: if (left < 8) return 2; bits = get_be64(b); pos += 8; kind = 2; break;
        case floatFlag: {
            if (left < 8) return 2;
            const double f = decodeCmpUintToFloat(get_be64(b));
            memcpy(&bits, &f, 8);
            pos += 8;
            kind = 3;
            break;
        }
        case varintFlag:
        case uvarintFlag:
        case compactBytesFlag: {
            uint64_t ux;
            const int kb = uvarint(b, left, &ux);
            if (kb < 0) return 3;
            if (kb == 0) return 2;
            pos += kb;
            int64_t x = (int64_t)(ux >> 1);
            if (ux & 1) x = ~x;
            if (flag == uvarintFlag) { bits = ux; kind = 2; }
            else if (flag == varintFlag) { bits = (uint64_t)x; kind = 1; }
            else {
                if (x < 0 || end - pos < x) return 2;  // "insufficient bytes to decode value, expected length"
                cell = {data + pos, x};
                pos += x;
                kind = 4;
            }
            break;
        }
        case NilFlag: break;
        case bytesFlag: {  // decodeBytes (bytes.go:69-112)
            std::string& buf = owned;
            for (;;) {
                if (end - pos < 9) return 2;
                const uint8_t* group = data + pos;
                const unsigned pad = 0xFFu - group[8];
                if (pad > 8) return 7;  // "invalid marker byte"
                buf.append((const char*)group, 8 - pad);
                pos += 9;
                if (pad != 0) {
                    for (unsigned q = 8 - pad; q < 8; q++)
                        if (group[q] != 0) return 8;  // "invalid padding byte"
                    break;
                }
            }
            cell = {(const uint8_t*)buf.data(), (int64_t)buf.size()};
            kind = 4;
            break;
        }
        default: return 4;
    }
    if (kind != 0 && (type == TSQ_BYTES) != (kind == 4)) return 6;
    nn = kind != 0;
    if (kind != 0 && type == TSQ_F32) {
        float f32;
        if (kind == 3) { double f; memcpy(&f, &bits, 8); f32 = (float)f; }
        else memcpy(&f32, &bits, 4);
        uint32_t w;
        memcpy(&w, &f32, 4);
        bits = w;
    }
    return 0;
}
}  // namespace

/* selectResult over the chunks of a response (select_result.go:102-155): every chunk is decoded to its end with DecodeOne
 * (codec.go:623-690), bytes datums included (compactBytesFlag -> DecodeCompactBytes, bytes.go:150-160 -> chk.AppendBytes).  Status as
 * orc_decode_rows, plus 6 = a datum whose kind cannot go into the column (a string for a number column or the reverse); a
 * memcomparable bytes datum (bytesFlag -> DecodeBytes, bytes.go:69-118): 7 = "invalid marker byte", 8 = "invalid padding byte".
 * The result holds the complete rows before the first offending value. */
orc_result* orc_decode_rows_chunks(const uint8_t* data, int64_t n_bytes, const int64_t* chunk_offsets, int64_t n_chunks, int32_t n_cols,
                                   const int32_t* types, int32_t* status) {
    orc_result* res = new orc_result();
    res->cols.resize((size_t)n_cols);
    for (int c = 0; c < n_cols; c++) res->cols[c].type = types[c];
    *status = 0;
    for (int64_t k = 0; k < n_chunks; k++) {
        int64_t pos = chunk_offsets[k];
        const int64_t end = chunk_offsets[k + 1];
        if (pos < 0 || end < pos || end > n_bytes) { *status = 1; return res; }
        while (pos < end) {
            std::vector<uint64_t> bits((size_t)n_cols, 0);
            std::vector<uint8_t> nn((size_t)n_cols, 0);
            std::vector<std::pair<const uint8_t*, int64_t>> cell((size_t)n_cols, {nullptr, 0});
            std::vector<std::string> owned((size_t)n_cols);  // DecodeBytes output (a memcomparable cell is not contiguous in the response)
            for (int c = 0; c < n_cols; c++) {
                const int st = decode_one(data, pos, end, types[c], bits[(size_t)c], nn[(size_t)c], cell[(size_t)c], owned[(size_t)c]);
                if (st != 0) { *status = st; return res; }
            }
            for (int c = 0; c < n_cols; c++) {
                if (types[c] == TSQ_BYTES && nn[c]) res->cols[c].append_bytes(cell[c].first, (size_t)cell[c].second);
                else res->cols[c].append_raw(bits[c], nn[c] != 0);
            }
            res->rows++;
        }
    }
    return res;
}

/* Index keys of the rows of a chunk: EncodeIndexSeekKey (tablecodec.go:87-93) = 't' | EncodeInt(tableID) | "_i" | EncodeInt(indexID) |
 * EncodeKey(values...) (codec.go:199-203), and behind it the handle as an int datum where handle_in_key[r] != 0 (a non-unique index:
 * `EncodeKey(.., append(values, handle))`).  key_offsets_out: nrows + 1 entries.  Returns the bytes written, -1 when cap is too small. */
int64_t orc_encode_index_keys(const tsq_col* cols, int32_t n_cols, int64_t nrows, int64_t table_id, int64_t index_id, const int64_t* handles,
                              const uint8_t* handle_in_key, uint8_t* out, int64_t cap, int64_t* key_offsets_out) {
    std::vector<uint8_t> vals((size_t)cap + 64);
    std::vector<int64_t> ends((size_t)nrows + 1, 0);
    const int64_t got = encode_rows_impl(cols, n_cols, nrows, 1, vals.data(), cap, ends.data());
    if (got < 0) return -1;
    int64_t n = 0, prev = 0;
    for (int64_t r = 0; r < nrows; r++) {
        key_offsets_out[r] = n;
        const int64_t len = ends[(size_t)r] - prev;
        if (n + 19 + len + 9 > cap) return -1;
        out[n++] = 't';
        put_be64(out + n, (uint64_t)table_id ^ signMask); n += 8;  // codec.EncodeInt = EncodeIntToCmpUint, big endian (number.go:24-42)
        out[n++] = '_'; out[n++] = 'i';
        put_be64(out + n, (uint64_t)index_id ^ signMask); n += 8;
        memcpy(out + n, vals.data() + prev, (size_t)len); n += len;
        prev = ends[(size_t)r];
        if (handle_in_key && handle_in_key[r]) { out[n++] = intFlag; put_be64(out + n, (uint64_t)handles[r] ^ signMask); n += 8; }
    }
    key_offsets_out[nrows] = n;
    return n;
}

/* indexScanExec (mocktikv/executor.go:191-320) over all pairs of a range: tablecodec.DecodeIndexKV (tablecodec.go:376-434) —
 * CutIndexKeyNew cuts n_index_cols values behind the 19-byte prefix, what remains is the handle datum (kept when pk_status != 0),
 * otherwise the pair's value holds the handle (DecodeIndexValueAsHandle, :456-465: 8 bytes big endian) — and every cut value through
 * DecodeOne into its column (what the executors above the scan do with it).  types: n_index_cols (+ 1: the handle column when
 * pk_status != 0; 1 = signed, 2 = unsigned).  Status as orc_decode_rows_chunks, plus 9 = no handle in key or value; the result holds
 * the pairs before the first offending one. */
orc_result* orc_decode_index_kv(const uint8_t* keys, int64_t n_bytes, const int64_t* key_offsets, int64_t n_keys, const uint8_t* values,
                                const int64_t* value_offsets, int32_t n_index_cols, const int32_t* types, int32_t pk_status, int32_t* status) {
    const int n_cols = n_index_cols + (pk_status != 0 ? 1 : 0);
    orc_result* res = new orc_result();
    res->cols.resize((size_t)n_cols);
    for (int c = 0; c < n_cols; c++) res->cols[(size_t)c].type = types[c];
    *status = 0;
    for (int64_t k = 0; k < n_keys; k++) {
        int64_t pos = key_offsets[k];
        const int64_t end = key_offsets[k + 1];
        if (pos < 0 || end < pos || end > n_bytes || end - pos < 19) { *status = 1; return res; }
        pos += 19;  // key[prefixLen+idLen:]
        std::vector<uint64_t> bits((size_t)n_cols, 0);
        std::vector<uint8_t> nn((size_t)n_cols, 0);
        std::vector<std::pair<const uint8_t*, int64_t>> cell((size_t)n_cols, {nullptr, 0});
        std::vector<std::string> owned((size_t)n_cols);
        for (int c = 0; c < n_index_cols; c++) {
            const int st = decode_one(keys, pos, end, types[c], bits[(size_t)c], nn[(size_t)c], cell[(size_t)c], owned[(size_t)c]);
            if (st != 0) { *status = st; return res; }
        }
        if (pk_status != 0) {
            const size_t hc = (size_t)n_index_cols;
            if (pos < end) {  // len(b) > 0: values = append(values, b)
                const int st = decode_one(keys, pos, end, TSQ_I64, bits[hc], nn[hc], cell[hc], owned[hc]);
                if (st != 0) { *status = st; return res; }
                if (!nn[hc]) { *status = 6; return res; }  // a NULL handle datum
            } else {
                const int64_t vlo = value_offsets ? value_offsets[k] : 0, vhi = value_offsets ? value_offsets[k + 1] : -1;
                if (!values || vhi - vlo < 8) { *status = 9; return res; }
                bits[hc] = get_be64(values + vlo);
                nn[hc] = 1;
            }
        }
        for (int c = 0; c < n_cols; c++) {
            if (types[c] == TSQ_BYTES && nn[(size_t)c]) res->cols[(size_t)c].append_bytes(cell[(size_t)c].first, (size_t)cell[(size_t)c].second);
            else res->cols[(size_t)c].append_raw(bits[(size_t)c], nn[(size_t)c] != 0);
        }
        res->rows++;
    }
    return res;
}
}
