/*
 * mocktikv.cpp — CPU restatement of the storage side's datum-level coprocessor aggregate and of tablecodec's record keys
 * (SURVEY.md §8 f, rank 4).  TEST INFRASTRUCTURE ONLY (see oracle.h): the product never links or calls this file.
 *
 * Follows, line by line:
 *   tablecodec.EncodeRowKeyWithHandle / appendTableRecordPrefix   tablecodec/tablecodec.go:57-70
 *   tablecodec.DecodeRowKey                                       tablecodec.go:235-242
 *   tablecodec.DecodeKeyHead                                      tablecodec.go:188-220
 *   tablecodec.DecodeRecordKey                                    tablecodec.go:73-77 — a course STUB in the reference ("Your code here");
 *       filled the way its own test demands (tablecodec_test.go:111-135): the table id and handle EncodeRowKeyWithHandle wrote, an
 *       error for anything that is not a record key of RecordRowKeyLen bytes
 *   codec.EncodeInt / DecodeInt / EncodeIntToCmpUint              util/codec/number.go:24-53
 *   mocktikv hashAggExec (Next / aggregate / getGroupKey / getContexts)   store/mockstore/mocktikv/aggregate.go:78-182
 *   aggregation.{count,sum,avg,maxMin,firstRow}Function + calculateSum    expression/aggregation/{count,sum,avg,max_min,first_row}.go,
 *       util.go:56-91, aggregation.go:118-135 (updateSum); NewDistAggFunc leaves Mode = CompleteMode (aggregation.go:46-72)
 *   types.ComputePlus / AddInt64 / ConvertUintToInt               types/datum_eval.go:22-56, types/overflow.go:33-40, types/convert.go:122-128
 *   codec.EncodeValue of a group-by datum                         util/codec/codec.go:74-99, 145-176 (varint forms), 101-109 (compact bytes)
 * The executor is sequential and row-at-a-time, so — unlike the SQL side's parallel HashAggExec — everything here is
 * deterministic: groups come out in first-seen order, FIRST_ROW is the first row of the scan with that key, and an int64 SUM
 * fails on the first RUNNING sum that leaves BIGINT.
 */
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "orc_result_internal.h"

namespace {
const uint64_t signMask = 0x8000000000000000ULL;
const uint8_t NilFlag = 0, compactBytesFlag = 2, floatFlag = 5, varintFlag = 8, uvarintFlag = 9;

void put_be64(uint8_t* b, uint64_t u) {
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(u >> (56 - 8 * i));
}
uint64_t get_be64(const uint8_t* b) {
    uint64_t u = 0;
    for (int i = 0; i < 8; i++) u = (u << 8) | b[i];
    return u;
}
void put_uvarint(std::string& b, uint64_t x) {  // encoding/binary.PutUvarint
    while (x >= 0x80) {
        b.push_back((char)((uint8_t)x | 0x80));
        x >>= 7;
    }
    b.push_back((char)(uint8_t)x);
}
void put_varint(std::string& b, int64_t x) {  // encoding/binary.PutVarint: zig-zag
    uint64_t ux = (uint64_t)x << 1;
    if (x < 0) ux = ~ux;
    put_uvarint(b, ux);
}

// types.Datum, the kinds a fixed-width / string chunk column can hold after DecodeOne (codec.go:623-690)
enum Kind { KNull = 0, KInt64, KUint64, KFloat64, KBytes };
struct Datum {
    Kind k = KNull;
    int64_t i = 0;
    uint64_t u = 0;
    double f = 0;
    std::string b;
    bool is_null() const { return k == KNull; }
};

bool cell_null(const tsq_col& c, int64_t r) { return c.null_bitmap && ((c.null_bitmap[r >> 3] >> (r & 7)) & 1) == 0; }
// tableScanExec hands DecodeOne's datums on: TypeFloat columns arrive as float64 (DecodeColumnValue does not unflatten)
Datum cell(const tsq_col& c, int64_t r) {
    Datum d;
    if (cell_null(c, r)) return d;
    switch (c.type) {
        case TSQ_I64: d.k = KInt64; d.i = ((const int64_t*)c.data)[r]; break;
        case TSQ_U64: d.k = KUint64; d.u = ((const uint64_t*)c.data)[r]; break;
        case TSQ_F32: d.k = KFloat64; d.f = (double)((const float*)c.data)[r]; break;
        case TSQ_F64: d.k = KFloat64; d.f = ((const double*)c.data)[r]; break;
        default: {  // TSQ_BYTES
            d.k = KBytes;
            const int64_t lo = c.offsets[r], hi = c.offsets[r + 1];
            d.b.assign((const char*)c.data + lo, (size_t)(hi - lo));
        }
    }
    return d;
}
// codec.EncodeValue (comparable = false) of one datum
void encode_value(std::string& out, const Datum& d) {
    switch (d.k) {
        case KNull: out.push_back((char)NilFlag); break;
        case KInt64: out.push_back((char)varintFlag); put_varint(out, d.i); break;
        case KUint64: out.push_back((char)uvarintFlag); put_uvarint(out, d.u); break;
        case KFloat64: {
            uint64_t u;
            memcpy(&u, &d.f, 8);
            if (d.f >= 0) u |= signMask; else u = ~u;  // encodeFloatToCmpUint64, float.go:22-30
            uint8_t be[8];
            put_be64(be, u);
            out.push_back((char)floatFlag);
            out.append((const char*)be, 8);
            break;
        }
        case KBytes:  // encodeBytes(..., comparable = false): compactBytesFlag + EncodeCompactBytes = varint(len) + bytes
            out.push_back((char)compactBytesFlag);
            put_varint(out, (int64_t)d.b.size());
            out.append(d.b);
            break;
    }
}
// Datum.CompareDatum for two datums of one column (same kind; NULL is smaller than everything, types/datum.go compareNull)
int compare_datum(const Datum& a, const Datum& b) {
    if (a.is_null()) return b.is_null() ? 0 : -1;
    if (b.is_null()) return 1;
    switch (a.k) {
        case KInt64: return a.i < b.i ? -1 : (a.i > b.i ? 1 : 0);
        case KUint64: return a.u < b.u ? -1 : (a.u > b.u ? 1 : 0);
        case KFloat64: return a.f < b.f ? -1 : (a.f > b.f ? 1 : 0);  // types.CompareFloat64
        default: {
            const int c = a.b.compare(b.b);
            return c < 0 ? -1 : (c > 0 ? 1 : 0);
        }
    }
}

// aggregation.AggEvaluateContext (aggregation.go:75-80)
struct AggCtx {
    int64_t Count = 0;
    Datum Value;
    bool GotFirstRow = false;
};

// calculateSum (util.go:56-91).  Returns false on error (err set).
bool calculate_sum(Datum& sum, const Datum& v, std::string& err) {
    Datum data;
    switch (v.k) {
        case KNull: break;
        case KInt64: data.k = KInt64; data.i = v.i; break;
        case KUint64:  // v.ToInt64: ConvertUintToInt(val, MaxInt64) — convert.go:122-128
            if (v.u > (uint64_t)INT64_MAX) { err = "constant overflows bigint"; return false; }
            data.k = KInt64;
            data.i = (int64_t)v.u;
            break;
        case KFloat64: data.k = KFloat64; data.f = v.f; break;
        default: err = "sum of a string: the planner casts first"; return false;
    }
    if (data.is_null()) return true;
    switch (sum.k) {
        case KNull: sum = data; return true;
        case KInt64: {  // ComputePlus -> AddInt64 (overflow.go:33-40)
            if (data.k != KInt64) { err = "invalid operation"; return false; }
            const int64_t a = sum.i, b = data.i;
            if ((a > 0 && b > 0 && INT64_MAX - a < b) || (a < 0 && b < 0 && INT64_MIN - a > b)) { err = "BIGINT value is out of range"; return false; }
            sum.i = a + b;
            return true;
        }
        case KFloat64:
            if (data.k != KFloat64) { err = "invalid operation"; return false; }
            sum.f = sum.f + data.f;
            return true;
        default: err = "invalid value for aggregate"; return false;
    }
}

void append_datum(OutCol& oc, const Datum& d) {
    switch (d.k) {
        case KNull: oc.append_raw(0, false); break;
        case KInt64: oc.append_raw((uint64_t)d.i, true); break;
        case KUint64: oc.append_raw(d.u, true); break;
        case KFloat64: { uint64_t u; memcpy(&u, &d.f, 8); oc.append_raw(u, true); break; }
        case KBytes: oc.append_bytes(d.b.data(), d.b.size()); break;
    }
}
int32_t datum_col_type(int32_t t) { return t == TSQ_F32 ? TSQ_F64 : t; }
}  // namespace

extern "C" {

/* EncodeRowKeyWithHandle (tablecodec.go:65-70): 't' + EncodeInt(tableID) + "_r" + EncodeInt(handle) = 19 bytes */
void orc_encode_row_key(int64_t table_id, int64_t handle, uint8_t* out19) {
    out19[0] = 't';
    put_be64(out19 + 1, (uint64_t)table_id ^ signMask);
    out19[9] = '_';
    out19[10] = 'r';
    put_be64(out19 + 11, (uint64_t)handle ^ signMask);
}
/* DecodeRowKey (tablecodec.go:235-242): 0 = ok, 1 = "invalid key" */
int32_t orc_decode_row_key(const uint8_t* key, int64_t len, int64_t* handle) {
    *handle = 0;
    if (len != 19 || key[0] != 't' || key[9] != '_' || key[10] != 'r') return 1;
    *handle = (int64_t)(get_be64(key + 11) ^ signMask);
    return 0;
}
/* DecodeKeyHead (tablecodec.go:188-220): 0 = ok, 1 = invalid key, 2 = insufficient bytes (DecodeInt, number.go:44-53) */
int32_t orc_decode_key_head(const uint8_t* key, int64_t len, int64_t* table_id, int64_t* index_id, int32_t* is_record) {
    *table_id = 0;
    *index_id = 0;
    *is_record = 0;
    if (len < 1 || key[0] != 't') return 1;
    if (len - 1 < 8) return 2;
    *table_id = (int64_t)(get_be64(key + 1) ^ signMask);
    const uint8_t* k = key + 9;
    const int64_t left = len - 9;
    if (left >= 2 && k[0] == '_' && k[1] == 'r') { *is_record = 1; return 0; }
    if (!(left >= 2 && k[0] == '_' && k[1] == 'i')) return 1;
    if (left - 2 < 8) return 2;
    *index_id = (int64_t)(get_be64(k + 2) ^ signMask);
    return 0;
}
/* DecodeRecordKey (tablecodec.go:73-77, STUB filled): a record key of RecordRowKeyLen bytes -> (tableID, handle) */
int32_t orc_decode_record_key(const uint8_t* key, int64_t len, int64_t* table_id, int64_t* handle) {
    int64_t idx;
    int32_t rec;
    *handle = 0;
    const int32_t st = orc_decode_key_head(key, len, table_id, &idx, &rec);
    if (st != 0) return st;
    if (!rec || len != 19) return 1;
    *handle = (int64_t)(get_be64(key + 11) ^ signMask);
    return 0;
}

/* hashAggExec (mocktikv/aggregate.go:78-182) over the rows of `cols`: every row is aggregated in scan order; output rows in
 * first-seen group order, each = the partial results of every function (AVG: count, then sum — avg.go:78-81) followed by the
 * group-by values.  cfg: group_key_col / aggs[].func, arg_col (-1 = the constant argument of COUNT(*)); modes are ignored
 * (NewDistAggFunc leaves CompleteMode).  *status: TSQ_OK | TSQ_ERR_OVERFLOW_BIGINT (a running int64 sum left BIGINT, or an unsigned
 * argument above MaxInt64) | TSQ_ERR_UNSUPPORTED. */
orc_result* orc_cop_hash_agg(const tsq_agg_cfg* cfg, const tsq_col* cols, int64_t nrows, tsq_status* status) {
    *status = TSQ_OK;
    std::unordered_map<std::string, size_t> groups;         // e.groups / e.aggCtxsMap keyed by the group key bytes
    std::vector<std::vector<Datum>> group_rows;             // e.groupKeyRows (the datums instead of their EncodeValue bytes)
    std::vector<std::vector<AggCtx>> ctxs;                  // getContexts
    std::string gk, err;
    for (int64_t r = 0; r < nrows; r++) {
        // getGroupKey (aggregate.go:119-143)
        gk.clear();
        std::vector<Datum> gvals;
        for (int g = 0; g < cfg->n_group_keys; g++) {
            gvals.push_back(cell(cols[cfg->group_key_col[g]], r));
            encode_value(gk, gvals.back());
        }
        auto it = groups.find(gk);
        size_t gi;
        if (it == groups.end()) {
            gi = group_rows.size();
            groups.emplace(gk, gi);
            group_rows.push_back(gvals);
            ctxs.emplace_back((size_t)cfg->n_aggs);
        } else {
            gi = it->second;
        }
        for (int a = 0; a < cfg->n_aggs; a++) {
            const tsq_agg_func& f = cfg->aggs[a];
            AggCtx& c = ctxs[gi][a];
            Datum v;
            if (f.arg_col >= 0) v = cell(cols[f.arg_col], r);
            else { v.k = KInt64; v.i = 1; }  // count(1)
            switch (f.func) {
                case TSQ_AGG_COUNT:  // count.go:28-45
                    if (!v.is_null()) c.Count++;
                    break;
                case TSQ_AGG_SUM:    // sum.go:27-29 -> updateSum (aggregation.go:118-135)
                case TSQ_AGG_AVG:    // avg.go:53-61, CompleteMode -> updateSum
                    if (v.is_null()) break;
                    if (!calculate_sum(c.Value, v, err)) {
                        orc_set_error(err);
                        *status = err == "sum of a string: the planner casts first" ? TSQ_ERR_UNSUPPORTED : TSQ_ERR_OVERFLOW_BIGINT;
                        return nullptr;
                    }
                    c.Count++;
                    break;
                case TSQ_AGG_MAX:
                case TSQ_AGG_MIN: {  // max_min.go:38-59
                    if (c.Value.is_null()) c.Value = v;
                    if (v.is_null()) break;
                    const int cmp = compare_datum(c.Value, v);
                    if ((f.func == TSQ_AGG_MAX && cmp == -1) || (f.func == TSQ_AGG_MIN && cmp == 1)) c.Value = v;
                    break;
                }
                case TSQ_AGG_FIRSTROW:  // first_row.go:28-42
                    if (c.GotFirstRow) break;
                    c.Value = v;
                    c.GotFirstRow = true;
                    break;
                default:
                    orc_set_error("Unknown aggregate function type");
                    *status = TSQ_ERR_UNSUPPORTED;
                    return nullptr;
            }
        }
    }
    // Next (aggregate.go:78-116): per group, GetPartialResult of every function, then the group-by row
    orc_result* res = new orc_result();
    for (int a = 0; a < cfg->n_aggs; a++) {
        const tsq_agg_func& f = cfg->aggs[a];
        const int32_t at = f.arg_col >= 0 ? cols[f.arg_col].type : TSQ_I64;
        auto add = [&](int32_t t) { res->cols.emplace_back(); res->cols.back().type = t; };
        switch (f.func) {
            case TSQ_AGG_COUNT: add(TSQ_I64); break;
            case TSQ_AGG_AVG: add(TSQ_I64);  // count first (avg.go:78-81), then the sum like SUM
                /* fallthrough */
            case TSQ_AGG_SUM: add((at == TSQ_F32 || at == TSQ_F64) ? TSQ_F64 : TSQ_I64); break;
            default: add(datum_col_type(at));
        }
    }
    for (int g = 0; g < cfg->n_group_keys; g++) {
        res->cols.emplace_back();
        res->cols.back().type = datum_col_type(cols[cfg->group_key_col[g]].type);
    }
    for (size_t gi = 0; gi < group_rows.size(); gi++) {
        size_t oc = 0;
        for (int a = 0; a < cfg->n_aggs; a++) {
            const tsq_agg_func& f = cfg->aggs[a];
            const AggCtx& c = ctxs[gi][a];
            if (f.func == TSQ_AGG_COUNT || f.func == TSQ_AGG_AVG) {
                Datum d;
                d.k = KInt64;
                d.i = c.Count;
                append_datum(res->cols[oc++], d);
            }
            if (f.func != TSQ_AGG_COUNT) append_datum(res->cols[oc++], c.Value);
        }
        for (int g = 0; g < cfg->n_group_keys; g++) append_datum(res->cols[oc++], group_rows[gi][g]);
    }
    res->rows = (int64_t)group_rows.size();
    return res;
}
}
