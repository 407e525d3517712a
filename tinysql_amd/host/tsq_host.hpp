// tsq_host.hpp — the host side ABOVE the C-ABI, in C++: what the cgo shim in package `executor` does in Go
// (INTEGRATION.md), written against the same volcano contract so that tests read like the reference's own.
//
// The reference is Go and this image has no Go toolchain, so the host side that a TinySQL maintainer would write as
//   type GPUHashJoinExec struct{ baseExecutor; ... }   (executor/join.go:31-60 shape)
// is given here as header-only C++ over include/tsq.h.  Names, argument meaning and error behaviour follow the
// reference:
//   Column / Chunk      util/chunk/column.go:28-34, chunk.go:31-46   (fixed-width columns; bit 1 = NOT NULL)
//   Executor            executor/executor.go:146-162                 (Open / Next(req) / Close; empty req = EOS)
//   MockDataSource      executor/benchmark_test.go:50-177
//   HashJoinExec        executor/join.go:110-146                     (left-child cols || right-child cols)
//   HashAggExec         executor/aggregate.go:559-588
//   SelectionExec       executor/executor.go:346-438
//   ProjectionExec      executor/projection.go:54-90, expression/evaluator.go:121-133
//   SortExec / TopNExec executor/sort.go:27-318
//   MergeJoinExec       executor/merge_join.go:31-373 (= HashJoinExec with ordered output)
//   Expression          expression/{column,constant,scalar_function}.go, lowered to tsq_expr_prog postfix
// All compute happens in libtsq (HIP); nothing here touches the oracle and there is no CPU fallback: without a
// device tsq_ctx_create fails and every constructor throws.
#ifndef TSQ_HOST_HPP
#define TSQ_HOST_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tsq.h"

namespace tsqhost {

// ---------------------------------------------------------------- errors (terror values of the reference)
struct Error : std::runtime_error {
    tsq_status code;
    Error(tsq_status c, const std::string& m) : std::runtime_error(m), code(c) {}
    // types.ErrOverflow (types/overflow.go:33-40, builtin_arithmetic_vec.go:51)
    bool IsOverflow() const { return code == TSQ_ERR_OVERFLOW_BIGINT || code == TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED || code == TSQ_ERR_OVERFLOW_DOUBLE; }
    bool IsUnsupported() const { return code == TSQ_ERR_UNSUPPORTED; }  // plan not eligible: run the Go operator
};
inline void check(tsq_status s, const void* handle) {
    if (s == TSQ_OK) return;
    const char* m = tsq_last_error(handle);
    throw Error(s, std::string(m && *m ? m : "libtsq error") + " (status " + std::to_string((int)s) + ")");
}

// ---------------------------------------------------------------- util/chunk
class Column {  // util/chunk/column.go:28-34: fixed-width cells, or offsets[n + 1] + data for var-len (TSQ_BYTES) columns
public:
    int32_t type = TSQ_I64;
    int64_t length = 0;
    std::vector<uint8_t> nullBitmap;  // bit = 1 => NOT NULL, LSB first (column.go:89-92)
    std::vector<int64_t> offsets;     // var-len only: length + 1 entries, first 0
    std::vector<uint8_t> data;

    explicit Column(int32_t tp = TSQ_I64) : type(tp) { if (isVar()) offsets.assign(1, 0); }
    bool isVar() const { return type == TSQ_BYTES; }
    int elemSize() const { return type == TSQ_F32 ? 4 : 8; }
    void Reset() { length = 0; nullBitmap.clear(); data.clear(); if (isVar()) offsets.assign(1, 0); }
    bool IsNull(int64_t i) const { return ((nullBitmap[i >> 3] >> (i & 7)) & 1) == 0; }
    void appendNullBit(bool notNull) {  // column.go:113-127
        if ((length & 7) == 0) nullBitmap.push_back(0);
        if (notNull) nullBitmap[length >> 3] |= (uint8_t)(1u << (length & 7));
    }
    void appendRaw(const void* p, bool notNull) {
        appendNullBit(notNull);
        const size_t off = data.size();
        data.resize(off + elemSize());
        if (notNull) memcpy(&data[off], p, elemSize());  // a NULL slot holds zero bytes (column.go:150-158)
        length++;
    }
    void AppendInt64(int64_t v) { appendRaw(&v, true); }
    void AppendUint64(uint64_t v) { appendRaw(&v, true); }
    void AppendFloat64(double v) { appendRaw(&v, true); }
    void AppendFloat32(float v) { appendRaw(&v, true); }
    void AppendBytes(const void* p, size_t n) {  // column.go:207-211
        appendNullBit(true);
        data.insert(data.end(), (const uint8_t*)p, (const uint8_t*)p + n);
        offsets.push_back((int64_t)data.size());
        length++;
    }
    void AppendString(const std::string& v) { AppendBytes(v.data(), v.size()); }
    void AppendNull() {
        if (isVar()) {  // a NULL var-len cell has no bytes: the offset repeats (column.go:150-158)
            appendNullBit(false);
            offsets.push_back((int64_t)data.size());
            length++;
            return;
        }
        uint64_t z = 0;
        appendRaw(&z, false);
    }
    // copies cell r of `src` (same type)
    void AppendCell(const Column& src, int64_t r) {
        if (src.IsNull(r)) AppendNull();
        else if (isVar()) AppendBytes(src.data.data() + src.offsets[r], (size_t)(src.offsets[r + 1] - src.offsets[r]));
        else appendRaw(&src.data[(size_t)r * src.elemSize()], true);
    }
    int64_t GetInt64(int64_t i) const { int64_t v; memcpy(&v, &data[i * 8], 8); return v; }
    uint64_t GetUint64(int64_t i) const { uint64_t v; memcpy(&v, &data[i * 8], 8); return v; }
    double GetFloat64(int64_t i) const { double v; memcpy(&v, &data[i * 8], 8); return v; }
    float GetFloat32(int64_t i) const { float v; memcpy(&v, &data[i * 4], 4); return v; }
    std::string GetString(int64_t i) const { return std::string((const char*)data.data() + offsets[i], (size_t)(offsets[i + 1] - offsets[i])); }
    // prepare for being filled with up to n rows by libtsq (var-len: `bytes` data bytes, from tsq_join_peek / tsq_agg_peek)
    void resizeFor(int64_t n, int64_t bytes = 0) {
        if (isVar()) {
            data.assign((size_t)bytes + 8, 0);
            offsets.assign((size_t)n + 1, 0);
        } else {
            data.assign((size_t)n * elemSize(), 0);
        }
        nullBitmap.assign((size_t)(n + 7) / 8, 0);
        length = 0;
    }
    tsq_col View(int64_t rows) {
        tsq_col c;
        memset(&c, 0, sizeof c);
        if (isVar() && data.empty()) data.assign(8, 0);  // libtsq wants a data pointer even when every cell is empty
        c.data = data.data();
        c.null_bitmap = nullBitmap.data();
        c.offsets = isVar() ? offsets.data() : nullptr;
        c.length = rows;
        c.elem_size = isVar() ? -1 : elemSize();
        c.type = type;
        return c;
    }
    void truncate(int64_t n) {
        length = n;
        if (isVar()) {
            offsets.resize((size_t)n + 1);
            data.resize((size_t)offsets[n]);
        } else {
            data.resize((size_t)n * elemSize());
        }
        nullBitmap.resize((size_t)(n + 7) / 8);
    }
};

using Schema = std::vector<int32_t>;  // column types (expression.Schema carries more; only types matter on this path)

class Chunk {  // util/chunk/chunk.go:31-46
public:
    std::vector<Column> columns;
    int requiredRows = 1024;  // tidb_max_chunk_size (tidb_vars.go:242)
    Chunk() {}
    explicit Chunk(const Schema& s, int maxChunkSize = 1024) : requiredRows(maxChunkSize) {
        for (int32_t t : s) columns.emplace_back(t);
    }
    int64_t NumRows() const { return columns.empty() ? 0 : columns[0].length; }  // chunk.go:308-316
    int NumCols() const { return (int)columns.size(); }
    void Reset() { for (auto& c : columns) c.Reset(); }                           // chunk.go:245-254
    void SwapColumns(Chunk& other) { columns.swap(other.columns); }               // chunk.go:231-235
    bool IsFull() const { return NumRows() >= requiredRows; }                     // chunk.go:165-167
    Schema schema() const { Schema s; for (auto& c : columns) s.push_back(c.type); return s; }
    std::vector<tsq_col> Views() {
        std::vector<tsq_col> v;
        for (auto& c : columns) v.push_back(c.View(c.length));
        return v;
    }
};

// ---------------------------------------------------------------- context
class Context {
public:
    tsq_ctx* h = nullptr;
    explicit Context(int device = 0) { check(tsq_ctx_create(device, &h), nullptr); }
    ~Context() { if (h) tsq_ctx_destroy(h); }
    // one slab of HBM for the buffers of every operator of this context, reserved when the process starts (tsq_ctx_reserve):
    // the first query does not pay hipMalloc (the Go process keeps its heap across queries the same way)
    void Reserve(int64_t bytes) { check(tsq_ctx_reserve(h, bytes), h); }
    struct Arena { int64_t size, used, peak; };
    Arena ArenaStats() const {
        Arena a{0, 0, 0};
        check(tsq_ctx_arena_stats(h, &a.size, &a.used, &a.peak), h);
        return a;
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
};

// ---------------------------------------------------------------- chunk.Codec / chunk.Decoder (util/chunk/codec.go:28-143, 233-353)
// colTypes are the schema's column types; getFixedLen (codec.go:169-181) is their element size
class Codec {
public:
    Context* ctx;
    Schema colTypes;
    Codec(Context* c, const Schema& t) : ctx(c), colTypes(t) {}
    std::vector<uint8_t> Encode(Chunk& chk) {  // codec.go:42-48
        std::vector<tsq_col> v = chk.Views();
        int64_t need = 0;
        check(tsq_chunk_encode(ctx->h, v.data(), (int32_t)v.size(), chk.NumRows(), nullptr, 0, 0, &need), ctx->h);
        std::vector<uint8_t> buffer((size_t)need);
        uint8_t dummy = 0;
        check(tsq_chunk_encode(ctx->h, v.data(), (int32_t)v.size(), chk.NumRows(), need ? buffer.data() : &dummy, need, 0, &need), ctx->h);
        return buffer;
    }
    // appends rows [first, first + maxRows) of the wire chunk to chk (Decoder.decodeColumn, codec.go:298-353); returns the rows
    // appended, *used = the length of the wire chunk
    int64_t decodeWindow(const uint8_t* buffer, int64_t n, int64_t first, int64_t maxRows, Chunk& chk, int64_t* used, uint32_t dataFlags = 0) {
        const int nc = (int)colTypes.size();
        std::vector<int64_t> bytes((size_t)nc);
        int64_t total = 0, take = 0;
        check(tsq_chunk_decode_peek(ctx->h, buffer, n, dataFlags, colTypes.data(), nc, first, maxRows, &total, &take, bytes.data(), used), ctx->h);
        std::vector<tsq_col> out;
        for (int c = 0; c < nc; c++) {  // room for the appended rows behind the ones the destination holds
            Column& col = chk.columns[(size_t)c];
            const int64_t rows = col.length + take;
            col.nullBitmap.resize((size_t)(rows + 7) / 8 + 1);
            if (col.isVar()) {
                col.offsets.resize((size_t)rows + 1);
                col.data.resize((size_t)(col.offsets[(size_t)col.length] + bytes[(size_t)c]) + 8);
            } else {
                col.data.resize((size_t)rows * col.elemSize());
            }
            out.push_back(col.View(col.length));
        }
        int64_t got = 0;
        check(tsq_chunk_decode(ctx->h, buffer, n, dataFlags, colTypes.data(), nc, first, maxRows, out.data(), &got, used), ctx->h);
        for (int c = 0; c < nc; c++) {
            Column& col = chk.columns[(size_t)c];
            col.length = out[(size_t)c].length;
            col.nullBitmap.resize((size_t)(col.length + 7) / 8);
            if (col.isVar()) col.data.resize((size_t)col.offsets[(size_t)col.length]);
        }
        return got;
    }
    // DecodeToChunk (codec.go:88-93): returns the remained bytes
    std::vector<uint8_t> DecodeToChunk(const std::vector<uint8_t>& buffer, Chunk& chk) {
        chk.Reset();
        int64_t used = 0;
        decodeWindow(buffer.data(), (int64_t)buffer.size(), 0, (int64_t)1 << 40, chk, &used);
        return std::vector<uint8_t>(buffer.begin() + used, buffer.end());
    }
};

class Decoder {  // codec.go:233-353; the intermediate chunk is the wire buffer itself — in HBM since Reset — decoded window by window
public:
    Codec codec;
    Chunk* intermChk;
    int64_t remainedRows = 0, next = 0, nBytes = 0;
    void* dev = nullptr;  // the response's bytes in HBM (one H2D copy per response; a host buffer would be staged again for every window)
    Decoder(Context* ctx, Chunk* chk, const Schema& colTypes) : codec(ctx, colTypes), intermChk(chk) {}
    Decoder(const Decoder&) = delete;
    Decoder& operator=(const Decoder&) = delete;
    ~Decoder() { release(); }
    void release() {
        if (dev) (void)tsq_dev_free(codec.ctx->h, dev);
        dev = nullptr;
    }
    void Reset(const std::vector<uint8_t>& d) {  // codec.go:272-275
        release();
        nBytes = (int64_t)d.size();
        next = 0;
        remainedRows = 0;
        if (nBytes == 0) return;
        check(tsq_dev_alloc(codec.ctx->h, nBytes + 64, &dev), codec.ctx->h);
        check(tsq_copy_h2d(codec.ctx->h, dev, d.data(), nBytes), codec.ctx->h);
        check(tsq_chunk_decode_peek(codec.ctx->h, (const uint8_t*)dev, nBytes, TSQ_COL_DEVICE, codec.colTypes.data(), (int32_t)codec.colTypes.size(), 0, 0, &remainedRows,
                                    nullptr, nullptr, nullptr), codec.ctx->h);
    }
    void Decode(Chunk& chk) {  // codec.go:257-269
        int64_t requiredRows = chk.requiredRows - chk.NumRows();
        requiredRows = (requiredRows + 7) >> 3 << 3;
        if (requiredRows > remainedRows) requiredRows = remainedRows;
        int64_t used = 0;
        codec.decodeWindow((const uint8_t*)dev, nBytes, next, requiredRows, chk, &used, TSQ_COL_DEVICE);
        next += requiredRows;
        remainedRows -= requiredRows;
    }
    bool IsFinished() const { return remainedRows == 0; }
    int64_t RemainedRows() const { return remainedRows; }
    void ReuseIntermChk(Chunk& chk) {  // codec.go:291-308: the rest of the rows without a second copy
        intermChk->Reset();
        int64_t used = 0;
        codec.decodeWindow((const uint8_t*)dev, nBytes, next, remainedRows, *intermChk, &used, TSQ_COL_DEVICE);
        chk.SwapColumns(*intermChk);
        next += remainedRows;
        remainedRows = 0;
    }
};

// ---------------------------------------------------------------- package executor
class Executor {  // executor/executor.go:146-152
public:
    virtual ~Executor() {}
    virtual void Open() { for (auto* c : children_) c->Open(); }      // baseExecutor.Open (executor.go:71-79)
    // Fills req with at most req->requiredRows rows; an EMPTY req means end of stream and Next stays idempotent
    // after it (executor.go:145 NOTE, server/conn.go:955-957).
    virtual void Next(Chunk* req) = 0;
    virtual void Close() { for (auto* c : children_) c->Close(); }
    const Schema& schema() const { return schema_; }
    int maxChunkSize = 1024;
protected:
    Executor(Context* ctx, Schema s, std::vector<Executor*> children) : ctx_(ctx), schema_(std::move(s)), children_(std::move(children)) {}
    Context* ctx_;
    Schema schema_;
    std::vector<Executor*> children_;
};

// recordSet + writeChunks (executor/adapter.go:93-119, server/conn.go:931-975)
inline std::vector<Chunk> Drain(Executor* e) {
    std::vector<Chunk> out;
    e->Open();
    try {
        for (;;) {
            Chunk req(e->schema(), e->maxChunkSize);
            e->Next(&req);
            if (req.NumRows() == 0) break;
            out.push_back(std::move(req));
        }
    } catch (...) {
        e->Close();
        throw;
    }
    e->Close();
    return out;
}

class MockDataSource : public Executor {  // executor/benchmark_test.go:50-144
public:
    MockDataSource(Context* ctx, Chunk table, int maxChunk = 1024) : Executor(ctx, table.schema(), {}), table_(std::move(table)) { maxChunkSize = maxChunk; }
    void Open() override { pos_ = 0; }
    void Next(Chunk* req) override {
        req->Reset();
        const int64_t n = table_.NumRows();
        const int64_t hi = std::min<int64_t>(n, pos_ + std::min(req->requiredRows, maxChunkSize));
        for (int64_t r = pos_; r < hi; r++)
            for (int c = 0; c < table_.NumCols(); c++) {
                req->columns[c].AppendCell(table_.columns[c], r);
            }
        pos_ = hi;
    }
private:
    Chunk table_;
    int64_t pos_ = 0;
};

// ---------------------------------------------------------------- package expression (tree -> postfix bytecode)
enum EvalType { ETInt, ETReal, ETString };  // types/eval_type.go:21-28
struct Expression {  // expression.Expression (expression.go:57-114), the vectorizable subset
    enum Kind { COLUMN, CONSTANT, FUNC } kind = COLUMN;
    EvalType evalType = ETInt;
    bool isUnsigned = false;
    int index = 0;             // COLUMN
    bool isNull = false;       // CONSTANT NULL
    int64_t bits = 0;          // CONSTANT: int64 or the bit pattern of a double
    std::string str;           // CONSTANT of ETString
    std::string name;          // FUNC (ast.* function names, lower case)
    std::vector<Expression> args;
};
inline Expression Col(int index, int32_t type) {  // expression.Column (column.go:56-130)
    Expression e;
    e.kind = Expression::COLUMN;
    e.index = index;
    e.evalType = type == TSQ_BYTES ? ETString : ((type == TSQ_F32 || type == TSQ_F64) ? ETReal : ETInt);
    e.isUnsigned = type == TSQ_U64;
    return e;
}
inline Expression Str(const std::string& v) { Expression e; e.kind = Expression::CONSTANT; e.evalType = ETString; e.str = v; return e; }
inline Expression Int(int64_t v) { Expression e; e.kind = Expression::CONSTANT; e.bits = v; return e; }
inline Expression Real(double v) { Expression e; e.kind = Expression::CONSTANT; e.evalType = ETReal; memcpy(&e.bits, &v, 8); return e; }
inline Expression Null(EvalType t) { Expression e; e.kind = Expression::CONSTANT; e.evalType = t; e.isNull = true; return e; }
// expression.NewFunction: result type inference of builtin_arithmetic.go:112-133,216-237,330-355,435-444,
// builtin_compare.go:60-110 for same-class arguments (TinySQL has no CAST: mixed classes are not vectorizable here)
inline Expression Func(const std::string& name, std::vector<Expression> args) {
    Expression e;
    e.kind = Expression::FUNC;
    e.name = name;
    e.args = std::move(args);
    auto& a = e.args;
    auto same = [&](size_t from) { for (size_t i = from + 1; i < a.size(); i++) if (a[i].evalType != a[from].evalType) return false; return true; };
    static const char* cmp[] = {"lt", "le", "gt", "ge", "eq", "ne"};
    bool isCmp = false;
    for (auto* c : cmp) isCmp |= name == c;
    if (name == "plus" || name == "minus" || name == "mul") {
        if (a.size() != 2 || !same(0) || a[0].evalType == ETString) throw Error(TSQ_ERR_UNSUPPORTED, "arithmetic over mixed or string arguments needs an implicit conversion");
        e.evalType = a[0].evalType;
        e.isUnsigned = e.evalType == ETInt && (a[0].isUnsigned || a[1].isUnsigned);
    } else if (name == "div") {
        if (a.size() != 2 || a[0].evalType != ETReal || a[1].evalType != ETReal) throw Error(TSQ_ERR_UNSUPPORTED, "DIV of non-real arguments");
        e.evalType = ETReal;
    } else if (isCmp) {
        if (a.size() != 2 || !same(0)) throw Error(TSQ_ERR_UNSUPPORTED, "mixed int/real comparison needs an implicit conversion");
    } else if (name == "and" || name == "or") {
        if (a.size() != 2 || a[0].evalType != ETInt || a[1].evalType != ETInt) throw Error(TSQ_ERR_UNSUPPORTED, "logic operator over non-int arguments");
    } else if (name == "not" || name == "isnull") {
        if (a.size() != 1 || (name == "not" && a[0].evalType == ETString)) throw Error(TSQ_ERR_INVALID, "arity / type");
    } else if (name == "strcmp") {  // builtin_string_vec.go:52
        if (a.size() != 2 || a[0].evalType != ETString || a[1].evalType != ETString) throw Error(TSQ_ERR_UNSUPPORTED, "STRCMP of non-string arguments");
    } else if (name == "length") {  // builtin_string_vec.go:89
        if (a.size() != 1 || a[0].evalType != ETString) throw Error(TSQ_ERR_UNSUPPORTED, "LENGTH of a non-string argument");
    } else if (name == "unaryminus") {
        if (a.size() != 1 || a[0].evalType == ETString) throw Error(TSQ_ERR_INVALID, "arity / type");
        e.evalType = a[0].evalType;
    } else if (name == "ifnull") {
        if (a.size() != 2 || !same(0)) throw Error(TSQ_ERR_UNSUPPORTED, "IFNULL of mixed types");
        e.evalType = a[0].evalType;
        e.isUnsigned = a[0].isUnsigned && a[1].isUnsigned;
    } else if (name == "if") {
        if (a.size() != 3 || a[0].evalType != ETInt || a[1].evalType != a[2].evalType) throw Error(TSQ_ERR_UNSUPPORTED, "IF needs an int condition and same-typed branches");
        e.evalType = a[1].evalType;
        e.isUnsigned = a[1].isUnsigned && a[2].isUnsigned;
    } else if (name == "in") {
        if (a.size() < 2 || a.size() > 32 || !same(0)) throw Error(TSQ_ERR_UNSUPPORTED, "IN list");
    } else {
        throw Error(TSQ_ERR_UNSUPPORTED, "function " + name + " has no GPU signature");
    }
    return e;
}
inline void emit(const Expression& e, tsq_expr_prog& p) {
    auto push = [&](int opcode, int flags, int arg, uint32_t aux) {
        if (p.n_ops >= TSQ_EXPR_MAX_OPS) throw Error(TSQ_ERR_UNSUPPORTED, "expression too large for the GPU interpreter");
        tsq_expr_op& o = p.ops[p.n_ops++];
        o.opcode = (uint8_t)opcode;
        o.flags = (uint8_t)flags;
        o.arg = (uint16_t)arg;
        o.aux = aux;
    };
    const bool real = e.evalType == ETReal, str = e.evalType == ETString;
    if (e.kind == Expression::COLUMN) { push(str ? TSQ_OP_COL_STR : (real ? TSQ_OP_COL_REAL : TSQ_OP_COL_INT), 0, e.index, 0); return; }
    if (e.kind == Expression::CONSTANT) {
        if (e.isNull) { push(str ? TSQ_OP_CONST_NULL_STR : (real ? TSQ_OP_CONST_NULL_REAL : TSQ_OP_CONST_NULL_INT), 0, 0, 0); return; }
        if (p.n_consts >= TSQ_EXPR_MAX_CONSTS) throw Error(TSQ_ERR_UNSUPPORTED, "too many constants");
        if (str) {  // the bytes go into the program's pool, the constant is (offset << 32) | length
            if ((size_t)p.n_str_bytes + e.str.size() > TSQ_EXPR_STR_POOL) throw Error(TSQ_ERR_UNSUPPORTED, "string constants exceed the program's pool");
            p.consts[p.n_consts] = ((int64_t)p.n_str_bytes << 32) | (int64_t)e.str.size();
            memcpy(p.str_pool + p.n_str_bytes, e.str.data(), e.str.size());
            p.n_str_bytes += (int32_t)e.str.size();
            push(TSQ_OP_CONST_STR, 0, p.n_consts++, 0);
            return;
        }
        p.consts[p.n_consts] = e.bits;
        push(real ? TSQ_OP_CONST_REAL : TSQ_OP_CONST_INT, 0, p.n_consts++, 0);
        return;
    }
    for (auto& x : e.args) emit(x, p);
    const auto& a = e.args;
    const bool areal = a[0].evalType == ETReal, astr = a[0].evalType == ETString;
    int flags = 0;
    if (a.size() >= 1 && a[0].isUnsigned) flags |= TSQ_F_LHS_UNSIGNED;
    if (a.size() >= 2 && a[1].isUnsigned) flags |= TSQ_F_RHS_UNSIGNED;
    const std::string& n = e.name;
    static const char* cmp[] = {"lt", "le", "gt", "ge", "eq", "ne"};
    for (int i = 0; i < 6; i++)
        if (n == cmp[i]) { push((astr ? TSQ_OP_LT_STR : (areal ? TSQ_OP_LT_REAL : TSQ_OP_LT_INT)) + i, astr ? 0 : flags, 0, 0); return; }
    if (n == "strcmp") { push(TSQ_OP_STRCMP, 0, 0, 0); return; }
    if (n == "length") { push(TSQ_OP_LENGTH, 0, 0, 0); return; }
    if (n == "plus") push(areal ? TSQ_OP_PLUS_REAL : TSQ_OP_PLUS_INT, flags, 0, 0);
    else if (n == "minus") push(areal ? TSQ_OP_MINUS_REAL : TSQ_OP_MINUS_INT, flags, 0, 0);
    else if (n == "mul") push(areal ? TSQ_OP_MUL_REAL : ((a[0].isUnsigned || a[1].isUnsigned) ? TSQ_OP_MUL_INT_UNSIGNED : TSQ_OP_MUL_INT), areal ? 0 : flags, 0, 0);
    else if (n == "div") push(TSQ_OP_DIV_REAL, 0, 0, 0);
    else if (n == "and") push(TSQ_OP_LOGIC_AND, 0, 0, 0);
    else if (n == "or") push(TSQ_OP_LOGIC_OR, 0, 0, 0);
    else if (n == "not") push(areal ? TSQ_OP_NOT_REAL : TSQ_OP_NOT_INT, 0, 0, 0);
    else if (n == "unaryminus") push(areal ? TSQ_OP_NEG_REAL : TSQ_OP_NEG_INT, flags, 0, 0);
    else if (n == "isnull") push(astr ? TSQ_OP_ISNULL_STR : (areal ? TSQ_OP_ISNULL_REAL : TSQ_OP_ISNULL_INT), 0, 0, 0);
    else if (n == "ifnull") push(astr ? TSQ_OP_IFNULL_STR : (areal ? TSQ_OP_IFNULL_REAL : TSQ_OP_IFNULL_INT), 0, 0, 0);
    else if (n == "if") push(a[1].evalType == ETString ? TSQ_OP_IF_STR : (a[1].evalType == ETReal ? TSQ_OP_IF_REAL : TSQ_OP_IF_INT), 0, 0, 0);
    else if (n == "in") {
        uint32_t aux = 0;
        for (size_t j = 1; j < a.size(); j++) if (a[j].isUnsigned) aux |= 1u << (j - 1);
        if (astr) push(TSQ_OP_IN_STR, 0, (int)a.size() - 1, 0);
        else push(areal ? TSQ_OP_IN_REAL : TSQ_OP_IN_INT, flags & TSQ_F_LHS_UNSIGNED, (int)a.size() - 1, aux);
    }
}
inline tsq_expr_prog Lower(const Expression& e) {
    tsq_expr_prog p;
    memset(&p, 0, sizeof p);
    emit(e, p);
    // a string-valued root (builtinIfStringSig / builtinIfNullStringSig.vecEvalString, a string column or constant) is declared as
    // TSQ_BYTES and evaluated by tsq_expr_eval_str
    p.result_type = e.evalType == ETString ? TSQ_BYTES : (e.evalType == ETReal ? TSQ_F64 : TSQ_I64);
    p.result_unsigned = e.isUnsigned ? 1 : 0;
    return p;
}
inline std::vector<tsq_expr_prog> LowerList(const std::vector<Expression>& es) {
    std::vector<tsq_expr_prog> v;
    for (auto& e : es) v.push_back(Lower(e));
    return v;
}

class CompiledExpr {  // a tsq_expr handle: one projection expression or one CNF filter list
public:
    tsq_expr* h = nullptr;
    int64_t divisionByZeroWarnings = 0;  // handleDivisionByZeroError (expression/errors.go:65-77)
    CompiledExpr(Context* ctx, const std::vector<Expression>& es) {
        auto progs = LowerList(es);
        check(tsq_expr_compile(ctx->h, progs.data(), (int32_t)progs.size(), &h), ctx->h);
    }
    ~CompiledExpr() { if (h) tsq_expr_destroy(h); }
    CompiledExpr(const CompiledExpr&) = delete;
    CompiledExpr& operator=(const CompiledExpr&) = delete;
};

// ---------------------------------------------------------------- SelectionExec (executor/executor.go:346-438)
class SelectionExec : public Executor {
public:
    SelectionExec(Context* ctx, Executor* child, std::vector<Expression> filters)
        : Executor(ctx, child->schema(), {child}), filters_(std::move(filters)) {}
    void Open() override {
        Executor::Open();
        expr_.reset(new CompiledExpr(ctx_, filters_));
        child_.reset(new Chunk(schema_, maxChunkSize));
        cursor_ = 0;
        selected_.clear();
    }
    void Next(Chunk* req) override {  // executor.go:393-438: copy selected rows of the child chunk until req is full
        req->Reset();
        for (;;) {
            for (; cursor_ < (int64_t)selected_.size(); cursor_++) {
                if (!selected_[cursor_]) continue;
                if (req->IsFull()) return;
                for (int c = 0; c < req->NumCols(); c++) req->columns[c].AppendCell(child_->columns[c], cursor_);
            }
            children_[0]->Next(child_.get());
            const int64_t n = child_->NumRows();
            if (n == 0) return;
            selected_.assign((size_t)n, 0);
            auto in = child_->Views();
            int64_t w = 0;
            check(tsq_filter_eval(expr_->h, in.data(), (int32_t)in.size(), n, nullptr, selected_.data(), nullptr, &w), expr_->h);
            expr_->divisionByZeroWarnings += w;
            cursor_ = 0;
        }
    }
    void Close() override { expr_.reset(); Executor::Close(); }
private:
    std::vector<Expression> filters_;
    std::unique_ptr<CompiledExpr> expr_;
    std::unique_ptr<Chunk> child_;
    std::vector<uint8_t> selected_;
    int64_t cursor_ = 0;
};

// ---------------------------------------------------------------- ProjectionExec (executor/projection.go:54-90)
class ProjectionExec : public Executor {
public:
    ProjectionExec(Context* ctx, Executor* child, std::vector<Expression> exprs) : Executor(ctx, types(exprs), {child}), exprs_(std::move(exprs)) {}
    void Open() override {
        Executor::Open();
        compiled_.clear();
        for (auto& e : exprs_) compiled_.emplace_back(new CompiledExpr(ctx_, {e}));
        child_.reset(new Chunk(children_[0]->schema(), maxChunkSize));
    }
    void Next(Chunk* req) override {  // EvaluatorSuite.Run (evaluator.go:121-133): one VecEval per output column
        req->Reset();
        children_[0]->Next(child_.get());
        const int64_t n = child_->NumRows();
        if (n == 0) return;
        auto in = child_->Views();
        for (size_t i = 0; i < compiled_.size(); i++) {
            Column& dst = req->columns[i];
            if (dst.type == TSQ_BYTES) {  // VecEvalString (expression.go:329-341): ask for the bytes of the result, then fill the column
                int64_t w = 0, need = 0;
                dst.resizeFor(n, 0);
                tsq_col ask = dst.View(n);
                tsq_status s0 = tsq_expr_eval_str(compiled_[i]->h, in.data(), (int32_t)in.size(), n, nullptr, &ask, 0, &need, &w);
                if (s0 != TSQ_OK && !(s0 == TSQ_ERR_INVALID && need > 0)) check(s0, compiled_[i]->h);
                if (need > 0) {
                    dst.resizeFor(n, need);
                    tsq_col out = dst.View(n);
                    check(tsq_expr_eval_str(compiled_[i]->h, in.data(), (int32_t)in.size(), n, nullptr, &out, need, &need, &w), compiled_[i]->h);
                }
                compiled_[i]->divisionByZeroWarnings += w;
                dst.length = n;
                dst.data.resize((size_t)need);
                continue;
            }
            dst.resizeFor(n);
            tsq_col out = dst.View(n);
            out.type = dst.type == TSQ_F64 ? TSQ_F64 : TSQ_I64;
            int64_t w = 0;
            check(tsq_expr_eval(compiled_[i]->h, in.data(), (int32_t)in.size(), n, nullptr, &out, &w), compiled_[i]->h);
            compiled_[i]->divisionByZeroWarnings += w;
            dst.length = n;
        }
    }
    void Close() override { compiled_.clear(); Executor::Close(); }
    int64_t DivisionByZeroWarnings() const { int64_t w = 0; for (auto& c : compiled_) w += c->divisionByZeroWarnings; return w; }
private:
    static Schema types(const std::vector<Expression>& es) {
        Schema s;
        for (auto& e : es) s.push_back(e.evalType == ETString ? TSQ_BYTES : (e.evalType == ETReal ? TSQ_F64 : (e.isUnsigned ? TSQ_U64 : TSQ_I64)));
        return s;
    }
    std::vector<Expression> exprs_;
    std::vector<std::unique_ptr<CompiledExpr>> compiled_;
    std::unique_ptr<Chunk> child_;
};

// ---------------------------------------------------------------- HashJoinExec (executor/join.go:31-146)
enum JoinType { InnerJoin = TSQ_JOIN_INNER, LeftOuterJoin = TSQ_JOIN_LEFT_OUTER, RightOuterJoin = TSQ_JOIN_RIGHT_OUTER };
class HashJoinExec : public Executor {
public:
    // children: left, right.  innerChildIdx picks the build side (planner/core/physical_plans.go:201-224).
    HashJoinExec(Context* ctx, Executor* left, Executor* right, std::vector<int> leftKeys, std::vector<int> rightKeys, JoinType jt, int innerChildIdx,
                 std::vector<Expression> otherConditions = {}, std::vector<Expression> outerFilter = {})
        : Executor(ctx, concat(left->schema(), right->schema()), {left, right}), other_(LowerList(otherConditions)), filter_(LowerList(outerFilter)) {
        memset(&cfg_, 0, sizeof cfg_);
        buildIsRight_ = innerChildIdx == 1;
        build_ = buildIsRight_ ? right : left;
        probe_ = buildIsRight_ ? left : right;
        const auto& bk = buildIsRight_ ? rightKeys : leftKeys;
        const auto& pk = buildIsRight_ ? leftKeys : rightKeys;
        if (bk.size() != pk.size() || bk.empty() || bk.size() > TSQ_MAX_KEYS) throw Error(TSQ_ERR_UNSUPPORTED, "1..4 join key columns supported");
        cfg_.join_type = jt;
        cfg_.build_is_right = buildIsRight_ ? 1 : 0;
        cfg_.n_keys = (int32_t)bk.size();
        for (size_t i = 0; i < bk.size(); i++) { cfg_.build_key_idx[i] = bk[i]; cfg_.probe_key_idx[i] = pk[i]; }
        cfg_.n_build_cols = (int32_t)build_->schema().size();
        cfg_.n_probe_cols = (int32_t)probe_->schema().size();
        for (int i = 0; i < cfg_.n_build_cols; i++) cfg_.build_types[i] = build_->schema()[i];
        for (int i = 0; i < cfg_.n_probe_cols; i++) cfg_.probe_types[i] = probe_->schema()[i];
        cfg_.max_chunk_size = 1024;
        cfg_.concurrency = 5;  // tidb_hash_join_concurrency default (tidb_vars.go:249); unused on the GPU
    }
    ~HashJoinExec() override { destroy(); }
    void Open() override {
        Executor::Open();
        cfg_.other_conds = other_.empty() ? nullptr : other_.data();
        cfg_.n_other_conds = (int32_t)other_.size();
        cfg_.outer_filters = filter_.empty() ? nullptr : filter_.data();
        cfg_.n_outer_filters = (int32_t)filter_.size();
        check(tsq_join_create(ctx_->h, &cfg_, &h_), ctx_->h);
        if (ordered_) check(tsq_join_set_ordered(h_, 1), h_);
        if (!used_.empty()) check(tsq_join_set_used_columns(h_, used_.data(), (int32_t)used_.size()), h_);
        warnings_ = 0;
        prepared_ = false;
        probeDone_ = false;
    }
    void Next(Chunk* req) override {  // join.go:125-146
        req->Reset();
        if (!prepared_) {  // fetchAndBuildHashTable (join.go:148-158)
            Chunk chk(build_->schema(), maxChunkSize);
            for (;;) {
                build_->Next(&chk);
                if (chk.NumRows() == 0) break;
                auto v = chk.Views();
                check(tsq_join_build_push(h_, v.data(), (int32_t)v.size(), chk.NumRows()), h_);
            }
            check(tsq_join_build_finish(h_), h_);
            prepared_ = true;
        }
        const int64_t cap = req->requiredRows;
        Chunk probeChk(probe_->schema(), maxChunkSize);
        bool anyVar = false;
        for (auto& c : req->columns) anyVar |= c.isVar();
        for (;;) {
            std::vector<int64_t> bytes(req->columns.size(), 0);
            if (anyVar) {  // data bytes of the var-len columns of the next pull
                int64_t pn = 0;
                check(tsq_join_peek(h_, cap, &pn, bytes.data(), (int32_t)bytes.size()), h_);
            }
            for (size_t c = 0; c < req->columns.size(); c++) req->columns[c].resizeFor(cap, bytes[c]);
            std::vector<tsq_col> out;
            for (auto& c : req->columns) out.push_back(c.View(cap));
            int64_t n = 0;
            int32_t eos = 0;
            check(tsq_join_pull(h_, out.data(), (int32_t)out.size(), cap, &n, &eos), h_);
            if (n > 0) { for (auto& c : req->columns) c.truncate(n); return; }
            if (eos) { for (auto& c : req->columns) c.truncate(0); return; }
            if (probeDone_) continue;
            probe_->Next(&probeChk);  // fetchOuterSideChunks (join.go:160-231)
            if (probeChk.NumRows() == 0) {
                check(tsq_join_probe_finish(h_), h_);
                probeDone_ = true;
                continue;
            }
            auto v = probeChk.Views();
            check(tsq_join_probe_push(h_, v.data(), (int32_t)v.size(), probeChk.NumRows(), nullptr), h_);
        }
    }
    // Close may race with Next on another thread (join_test.go:172-182, TestJoinLeak): cancel first.
    void Close() override {
        if (h_) {  // the division-by-zero warnings of OtherConditions / outer filters (expression/errors.go:65-77: handleDivisionByZeroError
                   // appends one warning per offending row to the statement context; the shim does that with this count)
            tsq_stats st;
            if (tsq_join_stats(h_, &st) == TSQ_OK) warnings_ = st.div_by_zero_warnings;
        }
        destroy();
        Executor::Close();
    }
    void Cancel() { if (h_) tsq_join_cancel(h_); }
    // Inline projection (planner/core/rule_column_pruning.go): used[c] == 0: the parent never reads output column c (left child's
    // columns, then the right child's).  The operator may leave such a column unmaterialised; its cells in `req` are unspecified.
    void SetUsedColumns(std::vector<uint8_t> used) {
        if (!used.empty() && used.size() != schema().size()) throw Error(TSQ_ERR_INVALID, "SetUsedColumns: one flag per output column");
        used_ = std::move(used);
    }
    int64_t DivisionByZeroWarnings() const { return warnings_; }  // after Close()
protected:
    bool ordered_ = false;  // MergeJoinExec: outer rows in order, each with its inner matches in order
private:
    static Schema concat(Schema a, const Schema& b) { a.insert(a.end(), b.begin(), b.end()); return a; }
    void destroy() {
        if (h_) { tsq_join_cancel(h_); tsq_join_destroy(h_); h_ = nullptr; }
    }
    tsq_join_cfg cfg_;
    std::vector<tsq_expr_prog> other_, filter_;
    Executor *build_, *probe_;
    bool buildIsRight_ = true, prepared_ = false, probeDone_ = false;
    std::vector<uint8_t> used_;
    int64_t warnings_ = 0;
    tsq_join* h_ = nullptr;
};

// ---------------------------------------------------------------- MergeJoinExec (executor/merge_join.go:31-373)
// Two sorted children walked with two cursors in the reference; its output — outer rows in order, each with its inner
// group in order, NULL-key inner rows skipped, unmatched outer rows padded for outer joins — is the hash join with ordered
// output (tsq_join_set_ordered): outer child = probe side, inner child = build side.
class MergeJoinExec : public HashJoinExec {
public:
    MergeJoinExec(Context* ctx, Executor* left, Executor* right, std::vector<int> leftKeys, std::vector<int> rightKeys, JoinType jt, int innerChildIdx,
                  std::vector<Expression> otherConditions = {}, std::vector<Expression> outerFilter = {})
        : HashJoinExec(ctx, left, right, std::move(leftKeys), std::move(rightKeys), jt, innerChildIdx, std::move(otherConditions), std::move(outerFilter)) {
        ordered_ = true;
    }
};

// ---------------------------------------------------------------- HashAggExec (executor/aggregate.go:134-588)
struct AggFuncDesc {  // expression/aggregation/descriptor.go:56-91
    int32_t func;     // TSQ_AGG_*
    int32_t argCol;   // -1: constant argument (COUNT(*) arrives as count(1), parser.y:3258-3262)
    int32_t argType;  // TSQ_I64 ...
    int32_t mode = TSQ_MODE_COMPLETE;
    int32_t argCol2 = -1;  // AVG in Final/Partial2 mode: (count column, sum column)
    Schema outTypes() const {
        const bool partial = mode == TSQ_MODE_PARTIAL1 || mode == TSQ_MODE_PARTIAL2;
        const bool real = argType == TSQ_F32 || argType == TSQ_F64;
        switch (func) {
            case TSQ_AGG_COUNT: return {TSQ_I64};
            case TSQ_AGG_SUM: return {real ? TSQ_F64 : TSQ_I64};  // no DECIMAL in TinySQL: base_func.go:119-131
            case TSQ_AGG_AVG: return partial ? Schema{TSQ_I64, real ? TSQ_F64 : TSQ_I64} : Schema{real ? TSQ_F64 : TSQ_I64};
            default: return {argType};
        }
    }
};
class HashAggExec : public Executor {
public:
    HashAggExec(Context* ctx, Executor* child, std::vector<int> groupByCols, std::vector<AggFuncDesc> aggFuncs)
        : Executor(ctx, types(aggFuncs), {child}) {
        memset(&cfg_, 0, sizeof cfg_);
        const Schema& in = child->schema();
        if (groupByCols.size() > TSQ_MAX_GROUP_KEYS || aggFuncs.size() > TSQ_MAX_AGGS) throw Error(TSQ_ERR_UNSUPPORTED, "too many group keys / aggregates");
        cfg_.n_group_keys = (int32_t)groupByCols.size();
        for (size_t i = 0; i < groupByCols.size(); i++) { cfg_.group_key_col[i] = groupByCols[i]; cfg_.group_key_type[i] = in[groupByCols[i]]; }
        cfg_.n_aggs = (int32_t)aggFuncs.size();
        for (size_t i = 0; i < aggFuncs.size(); i++) {
            cfg_.aggs[i].func = aggFuncs[i].func;
            cfg_.aggs[i].mode = aggFuncs[i].mode;
            cfg_.aggs[i].arg_col = aggFuncs[i].argCol;
            cfg_.aggs[i].arg_col2 = aggFuncs[i].argCol2;
            cfg_.aggs[i].arg_type = aggFuncs[i].argType;
        }
        cfg_.n_input_cols = (int32_t)in.size();
        for (size_t i = 0; i < in.size(); i++) cfg_.input_types[i] = in[i];
        cfg_.max_chunk_size = 1024;
        // empty input without GROUP BY yields one row of defaults unless every function is FIRST_ROW
        // (executor/builder.go:517-539, aggregate.go:572-574)
        defaultRow_ = groupByCols.empty();
        bool allFirstRow = true;
        for (auto& f : aggFuncs) allFirstRow &= f.func == TSQ_AGG_FIRSTROW;
        if (allFirstRow) defaultRow_ = false;
        funcs_ = std::move(aggFuncs);
    }
    ~HashAggExec() override { destroy(); }
    void Open() override {
        Executor::Open();
        check(tsq_agg_create(ctx_->h, &cfg_, &h_), ctx_->h);
        prepared_ = false;
        sawInput_ = false;
        done_ = false;
    }
    void Next(Chunk* req) override {  // parallelExec (aggregate.go:559-588)
        req->Reset();
        if (done_) return;
        if (!prepared_) {
            Chunk chk(children_[0]->schema(), maxChunkSize);
            for (;;) {
                children_[0]->Next(&chk);
                if (chk.NumRows() == 0) break;
                sawInput_ = true;
                auto v = chk.Views();
                check(tsq_agg_push(h_, v.data(), (int32_t)v.size(), chk.NumRows()), h_);
            }
            check(tsq_agg_finish(h_), h_);
            prepared_ = true;
        }
        if (!sawInput_) {
            done_ = true;
            if (!defaultRow_) return;
            // COUNT -> 0, everything else NULL (expression/aggregation/base_func.go:176-185)
            size_t c = 0;
            for (auto& f : funcs_)
                for (int32_t t : f.outTypes()) {
                    (void)t;
                    if (f.func == TSQ_AGG_COUNT) req->columns[c].AppendInt64(0);
                    else req->columns[c].AppendNull();
                    c++;
                }
            return;
        }
        const int64_t cap = req->requiredRows;
        std::vector<int64_t> bytes(req->columns.size(), 0);
        bool anyVar = false;
        for (auto& c : req->columns) anyVar |= c.isVar();
        if (anyVar) {  // firstRow4String / maxMin4String outputs: their data bytes for this pull
            int64_t pn = 0;
            check(tsq_agg_peek(h_, cap, &pn, bytes.data(), (int32_t)bytes.size()), h_);
        }
        for (size_t c = 0; c < req->columns.size(); c++) req->columns[c].resizeFor(cap, bytes[c]);
        std::vector<tsq_col> out;
        for (auto& c : req->columns) out.push_back(c.View(cap));
        int64_t n = 0;
        int32_t eos = 0;
        check(tsq_agg_pull(h_, out.data(), (int32_t)out.size(), cap, &n, &eos), h_);
        for (auto& c : req->columns) c.truncate(n);
        if (n == 0) done_ = true;
    }
    void Close() override { destroy(); Executor::Close(); }
protected:
    tsq_agg* h_ = nullptr;
private:
    static Schema types(const std::vector<AggFuncDesc>& fs) {
        Schema s;
        for (auto& f : fs) for (int32_t t : f.outTypes()) s.push_back(t);
        return s;
    }
    void destroy() {
        if (h_) { tsq_agg_cancel(h_); tsq_agg_destroy(h_); h_ = nullptr; }
    }
    tsq_agg_cfg cfg_;
    std::vector<AggFuncDesc> funcs_;
    bool defaultRow_ = false, prepared_ = false, sawInput_ = false, done_ = false;
};

// ---------------------------------------------------------------- StreamAggExec (the north star names it; the reference has only the
// plan name, planner/core/cbo_test.go:200-212): the child delivers rows ORDERED by the group-by columns, the groups come out in that
// order, FIRST_ROW is the first row of its group.  Same aggregate functions, modes and default row as HashAggExec — the same
// tsq_agg handle with tsq_agg_set_stream (csrc/tsq_streamagg.h).
class StreamAggExec : public HashAggExec {
public:
    using HashAggExec::HashAggExec;
    void Open() override {
        HashAggExec::Open();
        check(tsq_agg_set_stream(h_, 1), h_);
    }
};

// ---------------------------------------------------------------- SortExec / TopNExec (executor/sort.go:27-318)
struct ByItem {  // plannercore.ByItems: a bare column and its direction (sort.go:107-113)
    int col;
    bool desc;
};
class SortExec : public Executor {
public:
    SortExec(Context* ctx, Executor* child, std::vector<ByItem> byItems) : SortExec(ctx, child, std::move(byItems), 0, -1) {}
    ~SortExec() override { destroy(); }
    void Open() override {
        Executor::Open();
        check(tsq_sort_create(ctx_->h, &cfg_, &h_), ctx_->h);
        fetched_ = false;
    }
    void Next(Chunk* req) override {  // sort.go:58-78
        req->Reset();
        if (!fetched_) {  // fetchRowChunks (sort.go:80-97), then the sort
            Chunk chk(children_[0]->schema(), maxChunkSize);
            for (;;) {
                children_[0]->Next(&chk);
                if (chk.NumRows() == 0) break;
                auto v = chk.Views();
                check(tsq_sort_push(h_, v.data(), (int32_t)v.size(), chk.NumRows()), h_);
            }
            check(tsq_sort_finish(h_), h_);
            fetched_ = true;
        }
        const int64_t cap = req->requiredRows;
        std::vector<int64_t> bytes(req->columns.size(), 0);
        bool anyVar = false;
        for (auto& c : req->columns) anyVar |= c.isVar();
        if (anyVar) {  // string payload columns: their data bytes for this pull
            int64_t pn = 0;
            check(tsq_sort_peek(h_, cap, &pn, bytes.data(), (int32_t)bytes.size()), h_);
        }
        for (size_t c = 0; c < req->columns.size(); c++) req->columns[c].resizeFor(cap, bytes[c]);
        std::vector<tsq_col> out;
        for (auto& c : req->columns) out.push_back(c.View(cap));
        int64_t n = 0;
        int32_t eos = 0;
        check(tsq_sort_pull(h_, out.data(), (int32_t)out.size(), cap, &n, &eos), h_);
        for (auto& c : req->columns) c.truncate(n);
    }
    void Close() override { destroy(); Executor::Close(); }
protected:
    SortExec(Context* ctx, Executor* child, std::vector<ByItem> byItems, int64_t offset, int64_t count) : Executor(ctx, child->schema(), {child}) {
        memset(&cfg_, 0, sizeof cfg_);
        if (byItems.empty() || byItems.size() > TSQ_MAX_KEYS) throw Error(TSQ_ERR_UNSUPPORTED, "1..4 ORDER BY items supported");
        cfg_.n_cols = (int32_t)schema_.size();
        for (size_t i = 0; i < schema_.size(); i++) cfg_.col_types[i] = schema_[i];
        cfg_.n_keys = (int32_t)byItems.size();
        for (size_t i = 0; i < byItems.size(); i++) { cfg_.key_col[i] = byItems[i].col; cfg_.key_desc[i] = byItems[i].desc ? 1 : 0; }
        cfg_.limit_offset = offset;
        cfg_.limit_count = count;
        cfg_.max_chunk_size = 1024;
    }
private:
    void destroy() {
        if (h_) { tsq_sort_cancel(h_); tsq_sort_destroy(h_); h_ = nullptr; }
    }
    tsq_sort_cfg cfg_;
    tsq_sort* h_ = nullptr;
    bool fetched_ = false;
};
class TopNExec : public SortExec {  // rows [Offset, Offset + Count) of the order (sort.go:213-238)
public:
    TopNExec(Context* ctx, Executor* child, std::vector<ByItem> byItems, uint64_t offset, uint64_t count)
        : SortExec(ctx, child, std::move(byItems), (int64_t)offset, (int64_t)std::min<uint64_t>(count, (uint64_t)INT64_MAX)) {}
};


// ---------------------------------------------------------------- the storage side: package tablecodec + mocktikv's executors
// tablecodec.EncodeRowKeyWithHandle / DecodeRowKey for a batch (tablecodec/tablecodec.go:65-70, 235-242) through libtsq
inline std::vector<uint8_t> EncodeRowKeysWithHandles(Context* ctx, int64_t tableID, const std::vector<int64_t>& handles) {
    std::vector<uint8_t> keys(handles.size() * 19 + 1);
    check(tsq_rowkeys_encode(ctx->h, tableID, handles.data(), (int64_t)handles.size(), 0, keys.data()), ctx->h);
    keys.resize(handles.size() * 19);
    return keys;
}
inline std::vector<int64_t> DecodeRowKeys(Context* ctx, const std::vector<uint8_t>& keys) {
    if (keys.size() % 19) throw Error(TSQ_ERR_INVALID, "invalid key");
    std::vector<int64_t> handles(keys.size() / 19 + 1);
    int64_t n = 0;
    check(tsq_rowkeys_decode(ctx->h, keys.data(), (int64_t)keys.size(), nullptr, (int64_t)(keys.size() / 19), 0, handles.data(), nullptr, &n), ctx->h);
    handles.resize((size_t)n);
    return handles;
}

namespace mocktikv {
// rowcodec.ColInfo (util/rowcodec/decoder.go:45-55) as far as this path reads it
struct ColInfo {
    int64_t ID;
    int32_t type;     // TSQ_I64 / TSQ_U64 / TSQ_F32 / TSQ_F64 / TSQ_BYTES (the mysql type + UnsignedFlag folded)
    bool IsPKHandle;
};
// the KV pairs of the scanned ranges, in scan order: record keys back to back (19 bytes each) and the stored rows (rowcodec v2)
struct Pairs {
    std::vector<uint8_t> keys, values;
    std::vector<int64_t> valueOffsets;  // n + 1
};

// tableScanExec (store/mockstore/mocktikv/executor.go:48-196): DecodeRowKey + the row decode of every pair, a chunk at a time
class tableScanExec : public Executor {
public:
    tableScanExec(Context* ctx, std::vector<ColInfo> columns, const Pairs* pairs) : Executor(ctx, types(columns), {}), cols_(std::move(columns)), p_(pairs) {
        n_ = (int64_t)p_->valueOffsets.size() - 1;
        if ((int64_t)p_->keys.size() != 19 * n_) throw Error(TSQ_ERR_INVALID, "invalid key");
        for (auto& c : cols_) {
            tsq_rowcodec_col rc;
            memset(&rc, 0, sizeof rc);
            rc.col_id = c.ID;
            rc.type = c.type;
            rc.flags = c.IsPKHandle ? TSQ_RC_HANDLE : 0;
            rc_.push_back(rc);
        }
    }
    void Open() override {
        pos_ = 0;
        handles_.assign((size_t)n_ + 1, 0);
        int64_t got = 0;
        if (n_) check(tsq_rowkeys_decode(ctx_->h, p_->keys.data(), (int64_t)p_->keys.size(), nullptr, n_, 0, handles_.data(), nullptr, &got), ctx_->h);
    }
    void Next(Chunk* req) override {
        req->Reset();
        if (pos_ >= n_) return;
        const int64_t lo = pos_, hi = std::min<int64_t>(n_, pos_ + std::min(req->requiredRows, maxChunkSize));
        const int64_t b0 = p_->valueOffsets[lo], b1 = p_->valueOffsets[hi];
        std::vector<int64_t> offs((size_t)(hi - lo) + 1);  // the batch's rows, rebased onto its first byte
        for (int64_t r = lo; r <= hi; r++) offs[(size_t)(r - lo)] = p_->valueOffsets[r] - b0;
        for (auto& c : req->columns) c.resizeFor(hi - lo, b1 - b0);
        std::vector<tsq_col> out;
        for (auto& c : req->columns) out.push_back(c.View(hi - lo));
        int64_t n = 0;
        check(tsq_rowcodec_decode(ctx_->h, p_->values.data() + b0, b1 - b0, offs.data(), handles_.data() + lo, hi - lo, 0, (int32_t)rc_.size(), rc_.data(), out.data(), &n),
              ctx_->h);
        for (auto& c : req->columns) c.truncate(n);
        pos_ = hi;
        count_ += n;
    }
    int64_t Count() const { return count_; }  // Counts() of the one range (executor.go:76-85)
private:
    static Schema types(const std::vector<ColInfo>& cs) { Schema s; for (auto& c : cs) s.push_back(c.type); return s; }
    std::vector<ColInfo> cols_;
    const Pairs* p_;
    std::vector<tsq_rowcodec_col> rc_;
    std::vector<int64_t> handles_;
    int64_t n_ = 0, pos_ = 0, count_ = 0;
};

// the KV pairs of an index range, in scan order: index keys back to back + offsets, the pairs' values (the handle of a unique index)
struct IndexPairs {
    std::vector<uint8_t> keys, values;
    std::vector<int64_t> keyOffsets, valueOffsets;  // n + 1 each (valueOffsets may stay empty when every key carries its handle)
};
enum PrimaryKeyStatus { PrimaryKeyNotExists = 0, PrimaryKeyIsSigned = 1, PrimaryKeyIsUnsigned = 2 };  // tablecodec.go:394-403

// indexScanExec (store/mockstore/mocktikv/executor.go:191-320): tablecodec.DecodeIndexKV (tablecodec.go:376-434) of every pair, a chunk
// at a time.  types: the index columns (+ the handle column when pkStatus != PrimaryKeyNotExists); colsLen = the index columns
class indexScanExec : public Executor {
public:
    indexScanExec(Context* ctx, Schema types, int colsLen, PrimaryKeyStatus pkStatus, const IndexPairs* pairs)
        : Executor(ctx, std::move(types), {}), colsLen_(colsLen), pk_(pkStatus), p_(pairs) {
        n_ = (int64_t)p_->keyOffsets.size() - 1;
    }
    void Open() override { pos_ = count_ = 0; }
    void Next(Chunk* req) override {
        req->Reset();
        if (pos_ >= n_) return;
        const int64_t lo = pos_, hi = std::min<int64_t>(n_, pos_ + std::min(req->requiredRows, maxChunkSize));
        const int64_t keyBytes = p_->keyOffsets[(size_t)hi] - p_->keyOffsets[(size_t)lo];
        for (auto& c : req->columns) c.resizeFor(hi - lo, keyBytes);  // a string cell is a piece of its key
        std::vector<tsq_col> out;
        for (auto& c : req->columns) out.push_back(c.View(hi - lo));
        const Schema tp = req->schema();
        const bool vals = !p_->valueOffsets.empty();
        int64_t n = 0;
        check(tsq_indexkeys_decode(ctx_->h, p_->keys.data(), (int64_t)p_->keys.size(), p_->keyOffsets.data() + lo, hi - lo, vals ? p_->values.data() : nullptr,
                                   vals ? (int64_t)p_->values.size() : 0, vals ? p_->valueOffsets.data() + lo : nullptr, 0, colsLen_, tp.data(), (int32_t)pk_, out.data(), &n),
              ctx_->h);
        for (auto& c : req->columns) c.truncate(n);
        pos_ = hi;
        count_ += n;
    }
    int64_t Count() const { return count_; }
private:
    int colsLen_;
    PrimaryKeyStatus pk_;
    const IndexPairs* p_;
    int64_t n_ = 0, pos_ = 0, count_ = 0;
};

// selectionExec (executor.go:322-390) = SelectionExec; topNExec (:392-470, topn.go) = TopNExec with offset 0
using selectionExec = SelectionExec;
class topNExec : public TopNExec {
public:
    topNExec(Context* ctx, Executor* src, std::vector<ByItem> orderBy, uint64_t limit) : TopNExec(ctx, src, std::move(orderBy), 0, limit) {}
};
// hashAggExec (aggregate.go:30-182): per function its partial results (AVG: count, sum — avg.go:78-81), then the group-by values
class hashAggExec : public HashAggExec {
public:
    hashAggExec(Context* ctx, Executor* src, const std::vector<std::pair<int32_t, int>>& aggFuncs, const std::vector<int>& groupByCols)
        : HashAggExec(ctx, src, groupByCols, descs(src->schema(), aggFuncs, groupByCols)) {}
private:
    static std::vector<AggFuncDesc> descs(const Schema& in, const std::vector<std::pair<int32_t, int>>& fs, const std::vector<int>& gby) {
        std::vector<AggFuncDesc> d;
        for (auto& f : fs) {
            AggFuncDesc a;
            a.func = f.first;
            a.argCol = f.second;
            a.argType = f.second >= 0 ? in[(size_t)f.second] : TSQ_I64;
            a.mode = f.first == TSQ_AGG_AVG ? TSQ_MODE_PARTIAL1 : TSQ_MODE_COMPLETE;
            d.push_back(a);
        }
        for (int g : gby) {
            AggFuncDesc a;
            a.func = TSQ_AGG_FIRSTROW;
            a.argCol = g;
            a.argType = in[(size_t)g];
            d.push_back(a);
        }
        return d;
    }
};
// limitExec (executor.go:472-507)
class limitExec : public Executor {
public:
    limitExec(Context* ctx, Executor* src, uint64_t limit) : Executor(ctx, src->schema(), {src}), limit_(limit) {}
    void Open() override { Executor::Open(); cursor_ = 0; }
    void Next(Chunk* req) override {
        req->Reset();
        if (cursor_ >= limit_) return;
        Chunk chk(schema_, maxChunkSize);
        chk.requiredRows = (int)std::min<uint64_t>((uint64_t)req->requiredRows, limit_ - cursor_);
        children_[0]->Next(&chk);
        const int64_t n = std::min<int64_t>(chk.NumRows(), (int64_t)(limit_ - cursor_));
        for (int64_t r = 0; r < n; r++)
            for (int c = 0; c < chk.NumCols(); c++) req->columns[c].AppendCell(chk.columns[c], r);
        cursor_ += (uint64_t)n;
    }
private:
    uint64_t limit_, cursor_ = 0;
};

// handleCopDAGRequest's tail (cop_handler_dag.go:58-83, 414-425, 510-519): run the executor to its end, encode the requested columns of
// every row (codec.EncodeValue on the GPU), cut the rows into RowsData pieces of 64
inline std::vector<std::string> fillUpData4SelectResponse(Context* ctx, Executor* e, const std::vector<int>& outputOffsets) {
    std::vector<std::string> chunks;
    int64_t rowCnt = 0;
    for (Chunk& chk : Drain(e)) {
        const int64_t n = chk.NumRows();
        std::vector<tsq_col> cols;
        for (int o : outputOffsets) cols.push_back(chk.columns[(size_t)o].View(n));
        int64_t need = 0;
        const tsq_status st = tsq_rows_encode(ctx->h, cols.data(), (int32_t)cols.size(), nullptr, n, nullptr, 0, 0, nullptr, &need);  // the size first
        if (st != TSQ_ERR_INVALID && st != TSQ_OK) check(st, ctx->h);
        std::vector<uint8_t> raw((size_t)need + 8);
        std::vector<int64_t> offs((size_t)n + 1);
        check(tsq_rows_encode(ctx->h, cols.data(), (int32_t)cols.size(), nullptr, n, raw.data(), need, 0, offs.data(), &need), ctx->h);
        for (int64_t lo = 0; lo < n;) {
            const int64_t room = 64 - rowCnt % 64, hi = std::min<int64_t>(n, lo + room);
            if (rowCnt % 64 == 0) chunks.emplace_back();
            chunks.back().append((const char*)raw.data() + offs[(size_t)lo], (size_t)(offs[(size_t)hi] - offs[(size_t)lo]));
            rowCnt += hi - lo;
            lo = hi;
        }
    }
    return chunks;
}
}  // namespace mocktikv

}  // namespace tsqhost
#endif
