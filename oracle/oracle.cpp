// oracle.cpp — CPU restatement of TinySQL's hash join / hash aggregate / vectorized expression
// algorithms.  TEST INFRASTRUCTURE ONLY (see oracle.h): the checker, never the product.
//
// Every block cites the reference lines it follows (paths relative to the TinySQL tree).
// The restatement keeps the reference's *structure* where it determines results:
//   - FNV-1-64 over [flag][raw bytes] per key column      util/codec/codec.go:249-338
//   - hash -> chain head map + 16-byte chained entries    executor/hash_table.go:181-276
//   - key equality by (flag, bytes)                       util/codec/codec.go:363-382
//   - joiners: lhs||rhs, NULL padding, other conditions   executor/joiner.go:145-410
//   - group key = concatenated encoded datums             util/codec/codec.go:713-746
//   - partial -> shuffle -> final aggregation             executor/aggregate.go:307-457
//   - node-at-a-time vectorized expressions               expression/builtin_*_vec.go
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;

// ------------------------------------------------------------------ column access
// util/chunk/column.go:89-92 IsNull: nullBitmap bit==0 => NULL (LSB first)
inline bool col_is_null(const tsq_col& c, int64_t i) {
    if (!c.null_bitmap) return false;
    return ((c.null_bitmap[i >> 3] >> (i & 7)) & 1) == 0;
}
inline int64_t col_i64(const tsq_col& c, int64_t i) { return ((const int64_t*)c.data)[i]; }
inline double col_f64(const tsq_col& c, int64_t i) {
    if (c.type == TSQ_F32) return (double)((const float*)c.data)[i]; // column.go:81-104 widening
    return ((const double*)c.data)[i];
}
inline uint64_t col_raw64(const tsq_col& c, int64_t i) {
    if (c.elem_size == 4) return ((const uint32_t*)c.data)[i];
    return ((const uint64_t*)c.data)[i];
}

// ------------------------------------------------------------------ FNV-1 64 (Go hash/fnv New64)
// hash/fnv: offset64 = 14695981039346656037, prime64 = 1099511628211; New64 is FNV-1:
// hash *= prime; hash ^= byte   (executor/hash_table.go:18,64 uses fnv.New64()).
constexpr uint64_t FNV_OFFSET = 14695981039346656037ULL;
constexpr uint64_t FNV_PRIME = 1099511628211ULL;
inline uint64_t fnv1_write(uint64_t h, const uint8_t* p, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        h *= FNV_PRIME;
        h ^= p[i];
    }
    return h;
}

// util/codec/codec.go:36-49 flags
constexpr uint8_t NilFlag = 0, compactBytesFlag = 2, floatFlag = 5, varintFlag = 8, uvarintFlag = 9;

// util/codec/codec.go:212-240 encodeHashChunkRowIdx: (flag, 8 raw bytes) of one key cell
inline void encode_hash_cell(const tsq_col& c, int64_t row, uint8_t& flag, uint64_t& bits, bool& isnull) {
    isnull = col_is_null(c, row);
    if (isnull) {
        flag = NilFlag;
        bits = 0;
        return;
    }
    switch (c.type) {
        case TSQ_I64:
            flag = varintFlag;
            bits = (uint64_t)col_i64(c, row);
            break;
        case TSQ_U64: {  // codec.go:219-224: UNSIGNED column and value<0 as int64 => uvarintFlag
            int64_t v = col_i64(c, row);
            flag = v < 0 ? uvarintFlag : varintFlag;
            bits = (uint64_t)v;
            break;
        }
        case TSQ_F32: {  // codec.go:226-229: widened to float64
            double d = (double)((const float*)c.data)[row];
            flag = floatFlag;
            memcpy(&bits, &d, 8);
            break;
        }
        case TSQ_F64:
            flag = floatFlag;
            memcpy(&bits, &((const double*)c.data)[row], 8);
            break;
        default:
            flag = 0xff;
            bits = 0;
    }
}

// ------------------------------------------------------------------ splitmix64
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64_t gen_r(const tsq_gen_spec& s, uint64_t i, uint64_t c) {
    return splitmix64(s.seed ^ ((uint64_t)s.table << 56) ^ (c << 48) ^ i);
}

// order-independent row checksum (shared definition with tinysql_amd/csrc: tsq_rowhash)
constexpr uint64_t ROWHASH_SEED = 0x243F6A8885A308D3ULL;
constexpr uint64_t ROWHASH_NULL = 0xA5A5A5A55A5A5A5AULL;
inline uint64_t rowhash_step(uint64_t h, uint64_t v, uint32_t c) {
    return splitmix64(h ^ (v + (uint64_t)(c + 1) * 0x9E3779B97F4A7C15ULL));
}

// ------------------------------------------------------------------ rowHashMap (hash_table.go:181-276)
struct EntryAddr {
    uint32_t sliceIdx = 0, offset = 0;
    bool isNull() const { return sliceIdx == 0 && offset == 0; }
};
struct Entry {
    uint64_t ptr;  // chunk.RowPtr{ChkIdx,RowIdx} packed (list.go:22-38); here: global row index
    EntryAddr next;
};
struct EntryStore {  // hash_table.go:185-219: slabs growing 64 -> 8192 entries
    std::vector<std::vector<Entry>> slices;
    void init() {
        slices.clear();
        slices.emplace_back();
        slices.back().reserve(64);
        put(Entry{0, EntryAddr{}});  // reserved nullEntryAddr
    }
    EntryAddr put(const Entry& e) {
        uint32_t si = (uint32_t)slices.size() - 1;
        if (slices[si].size() == slices[si].capacity()) {
            size_t size = slices[si].capacity() * 2;
            if (size >= 8 * 1024) size = 8 * 1024;
            slices.emplace_back();
            slices.back().reserve(size);
            si++;
        }
        EntryAddr a;
        a.sliceIdx = si;
        a.offset = (uint32_t)slices[si].size();
        slices[si].push_back(e);
        return a;
    }
    const Entry& get(EntryAddr a) const { return slices[a.sliceIdx][a.offset]; }
};
struct RowHashMap {
    EntryStore store;
    std::unordered_map<uint64_t, EntryAddr> table;
    int64_t length = 0;
    explicit RowHashMap(size_t est = 0) {
        store.init();
        if (est) table.reserve(est);
    }
    void Put(uint64_t key, uint64_t ptr) {  // hash_table.go:247-256 push at head
        EntryAddr old{};
        auto it = table.find(key);
        if (it != table.end()) old = it->second;
        EntryAddr na = store.put(Entry{ptr, old});
        table[key] = na;
        length++;
    }
    // hash_table.go:259-272: walk chain then reverse => insertion order
    void Get(uint64_t key, std::vector<uint64_t>& out) const {
        out.clear();
        auto it = table.find(key);
        if (it == table.end()) return;
        EntryAddr a = it->second;
        while (!a.isNull()) {
            const Entry& e = store.get(a);
            a = e.next;
            out.push_back(e.ptr);
        }
        std::reverse(out.begin(), out.end());
    }
};

}  // namespace

#include "orc_result_internal.h"  // OutCol, orc_result (shared with mocktikv.cpp)

namespace {

// ------------------------------------------------------------------ expressions
// A materialised intermediate column (what every builtin*Sig.vecEval* produces).
struct ECol {
    bool real = false, str = false;
    std::vector<int64_t> i;
    std::vector<double> f;
    std::vector<std::string> s;  // ETString results: the evaluator copies the bytes, like Column.AppendString
    std::vector<uint8_t> null;  // 1 = NULL
    void resize(int64_t n, bool r) {
        real = r;
        str = false;
        null.assign(n, 0);
        if (r) f.assign(n, 0.0);
        else i.assign(n, 0);
    }
    void resize_str(int64_t n) {
        real = false;
        str = true;
        null.assign(n, 0);
        s.assign(n, std::string());
    }
};
// types/compare.go:115-123 CompareString: Go's string order is byte-wise lexicographic
inline int cmp_string(const std::string& x, const std::string& y) {
    const int c = x.compare(y);  // char_traits<char>::compare = memcmp, then the lengths
    return c < 0 ? -1 : (c == 0 ? 0 : 1);
}

inline int64_t wrap_neg(int64_t v) { return (int64_t)(0 - (uint64_t)v); }          // Go -v wraps
inline int64_t wrap_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
inline int64_t wrap_sub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
inline int64_t wrap_mul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
inline int64_t go_div(int64_t a, int64_t b) {  // Go spec: MinInt64 / -1 == MinInt64, no trap
    if (b == -1) return wrap_neg(a);
    return a / b;
}
constexpr int64_t I64MAX = std::numeric_limits<int64_t>::max();
constexpr int64_t I64MIN = std::numeric_limits<int64_t>::min();
constexpr uint64_t U64MAX = std::numeric_limits<uint64_t>::max();
constexpr double F64MAX = std::numeric_limits<double>::max();

// types/compare.go:44-101 VecCompare{UU,II,UI,IU}
inline int cmp_int(int64_t x, int64_t y, bool ux, bool uy) {
    if (ux && uy) {
        uint64_t a = (uint64_t)x, b = (uint64_t)y;
        return a < b ? -1 : (a == b ? 0 : 1);
    }
    if (ux && !uy) {  // VecCompareUI
        if (y < 0 || (uint64_t)x > (uint64_t)I64MAX) return 1;
        return x < y ? -1 : (x == y ? 0 : 1);
    }
    if (!ux && uy) {  // VecCompareIU
        if (x < 0 || (uint64_t)y > (uint64_t)I64MAX) return -1;
        return x < y ? -1 : (x == y ? 0 : 1);
    }
    return x < y ? -1 : (x == y ? 0 : 1);
}
// types/compare.go:104-112 CompareFloat64 (NaN compares as "greater")
inline int cmp_real(double x, double y) { return x < y ? -1 : (x == y ? 0 : 1); }

// types/helper.go:28 RoundFloat
inline double round_float(double f) {
    if (std::fabs(f) < 0.5) return 0;
    return std::trunc(f + std::copysign(0.5, f));
}

struct EvalCtx {
    const tsq_col* cols;
    int32_t n_cols;
    int64_t n;           // logical rows
    const int32_t* sel;  // logical -> physical (chunk.go:319-331) or null
    int64_t warnings = 0;
    int64_t phys(int64_t i) const { return sel ? sel[i] : i; }
};

// Evaluates one postfix program node-at-a-time over all n logical rows; the first node (in
// post-order == the reference's evaluation order: arg0, arg1, ..., op) with an offending row
// aborts with its error, exactly like `return types.ErrOverflow...` inside the vec loops.
tsq_status eval_prog(const tsq_expr_prog& p, EvalCtx& cx, ECol& out) {
    std::vector<ECol> st;
    const int64_t n = cx.n;
    for (int32_t k = 0; k < p.n_ops; k++) {
        const tsq_expr_op& op = p.ops[k];
        const bool ul = op.flags & TSQ_F_LHS_UNSIGNED, ur = op.flags & TSQ_F_RHS_UNSIGNED;
        switch (op.opcode) {
            case TSQ_OP_COL_INT: {  // expression/column.go:56-75 -> CopyReconstruct(sel)
                if (op.arg >= cx.n_cols) { g_err = "column index out of range"; return TSQ_ERR_INVALID; }
                const tsq_col& c = cx.cols[op.arg];
                ECol r;
                r.resize(n, false);
                for (int64_t i = 0; i < n; i++) {
                    int64_t ph = cx.phys(i);
                    r.null[i] = col_is_null(c, ph);
                    r.i[i] = col_i64(c, ph);
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_COL_REAL: {  // expression/column.go:77-104
                if (op.arg >= cx.n_cols) { g_err = "column index out of range"; return TSQ_ERR_INVALID; }
                const tsq_col& c = cx.cols[op.arg];
                ECol r;
                r.resize(n, true);
                for (int64_t i = 0; i < n; i++) {
                    int64_t ph = cx.phys(i);
                    r.null[i] = col_is_null(c, ph);
                    r.f[i] = col_f64(c, ph);
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_CONST_INT:
            case TSQ_OP_CONST_NULL_INT: {  // expression/vectorized.go:23-80 broadcast
                ECol r;
                r.resize(n, false);
                bool isnull = op.opcode == TSQ_OP_CONST_NULL_INT;
                for (int64_t i = 0; i < n; i++) {
                    r.null[i] = isnull;
                    r.i[i] = isnull ? 0 : p.consts[op.arg];
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_CONST_REAL:
            case TSQ_OP_CONST_NULL_REAL: {
                ECol r;
                r.resize(n, true);
                bool isnull = op.opcode == TSQ_OP_CONST_NULL_REAL;
                double d = 0;
                if (!isnull) memcpy(&d, &p.consts[op.arg], 8);
                for (int64_t i = 0; i < n; i++) {
                    r.null[i] = isnull;
                    r.f[i] = d;
                }
                st.push_back(std::move(r));
                break;
            }
            // ---------------- real arithmetic: builtin_arithmetic_vec.go:29-60,62-91,275-304,348-383
            case TSQ_OP_PLUS_REAL:
            case TSQ_OP_MINUS_REAL:
            case TSQ_OP_MUL_REAL:
            case TSQ_OP_DIV_REAL: {
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    a.null[i] |= b.null[i];  // MergeNulls (column.go:559-574)
                    if (a.null[i]) continue;
                    double x = a.f[i], y = b.f[i];
                    if (op.opcode == TSQ_OP_PLUS_REAL) {
                        if ((x > 0 && y > F64MAX - x) || (x < 0 && y < -F64MAX - x)) return TSQ_ERR_OVERFLOW_DOUBLE;
                        a.f[i] = x + y;
                    } else if (op.opcode == TSQ_OP_MINUS_REAL) {
                        if ((x > 0 && -y > F64MAX - x) || (x < 0 && -y < -F64MAX - x)) return TSQ_ERR_OVERFLOW_DOUBLE;
                        a.f[i] = x - y;
                    } else if (op.opcode == TSQ_OP_MUL_REAL) {
                        a.f[i] = x * y;
                        if (std::isinf(a.f[i])) return TSQ_ERR_OVERFLOW_DOUBLE;
                    } else {
                        if (y == 0) {  // :369-375 handleDivisionByZeroError -> warning, NULL
                            cx.warnings++;
                            a.null[i] = 1;
                            continue;
                        }
                        a.f[i] = x / y;
                        if (std::isinf(a.f[i])) return TSQ_ERR_OVERFLOW_DOUBLE;
                    }
                }
                st.push_back(std::move(a));
                break;
            }
            // ---------------- int plus: builtin_arithmetic_vec.go:389-495
            case TSQ_OP_PLUS_INT: {
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    a.null[i] |= b.null[i];
                    if (a.null[i]) continue;
                    int64_t lh = a.i[i], rh = b.i[i];
                    if (ul && ur) {  // plusUU :428
                        if ((uint64_t)lh > U64MAX - (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    } else if (ul && !ur) {  // plusUS :444 (second check compares lh with itself, :454 — reproduced)
                        if (rh < 0 && (uint64_t)wrap_neg(rh) > (uint64_t)lh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                        if (rh > 0 && (uint64_t)lh > U64MAX - (uint64_t)lh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    } else if (!ul && ur) {  // plusSU :463
                        if (lh < 0 && (uint64_t)wrap_neg(lh) > (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                        if (lh > 0 && (uint64_t)rh > U64MAX - (uint64_t)lh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    } else {  // plusSS :481
                        if ((lh > 0 && rh > I64MAX - lh) || (lh < 0 && rh < I64MIN - lh)) return TSQ_ERR_OVERFLOW_BIGINT;
                    }
                    a.i[i] = wrap_add(lh, rh);
                }
                st.push_back(std::move(a));
                break;
            }
            // ---------------- int minus: builtin_arithmetic_vec.go:95-271
            case TSQ_OP_MINUS_INT: {
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                const bool force = op.flags & TSQ_F_FORCE_SIGNED;
                for (int64_t i = 0; i < n; i++) {
                    a.null[i] |= b.null[i];
                    if (a.null[i]) continue;
                    int64_t lh = a.i[i], rh = b.i[i];
                    auto ss_over = [&]() {
                        return (lh > 0 && wrap_neg(rh) > I64MAX - lh) || (lh < 0 && wrap_neg(rh) < I64MIN - lh);
                    };
                    if (force && ul && ur) {  // minusFUU :142
                        if (lh < 0) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                        if (rh < 0) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                        if (ss_over()) return TSQ_ERR_OVERFLOW_BIGINT;
                    } else if (force && ul && !ur) {  // minusFUS :166
                        if (lh < 0) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                        if (ss_over()) return TSQ_ERR_OVERFLOW_BIGINT;
                    } else if (force && !ul && ur) {  // minusFSU :186
                        if (rh < 0) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                        if (ss_over()) return TSQ_ERR_OVERFLOW_BIGINT;
                    } else if (!force && ul && ur) {  // minusUU :205
                        if ((uint64_t)lh < (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    } else if (!force && ul && !ur) {  // minusUS :221
                        if (rh >= 0 && (uint64_t)lh < (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                        if (rh < 0 && (uint64_t)lh > U64MAX - (uint64_t)wrap_neg(rh)) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    } else if (!force && !ul && ur) {  // minusSU :240
                        if ((uint64_t)wrap_sub(lh, I64MIN) < (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    } else {  // minusSS :255
                        if (ss_over()) return TSQ_ERR_OVERFLOW_BIGINT;
                    }
                    a.i[i] = wrap_sub(lh, rh);
                }
                st.push_back(std::move(a));
                break;
            }
            case TSQ_OP_MUL_INT: {  // builtin_arithmetic_vec.go:308-342
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    a.null[i] |= b.null[i];
                    if (a.null[i]) continue;
                    int64_t x = a.i[i], y = b.i[i];
                    int64_t tmp = wrap_mul(x, y);
                    if (x != 0 && go_div(tmp, x) != y) return TSQ_ERR_OVERFLOW_BIGINT;
                    a.i[i] = tmp;
                }
                st.push_back(std::move(a));
                break;
            }
            case TSQ_OP_MUL_INT_UNSIGNED: {  // :501-532
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    a.null[i] |= b.null[i];
                    if (a.null[i]) continue;
                    uint64_t x = (uint64_t)a.i[i], y = (uint64_t)b.i[i];
                    uint64_t res = x * y;
                    if (x != 0 && res / x != y) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    a.i[i] = (int64_t)res;
                }
                st.push_back(std::move(a));
                break;
            }
            // ---------------- compares: builtin_compare_vec.go:26-292, _generated.go
            case TSQ_OP_LT_INT: case TSQ_OP_LE_INT: case TSQ_OP_GT_INT:
            case TSQ_OP_GE_INT: case TSQ_OP_EQ_INT: case TSQ_OP_NE_INT:
            case TSQ_OP_LT_REAL: case TSQ_OP_LE_REAL: case TSQ_OP_GT_REAL:
            case TSQ_OP_GE_REAL: case TSQ_OP_EQ_REAL: case TSQ_OP_NE_REAL: {
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                ECol r;
                r.resize(n, false);
                const bool isreal = op.opcode >= TSQ_OP_LT_REAL;
                const int rel = isreal ? op.opcode - TSQ_OP_LT_REAL : op.opcode - TSQ_OP_LT_INT;
                for (int64_t i = 0; i < n; i++) {
                    r.null[i] = a.null[i] | b.null[i];
                    if (r.null[i]) continue;  // (int path computes then MergeNulls; value of NULL rows is unobservable)
                    int c = isreal ? cmp_real(a.f[i], b.f[i]) : cmp_int(a.i[i], b.i[i], ul, ur);
                    bool v = false;
                    switch (rel) {
                        case 0: v = c < 0; break;
                        case 1: v = c <= 0; break;
                        case 2: v = c > 0; break;
                        case 3: v = c >= 0; break;
                        case 4: v = c == 0; break;
                        case 5: v = c != 0; break;
                    }
                    r.i[i] = v ? 1 : 0;
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_LOGIC_AND: {  // builtin_op_vec.go:173-216
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    bool n0 = a.null[i], n1 = b.null[i];
                    if (!n0 && a.i[i] == 0) { a.null[i] = 0; continue; }
                    if (!n1 && b.i[i] == 0) { a.i[i] = 0; a.null[i] = 0; continue; }
                    if (n0 || n1) { a.null[i] = 1; continue; }
                    a.i[i] = 1;
                }
                st.push_back(std::move(a));
                break;
            }
            case TSQ_OP_LOGIC_OR: {  // builtin_op_vec.go:29-68
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    bool n0 = a.null[i], n1 = b.null[i];
                    if ((!n0 && a.i[i] != 0) || (!n1 && b.i[i] != 0)) { a.i[i] = 1; a.null[i] = 0; }
                    else if (n0 || n1) a.null[i] = 1;
                    else { a.i[i] = 0; a.null[i] = 0; }
                }
                st.push_back(std::move(a));
                break;
            }
            case TSQ_OP_NOT_INT: {  // :249-270
                ECol& a = st.back();
                for (int64_t i = 0; i < n; i++) {
                    if (a.null[i]) continue;
                    a.i[i] = a.i[i] == 0 ? 1 : 0;
                }
                break;
            }
            case TSQ_OP_NOT_REAL: {  // :141-167
                ECol a = std::move(st.back()); st.pop_back();
                ECol r;
                r.resize(n, false);
                for (int64_t i = 0; i < n; i++) {
                    r.null[i] = a.null[i];
                    if (r.null[i]) continue;
                    r.i[i] = a.f[i] == 0 ? 1 : 0;
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_NEG_INT: {  // :221-243.  NULL rows: the reference loop does not skip them,
                // so a NULL cell whose stale payload is MinInt64 would raise; NULL payloads are
                // defined as 0 by Column.AppendNull (column.go:150-158) so this is unobservable
                // for well-formed input.  We skip NULL rows (row-path semantics).
                ECol& a = st.back();
                for (int64_t i = 0; i < n; i++) {
                    if (a.null[i]) continue;
                    if (ul) {
                        if ((uint64_t)a.i[i] > (uint64_t)1 << 63) return TSQ_ERR_OVERFLOW_BIGINT;
                    } else if (a.i[i] == I64MIN) return TSQ_ERR_OVERFLOW_BIGINT;
                    a.i[i] = wrap_neg(a.i[i]);
                }
                break;
            }
            case TSQ_OP_NEG_REAL: {  // :74-86
                ECol& a = st.back();
                for (int64_t i = 0; i < n; i++) a.f[i] = -a.f[i];
                break;
            }
            case TSQ_OP_ISNULL_INT:
            case TSQ_OP_ISNULL_REAL: {  // :92-135
                ECol a = std::move(st.back()); st.pop_back();
                ECol r;
                r.resize(n, false);
                for (int64_t i = 0; i < n; i++) r.i[i] = a.null[i] ? 1 : 0;
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_IFNULL_INT:
            case TSQ_OP_IFNULL_REAL: {  // builtin_control_vec_generated.go:23-80
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    if (a.null[i] && !b.null[i]) {
                        a.null[i] = 0;
                        if (a.real) a.f[i] = b.f[i];
                        else a.i[i] = b.i[i];
                    }
                }
                st.push_back(std::move(a));
                break;
            }
            case TSQ_OP_IF_INT:
            case TSQ_OP_IF_REAL: {  // :117-205
                ECol c2 = std::move(st.back()); st.pop_back();
                ECol c1 = std::move(st.back()); st.pop_back();
                ECol c0 = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    if (c0.null[i] || c0.i[i] == 0) {
                        c1.null[i] = c2.null[i];
                        if (!c2.null[i]) {
                            if (c1.real) c1.f[i] = c2.f[i];
                            else c1.i[i] = c2.i[i];
                        }
                    }
                }
                st.push_back(std::move(c1));
                break;
            }
            case TSQ_OP_IN_INT:
            case TSQ_OP_IN_REAL: {  // builtin_other_vec_generated.go:24-95,151-205
                int nitems = op.arg;
                if ((int)st.size() < nitems + 1) { g_err = "IN: stack underflow"; return TSQ_ERR_INVALID; }
                std::vector<ECol> items(nitems);
                for (int j = nitems - 1; j >= 0; j--) { items[j] = std::move(st.back()); st.pop_back(); }
                ECol x = std::move(st.back()); st.pop_back();
                ECol r;
                r.resize(n, false);
                std::vector<uint8_t> hasNull(n, 0), found(n, 0);
                for (int j = 0; j < nitems; j++) {
                    bool uj = (op.aux >> j) & 1;
                    for (int64_t i = 0; i < n; i++) {
                        if (items[j].null[i] || x.null[i]) { hasNull[i] = 1; continue; }
                        bool eq;
                        if (op.opcode == TSQ_OP_IN_REAL) eq = cmp_real(x.f[i], items[j].f[i]) == 0;
                        else {
                            int64_t a0 = x.i[i], a1 = items[j].i[i];
                            if (ul == uj) eq = a1 == a0;
                            else if (!ul && uj) eq = a0 >= 0 && a1 == a0;
                            else eq = a1 >= 0 && a1 == a0;
                        }
                        if (eq) found[i] = 1;
                    }
                }
                for (int64_t i = 0; i < n; i++) {
                    if (found[i]) { r.i[i] = 1; r.null[i] = 0; }
                    else { r.i[i] = 0; r.null[i] = hasNull[i]; }
                }
                st.push_back(std::move(r));
                break;
            }
            // ---------------- strings (ETString), binary collation
            case TSQ_OP_COL_STR: {  // expression/column.go:106-129 VecEvalString -> CopyReconstruct(sel)
                if (op.arg >= cx.n_cols) { g_err = "column index out of range"; return TSQ_ERR_INVALID; }
                const tsq_col& c = cx.cols[op.arg];
                if (c.type != TSQ_BYTES || !c.offsets) { g_err = "COL_STR on a fixed-width column"; return TSQ_ERR_INVALID; }
                ECol r;
                r.resize_str(n);
                for (int64_t i = 0; i < n; i++) {
                    const int64_t ph = cx.phys(i);
                    r.null[i] = col_is_null(c, ph);
                    if (!r.null[i]) r.s[i].assign((const char*)c.data + c.offsets[ph], (size_t)(c.offsets[ph + 1] - c.offsets[ph]));
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_CONST_STR:
            case TSQ_OP_CONST_NULL_STR: {  // expression/vectorized.go:23-80 broadcast
                ECol r;
                r.resize_str(n);
                const bool isnull = op.opcode == TSQ_OP_CONST_NULL_STR;
                std::string v;
                if (!isnull) {
                    const uint64_t ol = (uint64_t)p.consts[op.arg];
                    v.assign((const char*)p.str_pool + (ol >> 32), (size_t)(ol & 0xffffffffu));
                }
                for (int64_t i = 0; i < n; i++) {
                    r.null[i] = isnull;
                    r.s[i] = v;
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_LT_STR: case TSQ_OP_LE_STR: case TSQ_OP_GT_STR:
            case TSQ_OP_GE_STR: case TSQ_OP_EQ_STR: case TSQ_OP_NE_STR:  // builtin_compare_vec_generated.go:65-555
            case TSQ_OP_STRCMP: {                                        // builtin_string_vec.go:52-83
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                ECol r;
                r.resize(n, false);
                for (int64_t i = 0; i < n; i++) {
                    r.null[i] = a.null[i] || b.null[i];  // result.MergeNulls(buf0, buf1)
                    if (r.null[i]) continue;
                    const int c = cmp_string(a.s[i], b.s[i]);
                    if (op.opcode == TSQ_OP_STRCMP) { r.i[i] = c; continue; }
                    const int rel = op.opcode - TSQ_OP_LT_STR;
                    const bool v = rel == 0 ? c < 0 : rel == 1 ? c <= 0 : rel == 2 ? c > 0 : rel == 3 ? c >= 0 : rel == 4 ? c == 0 : c != 0;
                    r.i[i] = v ? 1 : 0;
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_LENGTH: {  // builtin_string_vec.go:89-92 is a course stub ("Your code here"): MySQL LENGTH() = bytes, NULL in -> NULL out
                ECol a = std::move(st.back()); st.pop_back();
                ECol r;
                r.resize(n, false);
                for (int64_t i = 0; i < n; i++) {
                    r.null[i] = a.null[i];
                    r.i[i] = a.null[i] ? 0 : (int64_t)a.s[i].size();
                }
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_ISNULL_STR: {  // builtin_string_vec.go:21-42
                ECol a = std::move(st.back()); st.pop_back();
                ECol r;
                r.resize(n, false);
                for (int64_t i = 0; i < n; i++) r.i[i] = a.null[i] ? 1 : 0;
                st.push_back(std::move(r));
                break;
            }
            case TSQ_OP_IFNULL_STR: {  // builtin_control_vec_generated.go:81-111
                ECol b = std::move(st.back()); st.pop_back();
                ECol a = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    if (a.null[i] && !b.null[i]) {
                        a.null[i] = 0;
                        a.s[i] = b.s[i];
                    }
                }
                st.push_back(std::move(a));
                break;
            }
            case TSQ_OP_IF_STR: {  // builtin_control_vec_generated.go:209-253
                ECol c2 = std::move(st.back()); st.pop_back();
                ECol c1 = std::move(st.back()); st.pop_back();
                ECol c0 = std::move(st.back()); st.pop_back();
                for (int64_t i = 0; i < n; i++) {
                    if (c0.null[i] || c0.i[i] == 0) {
                        c1.null[i] = c2.null[i];
                        c1.s[i] = c2.s[i];
                    }
                }
                st.push_back(std::move(c1));
                break;
            }
            case TSQ_OP_IN_STR: {  // builtin_other_vec_generated.go:97-145
                const int nitems = op.arg;
                if ((int)st.size() < nitems + 1) { g_err = "IN: stack underflow"; return TSQ_ERR_INVALID; }
                std::vector<ECol> items(nitems);
                for (int j = nitems - 1; j >= 0; j--) { items[j] = std::move(st.back()); st.pop_back(); }
                ECol x = std::move(st.back()); st.pop_back();
                ECol r;
                r.resize(n, false);
                std::vector<uint8_t> hasNull(n, 0), found(n, 0);
                for (int j = 0; j < nitems; j++) {
                    for (int64_t i = 0; i < n; i++) {
                        if (items[j].null[i] || x.null[i]) { hasNull[i] = 1; continue; }
                        if (cmp_string(x.s[i], items[j].s[i]) == 0) found[i] = 1;
                    }
                }
                for (int64_t i = 0; i < n; i++) {
                    if (found[i]) { r.i[i] = 1; r.null[i] = 0; }
                    else { r.i[i] = 0; r.null[i] = hasNull[i]; }
                }
                st.push_back(std::move(r));
                break;
            }
            default:
                g_err = "unknown opcode " + std::to_string(op.opcode);
                return TSQ_ERR_INVALID;
        }
    }
    if (st.size() != 1) { g_err = "malformed program: stack depth != 1"; return TSQ_ERR_INVALID; }
    // a string-valued root is a VecEvalString call (orc_expr_eval_str); VecEvalInt / VecEvalReal / VecEvalBool callers declare an
    // Int or Real result
    if (st.back().str != (p.result_type == TSQ_BYTES)) { g_err = "string-valued root <=> result_type TSQ_BYTES"; return TSQ_ERR_UNSUPPORTED; }
    out = std::move(st.back());
    return TSQ_OK;
}

// expression.go:205-279 VecEvalBool
tsq_status vec_eval_bool(const tsq_expr_prog* progs, int32_t n_progs, const tsq_col* cols, int32_t n_cols,
                         int64_t nrows, const int32_t* in_sel, std::vector<uint8_t>& selected,
                         std::vector<uint8_t>& nulls, int64_t* warnings) {
    // chunk_executor.go:227-244: with a pre-existing input.Sel(), logical row i of the input is
    // evaluated; `selected` is indexed by logical row.
    selected.assign(nrows, 0);
    nulls.assign(nrows, 0);
    std::vector<int32_t> sel(nrows);  // logical indices still alive
    for (int64_t i = 0; i < nrows; i++) sel[i] = (int32_t)i;
    for (int32_t e = 0; e < n_progs; e++) {
        // input.SetSel(sel): evaluate only surviving rows
        std::vector<int32_t> phys(sel.size());
        for (size_t k = 0; k < sel.size(); k++) phys[k] = in_sel ? in_sel[sel[k]] : sel[k];
        EvalCtx cx{cols, n_cols, (int64_t)sel.size(), phys.data()};
        ECol buf;
        tsq_status s = eval_prog(progs[e], cx, buf);
        if (warnings) *warnings += cx.warnings;
        if (s != TSQ_OK) return s;
        const bool isInt = !buf.real;
        size_t j = 0;
        for (size_t i = 0; i < sel.size(); i++) {
            int8_t isZero;  // toBool, expression.go:281-326
            if (buf.null[i]) isZero = -1;
            else if (isInt) isZero = buf.i[i] == 0 ? 0 : 1;
            else isZero = round_float(buf.f[i]) == 0 ? 0 : 1;
            if (isZero == -1) {
                if (!isInt) continue;
                nulls[sel[i]] = 1;
                sel[j++] = sel[i];
                continue;
            }
            if (isZero == 0) continue;
            sel[j++] = sel[i];
        }
        sel.resize(j);
    }
    for (int32_t i : sel)
        if (!nulls[i]) selected[i] = 1;
    return TSQ_OK;
}

// ------------------------------------------------------------------ hash join
struct JoinSide {
    const tsq_col* cols;
    int32_t ncols;
    const int32_t* key_idx;
};

// util/codec/codec.go:249-338 HashChunkSelected for all key cols of one row
inline uint64_t hash_row_keys(const JoinSide& s, int32_t n_keys, int64_t row, bool& hasNull) {
    uint64_t h = FNV_OFFSET;
    hasNull = false;
    for (int32_t k = 0; k < n_keys; k++) {
        uint8_t flag;
        uint64_t bits;
        bool isnull;
        const tsq_col& c = s.cols[s.key_idx[k]];
        if (c.type == TSQ_BYTES) {  // codec.go:233-235: flag = compactBytesFlag, b = row.GetBytes(idx)
            isnull = col_is_null(c, row);
            flag = isnull ? NilFlag : compactBytesFlag;
            h = fnv1_write(h, &flag, 1);
            if (isnull) hasNull = true;
            else h = fnv1_write(h, (const uint8_t*)c.data + c.offsets[row], c.offsets[row + 1] - c.offsets[row]);
            continue;
        }
        encode_hash_cell(c, row, flag, bits, isnull);
        h = fnv1_write(h, &flag, 1);
        if (isnull) hasNull = true;  // codec.go:261-263: flag only, no bytes
        else h = fnv1_write(h, (const uint8_t*)&bits, 8);
    }
    return h;
}
// util/codec/codec.go:363-382 EqualChunkRow
inline bool equal_row_keys(const JoinSide& a, int64_t ra, const JoinSide& b, int64_t rb, int32_t n_keys) {
    for (int32_t k = 0; k < n_keys; k++) {
        uint8_t f1, f2;
        uint64_t b1, b2;
        bool n1, n2;
        const tsq_col &ca = a.cols[a.key_idx[k]], &cb = b.cols[b.key_idx[k]];
        if (ca.type == TSQ_BYTES || cb.type == TSQ_BYTES) {  // flag1 == flag2 && bytes.Equal(b1, b2)
            if (ca.type != cb.type) return false;
            const bool na = col_is_null(ca, ra), nb = col_is_null(cb, rb);
            if (na || nb) {
                if (na != nb) return false;
                continue;
            }
            const int64_t la = ca.offsets[ra + 1] - ca.offsets[ra], lb = cb.offsets[rb + 1] - cb.offsets[rb];
            if (la != lb || memcmp((const char*)ca.data + ca.offsets[ra], (const char*)cb.data + cb.offsets[rb], (size_t)la) != 0) return false;
            continue;
        }
        encode_hash_cell(ca, ra, f1, b1, n1);
        encode_hash_cell(cb, rb, f2, b2, n2);
        if (!(f1 == f2 && b1 == b2)) return false;
    }
    return true;
}

struct JoinState {
    const tsq_join_cfg* cfg;
    JoinSide build, probe;
    int64_t n_build;
    RowHashMap* map;
};

void build_table(JoinState& js) {
    // fetchAndBuildHashTable (join.go:148-158 STUB; intended per proj5-part2 README): drain the
    // inner child chunk by chunk into hashRowContainer.PutChunk (hash_table.go:146-169).
    const int32_t chunk = js.cfg->max_chunk_size > 0 ? js.cfg->max_chunk_size : 1024;
    for (int64_t base = 0; base < js.n_build; base += chunk) {
        int64_t rows = std::min<int64_t>(chunk, js.n_build - base);
        for (int64_t i = 0; i < rows; i++) {
            bool hasNull;
            uint64_t h = hash_row_keys(js.build, js.cfg->n_keys, base + i, hasNull);
            if (hasNull) continue;  // hash_table.go:161-163
            js.map->Put(h, (uint64_t)(base + i));
        }
    }
}

void append_cell(OutCol& oc, const tsq_col& c, int64_t row) {
    bool isnull = col_is_null(c, row);
    if (c.type == TSQ_BYTES) {  // Chunk.AppendRow of a var-len cell (chunk.go:334-356, appendCellByCell)
        if (isnull) oc.append_raw(0, false);
        else oc.append_bytes((const char*)c.data + c.offsets[row], (size_t)(c.offsets[row + 1] - c.offsets[row]));
        return;
    }
    oc.append_raw(isnull ? 0 : col_raw64(c, row), !isnull);
}

}  // namespace

void orc_set_error(const std::string& msg) { g_err = msg; }

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }
int64_t orc_result_rows(const orc_result* r) { return r->rows; }
int32_t orc_result_cols(const orc_result* r) { return (int32_t)r->cols.size(); }
int32_t orc_result_col_type(const orc_result* r, int32_t c) { return r->cols[c].type; }
void orc_result_copy_col(const orc_result* r, int32_t c, void* data, uint8_t* notnull) {
    const OutCol& oc = r->cols[c];
    if (data) {
        if (oc.type == TSQ_F32) {
            for (int64_t i = 0; i < r->rows; i++) ((uint32_t*)data)[i] = (uint32_t)oc.v[i];
        } else {
            memcpy(data, oc.v.data(), r->rows * 8);
        }
    }
    if (notnull) memcpy(notnull, oc.notnull.data(), r->rows);
}
/* var-len result column: total data bytes; then offsets (rows + 1 entries, first 0) and the bytes */
int64_t orc_result_col_bytes(const orc_result* r, int32_t c) { return (int64_t)r->cols[c].bytes.size(); }
void orc_result_copy_varlen(const orc_result* r, int32_t c, int64_t* offsets, void* data, uint8_t* notnull) {
    const OutCol& oc = r->cols[c];
    offsets[0] = 0;
    for (int64_t i = 0; i < r->rows; i++) offsets[i + 1] = (int64_t)oc.v[i];
    if (!oc.bytes.empty()) memcpy(data, oc.bytes.data(), oc.bytes.size());
    if (notnull) memcpy(notnull, oc.notnull.data(), r->rows);
}
void orc_result_free(orc_result* r) { delete r; }

uint64_t orc_fnv1_64(const uint8_t* p, int64_t n) { return fnv1_write(FNV_OFFSET, p, n); }

void orc_hash_keys(const tsq_col* cols, const int32_t* key_idx, int32_t n_keys, int64_t nrows,
                   const uint8_t* selected, uint64_t* out_hash, uint8_t* out_has_null) {
    JoinSide s{cols, 0, key_idx};
    for (int64_t i = 0; i < nrows; i++) {
        if (selected && !selected[i]) {  // codec.go:253-255: skipped rows keep a fresh hash
            out_hash[i] = FNV_OFFSET;
            out_has_null[i] = 0;
            continue;
        }
        bool hn;
        out_hash[i] = hash_row_keys(s, n_keys, i, hn);
        out_has_null[i] = hn;
    }
}

// util/codec/codec.go:713-746 HashGroupKey (one cell).  number.go:122 EncodeVarint ==
// binary.PutVarint (zigzag + base-128); float.go:22-46 EncodeFloat (memcomparable, big endian).
int32_t orc_group_key_encode(const tsq_col* col, int64_t row, uint8_t* buf) {
    if (col_is_null(*col, row)) {
        buf[0] = NilFlag;
        return 1;
    }
    if (col->type == TSQ_I64 || col->type == TSQ_U64) {  // unsigned flag ignored (codec.go:715-723)
        int64_t v = col_i64(*col, row);
        uint64_t ux = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);  // zigzag
        int n = 0;
        buf[n++] = varintFlag;
        while (ux >= 0x80) {
            buf[n++] = (uint8_t)ux | 0x80;
            ux >>= 7;
        }
        buf[n++] = (uint8_t)ux;
        return n;
    }
    double f = col_f64(*col, row);
    uint64_t u;
    memcpy(&u, &f, 8);
    if (f >= 0) u |= 0x8000000000000000ULL;  // float.go:22-30 (note: -0.0 >= 0 is true)
    else u = ~u;
    buf[0] = floatFlag;
    for (int k = 0; k < 8; k++) buf[1 + k] = (uint8_t)(u >> (56 - 8 * k));  // EncodeUint big endian
    return 9;
}

void orc_gen_column(const tsq_gen_spec* spec, int64_t nrows, void* dst, uint8_t* null_bitmap, const void* src) {
    const tsq_gen_spec& s = *spec;
    if (null_bitmap) memset(null_bitmap, 0, (nrows + 7) / 8);
    for (int64_t k = 0; k < nrows; k++) {
        uint64_t i = (uint64_t)(s.start + k);
        bool isnull = s.null_pct > 0 && (gen_r(s, i, 7) % 100) < (uint64_t)s.null_pct;
        uint64_t v = 0;
        switch (s.kind) {
            case TSQ_GEN_SEQ: v = i; break;
            case TSQ_GEN_AFFINE: v = (s.a * (i % s.m) + s.b) % s.m; break;
            case TSQ_GEN_RAND_MOD: v = gen_r(s, i, (uint64_t)s.col) % s.m; break;
            case TSQ_GEN_RAND_F64: {
                double d = (double)(gen_r(s, i, (uint64_t)s.col) >> 11) * (1.0 / 9007199254740992.0);
                memcpy(&v, &d, 8);
                break;
            }
            case TSQ_GEN_HASH_OF_COL: v = splitmix64(((const uint64_t*)src)[k] ^ s.b); break;
            case TSQ_GEN_ZIPF_OCT: {  // an octave drawn uniformly, a value drawn uniformly inside it (include/tsq.h)
                const uint64_t r = gen_r(s, i, (uint64_t)s.col);
                const uint64_t lo = 1ull << (r % (s.a ? s.a : 1));
                v = (lo + (splitmix64(r) & (lo - 1)) - 1) % s.m;
                break;
            }
        }
        if (isnull) v = 0;  // NULL slot holds zero bytes (column.go:150-158)
        ((uint64_t*)dst)[k] = v;
        if (null_bitmap && !isnull) null_bitmap[k >> 3] |= (uint8_t)(1u << (k & 7));
    }
}

void orc_rows_checksum(const tsq_col* cols, int32_t n_cols, int64_t nrows, uint64_t* sum_out, uint64_t* xor_out) {
    uint64_t s = 0, x = 0;
    for (int64_t i = 0; i < nrows; i++) {
        uint64_t h = ROWHASH_SEED;
        for (int32_t c = 0; c < n_cols; c++) {
            uint64_t v = col_is_null(cols[c], i) ? ROWHASH_NULL : col_raw64(cols[c], i);
            h = rowhash_step(h, v, (uint32_t)c);
        }
        s += h;
        x ^= h;
    }
    *sum_out = s;
    *xor_out = x;
}

int64_t orc_rowhashmap_put_get(const uint64_t* keys, const uint64_t* ptrs, int64_t n, uint64_t probe_key,
                               uint64_t* out_ptrs, int64_t cap) {
    RowHashMap m;
    for (int64_t i = 0; i < n; i++) m.Put(keys[i], ptrs[i]);
    std::vector<uint64_t> got;
    m.Get(probe_key, got);
    for (int64_t i = 0; i < (int64_t)got.size() && i < cap; i++) out_ptrs[i] = got[i];
    return (int64_t)got.size();
}

// tryToMatchInners with OtherConditions (joiner.go:351-378 + filter :155-167): the outer row joined with every candidate inner row in a
// scratch chunk, VectorizedFilter over it; sel[r] = candidate r passes.  Shared by the hash join and the merge join (merge_join.go:291).
static tsq_status join_conds_select(const tsq_join_cfg* cfg, const tsq_col* probe_cols, const tsq_col* build_cols, int32_t probe_off, int32_t build_off,
                                    int64_t prow, const std::vector<int64_t>& matched, std::vector<uint8_t>& sel) {
    const int32_t nb = cfg->n_build_cols, np = cfg->n_probe_cols;
    const int64_t m = (int64_t)matched.size();
    std::vector<std::vector<uint64_t>> data(nb + np, std::vector<uint64_t>(m));
    std::vector<std::vector<uint8_t>> bm(nb + np, std::vector<uint8_t>((m + 7) / 8, 0));
    std::vector<tsq_col> jc(nb + np);
    auto fill = [&](int32_t oc, const tsq_col& src, int64_t srow, int64_t r) {
        bool isnull = col_is_null(src, srow);
        data[oc][r] = isnull ? 0 : col_raw64(src, srow);
        if (!isnull) bm[oc][r >> 3] |= (uint8_t)(1u << (r & 7));
    };
    for (int64_t r = 0; r < m; r++) {
        for (int32_t c = 0; c < np; c++) fill(probe_off + c, probe_cols[c], prow, r);
        for (int32_t c = 0; c < nb; c++) fill(build_off + c, build_cols[c], matched[r], r);
    }
    for (int32_t c = 0; c < nb + np; c++) {
        int32_t t = (c >= probe_off && c < probe_off + np) ? cfg->probe_types[c - probe_off] : cfg->build_types[c - build_off];
        jc[c] = tsq_col{};
        jc[c].length = m;
        jc[c].type = t;
        jc[c].elem_size = t == TSQ_F32 ? 4 : 8;
        jc[c].null_bitmap = bm[c].data();
        if (t == TSQ_F32) {  // repack 4-byte
            uint32_t* d32 = (uint32_t*)data[c].data();
            for (int64_t r = 0; r < m; r++) d32[r] = (uint32_t)data[c][r];
        }
        jc[c].data = data[c].data();
    }
    std::vector<uint8_t> nl2;
    int64_t w = 0;
    return vec_eval_bool(cfg->other_conds, cfg->n_other_conds, jc.data(), nb + np, m, nullptr, sel, nl2, &w);
}

orc_result* orc_hash_join(const tsq_join_cfg* cfg, const tsq_col* build_cols, int64_t n_build,
                          const tsq_col* probe_cols, int64_t n_probe, const uint8_t* selected_in,
                          tsq_status* status) {
    *status = TSQ_OK;
    RowHashMap map((size_t)std::max<int64_t>(0, cfg->est_build_rows / 8));  // hash_table.go:84-96
    JoinState js{cfg, {build_cols, cfg->n_build_cols, cfg->build_key_idx},
                 {probe_cols, cfg->n_probe_cols, cfg->probe_key_idx}, n_build, &map};
    build_table(js);

    const int32_t nb = cfg->n_build_cols, np = cfg->n_probe_cols;
    const bool outerIsRight = !cfg->build_is_right;  // outer(probe) side is the right child
    // output = left-child cols || right-child cols (joiner.go:145-150)
    const int32_t probe_off = outerIsRight ? nb : 0, build_off = outerIsRight ? 0 : np;
    orc_result* res = new orc_result();
    res->cols.resize(nb + np);
    for (int32_t c = 0; c < np; c++) res->cols[probe_off + c].type = cfg->probe_types[c];
    for (int32_t c = 0; c < nb; c++) res->cols[build_off + c].type = cfg->build_types[c];

    // outerSideFilter (join.go:328): VectorizedFilter over the probe chunk
    std::vector<uint8_t> selected(n_probe, 1), nulls;
    if (cfg->n_outer_filters > 0) {
        int64_t w = 0;
        tsq_status s = vec_eval_bool(cfg->outer_filters, cfg->n_outer_filters, probe_cols, np, n_probe, nullptr,
                                     selected, nulls, &w);
        if (s != TSQ_OK) { *status = s; delete res; return nullptr; }
    }
    if (selected_in)
        for (int64_t i = 0; i < n_probe; i++) selected[i] = selected[i] && selected_in[i];

    auto emit = [&](int64_t prow, int64_t brow) {  // makeJoinRowToChunk (joiner.go:145-150)
        for (int32_t c = 0; c < np; c++) append_cell(res->cols[probe_off + c], probe_cols[c], prow);
        for (int32_t c = 0; c < nb; c++) append_cell(res->cols[build_off + c], build_cols[c], brow);
        res->rows++;
    };
    auto on_miss = [&](int64_t prow) {  // onMissMatch (joiner.go:274-277,337-340,405-406)
        if (cfg->join_type == TSQ_JOIN_INNER) return;
        for (int32_t c = 0; c < np; c++) append_cell(res->cols[probe_off + c], probe_cols[c], prow);
        for (int32_t c = 0; c < nb; c++) res->cols[build_off + c].append_raw(0, false);  // defaultInner = NULLs
        res->rows++;
    };

    std::vector<uint64_t> ptrs;
    std::vector<int64_t> matched;
    for (int64_t i = 0; i < n_probe; i++) {  // join2Chunk (join.go:343-360)
        bool hasNull;
        uint64_t h = hash_row_keys(js.probe, cfg->n_keys, i, hasNull);
        if (!selected[i] || hasNull) { on_miss(i); continue; }
        // GetMatchedRows (hash_table.go:110-134)
        map.Get(h, ptrs);
        matched.clear();
        for (uint64_t p : ptrs)
            if (equal_row_keys(js.build, (int64_t)p, js.probe, i, cfg->n_keys)) matched.push_back((int64_t)p);
        if (matched.empty()) { on_miss(i); continue; }
        if (cfg->n_other_conds == 0) {
            for (int64_t b : matched) emit(i, b);
            continue;
        }
        // tryToMatchInners with conditions (join_conds_select): the selected candidates are joined
        bool hasMatch = false;
        {
            std::vector<uint8_t> sel2;
            tsq_status s = join_conds_select(cfg, probe_cols, build_cols, probe_off, build_off, i, matched, sel2);
            if (s != TSQ_OK) { *status = s; delete res; return nullptr; }
            for (size_t r = 0; r < matched.size(); r++)
                if (sel2[r]) { emit(i, matched[r]); hasMatch = true; }
        }
        if (!hasMatch) on_miss(i);  // join.go:319-321
    }
    return res;
}

// MergeJoinExec (executor/merge_join.go:31-373): both children arrive sorted on the join keys; the outer rows are taken in
// order, the inner table is consumed group by group ("rows with the same key", :93-127, rows with a NULL key skipped
// :148-156), joinToChunk (:257-310) compares the outer row with the current inner group: greater -> next group, smaller
// (or filtered out, or no group left) -> onMissMatch, equal -> the outer row joined with every row of the group in order.
// `build` plays the inner table, `probe` the outer one (cfg->build_is_right tells which child is which; output is
// left-child columns || right-child columns, joiner.go:145-150).  OtherConditions: tryToMatchInners filters the group's joined rows (:291,
// joiner.go:351-378); an outer row none of whose candidates passes is a miss (:296-299).
orc_result* orc_merge_join(const tsq_join_cfg* cfg, const tsq_col* inner_cols, int64_t n_inner, const tsq_col* outer_cols,
                           int64_t n_outer, tsq_status* status) {
    *status = TSQ_OK;
    const int32_t nb = cfg->n_build_cols, np = cfg->n_probe_cols;
    const bool outerIsRight = !cfg->build_is_right;
    const int32_t probe_off = outerIsRight ? nb : 0, build_off = outerIsRight ? 0 : np;
    orc_result* res = new orc_result();
    res->cols.resize(nb + np);
    for (int32_t c = 0; c < np; c++) res->cols[probe_off + c].type = cfg->probe_types[c];
    for (int32_t c = 0; c < nb; c++) res->cols[build_off + c].type = cfg->build_types[c];
    std::vector<uint8_t> selected(n_outer, 1), nulls;  // mergeJoinOuterTable.filter (:356-372)
    if (cfg->n_outer_filters > 0) {
        int64_t w = 0;
        tsq_status st = vec_eval_bool(cfg->outer_filters, cfg->n_outer_filters, outer_cols, np, n_outer, nullptr, selected, nulls, &w);
        if (st != TSQ_OK) { *status = st; delete res; return nullptr; }
    }
    // chunk.CompareFunc per key (compare.go:27-103) through expression.GetCmpFunction: NULL smallest, then by value
    auto cmp_cell = [&](const tsq_col& a, int64_t i, const tsq_col& b, int64_t j) -> int {
        const bool an = col_is_null(a, i), bn = col_is_null(b, j);
        if (an || bn) return (an && bn) ? 0 : (an ? -1 : 1);
        const bool areal = a.type == TSQ_F32 || a.type == TSQ_F64, breal = b.type == TSQ_F32 || b.type == TSQ_F64;
        if (areal || breal) {
            const double x = a.type == TSQ_F32 ? (double)((const float*)a.data)[i] : (a.type == TSQ_F64 ? ((const double*)a.data)[i] : (double)((const int64_t*)a.data)[i]);
            const double y = b.type == TSQ_F32 ? (double)((const float*)b.data)[j] : (b.type == TSQ_F64 ? ((const double*)b.data)[j] : (double)((const int64_t*)b.data)[j]);
            return x < y ? -1 : (x == y ? 0 : 1);
        }
        const uint64_t x = ((const uint64_t*)a.data)[i], y = ((const uint64_t*)b.data)[j];
        const bool au = a.type == TSQ_U64, bu = b.type == TSQ_U64;
        if (au == bu) return au ? (x < y ? -1 : x > y) : ((int64_t)x < (int64_t)y ? -1 : (int64_t)x > (int64_t)y);
        if (!au) return (int64_t)x < 0 ? -1 : (x < y ? -1 : x > y);   // signed vs unsigned (types.CompareInt)
        return (int64_t)y < 0 ? 1 : (x < y ? -1 : x > y);
    };
    auto cmp_keys = [&](int64_t orow, int64_t irow) {
        for (int k = 0; k < cfg->n_keys; k++) {
            const int c = cmp_cell(outer_cols[cfg->probe_key_idx[k]], orow, inner_cols[cfg->build_key_idx[k]], irow);
            if (c) return c;
        }
        return 0;
    };
    auto inner_has_null = [&](int64_t r) {
        for (int k = 0; k < cfg->n_keys; k++)
            if (col_is_null(inner_cols[cfg->build_key_idx[k]], r)) return true;
        return false;
    };
    auto emit = [&](int64_t prow, int64_t brow) {
        for (int32_t c = 0; c < np; c++) append_cell(res->cols[probe_off + c], outer_cols[c], prow);
        for (int32_t c = 0; c < nb; c++) append_cell(res->cols[build_off + c], inner_cols[c], brow);
        res->rows++;
    };
    auto on_miss = [&](int64_t prow) {
        if (cfg->join_type == TSQ_JOIN_INNER) return;
        for (int32_t c = 0; c < np; c++) append_cell(res->cols[probe_off + c], outer_cols[c], prow);
        for (int32_t c = 0; c < nb; c++) res->cols[build_off + c].append_raw(0, false);
        res->rows++;
    };
    // rowsWithSameKey (:93-127): [g0, g1) = the current inner group among the rows without a NULL key
    std::vector<int64_t> inner;
    for (int64_t r = 0; r < n_inner; r++)
        if (!inner_has_null(r)) inner.push_back(r);
    size_t g0 = 0, g1 = 0;
    auto next_group = [&]() {
        g0 = g1;
        if (g0 >= inner.size()) return;
        g1 = g0 + 1;
        while (g1 < inner.size()) {
            bool same = true;
            for (int k = 0; k < cfg->n_keys && same; k++)
                same = cmp_cell(inner_cols[cfg->build_key_idx[k]], inner[g1], inner_cols[cfg->build_key_idx[k]], inner[g0]) == 0;
            if (!same) break;
            g1++;
        }
    };
    next_group();
    for (int64_t o = 0; o < n_outer;) {  // joinToChunk (:257-310)
        int c = -1;
        if (selected[o] && g0 < inner.size()) c = cmp_keys(o, inner[g0]);
        if (c > 0) { next_group(); continue; }
        if (c < 0) { on_miss(o); o++; continue; }
        if (cfg->n_other_conds == 0) {
            for (size_t g = g0; g < g1; g++) emit(o, inner[g]);
        } else {
            const std::vector<int64_t> group(inner.begin() + (std::ptrdiff_t)g0, inner.begin() + (std::ptrdiff_t)g1);
            std::vector<uint8_t> sel2;
            const tsq_status st = join_conds_select(cfg, outer_cols, inner_cols, probe_off, build_off, o, group, sel2);
            if (st != TSQ_OK) { *status = st; delete res; return nullptr; }
            bool hasMatch = false;
            for (size_t r = 0; r < group.size(); r++)
                if (sel2[r]) { emit(o, group[r]); hasMatch = true; }
            if (!hasMatch) on_miss(o);
        }
        o++;
    }
    return res;
}

// build once (single build thread), then one timed probe pass per entry of threads[]: the CPU baseline of bench.py at the reference's
// concurrencies (executor/benchmark_test.go:357 runs 4 workers, tidb_hash_join_concurrency defaults to 5, sessionctx/variable/tidb_vars.go:249)
int64_t orc_hash_join_timed_multi(const tsq_join_cfg* cfg, const tsq_col* build_cols, int64_t n_build, const tsq_col* probe_cols, int64_t n_probe,
                                  const int32_t* thread_counts, int32_t n_runs, double* build_ms, double* probe_ms, uint64_t* sum_out, uint64_t* xor_out) {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    RowHashMap map((size_t)std::max<int64_t>(0, cfg->est_build_rows / 8));
    JoinState js{cfg, {build_cols, cfg->n_build_cols, cfg->build_key_idx},
                 {probe_cols, cfg->n_probe_cols, cfg->probe_key_idx}, n_build, &map};
    build_table(js);  // single build thread (join.go:148, "Main Thread" in proj5-part2 README)
    auto t1 = clk::now();
    if (build_ms) *build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();

    const int32_t nb = cfg->n_build_cols, np = cfg->n_probe_cols;
    const int32_t chunk = cfg->max_chunk_size > 0 ? cfg->max_chunk_size : 1024;
    const int64_t n_chunks = (n_probe + chunk - 1) / chunk;
    int64_t total = 0;
    for (int32_t run = 0; run < n_runs; run++) {
        const int32_t threads = thread_counts[run] > 0 ? thread_counts[run] : 1;
        auto tp0 = clk::now();
        std::atomic<int64_t> next_chunk{0};
        std::vector<int64_t> counts(threads, 0);
        std::vector<uint64_t> sums(threads, 0), xors(threads, 0);
        auto worker = [&](int32_t id) {  // runJoinWorker (join.go:243 STUB; intended loop per README)
            std::vector<uint64_t> ptrs;
            // per-worker result chunk, rows appended one at a time then handed over (dropped) when full
            std::vector<std::vector<uint64_t>> rchk(nb + np, std::vector<uint64_t>(chunk));
            int64_t fill = 0, cnt = 0;
            uint64_t s = 0, x = 0;
            for (;;) {
                int64_t c = next_chunk.fetch_add(1);  // fetchOuterSideChunks hands out chunks (join.go:194-221)
                if (c >= n_chunks) break;
                int64_t lo = c * chunk, hi = std::min<int64_t>(n_probe, lo + chunk);
                for (int64_t i = lo; i < hi; i++) {
                    bool hasNull;
                    uint64_t h = hash_row_keys(js.probe, cfg->n_keys, i, hasNull);
                    if (hasNull) continue;
                    map.Get(h, ptrs);
                    for (uint64_t p : ptrs) {
                        if (!equal_row_keys(js.build, (int64_t)p, js.probe, i, cfg->n_keys)) continue;
                        uint64_t rh = ROWHASH_SEED;
                        for (int32_t cc = 0; cc < np; cc++) {  // AppendRow / AppendPartialRow (chunk.go:334-356)
                            bool isnull = col_is_null(probe_cols[cc], i);
                            uint64_t v = isnull ? 0 : col_raw64(probe_cols[cc], i);
                            rchk[cc][fill] = v;
                            rh = rowhash_step(rh, isnull ? ROWHASH_NULL : v, (uint32_t)cc);
                        }
                        for (int32_t cc = 0; cc < nb; cc++) {
                            bool isnull = col_is_null(build_cols[cc], (int64_t)p);
                            uint64_t v = isnull ? 0 : col_raw64(build_cols[cc], (int64_t)p);
                            rchk[np + cc][fill] = v;
                            rh = rowhash_step(rh, isnull ? ROWHASH_NULL : v, (uint32_t)(np + cc));
                        }
                        s += rh;
                        x ^= rh;
                        cnt++;
                        if (++fill == chunk) fill = 0;  // IsFull -> joinResultCh <- chk (join.go:311-318)
                    }
                }
            }
            counts[id] = cnt;
            sums[id] = s;
            xors[id] = x;
        };
        std::vector<std::thread> th;
        for (int32_t t = 0; t < threads; t++) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
        auto tp1 = clk::now();
        if (probe_ms) probe_ms[run] = std::chrono::duration<double, std::milli>(tp1 - tp0).count();
        total = 0;
        uint64_t s = 0, x = 0;
        for (int32_t t = 0; t < threads; t++) {
            total += counts[t];
            s += sums[t];
            x ^= xors[t];
        }
        if (sum_out) *sum_out = s;
        if (xor_out) *xor_out = x;
    }
    return total;
}

int64_t orc_hash_join_timed(const tsq_join_cfg* cfg, const tsq_col* build_cols, int64_t n_build,
                            const tsq_col* probe_cols, int64_t n_probe, int32_t threads, double* build_ms,
                            double* probe_ms, uint64_t* sum_out, uint64_t* xor_out) {
    return orc_hash_join_timed_multi(cfg, build_cols, n_build, probe_cols, n_probe, &threads, 1, build_ms, probe_ms, sum_out, xor_out);
}

tsq_status orc_expr_eval(const tsq_expr_prog* prog, const tsq_col* cols, int32_t n_cols, int64_t nrows,
                         const int32_t* sel, void* out_data, uint8_t* out_notnull, int64_t* div_by_zero_warnings) {
    EvalCtx cx{cols, n_cols, nrows, sel};
    ECol r;
    tsq_status s = eval_prog(*prog, cx, r);
    if (div_by_zero_warnings) *div_by_zero_warnings = cx.warnings;
    if (s != TSQ_OK) return s;
    for (int64_t i = 0; i < nrows; i++) {
        if (out_notnull) out_notnull[i] = r.null[i] ? 0 : 1;
        if (r.real) ((double*)out_data)[i] = r.null[i] ? 0.0 : r.f[i];
        else ((int64_t*)out_data)[i] = r.null[i] ? 0 : r.i[i];
    }
    return TSQ_OK;
}

tsq_status orc_expr_eval_str(const tsq_expr_prog* prog, const tsq_col* cols, int32_t n_cols, int64_t nrows, const int32_t* sel,
                             int64_t* out_offsets, uint8_t* out_data, int64_t cap_bytes, uint8_t* out_notnull, int64_t* bytes_out,
                             int64_t* div_by_zero_warnings) {
    EvalCtx cx{cols, n_cols, nrows, sel};
    ECol r;
    tsq_status s = eval_prog(*prog, cx, r);
    if (div_by_zero_warnings) *div_by_zero_warnings = cx.warnings;
    if (s != TSQ_OK) return s;
    if (!r.str) { g_err = "the root of the program is not string-valued"; return TSQ_ERR_INVALID; }
    // Column.AppendNull of a var-len column repeats the last offset (no bytes), AppendString appends the bytes
    // (util/chunk/column.go:150-158, 220-226)
    int64_t pos = 0;
    out_offsets[0] = 0;
    for (int64_t i = 0; i < nrows; i++) {
        if (out_notnull) out_notnull[i] = r.null[i] ? 0 : 1;
        if (!r.null[i]) {
            const std::string& v = r.s[i];
            for (size_t b = 0; b < v.size(); b++)
                if (pos + (int64_t)b < cap_bytes) out_data[pos + b] = (uint8_t)v[b];
            pos += (int64_t)v.size();
        }
        out_offsets[i + 1] = pos;
    }
    if (bytes_out) *bytes_out = pos;
    return TSQ_OK;
}

tsq_status orc_filter_eval(const tsq_expr_prog* progs, int32_t n_progs, const tsq_col* cols, int32_t n_cols,
                           int64_t nrows, const int32_t* sel, uint8_t* selected_out, uint8_t* isnull_out,
                           int64_t* div_by_zero_warnings) {
    std::vector<uint8_t> selected, nulls;
    int64_t w = 0;
    tsq_status s = vec_eval_bool(progs, n_progs, cols, n_cols, nrows, sel, selected, nulls, &w);
    if (div_by_zero_warnings) *div_by_zero_warnings = w;
    if (s != TSQ_OK) return s;
    memcpy(selected_out, selected.data(), nrows);
    if (isnull_out) memcpy(isnull_out, nulls.data(), nrows);
    return TSQ_OK;
}

}  // extern "C"

// ====================================================================== hash aggregation
namespace {

// one aggregate function's partial result (aggfuncs/*.go partialResult4* structs)
struct Partial {
    int64_t i = 0;       // count / int sum / int max-min / first_row int bits
    double f = 0;        // float sum / float max-min
    int64_t count = 0;   // AVG count
    bool isNull = true;  // SUM/MAX/MIN: no non-NULL input yet;  FIRST_ROW: value is NULL
    bool got = false;    // FIRST_ROW gotFirstRow
    std::string s;       // partialResult4MaxMinString.val / partialResult4FirstRowString.val (deep copies: stringutil.Copy)
};

struct AggDesc {
    tsq_agg_func f;
    bool is_real;  // value domain real (F32/F64) vs int
    bool is_str;   // ... or string (maxMin4String, firstRow4String, countOriginal4String)
};

inline std::string col_string(const tsq_col& c, int64_t i) {  // Column.GetString (util/chunk/column.go)
    return std::string((const char*)c.data + c.offsets[i], (size_t)(c.offsets[i + 1] - c.offsets[i]));
}

inline bool add_int64_overflow(int64_t a, int64_t b) {  // types/overflow.go:33-40 AddInt64
    return (a > 0 && b > 0 && I64MAX - a < b) || (a < 0 && b < 0 && I64MIN - a > b);
}

// UpdatePartialResult for one input row
tsq_status agg_update(const AggDesc& d, const tsq_col* cols, int64_t row, Partial& p) {
    const tsq_agg_func& f = d.f;
    const bool merge_mode = f.mode == TSQ_MODE_FINAL || f.mode == TSQ_MODE_PARTIAL2;
    const bool arg_null = f.arg_col >= 0 ? col_is_null(cols[f.arg_col], row) : false;
    switch (f.func) {
        case TSQ_AGG_COUNT:
            if (merge_mode) {  // countPartial (func_count.go:99-113): += arg
                if (arg_null) return TSQ_OK;
                p.i += col_i64(cols[f.arg_col], row);
            } else {  // countOriginal4* (func_count.go:33-97)
                if (arg_null) return TSQ_OK;
                p.i++;
            }
            return TSQ_OK;
        case TSQ_AGG_SUM:  // func_sum.go:60-78,117-139 (partial sums use the same code)
            if (arg_null) return TSQ_OK;
            if (d.is_real) {
                double v = col_f64(cols[f.arg_col], row);
                if (p.isNull) { p.f = v; p.isNull = false; }
                else p.f += v;
            } else {
                int64_t v = col_i64(cols[f.arg_col], row);
                if (p.isNull) { p.i = v; p.isNull = false; }
                else {
                    if (add_int64_overflow(p.i, v)) return TSQ_ERR_OVERFLOW_BIGINT;
                    p.i += v;
                }
            }
            return TSQ_OK;
        case TSQ_AGG_AVG:
            if (merge_mode) {  // avgPartial4* (func_avg.go:86-113,186-209): args = (count, sum)
                if (col_is_null(cols[f.arg_col2], row)) return TSQ_OK;
                if (arg_null) return TSQ_OK;
                int64_t c = col_i64(cols[f.arg_col], row);
                if (d.is_real) p.f += col_f64(cols[f.arg_col2], row);
                else {
                    int64_t s = col_i64(cols[f.arg_col2], row);
                    if (add_int64_overflow(p.i, s)) return TSQ_ERR_OVERFLOW_BIGINT;
                    p.i += s;
                }
                p.count += c;
            } else {  // avgOriginal4* (func_avg.go:62-82,164-180)
                if (arg_null) return TSQ_OK;
                if (d.is_real) p.f += col_f64(cols[f.arg_col], row);
                else {
                    int64_t v = col_i64(cols[f.arg_col], row);
                    if (add_int64_overflow(p.i, v)) return TSQ_ERR_OVERFLOW_BIGINT;
                    p.i += v;
                }
                p.count++;
            }
            return TSQ_OK;
        case TSQ_AGG_MAX:
        case TSQ_AGG_MIN: {  // func_max_min.go:81-103 (+uint/float variants)
            if (arg_null) return TSQ_OK;
            const bool isMax = f.func == TSQ_AGG_MAX;
            if (d.is_str) {  // maxMin4String.UpdatePartialResult (func_max_min.go:337-362)
                std::string v = col_string(cols[f.arg_col], row);
                if (p.isNull) { p.s = v; p.isNull = false; }
                else {
                    const int c = cmp_string(v, p.s);
                    if ((isMax && c == 1) || (!isMax && c == -1)) p.s = v;
                }
            } else if (d.is_real) {
                double v = col_f64(cols[f.arg_col], row);
                if (p.isNull) { p.f = v; p.isNull = false; }
                else if ((isMax && v > p.f) || (!isMax && v < p.f)) p.f = v;
            } else if (f.arg_type == TSQ_U64) {
                uint64_t v = (uint64_t)col_i64(cols[f.arg_col], row);
                if (p.isNull) { p.i = (int64_t)v; p.isNull = false; }
                else if ((isMax && v > (uint64_t)p.i) || (!isMax && v < (uint64_t)p.i)) p.i = (int64_t)v;
            } else {
                int64_t v = col_i64(cols[f.arg_col], row);
                if (p.isNull) { p.i = v; p.isNull = false; }
                else if ((isMax && v > p.i) || (!isMax && v < p.i)) p.i = v;
            }
            return TSQ_OK;
        }
        case TSQ_AGG_FIRSTROW:  // func_first_row.go:67-81
            if (p.got) return TSQ_OK;
            p.got = true;
            p.isNull = arg_null;
            if (d.is_str) { if (!arg_null) p.s = col_string(cols[f.arg_col], row); }  // firstRow4String (func_first_row.go:206-220)
            else if (d.is_real) p.f = arg_null ? 0 : col_f64(cols[f.arg_col], row);
            else p.i = arg_null ? 0 : col_i64(cols[f.arg_col], row);
            return TSQ_OK;
    }
    return TSQ_ERR_INVALID;
}

// MergePartialResult(src, dst)
tsq_status agg_merge(const AggDesc& d, const Partial& src, Partial& dst) {
    const tsq_agg_func& f = d.f;
    switch (f.func) {
        case TSQ_AGG_COUNT: dst.i += src.i; return TSQ_OK;  // func_count.go:115-119
        case TSQ_AGG_SUM:                                   // func_sum.go:80-88,141-154
            if (src.isNull) return TSQ_OK;
            if (d.is_real) dst.f += src.f;
            else {
                if (add_int64_overflow(src.i, dst.i)) return TSQ_ERR_OVERFLOW_BIGINT;
                dst.i += src.i;
            }
            dst.isNull = false;
            return TSQ_OK;
        case TSQ_AGG_AVG:  // func_avg.go:115-128,211-216
            if (d.is_real) { dst.f += src.f; dst.count += src.count; return TSQ_OK; }
            if (src.count == 0) return TSQ_OK;
            if (add_int64_overflow(src.i, dst.i)) return TSQ_ERR_OVERFLOW_BIGINT;
            dst.i += src.i;
            dst.count += src.count;
            return TSQ_OK;
        case TSQ_AGG_MAX:
        case TSQ_AGG_MIN: {  // func_max_min.go:105-117
            if (src.isNull) return TSQ_OK;
            if (dst.isNull) { dst = src; return TSQ_OK; }
            const bool isMax = f.func == TSQ_AGG_MAX;
            if (d.is_str) {  // maxMin4String.MergePartialResult (func_max_min.go:364-378)
                const int c = cmp_string(src.s, dst.s);
                if ((isMax && c > 0) || (!isMax && c < 0)) dst.s = src.s;
            } else if (d.is_real) { if ((isMax && src.f > dst.f) || (!isMax && src.f < dst.f)) dst.f = src.f; }
            else if (f.arg_type == TSQ_U64) {
                if ((isMax && (uint64_t)src.i > (uint64_t)dst.i) || (!isMax && (uint64_t)src.i < (uint64_t)dst.i)) dst.i = src.i;
            } else if ((isMax && src.i > dst.i) || (!isMax && src.i < dst.i)) dst.i = src.i;
            return TSQ_OK;
        }
        case TSQ_AGG_FIRSTROW:  // func_first_row.go:83-89
            if (!dst.got) dst = src;
            return TSQ_OK;
    }
    return TSQ_ERR_INVALID;
}

using GroupMap = std::unordered_map<std::string, std::vector<Partial>>;

std::vector<AggDesc> make_descs(const tsq_agg_cfg* cfg) {
    std::vector<AggDesc> ds(cfg->n_aggs);
    for (int32_t a = 0; a < cfg->n_aggs; a++) {
        ds[a].f = cfg->aggs[a];
        ds[a].is_real = cfg->aggs[a].arg_type == TSQ_F32 || cfg->aggs[a].arg_type == TSQ_F64;
        ds[a].is_str = cfg->aggs[a].arg_type == TSQ_BYTES;
    }
    return ds;
}

std::vector<Partial> alloc_partials(const std::vector<AggDesc>& ds) {
    std::vector<Partial> ps(ds.size());
    for (size_t a = 0; a < ds.size(); a++) {
        // AllocPartialResult: SUM/MAX/MIN start isNull=true; COUNT/AVG zero; FIRST_ROW got=false,isNull=false
        Partial& p = ps[a];
        p.isNull = ds[a].f.func == TSQ_AGG_SUM || ds[a].f.func == TSQ_AGG_MAX || ds[a].f.func == TSQ_AGG_MIN;
    }
    return ps;
}

// partial worker: updatePartialResult over rows [lo,hi) (aggregate.go:332-350)
tsq_status partial_update(const tsq_agg_cfg* cfg, const std::vector<AggDesc>& ds, const tsq_col* cols, int64_t lo,
                          int64_t hi, GroupMap& m) {
    uint8_t kb[16];
    std::string key;
    for (int64_t r = lo; r < hi; r++) {
        key.clear();  // getGroupKey (aggregate.go:359-394)
        for (int32_t g = 0; g < cfg->n_group_keys; g++) {
            const tsq_col& kc = cols[cfg->group_key_col[g]];
            if (kc.type == TSQ_BYTES && !col_is_null(kc, r)) {  // encodeBytes(.., comparable=false): compactBytesFlag + varint(len) + bytes
                                                              // (codec.go:738-744, bytes.go:144-148)
                const int64_t n = kc.offsets[r + 1] - kc.offsets[r];
                uint64_t ux = ((uint64_t)n << 1) ^ (uint64_t)(n >> 63);
                key.push_back((char)compactBytesFlag);
                while (ux >= 0x80) {
                    key.push_back((char)((uint8_t)ux | 0x80));
                    ux >>= 7;
                }
                key.push_back((char)(uint8_t)ux);
                key.append((const char*)kc.data + kc.offsets[r], (size_t)n);
            } else {
                key.append((const char*)kb, (size_t)orc_group_key_encode(&kc, r, kb));
            }
        }
        auto it = m.find(key);  // getPartialResult (aggregate.go:396-410)
        if (it == m.end()) it = m.emplace(key, alloc_partials(ds)).first;
        for (size_t a = 0; a < ds.size(); a++) {
            tsq_status s = agg_update(ds[a], cols, r, it->second[a]);
            if (s != TSQ_OK) return s;
        }
    }
    return TSQ_OK;
}

// AppendFinalResult2Chunk for every group of one final map (aggregate.go:429-457)
void append_final(const tsq_agg_cfg* cfg, const std::vector<AggDesc>& ds, const GroupMap& m, orc_result* res) {
    for (auto& kv : m) {
        int32_t oc = 0;
        for (size_t a = 0; a < ds.size(); a++) {
            const Partial& p = kv.second[a];
            const tsq_agg_func& f = ds[a].f;
            const bool partial_out = f.mode == TSQ_MODE_PARTIAL1 || f.mode == TSQ_MODE_PARTIAL2;
            auto put_real = [&](double v, bool nn, int32_t t) {
                uint64_t bits = 0;
                if (t == TSQ_F32) { float fv = (float)v; uint32_t b32; memcpy(&b32, &fv, 4); bits = b32; }
                else memcpy(&bits, &v, 8);
                res->cols[oc].type = t;
                res->cols[oc++].append_raw(bits, nn);
            };
            auto put_int = [&](int64_t v, bool nn, int32_t t) {
                res->cols[oc].type = t;
                res->cols[oc++].append_raw((uint64_t)v, nn);
            };
            auto put_str = [&](const std::string& v, bool nn) {  // chk.AppendString / AppendNull (func_max_min.go:327-335)
                res->cols[oc].type = TSQ_BYTES;
                if (nn) res->cols[oc++].append_bytes(v.data(), v.size());
                else res->cols[oc++].append_raw(0, false);
            };
            switch (f.func) {
                case TSQ_AGG_COUNT: put_int(p.i, true, TSQ_I64); break;  // func_count.go:22-26
                case TSQ_AGG_SUM:                                        // func_sum.go:50-58,107-115
                    if (ds[a].is_real) put_real(p.f, !p.isNull, TSQ_F64);
                    else put_int(p.i, !p.isNull, TSQ_I64);
                    break;
                case TSQ_AGG_AVG:
                    if (partial_out) {  // partial AVG emits (count, sum) (descriptor.go:70-81)
                        put_int(p.count, true, TSQ_I64);
                        if (ds[a].is_real) put_real(p.f, true, TSQ_F64);
                        else put_int(p.i, true, TSQ_I64);
                    } else if (ds[a].is_real) {  // func_avg.go:154-162
                        put_real(p.count ? p.f / (double)p.count : 0, p.count != 0, TSQ_F64);
                    } else {  // func_avg.go:47-55: integer division, BIGINT
                        put_int(p.count ? go_div(p.i, p.count) : 0, p.count != 0, TSQ_I64);
                    }
                    break;
                case TSQ_AGG_MAX:
                case TSQ_AGG_MIN:
                    if (ds[a].is_str) put_str(p.s, !p.isNull);
                    else if (ds[a].is_real) put_real(p.f, !p.isNull, f.arg_type);
                    else put_int(p.i, !p.isNull, f.arg_type);
                    break;
                case TSQ_AGG_FIRSTROW:  // func_first_row.go:91-99
                    if (ds[a].is_str) put_str(p.s, !(p.isNull || !p.got));
                    else if (ds[a].is_real) put_real(p.f, !(p.isNull || !p.got), f.arg_type);
                    else put_int(p.i, !(p.isNull || !p.got), f.arg_type);
                    break;
            }
        }
        res->rows++;
    }
}

int32_t out_col_count(const tsq_agg_cfg* cfg) {
    int32_t n = 0;
    for (int32_t a = 0; a < cfg->n_aggs; a++) {
        const tsq_agg_func& f = cfg->aggs[a];
        bool partial_out = f.mode == TSQ_MODE_PARTIAL1 || f.mode == TSQ_MODE_PARTIAL2;
        n += (f.func == TSQ_AGG_AVG && partial_out) ? 2 : 1;
    }
    return n;
}

orc_result* agg_run(const tsq_agg_cfg* cfg, const tsq_col* cols, int64_t nrows, int32_t M, int32_t N, bool threaded,
                    double* ms, tsq_status* status) {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    *status = TSQ_OK;
    if (M < 1) M = 1;
    if (N < 1) N = 1;
    std::vector<AggDesc> ds = make_descs(cfg);
    const int32_t chunk = cfg->max_chunk_size > 0 ? cfg->max_chunk_size : 1024;
    const int64_t n_chunks = (nrows + chunk - 1) / chunk;
    std::vector<GroupMap> partial(M);
    std::vector<tsq_status> st(M, TSQ_OK);
    auto pworker = [&](int32_t w) {  // HashAggPartialWorker.run (aggregate.go:307-330): chunk c -> worker c % M
        for (int64_t c = w; c < n_chunks; c += M) {
            int64_t lo = c * chunk, hi = std::min<int64_t>(nrows, lo + chunk);
            tsq_status s = partial_update(cfg, ds, cols, lo, hi, partial[w]);
            if (s != TSQ_OK) { st[w] = s; return; }
        }
    };
    if (threaded) {
        std::vector<std::thread> th;
        for (int32_t w = 0; w < M; w++) th.emplace_back(pworker, w);
        for (auto& t : th) t.join();
    } else {
        for (int32_t w = 0; w < M; w++) pworker(w);
    }
    for (int32_t w = 0; w < M; w++)
        if (st[w] != TSQ_OK) { *status = st[w]; return nullptr; }

    // shuffleIntermData (aggregate.go:354 STUB; intended per proj5-part3 README): group key ->
    // final worker hash(groupKey) % finalConcurrency; consumeIntermData (:424 STUB): merge.
    std::vector<GroupMap> fin(N);
    std::vector<tsq_status> fst(N, TSQ_OK);
    auto fworker = [&](int32_t fw) {
        for (int32_t w = 0; w < M; w++) {
            for (auto& kv : partial[w]) {
                uint64_t h = fnv1_write(FNV_OFFSET, (const uint8_t*)kv.first.data(), (int64_t)kv.first.size());
                if ((int32_t)(h % (uint64_t)N) != fw) continue;
                auto it = fin[fw].find(kv.first);
                if (it == fin[fw].end()) it = fin[fw].emplace(kv.first, alloc_partials(ds)).first;
                for (size_t a = 0; a < ds.size(); a++) {
                    tsq_status s = agg_merge(ds[a], kv.second[a], it->second[a]);
                    if (s != TSQ_OK) { fst[fw] = s; return; }
                }
            }
        }
    };
    if (threaded) {
        std::vector<std::thread> th;
        for (int32_t w = 0; w < N; w++) th.emplace_back(fworker, w);
        for (auto& t : th) t.join();
    } else {
        for (int32_t w = 0; w < N; w++) fworker(w);
    }
    for (int32_t w = 0; w < N; w++)
        if (fst[w] != TSQ_OK) { *status = fst[w]; return nullptr; }

    orc_result* res = new orc_result();
    res->cols.resize(out_col_count(cfg));
    for (int32_t w = 0; w < N; w++) append_final(cfg, ds, fin[w], res);
    // empty input without GROUP BY: one row of defaults (aggregate.go:572-574, builder.go:517-539):
    // COUNT -> 0, everything else NULL.
    if (res->rows == 0 && cfg->n_group_keys == 0) {
        GroupMap one;
        one.emplace(std::string(), alloc_partials(ds));
        append_final(cfg, ds, one, res);
    }
    if (ms) *ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    return res;
}

}  // namespace

extern "C" {
orc_result* orc_hash_agg(const tsq_agg_cfg* cfg, const tsq_col* cols, int64_t nrows, int32_t partial_workers,
                         int32_t final_workers, tsq_status* status) {
    return agg_run(cfg, cols, nrows, partial_workers, final_workers, false, nullptr, status);
}
orc_result* orc_hash_agg_timed(const tsq_agg_cfg* cfg, const tsq_col* cols, int64_t nrows, int32_t threads, double* ms,
                               tsq_status* status) {
    return agg_run(cfg, cols, nrows, threads, threads, true, ms, status);
}
}
