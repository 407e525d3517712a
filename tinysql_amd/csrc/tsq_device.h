// tsq_device.h — scalar building blocks shared by every HIP kernel of libtsq.
//
// Everything here is TSQ_HD (__host__ __device__) and free of HIP runtime calls so that the
// exact same source can be compiled by g++ into tests/hostsim (a test-only shared object that
// checks the per-row semantics against the oracle without a GPU).  The product never uses a
// host build of these functions: libtsq's entry points launch HIP kernels or fail loudly.
#ifndef TSQ_DEVICE_H
#define TSQ_DEVICE_H

#include <stdint.h>
#include <string.h>

#include "../../include/tsq.h"

#if defined(__HIPCC__)
#define TSQ_HD __host__ __device__ __forceinline__
#else
#define TSQ_HD inline
#endif
// A value every lane of the wave agrees on (the expression program is the same for all rows): telling the
// compiler so turns the interpreter's switch into scalar branches instead of exec-masked regions for every case.
#if defined(TSQ_JIT)
// run-time specialisation (tsq_expr.hip): the program is a compile-time constant, the op loop unrolls and folds
#define TSQ_UNIFORM(x) ((int)(x))
#define TSQ_JIT_UNROLL _Pragma("unroll")
#elif defined(__HIP_DEVICE_COMPILE__)
#define TSQ_UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#define TSQ_JIT_UNROLL
#else
#define TSQ_UNIFORM(x) ((int)(x))
#define TSQ_JIT_UNROLL
#endif

// ------------------------------------------------------------------ hashing
// splitmix64 — the synthetic-table generator of SURVEY.md §8(d), also the row-checksum mixer.
TSQ_HD uint64_t tsq_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// Bucket hash of a 64-bit key word.  The reference hashes with FNV-1-64 (executor/hash_table.go:64)
// but the hash value never reaches a result (equality is by bytes, util/codec/codec.go:377), so the
// GPU is free to use a cheaper, better-mixing function.
TSQ_HD uint64_t tsq_mix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xFF51AFD7ED558CCDULL;
    k ^= k >> 33;
    k *= 0xC4CEB9FE1A85EC53ULL;
    k ^= k >> 33;
    return k;
}
// high 64 bits of a 64x64 multiply: range reduction of a hash to [0, n) without a power-of-two n
TSQ_HD uint64_t tsq_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// order-independent row checksum: sum and xor over rows of this per-row hash
#define TSQ_ROWHASH_SEED 0x243F6A8885A308D3ULL
#define TSQ_ROWHASH_NULL 0xA5A5A5A55A5A5A5AULL
TSQ_HD uint64_t tsq_rowhash_step(uint64_t h, uint64_t v, uint32_t c) {
    return tsq_splitmix64(h ^ (v + (uint64_t)(c + 1) * 0x9E3779B97F4A7C15ULL));
}

// hash of a var-len cell (string join keys and string group keys: equality is decided by the bytes, so any hash will do)
TSQ_HD uint64_t tsq_hash_bytes(const uint8_t* p, int64_t n) {
    uint64_t w = 0x9E3779B97F4A7C15ULL ^ (uint64_t)n;
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x;
        memcpy(&x, p + i, 8);
        w = tsq_splitmix64(w ^ x);
    }
    uint64_t tail = 0;
    for (int64_t q = i; q < n; q++) tail = (tail << 8) | p[q];
    return tsq_splitmix64(w ^ tail);
}

// one var-len cell copied by ONE lane: eight bytes at a time (global loads and stores take any byte address), then the tail
TSQ_HD void tsq_copy_cell(uint8_t* d, const uint8_t* s, int64_t n) {
    // (d and s never overlap: a cell is copied between two buffers.  No byte loop over memory: every byte load behind a byte store may
    // alias it, so a 7-byte tail was seven dependent round trips — round 6: the tail of a cell of 8 bytes or more is one more 8-byte copy
    // that overlaps the bytes before it, a shorter cell is gathered into a register first)
    if (n >= 8) {
        int64_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t x;
            memcpy(&x, s + i, 8);
            memcpy(d + i, &x, 8);
        }
        if (i < n) {
            uint64_t x;
            memcpy(&x, s + n - 8, 8);
            memcpy(d + n - 8, &x, 8);
        }
        return;
    }
    uint64_t x = 0;
    for (int64_t i = 0; i < n; i++) x |= (uint64_t)s[i] << (8 * i);
    for (int64_t i = 0; i < n; i++) d[i] = (uint8_t)(x >> (8 * i));
}

// rank of a key for the multi-GPU radix redistribute (tsq_radix_split)
TSQ_HD uint32_t tsq_key_rank(uint64_t kw, uint32_t n_parts) {
    return (uint32_t)(((tsq_mix64(kw) & 0xffffu) * (uint64_t)n_parts) >> 16);
}

// ------------------------------------------------------------------ bit patterns
TSQ_HD double tsq_bits_f64(uint64_t b) {
    double d;
    memcpy(&d, &b, 8);
    return d;
}
TSQ_HD uint64_t tsq_f64_bits(double d) {
    uint64_t b;
    memcpy(&b, &d, 8);
    return b;
}
TSQ_HD float tsq_bits_f32(uint32_t b) {
    float f;
    memcpy(&f, &b, 4);
    return f;
}

// util/chunk/column.go:89-92: bit == 1 => NOT NULL, LSB first; bitmap == nullptr => no NULLs
TSQ_HD bool tsq_is_null(const uint8_t* bm, int64_t i) {
    if (!bm) return false;
    return ((bm[i >> 3] >> (i & 7)) & 1) == 0;
}

// ------------------------------------------------------------------ synthetic tables
TSQ_HD uint64_t tsq_gen_r(uint64_t seed, uint32_t table, uint64_t c, uint64_t i) {
    return tsq_splitmix64(seed ^ ((uint64_t)table << 56) ^ (c << 48) ^ i);
}
TSQ_HD bool tsq_gen_is_null(const tsq_gen_spec& s, uint64_t i) {
    return s.null_pct > 0 && (tsq_gen_r(s.seed, (uint32_t)s.table, 7, i) % 100) < (uint64_t)s.null_pct;
}
TSQ_HD uint64_t tsq_gen_value(const tsq_gen_spec& s, uint64_t i, uint64_t src) {
    switch (s.kind) {
        case TSQ_GEN_SEQ: return i;
        case TSQ_GEN_AFFINE: return (s.a * (i % s.m) + s.b) % s.m;
        case TSQ_GEN_RAND_MOD: return tsq_gen_r(s.seed, (uint32_t)s.table, (uint64_t)s.col, i) % s.m;
        case TSQ_GEN_RAND_F64:
            return tsq_f64_bits((double)(tsq_gen_r(s.seed, (uint32_t)s.table, (uint64_t)s.col, i) >> 11) *
                                (1.0 / 9007199254740992.0));
        case TSQ_GEN_HASH_OF_COL: return tsq_splitmix64(src ^ s.b);
        case TSQ_GEN_ZIPF_OCT: {
            const uint64_t r = tsq_gen_r(s.seed, (uint32_t)s.table, (uint64_t)s.col, i);
            const uint64_t oct = s.a ? (s.a < 63 ? s.a : 63) : 1;  // octaves: a shift of 64 or more is undefined
            const uint64_t lo = 1ull << (r % oct);
            return (lo + (tsq_splitmix64(r) & (lo - 1)) - 1) % s.m;
        }
    }
    return 0;
}

// ------------------------------------------------------------------ join key words
// A join key cell becomes (class flag, 64-bit word) exactly like
// util/codec/codec.go:212-240 encodeHashChunkRowIdx: ints -> raw 8 bytes with flag 8, or 9 when the
// column is UNSIGNED and the value has its top bit set; float32 widened to float64, flag 5.
// Two cells are equal iff flags and words are equal (codec.go:363-382).
#define TSQ_FLAG_INT 8
#define TSQ_FLAG_UINT 9
#define TSQ_FLAG_FLOAT 5
TSQ_HD uint64_t tsq_key_word(const void* data, int32_t type, int64_t row, uint32_t* flag) {
    switch (type) {
        case TSQ_I64: {
            *flag = TSQ_FLAG_INT;
            return ((const uint64_t*)data)[row];
        }
        case TSQ_U64: {
            uint64_t v = ((const uint64_t*)data)[row];
            *flag = (v >> 63) ? TSQ_FLAG_UINT : TSQ_FLAG_INT;
            return v;
        }
        case TSQ_F32: {
            *flag = TSQ_FLAG_FLOAT;
            return tsq_f64_bits((double)((const float*)data)[row]);
        }
        default: {
            *flag = TSQ_FLAG_FLOAT;
            return ((const uint64_t*)data)[row];
        }
    }
}

// ------------------------------------------------------------------ expression interpreter
// One row through one postfix program.  Each opcode restates one builtin*Sig.vecEval* of the
// reference for a single row; see include/tsq.h for the file:line of every opcode.
struct tsq_val {
    int64_t v;   // int64 or the bits of a double
    bool null;
};

// Row sources for the interpreter: the caller decides how (column, logical row) maps to storage.
// tsq_chunk_src  : one chunk, one physical row (chunk.sel already applied by the caller);
// tsq_joined_src : a joined row lhs||rhs living in two chunks (joiner.go:145-150 column order).
struct tsq_colset {  // kernel-argument friendly description of a chunk's columns
    const void* data[TSQ_MAX_COLS];
    const uint8_t* nulls[TSQ_MAX_COLS];
    const int64_t* offs[TSQ_MAX_COLS];  // TSQ_BYTES columns: offsets[n + 1] into data (util/chunk/column.go:28-34)
    int32_t type[TSQ_MAX_COLS];
    int32_t n;
};
// A string VALUE on the evaluation stack is a reference, no bytes move: [63:56] source (column index, or TSQ_STR_POOL = the
// program's constant pool), [55:32] length (< 2^24), [31:0] byte offset into the source's data (< 2^32).
#define TSQ_STR_POOL 0xffu
#define TSQ_STR_MAXLEN 0xffffffu
TSQ_HD uint64_t tsq_str_ref(uint32_t src, uint64_t off, uint64_t len) { return ((uint64_t)src << 56) | (len << 32) | off; }
TSQ_HD uint32_t tsq_str_len(uint64_t h) { return (uint32_t)(h >> 32) & TSQ_STR_MAXLEN; }
// types.CompareString with the binary collation = bytes.Compare (types/compare.go:136-139): -1 / 0 / 1
TSQ_HD int tsq_cmp_bytes(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
    const uint32_t n = la < lb ? la : lb;
    for (uint32_t i = 0; i < n; i++) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return la < lb ? -1 : (la == lb ? 0 : 1);
}
// a TSQ_BYTES cell as a string reference; *bad is set when the cell does not fit the reference (TSQ_ERR_UNSUPPORTED)
TSQ_HD uint64_t tsq_cell_str(const tsq_colset& cs, int c, int gc, int64_t row, bool* isnull, bool* bad) {
    *isnull = tsq_is_null(cs.nulls[c], row);
    const int64_t o0 = cs.offs[c][row], o1 = cs.offs[c][row + 1];
    if ((uint64_t)o0 >> 32 || (uint64_t)(o1 - o0) > TSQ_STR_MAXLEN) *bad = true;
    return tsq_str_ref((uint32_t)gc, (uint64_t)o0 & 0xffffffffu, (uint64_t)(o1 - o0) & TSQ_STR_MAXLEN);
}
TSQ_HD tsq_val tsq_cell_int(const tsq_colset& cs, int c, int64_t row) {
    tsq_val r;
    r.null = tsq_is_null(cs.nulls[c], row);
    r.v = ((const int64_t*)cs.data[c])[row];
    return r;
}
TSQ_HD tsq_val tsq_cell_real(const tsq_colset& cs, int c, int64_t row) {
    tsq_val r;
    r.null = tsq_is_null(cs.nulls[c], row);
    if (cs.type[c] == TSQ_F32) r.v = (int64_t)tsq_f64_bits((double)((const float*)cs.data[c])[row]);
    else r.v = ((const int64_t*)cs.data[c])[row];
    return r;
}
// raw 64-bit image of a cell (F32 zero-extended) — what the row checksum and the gather use
TSQ_HD uint64_t tsq_cell_raw(const tsq_colset& cs, int c, int64_t row) {
    if (cs.type[c] == TSQ_F32) return ((const uint32_t*)cs.data[c])[row];
    return ((const uint64_t*)cs.data[c])[row];
}
struct tsq_chunk_src {
    const tsq_colset* cs;
    int64_t row;
    TSQ_HD tsq_val load_int(int c) const { return tsq_cell_int(*cs, c, row); }
    TSQ_HD tsq_val load_real(int c) const { return tsq_cell_real(*cs, c, row); }
    TSQ_HD tsq_val load_str(int c, bool* bad) const {
        tsq_val r;
        r.v = (int64_t)tsq_cell_str(*cs, c, c, row, &r.null, bad);
        return r;
    }
    TSQ_HD const uint8_t* str_base(uint32_t src) const { return (const uint8_t*)cs->data[src]; }
};
struct tsq_joined_src {
    const tsq_colset* left;
    const tsq_colset* right;
    int64_t lrow, rrow;
    TSQ_HD tsq_val load_int(int c) const {
        return c < left->n ? tsq_cell_int(*left, c, lrow) : tsq_cell_int(*right, c - left->n, rrow);
    }
    TSQ_HD tsq_val load_real(int c) const {
        return c < left->n ? tsq_cell_real(*left, c, lrow) : tsq_cell_real(*right, c - left->n, rrow);
    }
    TSQ_HD tsq_val load_str(int c, bool* bad) const {
        tsq_val r;
        r.v = (int64_t)(c < left->n ? tsq_cell_str(*left, c, c, lrow, &r.null, bad) : tsq_cell_str(*right, c - left->n, c, rrow, &r.null, bad));
        return r;
    }
    TSQ_HD const uint8_t* str_base(uint32_t src) const {
        return (const uint8_t*)((int)src < left->n ? left->data[src] : right->data[src - left->n]);
    }
};

#define TSQ_I64MAX 0x7fffffffffffffffLL
#define TSQ_I64MIN (-TSQ_I64MAX - 1)
#define TSQ_U64MAX 0xffffffffffffffffULL
#define TSQ_F64MAX 1.7976931348623157e308

TSQ_HD int64_t tsq_wneg(int64_t v) { return (int64_t)(0 - (uint64_t)v); }
TSQ_HD int64_t tsq_godiv(int64_t a, int64_t b) { return b == -1 ? tsq_wneg(a) : a / b; }
TSQ_HD bool tsq_isinf(double d) { return d > TSQ_F64MAX || d < -TSQ_F64MAX; }

// types/compare.go:44-101
TSQ_HD int tsq_cmp_int(int64_t x, int64_t y, bool ux, bool uy) {
    if (ux && uy) {
        uint64_t a = (uint64_t)x, b = (uint64_t)y;
        return a < b ? -1 : (a == b ? 0 : 1);
    }
    if (ux && !uy) {
        if (y < 0 || (uint64_t)x > (uint64_t)TSQ_I64MAX) return 1;
        return x < y ? -1 : (x == y ? 0 : 1);
    }
    if (!ux && uy) {
        if (x < 0 || (uint64_t)y > (uint64_t)TSQ_I64MAX) return -1;
        return x < y ? -1 : (x == y ? 0 : 1);
    }
    return x < y ? -1 : (x == y ? 0 : 1);
}
TSQ_HD int tsq_cmp_real(double x, double y) { return x < y ? -1 : (x == y ? 0 : 1); }  // compare.go:104
// types/helper.go:28 RoundFloat(f) == 0  <=>  |f| < 0.5 (and NaN is "non zero")
TSQ_HD bool tsq_real_is_zero(double f) {
    double a = f < 0 ? -f : f;
    return a < 0.5;
}

// Evaluates `p` for one row.  Returns TSQ_OK or an overflow status; *err_node receives the
// postfix index of the offending node.  *div0 is incremented per x/0 (errors.go:65-77 warning).
template <class Src>
TSQ_HD tsq_status tsq_eval_row(const tsq_expr_prog& p, const Src& src, tsq_val* out, int* err_node, int* div0) {
    // Evaluation stack with the top two entries cached in registers (r0 = top, r1 = below it); only deeper entries
    // touch `mem`, a dynamically indexed array that the GPU compiler has to place in scratch memory.  With the whole
    // stack in that array a 5-node expression ran at 7 % of the HBM roofline (profiles/r01_kernels.txt).
    tsq_val mem[TSQ_EXPR_MAX_STACK];
    tsq_val r0, r1;
    r0.v = r1.v = 0;
    r0.null = r1.null = true;
    int sp = 0;
    auto push = [&](const tsq_val& x) {
        if (sp >= 2) mem[sp - 2] = r1;
        r1 = r0;
        r0 = x;
        sp++;
    };
    auto pop = [&]() -> tsq_val {
        const tsq_val x = r0;
        r0 = r1;
        if (sp >= 3) r1 = mem[sp - 3];
        sp--;
        return x;
    };
    const int n_ops = TSQ_UNIFORM(p.n_ops);
    TSQ_JIT_UNROLL
    for (int k = 0; k < n_ops; k++) {
        tsq_expr_op op;
        op.opcode = TSQ_UNIFORM(p.ops[k].opcode);
        op.flags = TSQ_UNIFORM(p.ops[k].flags);
        op.arg = TSQ_UNIFORM(p.ops[k].arg);
        op.aux = (decltype(op.aux))TSQ_UNIFORM(p.ops[k].aux);
        const bool ul = op.flags & TSQ_F_LHS_UNSIGNED, ur = op.flags & TSQ_F_RHS_UNSIGNED;
        *err_node = k;
        switch (op.opcode) {
            case TSQ_OP_COL_INT: push(src.load_int(op.arg)); break;
            case TSQ_OP_COL_REAL: push(src.load_real(op.arg)); break;
            case TSQ_OP_CONST_INT:
            case TSQ_OP_CONST_REAL: { tsq_val c; c.v = p.consts[op.arg]; c.null = false; push(c); break; }
            case TSQ_OP_CONST_NULL_INT:
            case TSQ_OP_CONST_NULL_REAL: { tsq_val c; c.v = 0; c.null = true; push(c); break; }
            case TSQ_OP_PLUS_REAL:
            case TSQ_OP_MINUS_REAL:
            case TSQ_OP_MUL_REAL:
            case TSQ_OP_DIV_REAL: {
                tsq_val b = pop();
                tsq_val& a = r0;
                a.null = a.null || b.null;
                if (a.null) break;
                double x = tsq_bits_f64((uint64_t)a.v), y = tsq_bits_f64((uint64_t)b.v), r;
                if (op.opcode == TSQ_OP_PLUS_REAL) {
                    if ((x > 0 && y > TSQ_F64MAX - x) || (x < 0 && y < -TSQ_F64MAX - x)) return TSQ_ERR_OVERFLOW_DOUBLE;
                    r = x + y;
                } else if (op.opcode == TSQ_OP_MINUS_REAL) {
                    if ((x > 0 && -y > TSQ_F64MAX - x) || (x < 0 && -y < -TSQ_F64MAX - x)) return TSQ_ERR_OVERFLOW_DOUBLE;
                    r = x - y;
                } else if (op.opcode == TSQ_OP_MUL_REAL) {
                    r = x * y;
                    if (tsq_isinf(r)) return TSQ_ERR_OVERFLOW_DOUBLE;
                } else {
                    if (y == 0) {
                        (*div0)++;
                        a.null = true;
                        break;
                    }
                    r = x / y;
                    if (tsq_isinf(r)) return TSQ_ERR_OVERFLOW_DOUBLE;
                }
                a.v = (int64_t)tsq_f64_bits(r);
                break;
            }
            case TSQ_OP_PLUS_INT: {
                tsq_val b = pop();
                tsq_val& a = r0;
                a.null = a.null || b.null;
                if (a.null) break;
                int64_t lh = a.v, rh = b.v;
                if (ul && ur) {
                    if ((uint64_t)lh > TSQ_U64MAX - (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                } else if (ul && !ur) {  // plusUS incl. the reference's lh-vs-lh check (:454)
                    if (rh < 0 && (uint64_t)tsq_wneg(rh) > (uint64_t)lh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    if (rh > 0 && (uint64_t)lh > TSQ_U64MAX - (uint64_t)lh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                } else if (!ul && ur) {
                    if (lh < 0 && (uint64_t)tsq_wneg(lh) > (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    if (lh > 0 && (uint64_t)rh > TSQ_U64MAX - (uint64_t)lh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                } else {
                    if ((lh > 0 && rh > TSQ_I64MAX - lh) || (lh < 0 && rh < TSQ_I64MIN - lh)) return TSQ_ERR_OVERFLOW_BIGINT;
                }
                a.v = (int64_t)((uint64_t)lh + (uint64_t)rh);
                break;
            }
            case TSQ_OP_MINUS_INT: {
                tsq_val b = pop();
                tsq_val& a = r0;
                a.null = a.null || b.null;
                if (a.null) break;
                const bool force = op.flags & TSQ_F_FORCE_SIGNED;
                int64_t lh = a.v, rh = b.v;
                const int64_t nrh = tsq_wneg(rh);
                const bool ss_over = (lh > 0 && nrh > TSQ_I64MAX - lh) || (lh < 0 && nrh < TSQ_I64MIN - lh);
                if (force && ul && ur) {
                    if (lh < 0 || rh < 0) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    if (ss_over) return TSQ_ERR_OVERFLOW_BIGINT;
                } else if (force && ul && !ur) {
                    if (lh < 0) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    if (ss_over) return TSQ_ERR_OVERFLOW_BIGINT;
                } else if (force && !ul && ur) {
                    if (rh < 0) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    if (ss_over) return TSQ_ERR_OVERFLOW_BIGINT;
                } else if (!force && ul && ur) {
                    if ((uint64_t)lh < (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                } else if (!force && ul && !ur) {
                    if (rh >= 0 && (uint64_t)lh < (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                    if (rh < 0 && (uint64_t)lh > TSQ_U64MAX - (uint64_t)nrh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                } else if (!force && !ul && ur) {
                    if (((uint64_t)lh - (uint64_t)TSQ_I64MIN) < (uint64_t)rh) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                } else {
                    if (ss_over) return TSQ_ERR_OVERFLOW_BIGINT;
                }
                a.v = (int64_t)((uint64_t)lh - (uint64_t)rh);
                break;
            }
            case TSQ_OP_MUL_INT: {
                tsq_val b = pop();
                tsq_val& a = r0;
                a.null = a.null || b.null;
                if (a.null) break;
                int64_t x = a.v, y = b.v;
                int64_t tmp = (int64_t)((uint64_t)x * (uint64_t)y);
                if (x != 0 && tsq_godiv(tmp, x) != y) return TSQ_ERR_OVERFLOW_BIGINT;
                a.v = tmp;
                break;
            }
            case TSQ_OP_MUL_INT_UNSIGNED: {
                tsq_val b = pop();
                tsq_val& a = r0;
                a.null = a.null || b.null;
                if (a.null) break;
                uint64_t x = (uint64_t)a.v, y = (uint64_t)b.v, res = x * y;
                if (x != 0 && res / x != y) return TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED;
                a.v = (int64_t)res;
                break;
            }
            case TSQ_OP_LT_INT: case TSQ_OP_LE_INT: case TSQ_OP_GT_INT:
            case TSQ_OP_GE_INT: case TSQ_OP_EQ_INT: case TSQ_OP_NE_INT:
            case TSQ_OP_LT_REAL: case TSQ_OP_LE_REAL: case TSQ_OP_GT_REAL:
            case TSQ_OP_GE_REAL: case TSQ_OP_EQ_REAL: case TSQ_OP_NE_REAL: {
                tsq_val b = pop();
                tsq_val& a = r0;
                a.null = a.null || b.null;
                if (a.null) { a.v = 0; break; }
                const bool isreal = op.opcode >= TSQ_OP_LT_REAL;
                const int rel = isreal ? op.opcode - TSQ_OP_LT_REAL : op.opcode - TSQ_OP_LT_INT;
                int c = isreal ? tsq_cmp_real(tsq_bits_f64((uint64_t)a.v), tsq_bits_f64((uint64_t)b.v))
                               : tsq_cmp_int(a.v, b.v, ul, ur);
                bool v = rel == 0 ? c < 0 : rel == 1 ? c <= 0 : rel == 2 ? c > 0 : rel == 3 ? c >= 0 : rel == 4 ? c == 0 : c != 0;
                a.v = v ? 1 : 0;
                break;
            }
            case TSQ_OP_LOGIC_AND: {
                tsq_val b = pop();
                tsq_val& a = r0;
                if (!a.null && a.v == 0) break;
                if (!b.null && b.v == 0) { a.v = 0; a.null = false; break; }
                if (a.null || b.null) { a.null = true; break; }
                a.v = 1;
                break;
            }
            case TSQ_OP_LOGIC_OR: {
                tsq_val b = pop();
                tsq_val& a = r0;
                if ((!a.null && a.v != 0) || (!b.null && b.v != 0)) { a.v = 1; a.null = false; }
                else if (a.null || b.null) a.null = true;
                else { a.v = 0; a.null = false; }
                break;
            }
            case TSQ_OP_NOT_INT: {
                tsq_val& a = r0;
                if (!a.null) a.v = a.v == 0 ? 1 : 0;
                break;
            }
            case TSQ_OP_NOT_REAL: {
                tsq_val& a = r0;
                if (!a.null) a.v = tsq_bits_f64((uint64_t)a.v) == 0 ? 1 : 0;
                else a.v = 0;
                break;
            }
            case TSQ_OP_NEG_INT: {
                tsq_val& a = r0;
                if (a.null) break;
                if (ul) {
                    if ((uint64_t)a.v > ((uint64_t)1 << 63)) return TSQ_ERR_OVERFLOW_BIGINT;
                } else if (a.v == TSQ_I64MIN) return TSQ_ERR_OVERFLOW_BIGINT;
                a.v = tsq_wneg(a.v);
                break;
            }
            case TSQ_OP_NEG_REAL: {
                tsq_val& a = r0;
                a.v = (int64_t)tsq_f64_bits(-tsq_bits_f64((uint64_t)a.v));
                break;
            }
            case TSQ_OP_ISNULL_INT:
            case TSQ_OP_ISNULL_REAL: {
                tsq_val& a = r0;
                a.v = a.null ? 1 : 0;
                a.null = false;
                break;
            }
            case TSQ_OP_IFNULL_INT:
            case TSQ_OP_IFNULL_REAL: {
                tsq_val b = pop();
                tsq_val& a = r0;
                if (a.null && !b.null) a = b;
                break;
            }
            case TSQ_OP_IF_INT:
            case TSQ_OP_IF_REAL: {
                tsq_val c2 = pop();
                tsq_val c1 = pop();
                tsq_val& c0 = r0;
                if (c0.null || c0.v == 0) c0 = c2;
                else c0 = c1;
                break;
            }
            case TSQ_OP_IN_INT:
            case TSQ_OP_IN_REAL: {
                const int nitems = op.arg;
                // rare and wide: spill the cached entries so that the whole stack is addressable
                if (sp >= 2) mem[sp - 2] = r1;
                mem[sp - 1] = r0;
                tsq_val* st = mem;
                const tsq_val x = st[sp - nitems - 1];
                bool hasNull = false, found = false;
                for (int j = 0; j < nitems; j++) {
                    const tsq_val it = st[sp - nitems + j];
                    if (it.null || x.null) { hasNull = true; continue; }
                    bool eq;
                    if (op.opcode == TSQ_OP_IN_REAL) eq = tsq_cmp_real(tsq_bits_f64((uint64_t)x.v), tsq_bits_f64((uint64_t)it.v)) == 0;
                    else {
                        const bool uj = (op.aux >> j) & 1;
                        if (ul == uj) eq = it.v == x.v;
                        else if (!ul && uj) eq = x.v >= 0 && it.v == x.v;
                        else eq = it.v >= 0 && it.v == x.v;
                    }
                    found = found || eq;
                }
                sp -= nitems;
                r0.v = found ? 1 : 0;
                r0.null = found ? false : hasNull;
                if (sp >= 2) r1 = mem[sp - 2];
                break;
            }
            case TSQ_OP_COL_STR: {
                bool bad = false;
                const tsq_val x = src.load_str(op.arg, &bad);
                if (bad) return TSQ_ERR_UNSUPPORTED;  // a cell beyond 4 GB of column data or longer than 16 MB
                push(x);
                break;
            }
            case TSQ_OP_CONST_STR: {
                tsq_val c;
                const uint64_t ol = (uint64_t)p.consts[op.arg];
                c.v = (int64_t)tsq_str_ref(TSQ_STR_POOL, ol >> 32, ol & TSQ_STR_MAXLEN);
                c.null = false;
                push(c);
                break;
            }
            case TSQ_OP_CONST_NULL_STR: { tsq_val c; c.v = 0; c.null = true; push(c); break; }
            case TSQ_OP_LT_STR: case TSQ_OP_LE_STR: case TSQ_OP_GT_STR:
            case TSQ_OP_GE_STR: case TSQ_OP_EQ_STR: case TSQ_OP_NE_STR:
            case TSQ_OP_STRCMP: {
                tsq_val b = pop();
                tsq_val& a = r0;
                a.null = a.null || b.null;
                if (a.null) { a.v = 0; break; }
                const uint64_t ha = (uint64_t)a.v, hb = (uint64_t)b.v;
                const uint32_t sa = (uint32_t)(ha >> 56), sb = (uint32_t)(hb >> 56);
                const uint8_t* pa = (sa == TSQ_STR_POOL ? (const uint8_t*)p.str_pool : src.str_base(sa)) + (uint32_t)ha;
                const uint8_t* pb = (sb == TSQ_STR_POOL ? (const uint8_t*)p.str_pool : src.str_base(sb)) + (uint32_t)hb;
                const int c = tsq_cmp_bytes(pa, tsq_str_len(ha), pb, tsq_str_len(hb));
                if (op.opcode == TSQ_OP_STRCMP) { a.v = c; break; }
                const int rel = op.opcode - TSQ_OP_LT_STR;
                const bool v = rel == 0 ? c < 0 : rel == 1 ? c <= 0 : rel == 2 ? c > 0 : rel == 3 ? c >= 0 : rel == 4 ? c == 0 : c != 0;
                a.v = v ? 1 : 0;
                break;
            }
            case TSQ_OP_LENGTH: {
                tsq_val& a = r0;
                a.v = a.null ? 0 : (int64_t)tsq_str_len((uint64_t)a.v);
                break;
            }
            case TSQ_OP_ISNULL_STR: {
                tsq_val& a = r0;
                a.v = a.null ? 1 : 0;
                a.null = false;
                break;
            }
            case TSQ_OP_IFNULL_STR: {
                tsq_val b = pop();
                tsq_val& a = r0;
                if (a.null && !b.null) a = b;
                break;
            }
            case TSQ_OP_IF_STR: {
                tsq_val c2 = pop();
                tsq_val c1 = pop();
                tsq_val& c0 = r0;
                if (c0.null || c0.v == 0) c0 = c2;
                else c0 = c1;
                break;
            }
            case TSQ_OP_IN_STR: {
                const int nitems = op.arg;
                if (sp >= 2) mem[sp - 2] = r1;
                mem[sp - 1] = r0;
                tsq_val* st = mem;
                const tsq_val x = st[sp - nitems - 1];
                bool hasNull = false, found = false;
                for (int j = 0; j < nitems; j++) {
                    const tsq_val it = st[sp - nitems + j];
                    if (it.null || x.null) { hasNull = true; continue; }
                    const uint64_t ha = (uint64_t)x.v, hb = (uint64_t)it.v;
                    const uint32_t sa = (uint32_t)(ha >> 56), sb = (uint32_t)(hb >> 56);
                    const uint8_t* pa = (sa == TSQ_STR_POOL ? (const uint8_t*)p.str_pool : src.str_base(sa)) + (uint32_t)ha;
                    const uint8_t* pb = (sb == TSQ_STR_POOL ? (const uint8_t*)p.str_pool : src.str_base(sb)) + (uint32_t)hb;
                    found = found || tsq_cmp_bytes(pa, tsq_str_len(ha), pb, tsq_str_len(hb)) == 0;
                }
                sp -= nitems;
                r0.v = found ? 1 : 0;
                r0.null = found ? false : hasNull;
                if (sp >= 2) r1 = mem[sp - 2];
                break;
            }
            default: return TSQ_ERR_INVALID;
        }
    }
    *out = r0;
    if (out->null) out->v = 0;
    return TSQ_OK;
}

// expression.VecEvalBool for one row (expression/expression.go:205-279): conjuncts are evaluated in
// order and only while the row is still alive; an Int-typed NULL keeps the row alive but marks it
// (`nulls`), a Real-typed NULL drops it; truthiness per toBool (:281-326).
template <class Src>
TSQ_HD tsq_status tsq_filter_row(const tsq_expr_prog* progs, int n_progs, const Src& src, bool* selected,
                                 bool* isnull, int* err_conj, int* err_node, int* div0) {
    bool nulls = false, alive = true;
    TSQ_JIT_UNROLL
    for (int e = 0; e < n_progs && alive; e++) {
        tsq_val v;
        *err_conj = e;
        tsq_status s = tsq_eval_row(progs[e], src, &v, err_node, div0);
        if (s != TSQ_OK) return s;
        const bool isint = progs[e].result_type != TSQ_F64;
        if (v.null) {
            if (isint) nulls = true;
            else alive = false;
        } else if (isint ? (v.v == 0) : tsq_real_is_zero(tsq_bits_f64((uint64_t)v.v))) {
            alive = false;
        }
    }
    *selected = alive && !nulls;
    *isnull = nulls;
    return TSQ_OK;
}

// error word for "first offending node, then first offending row" semantics of the vectorized
// evaluator: smaller == earlier in the reference's evaluation order.  atomicMin'ed by kernels.
//   [63:58] conjunct  [57:52] node  [51:4] row  [3:0] status
#define TSQ_ERRWORD_NONE 0xffffffffffffffffULL
TSQ_HD uint64_t tsq_errword(int conj, int node, uint64_t row, tsq_status st) {
    return ((uint64_t)conj << 58) | ((uint64_t)node << 52) | ((row & 0xffffffffffffULL) << 4) | (uint64_t)(st & 15);
}

// Static validation of a program (stack discipline, value kinds, indices, result type): host side, at compile.
// col_types (optional, n_cols entries) lets the column leaves be checked against the schema.
inline tsq_status tsq_validate_prog(const tsq_expr_prog& p, int32_t n_cols, const char** why, const int32_t* col_types = nullptr) {
    if (p.n_ops <= 0 || p.n_ops > TSQ_EXPR_MAX_OPS) { *why = "n_ops out of range"; return TSQ_ERR_INVALID; }
    if (p.n_consts < 0 || p.n_consts > TSQ_EXPR_MAX_CONSTS) { *why = "n_consts out of range"; return TSQ_ERR_INVALID; }
    if (p.n_str_bytes < 0 || p.n_str_bytes > TSQ_EXPR_STR_POOL) { *why = "string pool size out of range"; return TSQ_ERR_INVALID; }
    bool is_str[TSQ_EXPR_MAX_STACK + 1];  // kind of every stack entry: string reference or number
    int sp = 0;
    for (int k = 0; k < p.n_ops; k++) {
        const tsq_expr_op& op = p.ops[k];
        int pop = 0;
        bool res_str = false;
        int str_args = -1;  // -1: all popped values must be numbers; otherwise a bit mask of the popped entries (bit 0 = deepest) that must be strings
        switch (op.opcode) {
            case TSQ_OP_COL_INT: case TSQ_OP_COL_REAL: case TSQ_OP_COL_STR:
                if (n_cols >= 0 && op.arg >= n_cols) { *why = "column index out of range"; return TSQ_ERR_INVALID; }
                if (col_types && ((col_types[op.arg] == TSQ_BYTES) != (op.opcode == TSQ_OP_COL_STR))) { *why = "column leaf does not match the column type"; return TSQ_ERR_INVALID; }
                res_str = op.opcode == TSQ_OP_COL_STR;
                break;
            case TSQ_OP_CONST_INT: case TSQ_OP_CONST_REAL:
                if (op.arg >= p.n_consts) { *why = "const index out of range"; return TSQ_ERR_INVALID; }
                break;
            case TSQ_OP_CONST_STR: {
                if (op.arg >= p.n_consts) { *why = "const index out of range"; return TSQ_ERR_INVALID; }
                const uint64_t ol = (uint64_t)p.consts[op.arg];
                if ((ol >> 32) + (ol & 0xffffffffu) > (uint64_t)p.n_str_bytes) { *why = "string constant outside the pool"; return TSQ_ERR_INVALID; }
                res_str = true;
                break;
            }
            case TSQ_OP_CONST_NULL_INT: case TSQ_OP_CONST_NULL_REAL: break;
            case TSQ_OP_CONST_NULL_STR: res_str = true; break;
            case TSQ_OP_PLUS_REAL: case TSQ_OP_MINUS_REAL: case TSQ_OP_MUL_REAL: case TSQ_OP_DIV_REAL:
            case TSQ_OP_PLUS_INT: case TSQ_OP_MINUS_INT: case TSQ_OP_MUL_INT: case TSQ_OP_MUL_INT_UNSIGNED:
            case TSQ_OP_LT_INT: case TSQ_OP_LE_INT: case TSQ_OP_GT_INT: case TSQ_OP_GE_INT:
            case TSQ_OP_EQ_INT: case TSQ_OP_NE_INT: case TSQ_OP_LT_REAL: case TSQ_OP_LE_REAL:
            case TSQ_OP_GT_REAL: case TSQ_OP_GE_REAL: case TSQ_OP_EQ_REAL: case TSQ_OP_NE_REAL:
            case TSQ_OP_LOGIC_AND: case TSQ_OP_LOGIC_OR: case TSQ_OP_IFNULL_INT: case TSQ_OP_IFNULL_REAL:
                pop = 2;
                break;
            case TSQ_OP_LT_STR: case TSQ_OP_LE_STR: case TSQ_OP_GT_STR: case TSQ_OP_GE_STR:
            case TSQ_OP_EQ_STR: case TSQ_OP_NE_STR: case TSQ_OP_STRCMP:
                pop = 2;
                str_args = 3;
                break;
            case TSQ_OP_IFNULL_STR:
                pop = 2;
                str_args = 3;
                res_str = true;
                break;
            case TSQ_OP_NOT_INT: case TSQ_OP_NOT_REAL: case TSQ_OP_NEG_INT: case TSQ_OP_NEG_REAL:
            case TSQ_OP_ISNULL_INT: case TSQ_OP_ISNULL_REAL:
                pop = 1;
                break;
            case TSQ_OP_LENGTH: case TSQ_OP_ISNULL_STR:
                pop = 1;
                str_args = 1;
                break;
            case TSQ_OP_IF_INT: case TSQ_OP_IF_REAL: pop = 3; break;
            case TSQ_OP_IF_STR:
                pop = 3;
                str_args = 6;  // cond is a number, both values are strings
                res_str = true;
                break;
            case TSQ_OP_IN_INT: case TSQ_OP_IN_REAL: case TSQ_OP_IN_STR:
                if (op.arg < 1 || op.arg > 31) { *why = "IN list size out of range"; return TSQ_ERR_INVALID; }
                pop = op.arg + 1;
                if (op.opcode == TSQ_OP_IN_STR) str_args = -2;  // every popped value is a string
                break;
            default: *why = "unknown opcode"; return TSQ_ERR_UNSUPPORTED;
        }
        if (sp < pop) { *why = "stack underflow"; return TSQ_ERR_INVALID; }
        for (int q = 0; q < pop; q++) {
            const bool want = str_args == -2 ? true : (str_args < 0 ? false : ((str_args >> q) & 1) != 0);
            if (is_str[sp - pop + q] != want) { *why = "operand kind (string / number) does not match the opcode"; return TSQ_ERR_INVALID; }
        }
        sp = sp - pop + 1;
        if (sp > TSQ_EXPR_MAX_STACK) { *why = "expression too deep"; return TSQ_ERR_UNSUPPORTED; }
        is_str[sp - 1] = res_str;
    }
    if (sp != 1) { *why = "program leaves stack depth != 1"; return TSQ_ERR_INVALID; }
    // a string-valued root (IF / IFNULL of strings, a string column, a string constant) is declared by result_type TSQ_BYTES and
    // evaluated by tsq_expr_eval_str only; conditions and filters need an Int or Real root
    // A string-valued root declared as a number is a plan the numeric evaluators cannot take — UNSUPPORTED, the fallback signal of
    // every other string-valued root a host may hand in (it keeps its Go evaluator); a numeric root declared TSQ_BYTES is malformed.
    if (is_str[0] && p.result_type != TSQ_BYTES) { *why = "a string-valued root: declare result_type TSQ_BYTES and evaluate it with tsq_expr_eval_str"; return TSQ_ERR_UNSUPPORTED; }
    if (!is_str[0] && p.result_type == TSQ_BYTES) { *why = "result_type TSQ_BYTES needs a string-valued root"; return TSQ_ERR_INVALID; }
    if (p.result_type != TSQ_I64 && p.result_type != TSQ_F64 && p.result_type != TSQ_BYTES) { *why = "result_type must be TSQ_I64, TSQ_F64 or TSQ_BYTES"; return TSQ_ERR_INVALID; }
    return TSQ_OK;
}

#endif  // TSQ_DEVICE_H
