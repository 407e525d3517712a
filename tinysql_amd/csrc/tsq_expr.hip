// tsq_expr.hip — fused vectorized expression evaluation for gfx950 (MI355X).
//
// The reference evaluates an expression tree node by node, materialising a 1024-row chunk.Column
// per node (expression/scalar_function.go:43-80, builtin_*_vec.go).  Here the whole tree is ONE
// kernel: each lane walks the postfix program for its row in registers (tsq_eval_row in
// tsq_device.h restates every builtin*Sig.vecEval* per row), so an expression over `a` input
// columns moves 8·a + 8 bytes per row instead of 16+ bytes per node.  Error semantics of the
// vectorized evaluator ("the first offending node aborts the statement") are kept with an
// atomicMin'ed error word ordered by (conjunct, node, row).
#include "tsq_stage.h"

#include <hip/hiprtc.h>
#include <chrono>

#include <memory>
#include <sstream>

#include "tsq_jit_src.inc"

struct ExprArgs {
    tsq_colset in;
    const tsq_expr_prog* progs;  // device
    int32_t n_progs;
    int64_t nrows;
    const int32_t* sel;          // optional logical -> physical (chunk.go:319-331)
    uint64_t* out_data;          // projection: 8 bytes per row
    uint8_t* out_notnull;        // projection: 1 byte per row
    uint8_t* out_selected;       // filter: 1 byte per row (Go []bool)
    uint8_t* out_isnull;         // filter: optional
    unsigned long long* counters;  // [0] = error word (min), [1] = division-by-zero warnings, [2] = rows whose NOT-NULL bits the kernel wrote into out_bits itself
    uint32_t* out_bits;          // projection, optional: the result's null bitmap (the specialised kernel writes whole 32-row words of it instead of byte flags)
};

// The postfix programs are copied into LDS once per workgroup: interpreting them out of global memory made every
// node a dependent ~1 us load (the next opcode is not known before the previous load returns), i.e. 3.7 ms per 1e8
// rows for a 7-node expression whatever the rest of the kernel did.
#define TSQ_EXPR_MAX_PROGS 16
__device__ __forceinline__ void stage_progs(tsq_expr_prog* dst, const tsq_expr_prog* src, int n_progs) {
    const uint32_t words = (uint32_t)(n_progs * sizeof(tsq_expr_prog) / 4);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
    __syncthreads();
}

// K9 — projection form: expression.VecEval (expression/expression.go:329-341)
__global__ void __launch_bounds__(256) k_expr_eval(ExprArgs a) {
    __shared__ tsq_expr_prog s_progs[1];
    stage_progs(s_progs, a.progs, 1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t errw = TSQ_ERRWORD_NONE;
    uint32_t div0 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nrows; i += stride) {
        tsq_chunk_src src{&a.in, a.sel ? (int64_t)a.sel[i] : i};
        tsq_val v;
        int node = 0, d0 = 0;
        tsq_status s = tsq_eval_row(s_progs[0], src, &v, &node, &d0);
        div0 += (uint32_t)d0;
        if (s != TSQ_OK) {
            uint64_t w = tsq_errword(0, node, (uint64_t)i, s);
            errw = w < errw ? w : errw;
            continue;
        }
        a.out_data[i] = (uint64_t)v.v;
        a.out_notnull[i] = v.null ? 0 : 1;
    }
    if (errw != TSQ_ERRWORD_NONE) atomicMin(&a.counters[0], (unsigned long long)errw);
    if (div0) atomicAdd(&a.counters[1], (unsigned long long)div0);
}

// K10 — filter form: expression.VecEvalBool / VectorizedFilter (expression.go:205-279,
// chunk_executor.go:196-245): CNF list -> selected[] (+ nulls[]).
__global__ void __launch_bounds__(256) k_filter_eval(ExprArgs a) {
    __shared__ tsq_expr_prog s_progs[TSQ_EXPR_MAX_PROGS];
    stage_progs(s_progs, a.progs, a.n_progs);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t errw = TSQ_ERRWORD_NONE;
    uint32_t div0 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nrows; i += stride) {
        tsq_chunk_src src{&a.in, a.sel ? (int64_t)a.sel[i] : i};
        bool selected = false, isnull = false;
        int conj = 0, node = 0, d0 = 0;
        tsq_status s = tsq_filter_row(s_progs, a.n_progs, src, &selected, &isnull, &conj, &node, &d0);
        div0 += (uint32_t)d0;
        if (s != TSQ_OK) {
            uint64_t w = tsq_errword(conj, node, (uint64_t)i, s);
            errw = w < errw ? w : errw;
            continue;
        }
        a.out_selected[i] = selected ? 1 : 0;
        if (a.out_isnull) a.out_isnull[i] = isnull ? 1 : 0;
    }
    if (errw != TSQ_ERRWORD_NONE) atomicMin(&a.counters[0], (unsigned long long)errw);
    if (div0) atomicAdd(&a.counters[1], (unsigned long long)div0);
}

// K9s — a string-valued root: the evaluation kernels leave a REFERENCE per row (source column or the program's constant pool,
// offset, length: tsq_device.h); the lengths of the non-NULL rows, their exclusive scan (= offsets[nrows + 1]) and one copy pass
// make the var-len column AppendNull / AppendString would have built (builtin_control_vec_generated.go:81-115, 209-255).
struct StrRootArgs {
    const uint64_t* refs;
    const uint8_t* notnull;
    int64_t nrows;
    tsq_colset in;
    const tsq_expr_prog* prog;  // device: the constant pool
    int64_t* offs;
    uint8_t* data;
};
__global__ void __launch_bounds__(256) k_expr_str_len(StrRootArgs a) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.nrows; i += (int64_t)gridDim.x * 256)
        a.offs[i] = a.notnull[i] ? (int64_t)tsq_str_len(a.refs[i]) : 0;
}
__device__ __forceinline__ const uint8_t* str_root_src(const StrRootArgs& a, uint64_t ref) {
    const uint32_t src = (uint32_t)(ref >> 56);
    const uint32_t off = (uint32_t)ref;
    return src == TSQ_STR_POOL ? a.prog->str_pool + off : (const uint8_t*)a.in.data[src] + off;
}
template <bool WAVE>
__global__ void __launch_bounds__(256) k_expr_str_copy(StrRootArgs a) {
    if (!WAVE) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.nrows; i += (int64_t)gridDim.x * 256) {
            const int64_t n = a.offs[i + 1] - a.offs[i];
            if (n > 0) tsq_copy_cell(a.data + a.offs[i], str_root_src(a, a.refs[i]), n);
        }
        return;
    }
    const int lane = threadIdx.x & 63;  // long cells: one row per wave, 64 lanes on consecutive bytes
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t i = wave; i < a.nrows; i += nwaves) {
        const int64_t n = a.offs[i + 1] - a.offs[i];
        if (n <= 0) continue;
        const uint8_t* s = str_root_src(a, a.refs[i]);
        uint8_t* d = a.data + a.offs[i];
        for (int64_t b = lane; b < n; b += 64) d[b] = s[b];
    }
}

struct tsq_expr {
    tsq_handle_hdr hdr;
    tsq_ctx* ctx = nullptr;
    std::vector<tsq_expr_prog> progs;
    DevBuf progs_d, counters;
    std::vector<ColStore> icols;  // device copies of host input chunks
    DevBuf sel_d, out_data, out_nn, out_bitmap, out_sel, out_isnull, str_offs, str_data, scan_tmp;
    PinnedBuf hout, hflags;
    int64_t launches = 0;
    // run-time specialised kernels (hiprtc): the postfix programs become compile-time constants, the interpreter
    // loop of tsq_eval_row unrolls and every switch folds — same source, same semantics, ~10x fewer instructions
    int32_t jit_mode = TSQ_JIT_AUTO;
    bool jit_tried = false;
    std::string jit_src;  // generated once per handle (the plan cache's key)
    hipModule_t jit_mod = nullptr;
    hipFunction_t jit_expr = nullptr, jit_filter = nullptr;
    int64_t rows_seen = 0, jit_launches = 0;
    double jit_compile_ms = 0;
    std::string jit_log;
};

namespace {

// brings the input chunk to the device when it is host resident; fills `cs`
tsq_status expr_inputs(tsq_expr* e, const tsq_col* cols, int32_t n_cols, int64_t phys_rows, tsq_colset& cs, bool* is_dev) {
    tsq_ctx* ctx = e->ctx;
    tsq_handle_hdr* h = &e->hdr;
    if (n_cols < 0 || n_cols > TSQ_MAX_COLS) return tsq_fail(h, TSQ_ERR_INVALID, "too many input columns");
    bool dev = false, host = false;
    for (int c = 0; c < n_cols; c++) {
        if (cols[c].type < TSQ_I64 || cols[c].type > TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, "unknown column type");
        if (cols[c].type == TSQ_BYTES && !cols[c].offsets) return tsq_fail(h, TSQ_ERR_INVALID, "var-len column without offsets");
        if (!cols[c].data && cols[c].length > 0 && cols[c].type != TSQ_BYTES) return tsq_fail(h, TSQ_ERR_INVALID, "column data == NULL");
        (cols[c].flags & TSQ_COL_DEVICE) ? dev = true : host = true;
    }
    if (dev && host) return tsq_fail(h, TSQ_ERR_INVALID, "mixing host and device columns");
    *is_dev = dev;
    if (dev) {
        tsq_colset_from_cols(cs, cols, n_cols);
        return TSQ_OK;
    }
    e->icols.resize(n_cols);
    DevBuf tmp;
    for (int c = 0; c < n_cols; c++) {
        ColStore& st = e->icols[c];
        st.type = cols[c].type;
        st.rows = 0;
        st.has_nulls = false;
        const int64_t n = std::min<int64_t>(cols[c].length, phys_rows);
        tsq_status s = TSQ_OK;
        if (cols[c].type == TSQ_BYTES) {
            // var-len column (util/chunk/column.go:28-34): the n + 1 offsets and the offsets[n] data bytes, as they are
            const int64_t nbytes = n > 0 ? cols[c].offsets[n] : 0;
            if (nbytes < 0 || (n > 0 && cols[c].offsets[0] != 0)) { tmp.release(); return tsq_fail(h, TSQ_ERR_INVALID, "var-len column: offsets must start at 0 and grow"); }
            if ((uint64_t)nbytes >> 32) { tmp.release(); return tsq_fail(h, TSQ_ERR_UNSUPPORTED, "var-len column beyond 4 GB of data in one batch"); }
            s = st.offs.reserve(ctx, h, (size_t)(n + 1) * 8 + 64);
            if (s == TSQ_OK) s = st.data.reserve(ctx, h, (size_t)nbytes + 64);
            if (s != TSQ_OK) { tmp.release(); return s; }
            static const int64_t zero = 0;
            hipError_t e1 = hipMemcpyAsync(st.offs.p, n > 0 ? cols[c].offsets : &zero, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
            if (e1 == hipSuccess && nbytes) e1 = hipMemcpyAsync(st.data.p, cols[c].data, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream);
            if (e1 != hipSuccess) { tmp.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("hipMemcpyAsync(var-len column): ") + hipGetErrorString(e1)); }
            if (cols[c].null_bitmap && n > 0) {  // the store starts at row 0: the host bitmap is copied as it is
                s = st.nulls.reserve(ctx, h, tsq_bitmap_bytes(n) + 64);
                if (s != TSQ_OK) { tmp.release(); return s; }
                e1 = hipMemcpyAsync(st.nulls.p, cols[c].null_bitmap, tsq_bitmap_bytes(n), hipMemcpyHostToDevice, ctx->stream);
                if (e1 != hipSuccess) { tmp.release(); return tsq_fail(h, TSQ_ERR_HIP, std::string("hipMemcpyAsync(bitmap): ") + hipGetErrorString(e1)); }
                st.has_nulls = true;
            }
            st.rows = n;
        } else {
            s = tsq_col_append(ctx, h, st, cols[c].data, cols[c].null_bitmap, n, false, tmp);
        }
        if (s != TSQ_OK) { tmp.release(); return s; }
    }
    hipError_t err = hipStreamSynchronize(ctx->stream);
    tmp.release();
    if (err != hipSuccess) return tsq_fail(h, TSQ_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(err));
    tsq_fill_colset(cs, e->icols);
    return TSQ_OK;
}

tsq_status expr_status(tsq_expr* e, uint64_t w) {
    if (w == TSQ_ERRWORD_NONE) return TSQ_OK;
    tsq_status s = (tsq_status)(w & 15);
    const char* what = s == TSQ_ERR_OVERFLOW_BIGINT            ? "BIGINT value is out of range"
                       : s == TSQ_ERR_OVERFLOW_BIGINT_UNSIGNED ? "BIGINT UNSIGNED value is out of range"
                       : s == TSQ_ERR_OVERFLOW_DOUBLE          ? "DOUBLE value is out of range"
                                                               : "expression error";
    char buf[160];
    snprintf(buf, sizeof buf, "%s (conjunct %d, node %d, row %llu)", what, (int)(w >> 58), (int)((w >> 52) & 63),
             (unsigned long long)((w >> 4) & 0xffffffffffffULL));
    return tsq_fail(&e->hdr, s, buf);
}

}  // namespace

TSQ_API tsq_status tsq_expr_compile(tsq_ctx* ctx, const tsq_expr_prog* progs, int32_t n_progs, tsq_expr** out) {
    tsq_ctx_lock _api_lock(ctx);
    if (!ctx || !progs || !out || n_progs < 1 || n_progs > 16) return tsq_fail(ctx ? &ctx->hdr : nullptr, TSQ_ERR_INVALID, "tsq_expr_compile: bad arguments");
    *out = nullptr;
    for (int i = 0; i < n_progs; i++) {
        const char* why = "";
        tsq_status s = tsq_validate_prog(progs[i], -1, &why);
        if (s != TSQ_OK) return tsq_fail(&ctx->hdr, s, std::string("tsq_expr_compile: ") + why);
    }
    TSQ_HIP(&ctx->hdr, hipSetDevice(ctx->device));
    std::unique_ptr<tsq_expr> e(new tsq_expr());
    e->hdr.magic = TSQ_MAGIC_EXPR;
    e->ctx = ctx;
    e->progs.assign(progs, progs + n_progs);
    tsq_status s = e->progs_d.reserve(ctx, &e->hdr, sizeof(tsq_expr_prog) * n_progs);
    if (s == TSQ_OK) s = e->counters.reserve(ctx, &e->hdr, 64);
    if (s == TSQ_OK) {
        hipError_t err = hipMemcpy(e->progs_d.p, progs, sizeof(tsq_expr_prog) * n_progs, hipMemcpyHostToDevice);
        if (err != hipSuccess) s = tsq_fail(&e->hdr, TSQ_ERR_HIP, hipGetErrorString(err));
    }
    if (s != TSQ_OK) {
        tsq_fail(&ctx->hdr, s, e->hdr.err);
        tsq_expr_destroy(e.release());
        return s;
    }
    *out = e.release();
    return TSQ_OK;
}

// ---------------------------------------------------------------- run-time specialisation
// The generic kernels above interpret the program: ~110 instructions per node per wave (decode, scalar branches,
// stack traffic), i.e. a 7-node expression runs at 6 % of the HBM roofline.  For large inputs the SAME source
// (tsq.h + tsq_device.h, embedded at build time) is compiled once per handle with the programs as a constant
// table; the compiler unrolls the node loop and folds every opcode switch, leaving straight-line code.
#define TSQ_JIT_VARIANT_DEFAULT (7 | 128 | 256)  // coalesced + non-temporal projection, non-temporal filter (profiles/r06_jit_sweep.txt)
static std::string jit_source(const std::vector<tsq_expr_prog>& progs, int variant) {
    std::ostringstream o;
    o << "#define TSQ_JIT 1\n";
    o << "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
         "typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;\n"
         "typedef unsigned long size_t;\n";
    o << TSQ_JIT_HDR_ABI << "\n" << TSQ_JIT_HDR_DEV << "\n";
    o << "struct ExprArgs { tsq_colset in; const tsq_expr_prog* progs; int32_t n_progs; int64_t nrows; const int32_t* sel; uint64_t* out_data;\n"
         "  uint8_t* out_notnull; uint8_t* out_selected; uint8_t* out_isnull; unsigned long long* counters; uint32_t* out_bits; };\n";
    o << "__device__ const tsq_expr_prog P[" << progs.size() << "] = {\n";
    for (const tsq_expr_prog& p : progs) {
        o << " { " << p.n_ops << ", " << p.n_consts << ", " << p.result_type << ", " << p.result_unsigned << ", {";
        for (int k = 0; k < TSQ_EXPR_MAX_OPS; k++) {
            const tsq_expr_op& op = p.ops[k < p.n_ops ? k : 0];
            if (k < p.n_ops) o << "{" << (int)op.opcode << "," << (int)op.flags << "," << (int)op.arg << "," << op.aux << "u},";
            else o << "{0,0,0,0u},";
        }
        o << "}, {";
        for (int c = 0; c < TSQ_EXPR_MAX_CONSTS; c++) o << "(int64_t)0x" << std::hex << (unsigned long long)(c < p.n_consts ? p.consts[c] : 0) << std::dec << "ULL,";
        o << "}, " << p.n_str_bytes << ", 0, {";
        for (int b = 0; b < TSQ_EXPR_STR_POOL; b++) o << (b < p.n_str_bytes ? (int)p.str_pool[b] : 0) << ",";
        o << "} },\n";
    }
    o << "};\n";
    o << "#define N_PROGS " << progs.size() << "\n#define JIT_VARIANT " << variant << "\n";
    // ---- round 6: TWO rows per lane.  The 8-byte cells of the columns the programs read are loaded beforehand, rows 2 p and 2 p + 1 of a
    // column with ONE 16-byte load (a float4-style stream: the 8-byte-per-lane loop reached 0.38 of the roofline on (a + b) * 3 - a), and
    // the two results leave with one 16-byte store (+ one 2-byte store of their NOT-NULL flags).  SLOT[c] = the register slot of column c.
    int slot_of[TSQ_MAX_COLS], n_slots = 0;
    for (int c = 0; c < TSQ_MAX_COLS; c++) slot_of[c] = -1;
    bool pairs_ok = true;
    for (const tsq_expr_prog& p : progs)
        for (int k = 0; k < p.n_ops; k++) {
            const int opc = p.ops[k].opcode, c = p.ops[k].arg;
            if (opc == TSQ_OP_COL_STR) pairs_ok = false;  // (var-len cells are not loaded beforehand: the one-row loop keeps such programs)
            if ((opc == TSQ_OP_COL_INT || opc == TSQ_OP_COL_REAL) && c >= 0 && c < TSQ_MAX_COLS && slot_of[c] < 0) slot_of[c] = n_slots++;
        }
    if (n_slots == 0 || n_slots > 8) pairs_ok = false;
    o << "#define PAIRS_OK " << (pairs_ok ? 1 : 0) << "\n#define N_SLOTS " << (n_slots > 0 ? n_slots : 1) << "\n";
    o << "__device__ const int SLOT[" << TSQ_MAX_COLS << "] = {";
    for (int c = 0; c < TSQ_MAX_COLS; c++) o << slot_of[c] << ",";
    o << "};\n__device__ const int SLOT_COL[N_SLOTS] = {";
    for (int sl = 0; sl < (n_slots > 0 ? n_slots : 1); sl++) {
        int col = 0;
        for (int c = 0; c < TSQ_MAX_COLS; c++)
            if (slot_of[c] == sl) col = c;
        o << col << ",";
    }
    o << "};\n";
    o << R"JIT(
typedef unsigned long long jit_v2u64 __attribute__((ext_vector_type(2)));
struct tsq_pre_src {  // a row whose 8-byte cells sit in registers already
    const tsq_colset* cs;
    int64_t row;
    uint64_t cell[N_SLOTS];
    bool isnull[N_SLOTS];
    __device__ tsq_val load_int(int c) const { tsq_val r; r.v = (int64_t)cell[SLOT[c]]; r.null = isnull[SLOT[c]]; return r; }
    __device__ tsq_val load_real(int c) const { return load_int(c); }
    __device__ tsq_val load_str(int c, bool* bad) const { tsq_val r; r.v = (int64_t)tsq_cell_str(*cs, c, c, row, &r.null, bad); return r; }
    __device__ const uint8_t* str_base(uint32_t src) const { return (const uint8_t*)cs->data[src]; }
};
// the pairs loop applies when the rows are not indirected, every column read is an 8-byte column on a 16-byte boundary (F32 is widened
// cell by cell in the one-row loop) and the outputs are aligned too
__device__ bool jit_pairs_usable(const ExprArgs& a, const void* out, unsigned out_align) {
    if (!PAIRS_OK || a.sel != nullptr || ((unsigned long)out & (out_align - 1u))) return false;
    for (int sl = 0; sl < N_SLOTS; sl++) {
        const int c = SLOT_COL[sl];
        if (a.in.type[c] == TSQ_F32 || a.in.type[c] == TSQ_BYTES || ((unsigned long)a.in.data[c] & 15u)) return false;
    }
    return true;
}
// rows 4 q .. 4 q + 3 of every column the tree reads: two 16-byte loads per column and lane (a wave takes 2 KB of a column at once), the
// four NOT-NULL bits = one nibble of the bitmap.  JIT_VARIANT & 4 (A/B): a wave's step of 256 rows as two whole-wave coalesced
// 16-byte accesses — lane l takes rows 2 l, 2 l + 1 of the first and of the second 128 rows
#define JIT_COAL ((JIT_VARIANT & 4) != 0)
__device__ __forceinline__ int64_t jit_row(int64_t q, int r) {
    if (!JIT_COAL) return 4 * q + r;
    return ((q >> 6) << 8) + ((r >> 1) << 7) + 2 * (q & 63) + (r & 1);
}
__device__ __forceinline__ jit_v2u64 jit_ld16(const jit_v2u64* p) {
    if (JIT_VARIANT & 1) return __builtin_nontemporal_load(p);
    return *p;
}
__device__ __forceinline__ void jit_st16(jit_v2u64* p, jit_v2u64 v) {
    if (JIT_VARIANT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ void jit_load_quad(const ExprArgs& a, int64_t q, tsq_pre_src (&s)[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        s[r].cs = &a.in;
        s[r].row = jit_row(q, r);
    }
    const int64_t i0 = JIT_COAL ? ((q >> 6) << 7) + (q & 63) : 2 * q, i1 = JIT_COAL ? i0 + 64 : i0 + 1;
#pragma unroll
    for (int sl = 0; sl < N_SLOTS; sl++) {
        const int c = SLOT_COL[sl];
        const jit_v2u64* src = reinterpret_cast<const jit_v2u64*>(a.in.data[c]);
        const jit_v2u64 x = jit_ld16(src + i0), y = jit_ld16(src + i1);
        s[0].cell[sl] = x.x;
        s[1].cell[sl] = x.y;
        s[2].cell[sl] = y.x;
        s[3].cell[sl] = y.y;
        uint32_t b = 0xfu;
        if (a.in.nulls[c]) {
            if (JIT_COAL) {
                const int64_t by = ((q >> 6) << 5) + ((q & 63) >> 2);
                const uint32_t sh = 2u * (uint32_t)(q & 3);
                b = (((uint32_t)a.in.nulls[c][by] >> sh) & 3u) | ((((uint32_t)a.in.nulls[c][by + 16] >> sh) & 3u) << 2);
            } else b = (uint32_t)a.in.nulls[c][q >> 1] >> ((uint32_t)(q & 1) * 4u);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) s[r].isnull[sl] = !((b >> r) & 1u);
    }
}
// one lane's four rows of a step: evaluated, stored; the NOT-NULL bits as whole 32-row words of the result's bitmap (bits) or as flag bytes
__device__ __forceinline__ void jit_quad(const ExprArgs& a, int64_t q, tsq_pre_src (&s)[4], bool bits, uint64_t& errw, uint32_t& div0) {
    tsq_val v[4];
    int n[4] = {0, 0, 0, 0}, d0 = 0;
    tsq_status t[4];
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = tsq_eval_row(P[0], s[r], &v[r], &n[r], &d0);
    div0 += (uint32_t)d0;
    const bool ok = t[0] == TSQ_OK && t[1] == TSQ_OK && t[2] == TSQ_OK && t[3] == TSQ_OK;
    if (bits) {  // (a row that raised an error: the whole result is discarded)
        if (JIT_COAL) {  // sixteen lanes make one word of each half
            const uint32_t l16 = (uint32_t)q & 15u;
            uint32_t w0 = ((v[0].null ? 0u : 1u) | (v[1].null ? 0u : 2u)) << (2u * l16), w1 = ((v[2].null ? 0u : 1u) | (v[3].null ? 0u : 2u)) << (2u * l16);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                w0 |= (uint32_t)__shfl_xor((int)w0, o);
                w1 |= (uint32_t)__shfl_xor((int)w1, o);
            }
            if (l16 == 0) {
                const int64_t wi = ((q >> 6) << 3) + ((q & 63) >> 4);
                a.out_bits[wi] = w0;
                a.out_bits[wi + 4] = w1;
            }
        } else {  // a lane's four rows are a nibble; eight neighbouring lanes make one word
            const uint32_t lane8 = (uint32_t)q & 7u;
            uint32_t w = ((v[0].null ? 0u : 1u) | (v[1].null ? 0u : 2u) | (v[2].null ? 0u : 4u) | (v[3].null ? 0u : 8u)) << (4u * lane8);
            w |= (uint32_t)__shfl_xor((int)w, 1);
            w |= (uint32_t)__shfl_xor((int)w, 2);
            w |= (uint32_t)__shfl_xor((int)w, 4);
            if (lane8 == 0) a.out_bits[q >> 3] = w;
        }
    }
    if (ok) {
        jit_v2u64 y0, y1;
        y0.x = (uint64_t)v[0].v;
        y0.y = (uint64_t)v[1].v;
        y1.x = (uint64_t)v[2].v;
        y1.y = (uint64_t)v[3].v;
        jit_v2u64* dst = reinterpret_cast<jit_v2u64*>(a.out_data);
        const int64_t i0 = JIT_COAL ? ((q >> 6) << 7) + (q & 63) : 2 * q, i1 = JIT_COAL ? i0 + 64 : i0 + 1;
        jit_st16(dst + i0, y0);
        jit_st16(dst + i1, y1);
        if (!bits) {
            if (JIT_COAL) {
                *reinterpret_cast<unsigned short*>(a.out_notnull + jit_row(q, 0)) = (unsigned short)((v[0].null ? 0u : 1u) | (v[1].null ? 0u : 0x100u));
                *reinterpret_cast<unsigned short*>(a.out_notnull + jit_row(q, 2)) = (unsigned short)((v[2].null ? 0u : 1u) | (v[3].null ? 0u : 0x100u));
            } else
                reinterpret_cast<uint32_t*>(a.out_notnull)[q] = (v[0].null ? 0u : 1u) | (v[1].null ? 0u : 0x100u) | (v[2].null ? 0u : 0x10000u) | (v[3].null ? 0u : 0x1000000u);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int64_t row = jit_row(q, r);
        if (t[r] != TSQ_OK) { const uint64_t w = tsq_errword(0, n[r], (uint64_t)row, t[r]); errw = w < errw ? w : errw; }
        else { a.out_data[row] = (uint64_t)v[r].v; if (!bits) a.out_notnull[row] = v[r].null ? 0 : 1; }
    }
}
extern "C" __global__ void __launch_bounds__(256) jit_expr(ExprArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t errw = TSQ_ERRWORD_NONE;
    uint32_t div0 = 0;
    int64_t first = 0;  // rows the four-row loop has done
    if (jit_pairs_usable(a, a.out_data, 16u) && !((unsigned long)a.out_notnull & 3u)) {
        // whole 32-row words of the result's bitmap are written by the lanes themselves (out_bits: the host's pack pass then starts at row
        // counters[2]) — otherwise flag bytes
        const bool bits = a.out_bits != nullptr;
        const int64_t nq = JIT_COAL ? (a.nrows >> 8) << 6 : (bits ? (a.nrows >> 5) << 3 : a.nrows >> 2);  // (whole words / whole waves stay together)
        int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (JIT_VARIANT & 8)
            for (; q + stride < nq; q += 2 * stride) {  // two steps' loads in flight
                tsq_pre_src s0[4], s1[4];
                jit_load_quad(a, q, s0);
                jit_load_quad(a, q + stride, s1);
                jit_quad(a, q, s0, bits, errw, div0);
                jit_quad(a, q + stride, s1, bits, errw, div0);
            }
        for (; q < nq; q += stride) {
            tsq_pre_src s[4];
            jit_load_quad(a, q, s);
            jit_quad(a, q, s, bits, errw, div0);
        }
        first = nq * 4;
        if (bits && blockIdx.x == 0 && threadIdx.x == 0) a.counters[2] = (unsigned long long)first;
    }
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nrows; i += stride) {
        tsq_chunk_src src{&a.in, a.sel ? (int64_t)a.sel[i] : i};
        tsq_val v;
        int node = 0, d0 = 0;
        tsq_status s = tsq_eval_row(P[0], src, &v, &node, &d0);
        div0 += (uint32_t)d0;
        if (s != TSQ_OK) {
            uint64_t w = tsq_errword(0, node, (uint64_t)i, s);
            errw = w < errw ? w : errw;
            continue;
        }
        a.out_data[i] = (uint64_t)v.v;
        a.out_notnull[i] = v.null ? 0 : 1;
    }
    if (errw != TSQ_ERRWORD_NONE) atomicMin(&a.counters[0], (unsigned long long)errw);
    if (div0) atomicAdd(&a.counters[1], (unsigned long long)div0);
}
// JIT_VARIANT & 128 (A/B): the filter's 8-byte cells through non-temporal loads (the columns are read once)
struct tsq_nt_src {
    const tsq_colset* cs;
    int64_t row;
    __device__ tsq_val load_int(int c) const {
        tsq_val r;
        r.null = tsq_is_null(cs->nulls[c], row);
        r.v = __builtin_nontemporal_load((const int64_t*)cs->data[c] + row);
        return r;
    }
    __device__ tsq_val load_real(int c) const {
        if (cs->type[c] == TSQ_F32) return tsq_cell_real(*cs, c, row);
        return load_int(c);
    }
    __device__ tsq_val load_str(int c, bool* bad) const { tsq_val r; r.v = (int64_t)tsq_cell_str(*cs, c, c, row, &r.null, bad); return r; }
    __device__ const uint8_t* str_base(uint32_t src) const { return (const uint8_t*)cs->data[src]; }
};
extern "C" __global__ void __launch_bounds__(256) jit_filter(ExprArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t errw = TSQ_ERRWORD_NONE;
    uint32_t div0 = 0;
    // (one row per lane: a CNF list stops at its first false conjunct, so the columns of the later conjuncts are not read for the rows that
    // failed — loading every column of two rows beforehand, as jit_expr does, measured 0.53 vs 0.47 ms on `a < b AND c > 0.5`, 1e8 rows)
    const int64_t first = 0;
    for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nrows; i += stride) {
        bool selected = false, isnull = false;
        int conj = 0, node = 0, d0 = 0;
        tsq_status s;
        if (JIT_VARIANT & 128) {
            tsq_nt_src src{&a.in, a.sel ? (int64_t)a.sel[i] : i};
            s = tsq_filter_row(P, N_PROGS, src, &selected, &isnull, &conj, &node, &d0);
        } else {
            tsq_chunk_src src{&a.in, a.sel ? (int64_t)a.sel[i] : i};
            s = tsq_filter_row(P, N_PROGS, src, &selected, &isnull, &conj, &node, &d0);
        }
        div0 += (uint32_t)d0;
        if (s != TSQ_OK) {
            uint64_t w = tsq_errword(conj, node, (uint64_t)i, s);
            errw = w < errw ? w : errw;
            continue;
        }
        if (JIT_VARIANT & 256) __builtin_nontemporal_store((uint8_t)(selected ? 1 : 0), &a.out_selected[i]);
        else a.out_selected[i] = selected ? 1 : 0;
        if (a.out_isnull) a.out_isnull[i] = isnull ? 1 : 0;
    }
    if (errw != TSQ_ERRWORD_NONE) atomicMin(&a.counters[0], (unsigned long long)errw);
    if (div0) atomicAdd(&a.counters[1], (unsigned long long)div0);
}
)JIT";
    return o.str();
}

// compiles once per distinct program set and context; on any failure the generic kernels keep serving (still the GPU, never a CPU path).
// jit_compile_code: source -> code object (hiprtc only: no device call, so a helper thread may run it); jit_load: code object -> module
static void jit_compile_code(const std::string& arch_name, const std::string& src, tsq_ctx::JitEntry& out) {
    const auto t0 = std::chrono::steady_clock::now();
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "tsq_expr_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        out.log = "hiprtcCreateProgram failed";
        return;
    }
    std::string arch = std::string("--offload-arch=") + arch_name;
    const char* opts[] = {arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 5, opts);
    size_t logsz = 0;
    if (hiprtcGetProgramLogSize(prog, &logsz) == HIPRTC_SUCCESS && logsz > 1) {
        out.log.resize(logsz);
        (void)hiprtcGetProgramLog(prog, &out.log[0]);
    }
    size_t codesz = 0;
    if (rc == HIPRTC_SUCCESS && hiprtcGetCodeSize(prog, &codesz) == HIPRTC_SUCCESS && codesz) {
        out.code.resize(codesz);
        if (hiprtcGetCode(prog, out.code.data()) != HIPRTC_SUCCESS) out.code.clear();
    }
    (void)hiprtcDestroyProgram(&prog);
    out.compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static void jit_load(tsq_ctx::JitEntry& out) {
    if (out.code.empty()) return;
    const auto t0 = std::chrono::steady_clock::now();
    if (hipModuleLoadData(&out.mod, out.code.data()) != hipSuccess) {
        out.mod = nullptr;
        out.log += "\nhipModuleLoadData failed";
    } else {
        if (hipModuleGetFunction(&out.f_expr, out.mod, "jit_expr") != hipSuccess) out.f_expr = nullptr;
        if (hipModuleGetFunction(&out.f_filter, out.mod, "jit_filter") != hipSuccess) out.f_filter = nullptr;
    }
    std::vector<char>().swap(out.code);
    out.compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// The module belongs to the context's plan cache (tsq_internal.h) and lives until the context is destroyed.  wait (TSQ_JIT_FORCE): the
// caller compiles (or waits for the helper thread's compile); otherwise (TSQ_JIT_AUTO, round 6) the compile starts on a helper thread and
// the call returns at once — the interpreter kernels serve the handle until the code object is there, and no Next call ever waits for
// hiprtc (~250 ms per distinct tree).  Returns true once the entry is final (loaded or failed).
static bool jit_prepare(tsq_expr* e, bool wait) {
    if (e->jit_tried) return true;
    tsq_ctx* ctx = e->ctx;
    if (e->jit_src.empty()) e->jit_src = jit_source(e->progs, (int)tsq_knob(ctx, TSQ_KNOB_JIT_VARIANT, TSQ_JIT_VARIANT_DEFAULT));
    std::lock_guard<std::mutex> g(ctx->jit_mu);
    tsq_ctx::JitEntry& ent = ctx->jit_cache.try_emplace(e->jit_src).first->second;
    int st = ent.state.load(std::memory_order_acquire);
    if (st == 0) {
        if (wait) {
            jit_compile_code(ctx->prop.gcnArchName, e->jit_src, ent);
            st = 2;
        } else {
            ent.state.store(1, std::memory_order_release);
            const std::string arch = ctx->prop.gcnArchName;
            const std::string* src = &ctx->jit_cache.find(e->jit_src)->first;  // (the map's own copy: nodes of an unordered_map never move)
            // a process that ends while a compile is still running (a short script that never destroys its context) waits for it first:
            // hiprtc's own teardown must not start under a running compile.  Handlers registered later run earlier, so this one runs
            // before the destructors of the libraries loaded at start-up.
            static std::atomic<int> workers{0};
            static std::once_flag at_exit_once;
            std::call_once(at_exit_once, [] {
                std::atexit([] {
                    for (int i = 0; i < 20000 && workers.load(std::memory_order_acquire) > 0; i++) std::this_thread::sleep_for(std::chrono::milliseconds(1));
                });
            });
            workers.fetch_add(1, std::memory_order_acq_rel);
            ent.worker = std::thread([arch, src, &ent]() {
                jit_compile_code(arch, *src, ent);
                ent.state.store(2, std::memory_order_release);
                workers.fetch_sub(1, std::memory_order_acq_rel);
            });
            return false;
        }
    }
    if (st == 1) {
        if (!wait) return false;
        if (ent.worker.joinable()) ent.worker.join();
        st = 2;
    }
    if (st == 2) {
        if (ent.worker.joinable()) ent.worker.join();
        jit_load(ent);
        ent.state.store(3, std::memory_order_release);
    }
    e->jit_tried = true;
    e->jit_mod = ent.mod;
    e->jit_expr = ent.f_expr;
    e->jit_filter = ent.f_filter;
    e->jit_log = ent.log;
    e->jit_compile_ms = ent.compile_ms;
    return true;
}

// launches the specialised kernel when policy and availability allow it; returns false -> use the generic kernel
#define TSQ_JIT_AUTO_ROWS (256 << 10)
static bool jit_launch(tsq_expr* e, bool filter, ExprArgs& a, int grid) {
    if (e->jit_mode == TSQ_JIT_OFF) return false;
    if (e->jit_mode == TSQ_JIT_AUTO && e->rows_seen + a.nrows < TSQ_JIT_AUTO_ROWS) return false;  // (a compile costs ~250 ms of one host thread: not for a point query)
    if (!jit_prepare(e, e->jit_mode == TSQ_JIT_FORCE)) return false;
    hipFunction_t f = filter ? e->jit_filter : e->jit_expr;
    if (!f) return false;
    void* params[] = {&a};
    if (hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, 256, 1, 1, 0, e->ctx->stream, params, nullptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    e->jit_launches++;
    return true;
}

static tsq_status expr_run(tsq_expr* e, bool filter, const tsq_col* in_cols, int32_t n_cols, int64_t nrows, const int32_t* sel,
                           tsq_col* out, uint8_t* selected_out, uint8_t* isnull_out, int64_t* div0_out, bool str_root = false,
                           int64_t cap_bytes = 0, int64_t* bytes_out = nullptr) {
    tsq_ctx* ctx = e->ctx;
    tsq_handle_hdr* h = &e->hdr;
    if (nrows < 0 || (nrows > 0 && n_cols > 0 && !in_cols)) return tsq_fail(h, TSQ_ERR_INVALID, "bad arguments");
    if (div0_out) *div0_out = 0;
    if (bytes_out) *bytes_out = 0;
    if (nrows == 0) {
        if (str_root && out && out->offsets && !(out->flags & TSQ_COL_DEVICE)) out->offsets[0] = 0;
        if (str_root && out) { out->length = 0; out->type = TSQ_BYTES; }
        return TSQ_OK;
    }
    for (size_t p = 0; p < e->progs.size(); p++) {
        if ((e->progs[p].result_type == TSQ_BYTES) != str_root)
            return tsq_fail(h, str_root ? TSQ_ERR_INVALID : TSQ_ERR_UNSUPPORTED,
                            str_root ? "tsq_expr_eval_str needs a string-valued root (result_type TSQ_BYTES)" : "a string-valued root is evaluated by tsq_expr_eval_str");
        const char* why = "";
        int32_t ctypes[TSQ_MAX_COLS];
        for (int c = 0; c < n_cols && c < TSQ_MAX_COLS; c++) ctypes[c] = in_cols[c].type;
        tsq_status s = tsq_validate_prog(e->progs[p], n_cols, &why, n_cols <= TSQ_MAX_COLS ? ctypes : nullptr);
        if (s != TSQ_OK) return tsq_fail(h, s, why);
    }
    TSQ_HIP(h, hipSetDevice(ctx->device));
    // physical rows needed: with sel, rows up to max(sel)+1 are touched; the caller's columns say how long they are
    int64_t phys = nrows;
    for (int c = 0; c < n_cols; c++) phys = std::max<int64_t>(phys, in_cols[c].length);
    ExprArgs a;
    memset(&a, 0, sizeof a);
    bool dev = false;
    TSQ_TRY(expr_inputs(e, in_cols, n_cols, phys, a.in, &dev));
    if (!sel)
        for (int c = 0; c < n_cols; c++)
            if (in_cols[c].length < nrows) return tsq_fail(h, TSQ_ERR_INVALID, "column shorter than nrows");
    a.progs = e->progs_d.as<tsq_expr_prog>();
    a.n_progs = (int32_t)e->progs.size();
    a.nrows = nrows;
    if (sel) {
        if (dev) a.sel = sel;  // device-resident selection vector
        else {
            TSQ_TRY(e->sel_d.reserve(ctx, h, (size_t)nrows * 4 + 16));
            TSQ_HIP(h, hipMemcpyAsync(e->sel_d.p, sel, (size_t)nrows * 4, hipMemcpyHostToDevice, ctx->stream));
            a.sel = e->sel_d.as<int32_t>();
        }
    }
    a.counters = e->counters.as<unsigned long long>();
    ctx->pinned[0] = TSQ_ERRWORD_NONE;
    ctx->pinned[1] = 0;
    ctx->pinned[2] = 0;
    TSQ_HIP(h, hipMemcpyAsync(a.counters, ctx->pinned, 24, hipMemcpyHostToDevice, ctx->stream));
    int grid = tsq_grid_for(ctx, nrows, 256);
    {
        static const int per_cu[8] = {8, 4, 16, 32, 2, 8, 8, 8};
        const int gv = ((int)tsq_knob(ctx, TSQ_KNOB_JIT_VARIANT, TSQ_JIT_VARIANT_DEFAULT) >> 4) & 7;
        if (gv) grid = (int)std::min<int64_t>((nrows + 255) / 256, (int64_t)ctx->num_cus * per_cu[gv]);
    }
    if (!filter) {
        const bool odev = out->flags & TSQ_COL_DEVICE;
        if (odev != dev) return tsq_fail(h, TSQ_ERR_INVALID, "output placement (host/device) must match the inputs");
        if (!out->null_bitmap || (!str_root && !out->data) || (str_root && (!out->offsets || (!out->data && cap_bytes > 0))))
            return tsq_fail(h, TSQ_ERR_INVALID, "out needs data and null_bitmap buffers (and offsets for a string-valued root)");
        uint64_t* od = (uint64_t*)out->data;
        if (!odev || str_root) {
            TSQ_TRY(e->out_data.reserve(ctx, h, (size_t)nrows * 8 + 16));
            od = e->out_data.as<uint64_t>();
        }
        TSQ_TRY(e->out_nn.reserve(ctx, h, (size_t)nrows + 16));
        a.out_data = od;
        a.out_notnull = e->out_nn.as<uint8_t>();
        uint8_t* ob = out->null_bitmap;
        if (!odev) {
            TSQ_TRY(e->out_bitmap.reserve(ctx, h, tsq_bitmap_bytes(nrows) + 16));
            ob = e->out_bitmap.as<uint8_t>();
        }
        // the specialised kernel may write the bitmap's whole words itself (four rows per lane, eight lanes per word) and says in counters[2]
        // where the pack pass has to start; the interpreter leaves the word at 0 and every row comes from its flag byte
        const bool bits_ok = !str_root && nrows >= 4096 && ((uintptr_t)ob & 3u) == 0 && ((uintptr_t)a.out_notnull & 15u) == 0;
        a.out_bits = bits_ok ? reinterpret_cast<uint32_t*>(ob) : nullptr;
        if (!jit_launch(e, false, a, grid)) hipLaunchKernelGGL(k_expr_eval, dim3(grid), dim3(256), 0, ctx->stream, a);
        TSQ_HIP(h, hipGetLastError());
        if (str_root) {
            // references -> lengths -> offsets (exclusive scan) -> bytes; the caller's buffers are written only when the bytes fit
            StrRootArgs sa;
            memset(&sa, 0, sizeof sa);
            sa.refs = od;
            sa.notnull = a.out_notnull;
            sa.nrows = nrows;
            sa.in = a.in;
            sa.prog = e->progs_d.as<tsq_expr_prog>();
            TSQ_TRY(e->str_offs.reserve(ctx, h, ((size_t)nrows + 2) * 8));
            sa.offs = e->str_offs.as<int64_t>();
            hipLaunchKernelGGL(k_expr_str_len, dim3(grid), dim3(256), 0, ctx->stream, sa);
            TSQ_HIP(h, hipGetLastError());
            TSQ_TRY(tsq_launch_scan64(ctx, h, sa.offs, nrows, e->scan_tmp));
            TSQ_HIP(h, hipMemcpyAsync(ctx->pinned + 2, sa.offs + nrows, 8, hipMemcpyDeviceToHost, ctx->stream));
            TSQ_HIP(h, hipMemcpyAsync(ctx->pinned, a.counters, 16, hipMemcpyDeviceToHost, ctx->stream));
            TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
            e->launches++;
            e->rows_seen += nrows;
            if (div0_out) *div0_out = (int64_t)ctx->pinned[1];
            TSQ_TRY(expr_status(e, ctx->pinned[0]));
            const int64_t nbytes = (int64_t)ctx->pinned[2];
            if (bytes_out) *bytes_out = nbytes;
            if (nbytes > cap_bytes) return tsq_fail(h, TSQ_ERR_INVALID, "tsq_expr_eval_str: the result needs " + std::to_string(nbytes) + " data bytes");
            uint8_t* dd = (uint8_t*)out->data;
            if (!odev) {
                TSQ_TRY(e->str_data.reserve(ctx, h, (size_t)nbytes + 64));
                dd = e->str_data.as<uint8_t>();
            }
            sa.data = dd;
            if (nbytes > 0) {
                if (nbytes / nrows > 32) hipLaunchKernelGGL(k_expr_str_copy<true>, dim3(ctx->num_cus * 8), dim3(256), 0, ctx->stream, sa);
                else hipLaunchKernelGGL(k_expr_str_copy<false>, dim3(grid), dim3(256), 0, ctx->stream, sa);
                TSQ_HIP(h, hipGetLastError());
            }
            uint8_t* sb = out->null_bitmap;
            if (!odev) {
                TSQ_TRY(e->out_bitmap.reserve(ctx, h, tsq_bitmap_bytes(nrows) + 16));
                sb = e->out_bitmap.as<uint8_t>();
            }
            TSQ_TRY(tsq_launch_pack_bitmap(ctx, h, a.out_notnull, sb, nrows));
            if (odev) {
                TSQ_HIP(h, hipMemcpyAsync(out->offsets, sa.offs, ((size_t)nrows + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
            } else {
                TSQ_HIP(h, hipMemcpyAsync(out->offsets, sa.offs, ((size_t)nrows + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
                if (nbytes > 0) TSQ_HIP(h, hipMemcpyAsync(out->data, dd, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream));
                TSQ_HIP(h, hipMemcpyAsync(out->null_bitmap, sb, tsq_bitmap_bytes(nrows), hipMemcpyDeviceToHost, ctx->stream));
            }
            TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
            out->length = nrows;
            out->elem_size = -1;
            out->type = TSQ_BYTES;
            return TSQ_OK;
        }
        TSQ_TRY(tsq_launch_pack_bitmap(ctx, h, a.out_notnull, ob, nrows, bits_ok ? a.counters + 2 : nullptr));
        if (!odev) {
            TSQ_TRY(e->hout.reserve(h, (size_t)nrows * 8 + tsq_bitmap_bytes(nrows) + 32));
            TSQ_HIP(h, hipMemcpyAsync(e->hout.p, od, (size_t)nrows * 8, hipMemcpyDeviceToHost, ctx->stream));
            TSQ_HIP(h, hipMemcpyAsync((char*)e->hout.p + (size_t)nrows * 8, ob, tsq_bitmap_bytes(nrows), hipMemcpyDeviceToHost, ctx->stream));
        }
    } else {
        TSQ_TRY(e->out_sel.reserve(ctx, h, (size_t)nrows + 16));
        a.out_selected = dev ? selected_out : e->out_sel.as<uint8_t>();
        if (isnull_out) {
            TSQ_TRY(e->out_isnull.reserve(ctx, h, (size_t)nrows + 16));
            a.out_isnull = dev ? isnull_out : e->out_isnull.as<uint8_t>();
        }
        if (!jit_launch(e, true, a, grid)) hipLaunchKernelGGL(k_filter_eval, dim3(grid), dim3(256), 0, ctx->stream, a);
        TSQ_HIP(h, hipGetLastError());
        if (!dev) {
            TSQ_TRY(e->hflags.reserve(h, (size_t)nrows * 2 + 32));
            TSQ_HIP(h, hipMemcpyAsync(e->hflags.p, a.out_selected, (size_t)nrows, hipMemcpyDeviceToHost, ctx->stream));
            if (isnull_out) TSQ_HIP(h, hipMemcpyAsync((char*)e->hflags.p + nrows, a.out_isnull, (size_t)nrows, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    e->launches++;
    e->rows_seen += nrows;
    TSQ_HIP(h, hipMemcpyAsync(ctx->pinned, a.counters, 16, hipMemcpyDeviceToHost, ctx->stream));
    TSQ_HIP(h, hipStreamSynchronize(ctx->stream));
    if (div0_out) *div0_out = (int64_t)ctx->pinned[1];
    TSQ_TRY(expr_status(e, ctx->pinned[0]));
    if (!filter) {
        if (!(out->flags & TSQ_COL_DEVICE)) {
            memcpy(out->data, e->hout.p, (size_t)nrows * 8);
            memcpy(out->null_bitmap, (char*)e->hout.p + (size_t)nrows * 8, tsq_bitmap_bytes(nrows));
        }
        out->length = nrows;
        out->elem_size = 8;
        out->type = e->progs[0].result_type == TSQ_F64 ? TSQ_F64 : (e->progs[0].result_unsigned ? TSQ_U64 : TSQ_I64);
    } else if (!dev) {
        memcpy(selected_out, e->hflags.p, (size_t)nrows);
        if (isnull_out) memcpy(isnull_out, (char*)e->hflags.p + nrows, (size_t)nrows);
    }
    return TSQ_OK;
}

TSQ_API tsq_status tsq_expr_eval(tsq_expr* e, const tsq_col* in_cols, int32_t n_cols, int64_t nrows, const int32_t* sel, tsq_col* out,
                                 int64_t* div_by_zero_warnings) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(e, TSQ_MAGIC_EXPR));
    if (!e || e->hdr.magic != TSQ_MAGIC_EXPR) return TSQ_ERR_INVALID;
    if (!out) return tsq_fail(&e->hdr, TSQ_ERR_INVALID, "out == NULL");
    if (e->progs.size() != 1) return tsq_fail(&e->hdr, TSQ_ERR_INVALID, "tsq_expr_eval needs a single program (use tsq_filter_eval for CNF lists)");
    return expr_run(e, false, in_cols, n_cols, nrows, sel, out, nullptr, nullptr, div_by_zero_warnings);
}

TSQ_API tsq_status tsq_expr_eval_str(tsq_expr* e, const tsq_col* in_cols, int32_t n_cols, int64_t nrows, const int32_t* sel, tsq_col* out,
                                     int64_t cap_bytes, int64_t* bytes_out, int64_t* div_by_zero_warnings) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(e, TSQ_MAGIC_EXPR));
    if (!e || e->hdr.magic != TSQ_MAGIC_EXPR) return TSQ_ERR_INVALID;
    if (!out || cap_bytes < 0) return tsq_fail(&e->hdr, TSQ_ERR_INVALID, "out == NULL or cap_bytes < 0");
    if (e->progs.size() != 1) return tsq_fail(&e->hdr, TSQ_ERR_INVALID, "tsq_expr_eval_str needs a single program");
    return expr_run(e, false, in_cols, n_cols, nrows, sel, out, nullptr, nullptr, div_by_zero_warnings, true, cap_bytes, bytes_out);
}

TSQ_API tsq_status tsq_filter_eval(tsq_expr* e, const tsq_col* in_cols, int32_t n_cols, int64_t nrows, const int32_t* sel,
                                   uint8_t* selected_out, uint8_t* isnull_out, int64_t* div_by_zero_warnings) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(e, TSQ_MAGIC_EXPR));
    if (!e || e->hdr.magic != TSQ_MAGIC_EXPR) return TSQ_ERR_INVALID;
    if (!selected_out) return tsq_fail(&e->hdr, TSQ_ERR_INVALID, "selected_out == NULL");
    return expr_run(e, true, in_cols, n_cols, nrows, sel, nullptr, selected_out, isnull_out, div_by_zero_warnings);
}

TSQ_API tsq_status tsq_expr_set_jit(tsq_expr* e, int32_t mode) {
    if (!e || e->hdr.magic != TSQ_MAGIC_EXPR) return TSQ_ERR_INVALID;
    if (mode < TSQ_JIT_AUTO || mode > TSQ_JIT_FORCE) return tsq_fail(&e->hdr, TSQ_ERR_INVALID, "mode must be -1 (auto), 0 (off) or 1 (force)");
    e->jit_mode = mode;
    return TSQ_OK;
}
TSQ_API double tsq_expr_jit_compile_ms(tsq_expr* e) {
    if (!e || e->hdr.magic != TSQ_MAGIC_EXPR) return -1.0;
    return e->jit_tried ? e->jit_compile_ms : 0.0;
}
TSQ_API int64_t tsq_expr_jit_launches(tsq_expr* e) {
    if (!e || e->hdr.magic != TSQ_MAGIC_EXPR) return -1;
    if (e->jit_tried && !e->jit_mod) tsq_fail(&e->hdr, TSQ_OK, std::string("expression JIT unavailable: ") + e->jit_log.substr(0, 600));
    return e->jit_launches;
}

TSQ_API void tsq_expr_destroy(tsq_expr* e) {
    tsq_ctx_lock _api_lock(tsq_ctx_of(e, TSQ_MAGIC_EXPR));
    if (!e || e->hdr.magic != TSQ_MAGIC_EXPR) return;
    (void)hipSetDevice(e->ctx->device);
    (void)hipStreamSynchronize(e->ctx->stream);
    e->progs_d.release();
    e->str_offs.release();
    e->str_data.release();
    e->scan_tmp.release();
    e->counters.release();
    for (auto& c : e->icols) c.release();
    e->sel_d.release();
    e->out_data.release();
    e->out_nn.release();
    e->out_bitmap.release();
    e->out_sel.release();
    e->out_isnull.release();
    e->hout.release();
    e->hflags.release();
    // e->jit_mod stays in the context's plan cache (unloading a module right after its handle died made later,
    // unrelated kernels fault intermittently on ROCm 7.2 — see tsq_internal.h)
    e->hdr.magic = 0;
    delete e;
}
