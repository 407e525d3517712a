"""Mirror of package `expression` for the harness: expression trees -> tsq_expr_prog bytecode.

The Go shim does exactly this flattening inside `expression`: every
ScalarFunction whose signature has a vectorized implementation (expression/builtin_*_vec.go)
becomes one postfix opcode; anything else (strings, Set/GetVar, mixed int/real arguments that
the reference routes through implicit conversions) raises Unsupported and the caller keeps the
Go evaluator.  Evaluation itself always happens in libtsq on the GPU.
"""
import ctypes as C
import struct

import numpy as np

from . import _abi as abi
from . import _lib
from .chunk import Column as ChunkColumn, make_cols, pack_bitmap, unpack_bitmap  # noqa: F401

ETInt, ETReal, ETString = "int", "real", "string"


class Unsupported(Exception):
    """Expression is not eligible for the GPU path (fall back to the Go evaluator)."""


class Expression:
    eval_type = ETInt
    unsigned = False

    def Vectorized(self):  # expression.go:43-55 VecExpr.Vectorized
        return True


class Column(Expression):
    """expression.Column (expression/column.go): a bare input column."""

    def __init__(self, index, tp):
        self.index = index
        self.tp = tp
        self.eval_type = ETString if tp == abi.BYTES else (ETReal if tp in (abi.F32, abi.F64) else ETInt)
        self.unsigned = tp == abi.U64


class Constant(Expression):
    """expression.Constant (expression/constant.go); value None = NULL."""

    def __init__(self, value, eval_type=None, unsigned=False):
        if eval_type is None:
            eval_type = ETString if isinstance(value, (str, bytes)) else (ETReal if isinstance(value, float) else ETInt)
        self.value = value
        self.eval_type = eval_type
        self.unsigned = unsigned


_CMP = {"lt": 0, "le": 1, "gt": 2, "ge": 3, "eq": 4, "ne": 5}


class ScalarFunction(Expression):
    """expression.ScalarFunction; `name` is the ast.* function name (expression/builtin.go:332-363)."""

    def __init__(self, name, *args, no_unsigned_subtraction=False):
        self.name = name.lower()
        self.args = list(args)
        self.force_signed = no_unsigned_subtraction
        self._infer()

    def _infer(self):
        n, a = self.name, self.args
        ets = [x.eval_type for x in a]
        if n in ("plus", "minus", "mul"):
            if len(set(ets)) != 1 or ets[0] == ETString:
                raise Unsupported("mixed int/real/string arithmetic needs an implicit conversion")
            self.eval_type = ets[0]
            # builtin_arithmetic.go:112-133,216-237,330-355: int result is UNSIGNED if either side is
            self.unsigned = self.eval_type == ETInt and (a[0].unsigned or a[1].unsigned)
            if n == "minus" and self.force_signed:
                self.unsigned = False
        elif n == "div":  # builtin_arithmetic.go:435-444: always real
            if ets != [ETReal, ETReal]:
                raise Unsupported("DIV of non-real arguments needs an implicit conversion")
            self.eval_type = ETReal
        elif n in _CMP:
            if len(set(ets)) != 1:  # getBaseCmpType (builtin_compare.go:60-110) would pick Real + conversions
                raise Unsupported("mixed int/real comparison needs an implicit conversion")
            self.eval_type = ETInt
        elif n in ("and", "or"):
            if ets != [ETInt, ETInt]:
                raise Unsupported("logic operators over non-int arguments")
            self.eval_type = ETInt
        elif n in ("not", "unaryminus"):
            if ets[0] == ETString:
                raise Unsupported("NOT / unary minus of a string needs an implicit conversion")
            self.eval_type = ETInt if n == "not" else ets[0]
            self.unsigned = False
        elif n == "strcmp":  # builtin_string.go strcmpFunctionClass -> builtinStrcmpSig
            if ets != [ETString, ETString]:
                raise Unsupported("STRCMP of non-string arguments needs an implicit conversion")
            self.eval_type = ETInt
        elif n == "length":  # builtinLengthSig
            if ets != [ETString]:
                raise Unsupported("LENGTH of a non-string argument needs an implicit conversion")
            self.eval_type = ETInt
        elif n == "isnull":
            self.eval_type = ETInt
        elif n == "ifnull":
            if len(set(ets)) != 1:
                raise Unsupported("IFNULL of mixed types")
            self.eval_type = ets[0]
            self.unsigned = a[0].unsigned and a[1].unsigned
        elif n == "if":
            if ets[0] != ETInt or ets[1] != ets[2]:
                raise Unsupported("IF needs an int condition and same-typed branches")
            self.eval_type = ets[1]
            self.unsigned = a[1].unsigned and a[2].unsigned
        elif n == "in":
            if len(set(ets)) != 1:
                raise Unsupported("IN of mixed types")
            if len(a) - 1 > 31:
                raise Unsupported("IN list longer than 31 items")
            self.eval_type = ETInt
        else:
            raise Unsupported("function %r has no GPU signature" % n)


def _emit(e, ops, consts):
    if isinstance(e, Column):
        op = abi.OP_COL_STR if e.eval_type == ETString else (abi.OP_COL_REAL if e.eval_type == ETReal else abi.OP_COL_INT)
        ops.append((op, 0, e.index, 0))
        return
    if isinstance(e, Constant):
        if e.value is None:
            op = abi.OP_CONST_NULL_STR if e.eval_type == ETString else (abi.OP_CONST_NULL_REAL if e.eval_type == ETReal else abi.OP_CONST_NULL_INT)
            ops.append((op, 0, 0, 0))
            return
        if e.eval_type == ETString:
            b = e.value.encode() if isinstance(e.value, str) else bytes(e.value)
            pool = consts.pool
            if len(pool) + len(b) > abi.EXPR_STR_POOL:
                raise Unsupported("string constants exceed the program's %d-byte pool" % abi.EXPR_STR_POOL)
            consts.append((len(pool) << 32) | len(b))
            pool.extend(b)
            ops.append((abi.OP_CONST_STR, 0, len(consts) - 1, 0))
            return
        if e.eval_type == ETReal:
            bits = struct.unpack("<q", struct.pack("<d", float(e.value)))[0]
            op = abi.OP_CONST_REAL
        else:
            v = int(e.value)
            bits = v - (1 << 64) if v >= (1 << 63) else v
            op = abi.OP_CONST_INT
        consts.append(bits)
        ops.append((op, 0, len(consts) - 1, 0))
        return
    n, a = e.name, e.args
    for x in a:
        _emit(x, ops, consts)
    real = a[0].eval_type == ETReal
    isstr = a[0].eval_type == ETString
    flags = 0
    if len(a) >= 1 and a[0].unsigned:
        flags |= abi.F_LHS_UNSIGNED
    if len(a) >= 2 and a[1].unsigned:
        flags |= abi.F_RHS_UNSIGNED
    if n == "plus":
        ops.append((abi.OP_PLUS_REAL if real else abi.OP_PLUS_INT, flags, 0, 0))
    elif n == "minus":
        if e.force_signed:
            flags |= abi.F_FORCE_SIGNED
        ops.append((abi.OP_MINUS_REAL if real else abi.OP_MINUS_INT, flags, 0, 0))
    elif n == "mul":
        if real:
            ops.append((abi.OP_MUL_REAL, 0, 0, 0))
        elif a[0].unsigned or a[1].unsigned:  # builtin_arithmetic.go:330-355
            ops.append((abi.OP_MUL_INT_UNSIGNED, flags, 0, 0))
        else:
            ops.append((abi.OP_MUL_INT, flags, 0, 0))
    elif n == "div":
        ops.append((abi.OP_DIV_REAL, 0, 0, 0))
    elif n in _CMP:
        base = abi.OP_LT_STR if isstr else (abi.OP_LT_REAL if real else abi.OP_LT_INT)
        ops.append((base + _CMP[n], 0 if isstr else flags, 0, 0))
    elif n == "strcmp":
        ops.append((abi.OP_STRCMP, 0, 0, 0))
    elif n == "length":
        ops.append((abi.OP_LENGTH, 0, 0, 0))
    elif n == "and":
        ops.append((abi.OP_LOGIC_AND, 0, 0, 0))
    elif n == "or":
        ops.append((abi.OP_LOGIC_OR, 0, 0, 0))
    elif n == "not":
        ops.append((abi.OP_NOT_REAL if real else abi.OP_NOT_INT, 0, 0, 0))
    elif n == "unaryminus":
        ops.append((abi.OP_NEG_REAL if real else abi.OP_NEG_INT, flags, 0, 0))
    elif n == "isnull":
        ops.append((abi.OP_ISNULL_STR if isstr else (abi.OP_ISNULL_REAL if real else abi.OP_ISNULL_INT), 0, 0, 0))
    elif n == "ifnull":
        ops.append((abi.OP_IFNULL_STR if isstr else (abi.OP_IFNULL_REAL if real else abi.OP_IFNULL_INT), 0, 0, 0))
    elif n == "if":
        ops.append((abi.OP_IF_STR if a[1].eval_type == ETString else (abi.OP_IF_REAL if a[1].eval_type == ETReal else abi.OP_IF_INT), 0, 0, 0))
    elif n == "in":
        aux = 0
        for j, item in enumerate(a[1:]):
            if item.unsigned:
                aux |= 1 << j
        if isstr:
            ops.append((abi.OP_IN_STR, 0, len(a) - 1, 0))
        else:
            ops.append((abi.OP_IN_REAL if real else abi.OP_IN_INT, flags & abi.F_LHS_UNSIGNED, len(a) - 1, aux))


def compile_expr(e):
    """Expression tree -> abi.ExprProg (postfix)."""
    class _Consts(list):
        pass

    ops, consts = [], _Consts()
    consts.pool = bytearray()
    _emit(e, ops, consts)
    if len(ops) > abi.EXPR_MAX_OPS or len(consts) > abi.EXPR_MAX_CONSTS:
        raise Unsupported("expression too large for the GPU interpreter")
    p = abi.ExprProg()
    p.n_ops = len(ops)
    p.n_consts = len(consts)
    # a string-valued root (IF / IFNULL of strings, a string column or constant) is evaluated by tsq_expr_eval_str
    p.result_type = abi.BYTES if e.eval_type == ETString else (abi.F64 if e.eval_type == ETReal else abi.I64)
    p.result_unsigned = 1 if e.unsigned else 0
    for i, (opc, fl, arg, aux) in enumerate(ops):
        p.ops[i].opcode, p.ops[i].flags, p.ops[i].arg, p.ops[i].aux = opc, fl, arg, aux
    for i, c in enumerate(consts):
        p.consts[i] = c
    p.n_str_bytes = len(consts.pool)
    for i, b in enumerate(consts.pool):
        p.str_pool[i] = b
    return p


def compile_list(exprs):
    arr = (abi.ExprProg * max(1, len(exprs)))()
    for i, e in enumerate(exprs):
        arr[i] = compile_expr(e)
    return arr


class CompiledExpr:
    """tsq_expr handle: one projection expression or one CNF filter list."""

    def __init__(self, ctx, exprs, jit=None):
        self.ctx = ctx
        self.lib = ctx.lib
        self.exprs = list(exprs)
        self.progs = compile_list(self.exprs)
        h = C.c_void_p()
        _lib.check(self.lib.tsq_expr_compile(ctx.h, self.progs, len(self.exprs), C.byref(h)), ctx.h)
        self.h = h
        if jit is not None:
            _lib.check(self.lib.tsq_expr_set_jit(h, jit), h)
        self.warnings = 0  # StmtCtx.AppendWarning(ErrDivisionByZero) count (errors.go:65-77)

    def jit_compile_ms(self):
        """what hiprtc + the module load of this handle's programs took (0.0: served from the context's cache, or not compiled yet)"""
        return float(self.lib.tsq_expr_jit_compile_ms(self.h))

    def jit_launches(self):
        """launches served by run-time specialised kernels; raises with the hiprtc log when the JIT is unavailable."""
        n = self.lib.tsq_expr_jit_launches(self.h)
        if n == 0:
            msg = self.lib.tsq_last_error(self.h)
            if msg and b"JIT unavailable" in msg:
                raise RuntimeError(msg.decode(errors="replace"))
        return n

    def close(self):
        if self.h:
            self.lib.tsq_expr_destroy(self.h)
            self.h = None

    def VecEval(self, chk):
        """expression.VecEval (expression.go:329-341): returns a chunk.Column with NumRows() rows."""
        n = chk.NumRows()
        e = self.exprs[0]
        tp = abi.F64 if e.eval_type == ETReal else (abi.U64 if e.unsigned else abi.I64)
        keep = []
        cols = make_cols(chk.columns, keep)
        data = np.zeros(max(n, 1), dtype=np.float64 if tp == abi.F64 else np.int64)
        bm = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
        out = abi.Col()
        out.data = data.ctypes.data_as(C.c_void_p)
        out.null_bitmap = bm.ctypes.data_as(C.c_void_p)
        out.length = n
        out.elem_size = 8
        out.type = tp
        w = C.c_int64(0)
        sel = chk.sel.ctypes.data_as(C.c_void_p) if chk.sel is not None else None
        st = self.lib.tsq_expr_eval(self.h, cols, len(chk.columns), n, sel, C.byref(out), C.byref(w))
        self.warnings += w.value
        _lib.check(st, self.h)
        arr = data[:n]
        if tp == abi.U64:
            arr = arr.view(np.uint64)
        return ChunkColumn(tp, arr.copy(), unpack_bitmap(bm, n))

    def VecEvalString(self, chk):
        """expression.VecEvalString (expression.go:329-341 -> builtinIfStringSig / builtinIfNullStringSig.vecEvalString,
        Column.VecEvalString, Constant.VecEvalString): returns (offsets[n + 1], data bytes, notnull bool[n]) — the state of the
        var-len result column (util/chunk/column.go:28-34).  First call asks for the size, second one fills the buffers."""
        n = chk.NumRows()
        keep = []
        cols = make_cols(chk.columns, keep)
        offs = np.zeros(n + 1, dtype=np.int64)
        bm = np.zeros((n + 7) // 8 + 8, dtype=np.uint8)
        sel = chk.sel.ctypes.data_as(C.c_void_p) if chk.sel is not None else None
        out = abi.Col()
        out.offsets = offs.ctypes.data_as(C.c_void_p)
        out.null_bitmap = bm.ctypes.data_as(C.c_void_p)
        out.length, out.elem_size, out.type = n, -1, abi.BYTES
        w, need = C.c_int64(0), C.c_int64(0)
        # ONE evaluation in the common case: a result cell is one of the row's input strings or a constant of the program,
        # so the bytes of the string input columns (+ a constant per row, guessed at 32 B) bound the result; only when the guess was
        # too small the operator reports the size (TSQ_ERR_INVALID with bytes_out > 0) and runs again
        guess = 64 + 32 * n
        for c in chk.columns:
            if getattr(c, "offsets", None) is not None and len(c.offsets):
                guess += int(c.offsets[-1])
        data = np.zeros(guess + 8, dtype=np.uint8)
        out.data = data.ctypes.data_as(C.c_void_p)
        st = self.lib.tsq_expr_eval_str(self.h, cols, len(chk.columns), n, sel, C.byref(out), guess, C.byref(need), C.byref(w))
        if st == abi.ERR_INVALID and need.value > guess:
            data = np.zeros(need.value + 8, dtype=np.uint8)
            out.data = data.ctypes.data_as(C.c_void_p)
            st = self.lib.tsq_expr_eval_str(self.h, cols, len(chk.columns), n, sel, C.byref(out), need.value, C.byref(need), C.byref(w))
        self.warnings += w.value
        _lib.check(st, self.h)
        return offs, data[:need.value], unpack_bitmap(bm, n)

    def VectorizedFilter(self, chk, want_nulls=False):
        """expression.VectorizedFilter (chunk_executor.go:196): returns selected[] (and nulls[])."""
        n = chk.NumRows()
        keep = []
        cols = make_cols(chk.columns, keep)
        selected = np.zeros(max(n, 1), dtype=np.uint8)
        nulls = np.zeros(max(n, 1), dtype=np.uint8)
        w = C.c_int64(0)
        sel = chk.sel.ctypes.data_as(C.c_void_p) if chk.sel is not None else None
        st = self.lib.tsq_filter_eval(self.h, cols, len(chk.columns), n, sel, selected.ctypes.data_as(C.c_void_p),
                                      nulls.ctypes.data_as(C.c_void_p) if want_nulls else None, C.byref(w))
        self.warnings += w.value
        _lib.check(st, self.h)
        if want_nulls:
            return selected[:n].astype(bool), nulls[:n].astype(bool)
        return selected[:n].astype(bool)
