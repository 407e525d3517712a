"""GPU parity of the twelve string signatures (LT/LE/GT/GE/EQ/NEString, IfNullString, IfString, InString,
StringIsNull, Strcmp, Length) through the C-ABI (tsq_expr_eval / tsq_filter_eval with TSQ_BYTES input columns) against the
oracle's node-at-a-time restatement, bit for bit: values, NULL flags, selection vectors, CNF filters, interpreter and
run-time specialised kernels."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column, StrColumn

from .test_oracle_string_golden import rand_strs

pytestmark = pytest.mark.gpu

S0, S1, S2 = E.Column(0, abi.BYTES), E.Column(1, abi.BYTES), E.Column(2, abi.BYTES)
I3 = E.Column(3, abi.I64)


def inputs(seed, n):
    rng = np.random.default_rng(seed)
    long_a = [None if v is None else v * int(rng.integers(1, 9)) for v in rand_strs(rng, n)]  # up to ~280 bytes, shared prefixes
    return Chunk([StrColumn(long_a), StrColumn(rand_strs(rng, n)), StrColumn(rand_strs(rng, n, 0.4)),
                  Column(abi.I64, rng.integers(-2, 3, n), rng.random(n) > 0.2)])


EXPRS = {
    "lt(col,col)": E.ScalarFunction("lt", S0, S1), "le": E.ScalarFunction("le", S0, S1), "gt": E.ScalarFunction("gt", S1, S0),
    "ge": E.ScalarFunction("ge", S0, S1), "eq": E.ScalarFunction("eq", S1, S2), "ne": E.ScalarFunction("ne", S1, S2),
    "lt(col,const)": E.ScalarFunction("lt", S1, E.Constant(b"ab\x80")), "eq(const,col)": E.ScalarFunction("eq", E.Constant(""), S2),
    "strcmp": E.ScalarFunction("strcmp", S0, S1), "strcmp(col,null)": E.ScalarFunction("strcmp", S0, E.Constant(None, E.ETString)),
    "length": E.ScalarFunction("length", S0), "isnull": E.ScalarFunction("isnull", S2),
    "ifnull": E.ScalarFunction("length", E.ScalarFunction("ifnull", S2, S0)),
    "if": E.ScalarFunction("strcmp", E.ScalarFunction("if", I3, S0, S1), E.ScalarFunction("ifnull", S2, E.Constant("abc"))),
    "in(consts)": E.ScalarFunction("in", S1, E.Constant("a"), E.Constant("ab"), E.Constant(b"\xff"), E.Constant("")),
    "in(cols,null)": E.ScalarFunction("in", S1, S2, E.Constant(None, E.ETString), S0),
    "mixed": E.ScalarFunction("plus", E.ScalarFunction("length", S0), E.ScalarFunction("mul", I3, E.ScalarFunction("strcmp", S1, S2))),
}


@pytest.mark.parametrize("name", sorted(EXPRS))
@pytest.mark.parametrize("jit", [abi.JIT_OFF, abi.JIT_FORCE])
def test_string_expression_vs_oracle(ctx, orc, name, jit):
    chk = inputs(7, 20_000)
    e = EXPRS[name]
    want, _ = orc.expr_eval(E.compile_expr(e), chk)
    ce = E.CompiledExpr(ctx, [e], jit=jit)
    try:
        got = ce.VecEval(chk)
        assert got.values() == want.values(), name
        if jit == abi.JIT_FORCE:
            assert ce.jit_launches() >= 1
        # with a selection vector (chunk.go:319-331): logical rows only
        sel = np.random.default_rng(9).permutation(20_000)[:7777].astype(np.int32)
        sel.sort()
        chk_sel = Chunk(chk.columns, sel=sel)
        want_sel, _ = orc.expr_eval(E.compile_expr(e), chk_sel)
        assert ce.VecEval(chk_sel).values() == want_sel.values()
    finally:
        ce.close()


def test_string_cnf_filter_vs_oracle(ctx, orc):
    chk = inputs(11, 50_000)
    conj = [E.ScalarFunction("ge", S0, S1), E.ScalarFunction("or", E.ScalarFunction("in", S1, E.Constant("a"), S2), E.ScalarFunction("gt", E.ScalarFunction("length", S0), I3)),
            E.ScalarFunction("not", E.ScalarFunction("isnull", S2))]
    progs = E.compile_list(conj)
    want_sel, want_nul, _ = orc.filter_eval(progs, len(conj), chk)
    for jit in (abi.JIT_OFF, abi.JIT_FORCE):
        ce = E.CompiledExpr(ctx, conj, jit=jit)
        try:
            sel, nul = ce.VectorizedFilter(chk, want_nulls=True)
            assert (sel == want_sel).all() and (nul == want_nul).all()
        finally:
            ce.close()


def test_empty_and_all_null_string_columns(ctx, orc):
    for vals in ([], [None, None, None], ["", "", None]):
        chk = Chunk([StrColumn(vals), StrColumn(vals), StrColumn(vals), Column(abi.I64, np.zeros(len(vals), np.int64))])
        e = E.ScalarFunction("eq", S0, S1)
        ce = E.CompiledExpr(ctx, [e])
        try:
            assert ce.VecEval(chk).values() == orc.expr_eval(E.compile_expr(e), chk)[0].values()
        finally:
            ce.close()


def test_string_leaf_on_a_fixed_width_column_is_rejected(ctx):
    from tinysql_amd import _lib
    chk = Chunk([Column(abi.I64, np.arange(4)), StrColumn(["a"] * 4), StrColumn(["a"] * 4), Column(abi.I64, np.arange(4))])
    ce = E.CompiledExpr(ctx, [E.ScalarFunction("length", S0)])  # column 0 is BIGINT here
    try:
        with pytest.raises(_lib.TsqError) as ei:
            ce.VecEval(chk)
        assert ei.value.status == abi.ERR_INVALID
    finally:
        ce.close()


# ------------------------------------------------------------------ string-valued ROOTS (tsq_expr_eval_str)
# builtinIfStringSig / builtinIfNullStringSig.vecEvalString (builtin_control_vec_generated.go:209, :81), Column.VecEvalString with a
# selection vector (column.go:111), Constant.VecEvalString (constant.go:86): the result is a var-len COLUMN; its offsets, its data
# bytes and its NULL flags are compared with the oracle's.
STR_ROOTS = {
    "if(int,col,col)": E.ScalarFunction("if", I3, S0, S1),
    "if(cmp,col,const)": E.ScalarFunction("if", E.ScalarFunction("lt", S0, S1), S2, E.Constant("otherwise")),
    "if(isnull,const,col)": E.ScalarFunction("if", E.ScalarFunction("isnull", S2), E.Constant(""), S2),
    "ifnull(col,col)": E.ScalarFunction("ifnull", S2, S0),
    "ifnull(col,const)": E.ScalarFunction("ifnull", S2, E.Constant(b"\x00nul\xff")),
    "ifnull(ifnull)": E.ScalarFunction("ifnull", S2, E.ScalarFunction("ifnull", S1, E.Constant(None, E.ETString))),
    "if(nested)": E.ScalarFunction("if", I3, E.ScalarFunction("ifnull", S2, S1), E.ScalarFunction("if", E.ScalarFunction("eq", S1, S2), S0, E.Constant("x" * 40))),
    "column": S1,
    "constant": E.Constant("a constant"),
    "null constant": E.Constant(None, E.ETString),
}


def _same_column(got, want):
    go, gd, gn = got
    wo, wd, wn = want[0], want[1], want[2]
    assert (gn == wn).all()
    assert (go == wo).all()
    assert bytes(gd) == bytes(wd)


@pytest.mark.parametrize("name", sorted(STR_ROOTS))
@pytest.mark.parametrize("jit", [abi.JIT_OFF, abi.JIT_FORCE])
def test_string_valued_root_column_bytes_vs_oracle(ctx, orc, name, jit):
    chk = inputs(13, 20_000)
    e = STR_ROOTS[name]
    prog = E.compile_expr(e)
    assert prog.result_type == abi.BYTES
    ce = E.CompiledExpr(ctx, [e], jit=jit)
    try:
        _same_column(ce.VecEvalString(chk), orc.expr_eval_str(prog, chk))
        sel = np.random.default_rng(5).permutation(20_000)[:6001].astype(np.int32)  # an UNSORTED selection vector: rows are gathered in its order
        chk_sel = Chunk(chk.columns, sel=sel)
        _same_column(ce.VecEvalString(chk_sel), orc.expr_eval_str(prog, chk_sel))
    finally:
        ce.close()


def test_string_valued_root_long_cells_and_edges(ctx, orc):
    # cells of a few KB (one wave copies one cell), an empty chunk, one row, all-NULL results
    rng = np.random.default_rng(3)
    big = [None if rng.random() < 0.1 else bytes(rng.integers(0, 256, int(rng.integers(0, 5000)), dtype=np.uint8)) for _ in range(300)]
    chk = Chunk([StrColumn(big), StrColumn(big[::-1]), StrColumn([None] * 300), Column(abi.I64, rng.integers(0, 2, 300), rng.random(300) > 0.3)])
    for e in (E.ScalarFunction("if", I3, S0, S1), E.ScalarFunction("ifnull", S2, S0), S2):
        ce = E.CompiledExpr(ctx, [e])
        try:
            _same_column(ce.VecEvalString(chk), orc.expr_eval_str(E.compile_expr(e), chk))
            for m in (0, 1):
                part = chk.slice(0, m)
                _same_column(ce.VecEvalString(part), orc.expr_eval_str(E.compile_expr(e), part))
        finally:
            ce.close()


def test_string_root_and_numeric_entry_points_do_not_mix(ctx):
    from tinysql_amd import _lib
    chk = inputs(1, 100)
    ce = E.CompiledExpr(ctx, [E.ScalarFunction("ifnull", S2, S0)])
    try:
        with pytest.raises(_lib.TsqError) as ei:
            ce.VecEval(chk)  # tsq_expr_eval: Int / Real roots only
        assert ei.value.status == abi.ERR_UNSUPPORTED
    finally:
        ce.close()
    ce = E.CompiledExpr(ctx, [E.ScalarFunction("length", S0)])
    try:
        with pytest.raises(_lib.TsqError) as ei:
            ce.VecEvalString(chk)
        assert ei.value.status == abi.ERR_INVALID
    finally:
        ce.close()
