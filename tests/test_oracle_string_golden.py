"""CPU: the oracle's string builtins (oracle/oracle.cpp, ETString opcodes) pinned on the reference's own vectors —
expression/builtin_string_test.go:24-57 (Length: bytes, "你好" = 6), :62-100 (Strcmp incl. NULL and empty arguments), the
NULL protocol of builtin_compare_vec_generated.go:65-555 / builtin_other_vec_generated.go:97-145 (InString) / builtin_control_vec_
generated.go:81-111,209-253 (IfNullString, IfString) — and checked against Python's own bytes order on random columns
(Go's string `<` is byte-wise lexicographic, types/compare.go:115-123)."""
import numpy as np
import pytest

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column, StrColumn

S0, S1 = E.Column(0, abi.BYTES), E.Column(1, abi.BYTES)


def ev(expr, chunk):
    return orc.expr_eval(E.compile_expr(expr), chunk)[0].values()


def test_strcmp_vectors_of_the_reference():
    # builtin_string_test.go:69-82 (the string / NULL rows; the int and float rows go through implicit casts, not this signature)
    rows = [("123", "123", 0), ("123", "1", 1), ("1", "123", -1), ("123", "45", -1), (None, "123", None), ("123", None, None), ("", "123", -1),
            ("123", "", 1), ("", "", 0), ("", None, None), (None, "", None), (None, None, None)]
    chk = Chunk([StrColumn([r[0] for r in rows]), StrColumn([r[1] for r in rows])])
    assert ev(E.ScalarFunction("strcmp", S0, S1), chk) == [r[2] for r in rows]


def test_length_vectors_of_the_reference():
    # builtin_string_test.go:31-35: bytes, not characters; NULL -> NULL
    chk = Chunk([StrColumn(["abc", "你好", None, ""])])
    assert ev(E.ScalarFunction("length", S0), chk) == [3, 6, None, 0]


def rand_strs(rng, n, null_p=0.15):
    alphabet = [b"a", b"b", b"ab", b"\x00", b"\x7f", b"\x80", b"\xff", b"z", b"", b"abcabc"]
    out = []
    for _ in range(n):
        if rng.random() < null_p:
            out.append(None)
        else:
            out.append(b"".join(alphabet[i] for i in rng.integers(0, len(alphabet), rng.integers(0, 6))))
    return out


@pytest.mark.parametrize("name,pyop", [("lt", lambda a, b: a < b), ("le", lambda a, b: a <= b), ("gt", lambda a, b: a > b), ("ge", lambda a, b: a >= b),
                                       ("eq", lambda a, b: a == b), ("ne", lambda a, b: a != b)])
def test_compare_signatures_equal_python_bytes_order(name, pyop):
    rng = np.random.default_rng(5)
    a, b = rand_strs(rng, 3000), rand_strs(rng, 3000)
    got = ev(E.ScalarFunction(name, S0, S1), Chunk([StrColumn(a), StrColumn(b)]))
    assert got == [None if x is None or y is None else int(pyop(x, y)) for x, y in zip(a, b)]


def test_in_ifnull_if_isnull_protocols():
    a = ["x", None, "y", "q", None]
    b = ["y", "y", None, None, None]
    chk = Chunk([StrColumn(a), StrColumn(b), Column(abi.I64, np.array([1, 0, 5, 0, 0]), np.array([True, True, True, False, True]))])
    # InString (:97-145): 1 on a match, else NULL if any NULL took part, else 0
    assert ev(E.ScalarFunction("in", S0, E.Constant("x"), S1), chk) == [1, None, None, None, None]
    assert ev(E.ScalarFunction("in", S0, E.Constant("q"), E.Constant("zz")), chk) == [0, None, 0, 1, None]
    # IfNullString (:81-111) below a compare
    assert ev(E.ScalarFunction("eq", E.ScalarFunction("ifnull", S0, S1), E.Constant("y")), chk) == [0, 1, 1, 0, None]
    # IfString (:209-253): NULL or 0 condition takes the third argument
    assert ev(E.ScalarFunction("length", E.ScalarFunction("if", E.Column(2, abi.I64), S0, E.Constant("four"))), chk) == [1, 4, 1, 4, 4]
    assert ev(E.ScalarFunction("isnull", S0), chk) == [0, 1, 0, 0, 1]
    assert ev(E.ScalarFunction("isnull", E.ScalarFunction("if", E.Column(2, abi.I64), S0, S1)), chk) == [0, 0, 0, 1, 1]


def test_string_valued_root_is_declared_as_such():
    # round 3: a string-valued root compiles to result_type TSQ_BYTES (tsq_expr_eval_str evaluates it into a var-len column)
    assert E.compile_expr(E.ScalarFunction("ifnull", S0, S1)).result_type == abi.BYTES
    assert E.compile_expr(E.ScalarFunction("length", E.ScalarFunction("ifnull", S0, S1))).result_type == abi.I64


def test_oracle_string_valued_roots_vs_python(orc):
    """builtinIfStringSig / builtinIfNullStringSig.vecEvalString (builtin_control_vec_generated.go:209-255, :81-115): per row
    AppendNull or AppendString of the chosen argument; Column.VecEvalString through a selection vector (column.go:111-130);
    Constant.VecEvalString (constant.go:86).  The reference holds no golden bytes for them (its tests are vec == row on random data),
    so the restatement is pinned on a row-at-a-time Python statement of the same rules, on the column STATE (offsets, bytes, NULLs)."""
    rng = np.random.default_rng(77)
    n = 3000
    a, b, c = rand_strs(rng, n), rand_strs(rng, n, 0.4), rand_strs(rng, n)
    cond = rng.integers(-1, 2, n)
    cnn = rng.random(n) > 0.2
    chk = Chunk([StrColumn(a), StrColumn(b), StrColumn(c), Column(abi.I64, cond, cnn)])
    S0, S1, S2, I3 = E.Column(0, abi.BYTES), E.Column(1, abi.BYTES), E.Column(2, abi.BYTES), E.Column(3, abi.I64)
    sel = rng.permutation(n)[:1111].astype(np.int32)
    cases = [
        (E.ScalarFunction("if", I3, S0, S1), lambda i: a[i] if (cnn[i] and cond[i] != 0) else b[i]),   # :237-249: NULL or 0 condition -> third argument
        (E.ScalarFunction("ifnull", S1, S2), lambda i: b[i] if b[i] is not None else c[i]),            # :100-110
        (E.ScalarFunction("ifnull", S1, E.Constant("dflt")), lambda i: b[i] if b[i] is not None else b"dflt"),
        (S2, lambda i: c[i]),
        (E.Constant("k"), lambda i: b"k"),
        (E.Constant(None, E.ETString), lambda i: None),
    ]
    for e, rule in cases:
        for ch, idx in ((chk, range(n)), (Chunk(chk.columns, sel=sel), sel.tolist())):
            offs, data, nn, _ = orc.expr_eval_str(E.compile_expr(e), ch)
            want = [rule(i) for i in idx]
            assert nn.tolist() == [v is not None for v in want]
            assert bytes(data) == b"".join(v for v in want if v is not None)
            assert offs.tolist() == [0] + np.cumsum([0 if v is None else len(v) for v in want]).tolist()
