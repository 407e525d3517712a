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


def test_string_valued_root_is_refused():
    with pytest.raises(E.Unsupported):
        E.compile_expr(E.ScalarFunction("ifnull", S0, S1))
