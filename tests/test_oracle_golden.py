"""Pins the CPU oracle against the reference's own golden vectors.

CPU only.  Every case cites the reference test it was transcribed from (see
tests/golden/make_golden.py).  The GPU parity tests trust the oracle only because these pass.
"""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column

from . import helpers as H


# ------------------------------------------------------------------ rowHashMap (hash_table_test.go:21-50)
def test_rowhashmap_insertion_order(orc):
    assert orc.rowhashmap_put_get([1], [(1 << 32) | 1], 1).tolist() == [(1 << 32) | 1]
    slice_len = 64  # initialEntrySliceLen
    raw = {i: [(i << 32) | j for j in range(slice_len * i)] for i in range(10)}
    keys, ptrs = [], []
    for j in range(slice_len * 9):  # "put all rawData into m vertically"
        for i in range(9, -1, -1):
            if not j < slice_len * i:
                break
            keys.append(i)
            ptrs.append(raw[i][j])
    for i in range(10):
        got = orc.rowhashmap_put_get(keys, ptrs, i).tolist()
        assert got == raw[i], "Get must return insertion order across entryStore slab growth"
    assert len(keys) == sum(len(v) for v in raw.values())


# ------------------------------------------------------------------ hashing (codec_test.go:735-769,811-866)
def test_fnv1_64_vectors(orc):
    for v in H.golden("hash_cases.json")["fnv1_64"]:
        assert "%016x" % orc.fnv1_64(v["in"].encode()) == v["hex"]


def _one_cell_chunk(tp_name, value):
    tp = H.TYPES[tp_name]
    return H.chunk_from_rows([[value]], [tp])


def test_hash_chunk_row_equal(orc):
    for case in H.golden("hash_cases.json")["equal_cases"]:
        a = _one_cell_chunk(*case["a"])
        b = _one_cell_chunk(*case["b"])
        ha, _ = orc.hash_keys(a, [0])
        hb, _ = orc.hash_keys(b, [0])
        # equal cells hash equally (HashChunkRow) and join (EqualChunkRow)
        cfg = H.join_cfg([a.columns[0].tp], [b.columns[0].tp], [0], [0], abi.JOIN_INNER, 1)
        joined = orc.hash_join(cfg, b, a)
        assert (ha[0] == hb[0]) == case["equal"], case["ref"]
        assert (joined.NumRows() == 1) == case["equal"], case["ref"]


def test_hash_chunk_columns_matches_row_hash_and_flags_nulls(orc):
    # codec_test.go:811-866: vectorised column hash == per-row hash; NULL rows flagged
    rng = np.random.default_rng(7)
    cols = [H.random_column(rng, abi.I64, 257), H.random_column(rng, abi.F64, 257), H.random_column(rng, abi.U64, 257)]
    chk = Chunk(cols)
    h, hn = orc.hash_keys(chk, [0, 1, 2])
    for i in range(chk.NumRows()):
        row = Chunk([c.slice(i, i + 1) for c in cols])
        h1, hn1 = orc.hash_keys(row, [0, 1, 2])
        assert h1[0] == h[i] and hn1[0] == hn[i]
        assert hn[i] == any(c.IsNull(i) for c in cols)
    # manual FNV-1 over [flag][8 LE bytes] for a NOT NULL int64 cell (codec.go:264-276)
    i = int(np.nonzero(cols[0].notnull if cols[0].notnull is not None else np.ones(257, bool))[0][0])
    one = Chunk([cols[0].slice(i, i + 1)])
    h1, _ = orc.hash_keys(one, [0])
    assert h1[0] == orc.fnv1_64(bytes([8]) + int(cols[0].data[i]).to_bytes(8, "little", signed=True))


def test_group_key_encoding(orc):
    for v in H.golden("hash_cases.json")["group_key"]:
        col = _one_cell_chunk(v["type"], v["value"]).columns[0]
        assert orc.group_key_encode(col, 0).hex() == v["hex"], v


# ------------------------------------------------------------------ hash join (join_test.go)
@pytest.mark.parametrize("case", H.golden("join_cases.json"), ids=lambda c: c["ref"][:48])
def test_join_golden(orc, case):
    keep = []
    cfg, left, right, build, probe, _, _ = H.lower_join_case(case, keep)
    out = orc.hash_join(cfg, build, probe)
    assert H.rows_equal_unordered(out, [tuple(r) for r in case["expect"]]), case["ref"]


def test_join_matches_keep_build_insertion_order(orc):
    # hash_table.go:266-271: matches of one probe row come back in build insertion order
    build = H.chunk_from_rows([[1, 10], [2, 20], [1, 11], [1, 12]], [abi.I64, abi.I64])
    probe = H.chunk_from_rows([[1]], [abi.I64])
    cfg = H.join_cfg([abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
    out = orc.hash_join(cfg, build, probe)
    assert out.rows() == [(1, 1, 10), (1, 1, 11), (1, 1, 12)]


def test_join_null_keys_never_match(orc):
    # hash_table.go:161-163 (build) / join.go:344-345 (probe)
    build = H.chunk_from_rows([[None, 1], [5, 2]], [abi.I64, abi.I64])
    probe = H.chunk_from_rows([[None, 7], [5, 8]], [abi.I64, abi.I64])
    cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_LEFT_OUTER, 1)
    out = orc.hash_join(cfg, build, probe)
    assert H.rows_equal_unordered(out, [(None, 7, None, None), (5, 8, 5, 2)])


def test_join_int_vs_varlen_class_never_equal(orc):
    # join_test.go:345-353 TestIssue5278: an int key never equals a key of another class (flag byte differs)
    build = H.chunk_from_rows([[1.0]], [abi.F64])
    probe = H.chunk_from_rows([[1]], [abi.I64])
    cfg = H.join_cfg([abi.I64], [abi.F64], [0], [0], abi.JOIN_INNER, 1)
    assert orc.hash_join(cfg, build, probe).NumRows() == 0


# ------------------------------------------------------------------ aggregate functions
def _agg_input(tp_name, n, trailing_null=False):
    tp = H.TYPES[tp_name]
    if tp == abi.BYTES:  # aggfunc_test.go:157-158: the decimal digits of the row number
        rows = [[str(i)] for i in range(n)]
    else:
        rows = [[i] for i in range(n)] if tp in (abi.I64, abi.U64) else [[float(i)] for i in range(n)]
    if trailing_null:
        rows.append([None])
    return H.chunk_from_rows(rows, [tp]), tp


@pytest.mark.parametrize("case", H.golden("agg_cases.json")["funcs"], ids=lambda c: "%s-%s" % (c["func"], c["type"]))
def test_aggfunc_kats(orc, case):
    func = H.AGG_FUNCS[case["func"]]
    if "complete" in case:  # aggfunc_test.go:163-205
        chk, tp = _agg_input(case["type"], 5, trailing_null=True)
        cfg = H.agg_cfg([tp], [], [(func, 0, tp)])
        out = orc.hash_agg(cfg, chk, 1, 1)
        assert out.NumRows() == 1
        assert H.approx_equal(out.rows()[0][0], case["complete"][1], 0), case["ref"]
        empty = H.chunk_from_rows([], [tp])
        out = orc.hash_agg(cfg, empty, 1, 1)
        assert out.rows() == [(case["complete"][0],)], case["ref"]
    if "merge" in case:  # aggfunc_test.go:72-130
        chk, tp = _agg_input(case["type"], 5)
        pcfg = H.agg_cfg([tp], [], [(func, 0, tp, abi.MODE_PARTIAL1)])
        p1 = orc.hash_agg(pcfg, chk, 1, 1)
        p2 = orc.hash_agg(pcfg, chk.slice(2, 5), 1, 1)
        # the value AppendFinalResult2Chunk reports for each partial (complete-mode view of the same rows)
        ccfg = H.agg_cfg([tp], [], [(func, 0, tp)])
        assert H.approx_equal(orc.hash_agg(ccfg, chk, 1, 1).rows()[0][0], case["merge"][0], 0)
        assert H.approx_equal(orc.hash_agg(ccfg, chk.slice(2, 5), 1, 1).rows()[0][0], case["merge"][1], 0)
        # final phase merges the two partial rows
        from tinysql_amd.chunk import concat
        both = concat([p1, p2], [c.tp for c in p1.columns])
        ptypes = [c.tp for c in p1.columns]
        if func == abi.AGG_AVG:
            fcfg = H.agg_cfg(ptypes, [], [(func, 0, tp, abi.MODE_FINAL, 1)])
        else:
            fcfg = H.agg_cfg(ptypes, [], [(func, 0, ptypes[0] if func != abi.AGG_COUNT else tp, abi.MODE_FINAL)])
        if func == abi.AGG_FIRSTROW:
            # partial worker order decides which partial is merged first; the test merges p1 then p2
            out = orc.hash_agg(fcfg, both, 1, 1)
        else:
            out = orc.hash_agg(fcfg, both, 2, 2)
        assert H.approx_equal(out.rows()[0][0], case["merge"][2], 0), case["ref"]


@pytest.mark.parametrize("case", H.golden("agg_cases.json")["sql"], ids=lambda c: c["ref"][:40])
def test_agg_sql_rows(orc, case):
    types = [H.TYPES[t] for t in case["types"]]
    chk = H.chunk_from_rows(case["rows"], types)
    aggs = [(H.AGG_FUNCS[f], col, H.TYPES[t]) for f, col, t in case["aggs"]]
    cfg = H.agg_cfg(types, case["group_by"], aggs)
    out = orc.hash_agg(cfg, chk, 4, 4)
    assert H.rows_equal_unordered(out, [tuple(r) for r in case["expect"]]), case["ref"]


def test_agg_partial_final_split_equals_complete(orc):
    # descriptor.go:56-91 Split: Complete == Final(Partial1) for every function
    rng = np.random.default_rng(11)
    n = 5000
    k = Column(abi.I64, rng.integers(0, 37, n), rng.random(n) > 0.05)
    v = H.random_column(rng, abi.I64, n, lo=-1000, hi=1000)
    d = H.random_column(rng, abi.F64, n)
    chk = Chunk([k, v, d])
    types = [abi.I64, abi.I64, abi.F64]
    aggs = [(abi.AGG_COUNT, 1, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_AVG, 1, abi.I64), (abi.AGG_MAX, 1, abi.I64),
            (abi.AGG_MIN, 2, abi.F64), (abi.AGG_SUM, 2, abi.F64), (abi.AGG_FIRSTROW, 0, abi.I64)]
    a = orc.hash_agg(H.agg_cfg(types, [0], aggs), chk, 1, 1)
    b = orc.hash_agg(H.agg_cfg(types, [0], aggs), chk, 4, 3)
    ra, rb = sorted(a.rows(), key=lambda r: (r[6] is None, r[6])), sorted(b.rows(), key=lambda r: (r[6] is None, r[6]))
    assert len(ra) == len(rb) == 38  # 37 keys + the NULL group (codec.go:718-719)
    for x, y in zip(ra, rb):
        for i, (p, q) in enumerate(zip(x, y)):
            assert H.approx_equal(p, q, 1e-6 if i == 5 else 0)


def test_agg_sum_overflow_is_an_error(orc):
    # func_sum.go:133-137 -> types.AddInt64 (types/overflow.go:33-40)
    chk = H.chunk_from_rows([[(1 << 63) - 1], [1]], [abi.I64])
    cfg = H.agg_cfg([abi.I64], [], [(abi.AGG_SUM, 0, abi.I64)])
    with pytest.raises(orc.OracleError) as ei:
        orc.hash_agg(cfg, chk, 1, 1)
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT


def test_agg_group_key_float_zero_signs_share_a_group(orc):
    # float.go:22-30: -0.0 >= 0 so both zeros encode identically
    chk = H.chunk_from_rows([[0.0], [-0.0], [1.0]], [abi.F64])
    cfg = H.agg_cfg([abi.F64], [0], [(abi.AGG_COUNT, -1, abi.I64)])
    out = orc.hash_agg(cfg, chk, 1, 1)
    assert sorted(r[0] for r in out.rows()) == [1, 2]


# ------------------------------------------------------------------ expressions
def test_expr_kats(orc):
    one_row = Chunk([Column(abi.I64, np.zeros(1, np.int64))])
    for case in H.golden("expr_cases.json"):
        if "table" in case:
            a = [p[0] for p in case["pairs"]]
            b = [p[1] for p in case["pairs"]]
            chk = H.chunk_from_rows([[x, y] for x, y in zip(a, b)], [abi.I64, abi.I64])
            e = E.ScalarFunction(case["table"], E.Column(0, abi.I64), E.Column(1, abi.I64))
            col, _ = orc.expr_eval(E.compile_expr(e), chk)
            assert col.values() == [p[2] for p in case["pairs"]], case["ref"]
        else:
            e = H.expr_from_json(case["expr"], [abi.I64])
            col, _ = orc.expr_eval(E.compile_expr(e), one_row)
            assert col.values() == [case["expect"]], case["ref"]


def test_expr_overflow_errors(orc):
    i64max, i64min = (1 << 63) - 1, -(1 << 63)
    chk = H.chunk_from_rows([[i64max, 1, i64min, -1]], [abi.I64] * 4)
    c = [E.Column(i, abi.I64) for i in range(4)]
    for e in [E.ScalarFunction("plus", c[0], c[1]), E.ScalarFunction("minus", c[2], c[1]), E.ScalarFunction("mul", c[0], c[0]),
              E.ScalarFunction("unaryminus", c[2])]:
        with pytest.raises(orc.OracleError) as ei:
            orc.expr_eval(E.compile_expr(e), chk)
        assert ei.value.status == abi.ERR_OVERFLOW_BIGINT
    # Go: MinInt64 * -1 wraps and (MinInt64 / -1 == MinInt64) hides it (builtin_arithmetic_vec.go:333)
    col, _ = orc.expr_eval(E.compile_expr(E.ScalarFunction("mul", c[3], c[2])), chk)
    assert col.values() == [i64min]
    # unsigned
    u = H.chunk_from_rows([[(1 << 64) - 1, 1]], [abi.U64, abi.U64])
    cu = [E.Column(0, abi.U64), E.Column(1, abi.U64)]
    with pytest.raises(orc.OracleError) as ei:
        orc.expr_eval(E.compile_expr(E.ScalarFunction("plus", cu[0], cu[1])), u)
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT_UNSIGNED
    with pytest.raises(orc.OracleError) as ei:
        orc.expr_eval(E.compile_expr(E.ScalarFunction("minus", cu[1], cu[0])), u)
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT_UNSIGNED
    # real
    r = H.chunk_from_rows([[1.7e308, 1.7e308, 0.0]], [abi.F64] * 3)
    cr = [E.Column(i, abi.F64) for i in range(3)]
    for name in ("plus", "mul"):
        with pytest.raises(orc.OracleError) as ei:
            orc.expr_eval(E.compile_expr(E.ScalarFunction(name, cr[0], cr[1])), r)
        assert ei.value.status == abi.ERR_OVERFLOW_DOUBLE
    col, warns = orc.expr_eval(E.compile_expr(E.ScalarFunction("div", cr[0], cr[2])), r)
    assert col.values() == [None] and warns == 1  # builtin_arithmetic_vec.go:369-375


def test_compare_signed_unsigned_matrix(orc):
    # types/compare.go:44-101
    big = (1 << 64) - 1
    chk = H.chunk_from_rows([[big, -1], [5, 5], [0, -1]], [abi.U64, abi.I64])
    u, s = E.Column(0, abi.U64), E.Column(1, abi.I64)
    gt, _ = orc.expr_eval(E.compile_expr(E.ScalarFunction("gt", u, s)), chk)
    eq, _ = orc.expr_eval(E.compile_expr(E.ScalarFunction("eq", u, s)), chk)
    lt, _ = orc.expr_eval(E.compile_expr(E.ScalarFunction("lt", s, u)), chk)
    assert gt.values() == [1, 0, 1] and eq.values() == [0, 1, 0] and lt.values() == [1, 0, 1]


def test_vec_eval_bool_null_semantics(orc):
    # expression.go:247-276: NULL of an Int conjunct keeps the row in sel but flags it; a Real NULL drops it
    chk = H.chunk_from_rows([[1, 1.0], [None, 1.0], [1, None], [0, 1.0], [1, 0.4], [1, 0.5]], [abi.I64, abi.F64])
    progs = E.compile_list([E.Column(0, abi.I64), E.Column(1, abi.F64)])
    sel, nulls, _ = orc.filter_eval(progs, 2, chk)
    assert sel.tolist() == [True, False, False, False, False, True]   # 0.4 rounds to 0 (types/helper.go:28)
    assert nulls.tolist() == [False, True, False, False, False, False]


def test_filter_later_conjunct_only_sees_survivors(orc):
    # an overflow in conjunct 2 must not fire for rows conjunct 1 already dropped
    i64max = (1 << 63) - 1
    chk = H.chunk_from_rows([[0, i64max], [1, 5]], [abi.I64, abi.I64])
    c0, c1 = E.Column(0, abi.I64), E.Column(1, abi.I64)
    progs = E.compile_list([c0, E.ScalarFunction("gt", E.ScalarFunction("plus", c1, E.Constant(1)), E.Constant(0))])
    sel, _, _ = orc.filter_eval(progs, 2, chk)
    assert sel.tolist() == [False, True]
    chk2 = H.chunk_from_rows([[1, i64max]], [abi.I64, abi.I64])
    with pytest.raises(orc.OracleError):
        orc.filter_eval(progs, 2, chk2)
