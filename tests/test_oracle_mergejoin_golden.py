"""The oracle's restatement of MergeJoinExec (oracle/oracle.cpp orc_merge_join) pinned on executor/merge_join_test.go
(:245-321) row for row IN ORDER, and shown equal to the order the single-threaded hash-join restatement produces (probe rows
in order, matches in build insertion order) — the order tsq_join_set_ordered promises.  CPU only."""
import numpy as np

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column

from . import helpers as H


def _t(rows, ncols=2):
    if not rows:
        return Chunk([Column(abi.I64, np.zeros(0, np.int64)) for _ in range(ncols)])
    cols = list(zip(*rows))
    out = []
    for c in cols:
        nn = np.array([v is not None for v in c])
        out.append(Column(abi.I64, np.array([0 if v is None else v for v in c], dtype=np.int64), None if nn.all() else nn))
    return Chunk(out)


def _cfg(nl, nr, jt, inner, outer_filter=(), keep=None):
    return H.join_cfg([abi.I64] * nl, [abi.I64] * nr, [0], [0], jt, inner, (), outer_filter, keep if keep is not None else [])


def test_merge_join_test_rows_in_order():
    t, t1 = _t([(1, 1), (2, 2)]), _t([(2, 3), (4, 4)])                                             # merge_join_test.go:245-246
    # :257 t left outer join t1 on t.c1 = t1.c1 and t.c1 != 1 (a condition on the outer table = outer filter) order by t1.c1
    keep = []
    cfg = _cfg(2, 2, abi.JOIN_LEFT_OUTER, 1, [E.ScalarFunction("ne", E.Column(0, abi.I64), E.Constant(1))], keep)
    assert orc.merge_join(cfg, t1, t).rows() == [(1, 1, None, None), (2, 2, 2, 3)]                 # :258
    # :249 left outer join ... where t.c1 = 1 or t1.c2 > 20 (the WHERE runs above the join): join rows first
    assert orc.merge_join(_cfg(2, 2, abi.JOIN_LEFT_OUTER, 1), t1, t).rows() == [(1, 1, None, None), (2, 2, 2, 3)]
    # :251 t1 right outer join t: the outer table is the RIGHT child, output t1 columns || t columns
    assert orc.merge_join(_cfg(2, 2, abi.JOIN_RIGHT_OUTER, 0), t1, t).rows() == [(None, None, 1, 1), (2, 3, 2, 2)]
    # :276-279 duplicates: three rows of (1) joined with themselves -> nine "1 1"
    d = _t([(1,), (1,), (1,)], 1)
    assert orc.merge_join(_cfg(1, 1, abi.JOIN_INNER, 1), d, d).rows() == [(1, 1)] * 9
    # :284-288 1..7 join 1..7 -> 1..7 in order
    s = _t([(i,) for i in range(1, 8)], 1)
    assert [r[0] for r in orc.merge_join(_cfg(1, 1, abi.JOIN_INNER, 1), s, s).rows()] == list(range(1, 8))
    # :313 t right join t t1 on t.a = t1.b with t = (1, 2): no match -> "<nil> 2" (projected a, b); full row: NULL NULL 1 2
    one = _t([(1, 2)])
    cfg = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [1], abi.JOIN_RIGHT_OUTER, 0)
    assert orc.merge_join(cfg, one, one).rows() == [(None, None, 1, 2)]
    # :316-321 t(a, b) = (1,1..4) join s(a) = (1): count(*) = 4, rows in t order
    tt, ss = _t([(1, 1), (1, 2), (1, 3), (1, 4)]), _t([(1,)], 1)
    cfg = H.join_cfg([abi.I64] * 2, [abi.I64], [0], [0], abi.JOIN_INNER, 1)
    assert orc.merge_join(cfg, ss, tt).rows() == [(1, 1, 1), (1, 2, 1), (1, 3, 1), (1, 4, 1)]


def test_null_keys_and_group_walk():
    # inner rows with a NULL key are skipped (merge_join.go:148-156); an outer NULL key sorts first and never matches
    outer = _t([(None, 0), (1, 1), (1, 2), (3, 3), (5, 4), (5, 5), (9, 6)])
    inner = _t([(None, 10), (None, 11), (1, 12), (2, 13), (5, 14), (5, 15), (7, 16)])
    got = orc.merge_join(_cfg(2, 2, abi.JOIN_LEFT_OUTER, 1), inner, outer).rows()
    assert got == [(None, 0, None, None), (1, 1, 1, 12), (1, 2, 1, 12), (3, 3, None, None), (5, 4, 5, 14), (5, 4, 5, 15), (5, 5, 5, 14), (5, 5, 5, 15),
                   (9, 6, None, None)]
    assert orc.merge_join(_cfg(2, 2, abi.JOIN_INNER, 1), inner, outer).rows() == [r for r in got if r[2] is not None]


def test_merge_join_order_equals_the_ordered_hash_join_restatement():
    rng = np.random.default_rng(2)
    no, ni = 3000, 2500
    ok = np.sort(rng.integers(0, 700, no))
    ik = np.sort(rng.integers(0, 700, ni))
    outer = Chunk([Column(abi.I64, ok, np.arange(no) >= 40), Column(abi.I64, np.arange(no))])   # sorted: NULLs first
    inner = Chunk([Column(abi.I64, ik, np.arange(ni) >= 25), Column(abi.I64, np.arange(ni))])
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = _cfg(2, 2, jt, 1)
        assert orc.merge_join(cfg, inner, outer).rows() == orc.hash_join(cfg, inner, outer).rows()


def test_other_conditions_filter_the_group_and_a_row_without_a_passing_candidate_is_a_miss():
    # merge_join.go:291-299: tryToMatchInners runs the joiner's filter (joiner.go:155-167, 351-378) over the outer row joined with its inner
    # group; hasMatch stays false when no candidate passes -> onMissMatch.  merge_join_test.go holds no ON-condition beyond the outer
    # filter, so the restatement is pinned on its equality with the hash join's (join_test.go:69-116 are its golden rows) on sorted children
    outer = _t([(1, 5), (1, -5), (2, 1), (3, 0), (3, 9)])
    inner = _t([(1, 1), (1, 10), (3, -20), (3, 2)])
    cond = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(1, abi.I64), E.Column(3, abi.I64)), E.Constant(0))]  # outer.c2 + inner.c2 > 0
    keep = []
    cfg = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [0], abi.JOIN_LEFT_OUTER, 1, cond, (), keep)
    assert orc.merge_join(cfg, inner, outer).rows() == [(1, 5, 1, 1), (1, 5, 1, 10), (1, -5, 1, 10), (2, 1, None, None), (3, 0, 3, 2), (3, 9, 3, 2)]
    cfg_in = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [0], abi.JOIN_INNER, 1, cond, (), keep)
    assert orc.merge_join(cfg_in, inner, outer).rows() == [(1, 5, 1, 1), (1, 5, 1, 10), (1, -5, 1, 10), (3, 0, 3, 2), (3, 9, 3, 2)]
    # an outer row whose every candidate fails: NULL-padded once
    cond2 = [E.ScalarFunction("gt", E.Column(3, abi.I64), E.Constant(100))]
    cfg2 = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [0], abi.JOIN_LEFT_OUTER, 1, cond2, (), keep)
    assert orc.merge_join(cfg2, inner, outer).rows() == [(1, 5, None, None), (1, -5, None, None), (2, 1, None, None), (3, 0, None, None), (3, 9, None, None)]
    # random sorted children, duplicates on both sides, NULL keys and NULL condition operands: merge join == hash join, row for row in order
    rng = np.random.default_rng(12)
    for jt, inner_child in ((abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)):
        no, ni = 3000, 2500
        o = Chunk([Column(abi.I64, np.sort(rng.integers(0, 400, no)), rng.random(no) > 0.03), Column(abi.I64, rng.integers(-50, 50, no), rng.random(no) > 0.1)])
        i = Chunk([Column(abi.I64, np.sort(rng.integers(0, 400, ni)), rng.random(ni) > 0.03), Column(abi.I64, rng.integers(-50, 50, ni), rng.random(ni) > 0.1)])
        # (NULL keys sort first: put them there, as a sorted child would deliver them)
        for ch in (o, i):
            k = ch.columns[0]
            if k.notnull is not None:
                order = np.argsort(np.where(k.notnull, k.data, -1), kind="stable")
                for c in ch.columns:
                    c.data = c.data[order]
                    if c.notnull is not None:
                        c.notnull = c.notnull[order]
                    c._bitmap = None
        a, b = (1, 3) if inner_child == 1 else (3, 1)  # the outer child's c2 and the inner child's c2 in the joined schema
        cond = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(a, abi.I64), E.Column(b, abi.I64)), E.Constant(0))]
        keep = []
        cfg = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [0], jt, inner_child, cond, (), keep)
        assert orc.merge_join(cfg, i, o).rows() == orc.hash_join(cfg, i, o).rows()
