"""The oracle's restatement of SortExec / TopNExec row ordering (oracle/sort_rows.cpp) pinned on the reference's own
ORDER BY results: executor/union_scan_test.go:33-36, executor/executor_test.go:531-547, and the NULL / unsigned / float
rules of util/chunk/compare.go:27-103.  CPU only."""
import numpy as np

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column


def _t(rows, types=None):
    cols = list(zip(*rows))
    types = types or [abi.I64] * len(cols)
    out = []
    for t, c in zip(types, cols):
        dt = {abi.I64: np.int64, abi.U64: np.uint64, abi.F64: np.float64, abi.F32: np.float32}[t]
        nn = np.array([v is not None for v in c])
        out.append(Column(t, np.array([0 if v is None else v for v in c], dtype=dt), None if nn.all() else nn))
    return Chunk(out)


def test_union_scan_test_order_by_rows():
    t = _t([(1, 5), (2, 3), (3, 4), (4, 8), (6, 8), (7, 6)])                               # union_scan_test.go:31
    assert orc.sort_rows(t, [0], [True]).rows() == [(7, 6), (6, 8), (4, 8), (3, 4), (2, 3), (1, 5)]            # :33 order by a desc
    assert orc.sort_rows(t, [1, 0], [False, False]).rows() == [(2, 3), (3, 4), (1, 5), (7, 6), (4, 8), (6, 8)]  # :34 order by b, a
    assert orc.sort_rows(t, [1, 0], [True, True]).rows() == [(6, 8), (4, 8), (7, 6), (1, 5), (3, 4), (2, 3)]    # :35 order by b desc, a desc


def test_executor_test_order_by_expression_results():
    t = _t([(-0,), (-1,), (-2,)])  # executor_test.go:531-535: 1-d for d = 1,2,3 ordered by d -> -2 -1 0
    assert orc.sort_rows(t, [0], [False]).rows() == [(-2,), (-1,), (0,)]
    assert orc.sort_rows(_t([(1, 1), (2, 2)]), [1], [False]).rows() == [(1, 1), (2, 2)]   # :545-547


def test_compare_go_rules_null_unsigned_float():
    # cmpNull (compare.go:48-56): NULL is smaller than every value, two NULLs are equal; DESC negates (sort.go:121-123)
    t = _t([(3,), (None,), (-7,), (None,), (0,)])
    assert orc.sort_rows(t, [0], [False]).rows() == [(None,), (None,), (-7,), (0,), (3,)]
    assert orc.sort_rows(t, [0], [True]).rows() == [(3,), (0,), (-7,), (None,), (None,)]
    # cmpUint64: values above MaxInt64 are the largest (compare.go:66-72)
    u = _t([(1,), ((1 << 64) - 1,), (1 << 63,), (0,)], [abi.U64])
    assert orc.sort_rows(u, [0], [False]).rows() == [(0,), (1,), (1 << 63,), ((1 << 64) - 1,)]
    # cmpFloat64 / cmpFloat32 (widened): -inf < -1 < -0.0 == 0.0 < tiny < inf ; equal keys keep any order
    f = _t([(1.5,), (float("-inf"),), (0.0,), (-1.0,), (5e-324,), (float("inf"),)], [abi.F64])
    assert orc.sort_rows(f, [0], [False]).rows() == [(float("-inf"),), (-1.0,), (0.0,), (5e-324,), (1.5,), (float("inf"),)]
    assert orc.row_compare(_t([(0.0,), (-0.0,)], [abi.F64]), [0], [False], 0, 1) == 0
    g = _t([(2.5, 1), (2.5, 0), (-2.5, 9)], [abi.F32, abi.I64])
    assert orc.sort_rows(g, [0, 1], [True, False]).rows() == [(2.5, 0), (2.5, 1), (-2.5, 9)]
