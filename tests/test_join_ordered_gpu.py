"""GPU: ordered join output (tsq_join_set_ordered) = MergeJoinExec's output order (executor/merge_join.go:257-310): probe /
outer rows in order, each with its matches in build / inner row order.  The oracle's single-threaded hash-join restatement
produces exactly that order (and equals its merge-join restatement on sorted inputs, tests/test_oracle_mergejoin_golden.py),
so rows are compared one by one IN ORDER — on sorted and on unsorted inputs, with duplicate chains whose table slots are
filled in arbitrary order, NULL keys, outer joins, OtherConditions, many probe pushes and device-resident slices."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import executor as X
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu


def _rows(chunks):
    out = []
    for c in chunks:
        out += c.rows()
    return out


def test_merge_join_test_rows_through_the_executor_mirror(ctx):
    t = Chunk([Column(abi.I64, np.array([1, 2])), Column(abi.I64, np.array([1, 2]))])
    t1 = Chunk([Column(abi.I64, np.array([2, 4])), Column(abi.I64, np.array([3, 4]))])
    mj = X.MergeJoinExec(ctx, X.MockDataSource(ctx, t), X.MockDataSource(ctx, t1), [0], [0], abi.JOIN_LEFT_OUTER, 1,
                         outer_filter=[E.ScalarFunction("ne", E.Column(0, abi.I64), E.Constant(1))])
    assert _rows(X.drain(mj)) == [(1, 1, None, None), (2, 2, 2, 3)]                                  # merge_join_test.go:257-258
    d = Chunk([Column(abi.I64, np.array([1, 1, 1]))])
    assert _rows(X.drain(X.MergeJoinExec(ctx, X.MockDataSource(ctx, d), X.MockDataSource(ctx, d), [0], [0]))) == [(1, 1)] * 9   # :276-279


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0), (abi.JOIN_INNER, 0)])
@pytest.mark.parametrize("sorted_inputs", [True, False])
def test_ordered_rows_equal_the_oracle_order(ctx, orc, jt, inner, sorted_inputs):
    rng = np.random.default_rng(jt * 10 + inner + (5 if sorted_inputs else 0))
    no, ni = 40_000, 30_000
    ok, ik = rng.integers(0, 6000, no), rng.integers(0, 6000, ni)     # ~5 duplicates per key on both sides
    ik[:200] = 4242                                                   # one inner key with 200 rows
    if sorted_inputs:
        ok, ik = np.sort(ok), np.sort(ik)
    outer = Chunk([Column(abi.I64, ok, rng.random(no) > 0.05), Column(abi.I64, np.arange(no))])
    innr = Chunk([Column(abi.I64, ik, rng.random(ni) > 0.05), Column(abi.I64, np.arange(ni))])
    t = [abi.I64, abi.I64]
    cfg = H.join_cfg(t, t, [0], [0], jt, inner)
    want = orc.hash_join(cfg, innr, outer)
    got = G.run_join(ctx, cfg, innr, outer, chunk_rows=1024, pull_rows=1000, ordered=True)
    assert got.rows() == want.rows()
    if sorted_inputs and jt != abi.JOIN_RIGHT_OUTER:  # sorted children: that order is MergeJoinExec's
        assert orc.merge_join(cfg, innr, outer).rows() == want.rows()
    # one push of everything = device slices instead of 1024-row staging: same order
    assert G.run_join(ctx, cfg, innr, outer, chunk_rows=1 << 22, pull_rows=1 << 20, ordered=True).rows() == want.rows()


def test_ordered_with_other_conditions_and_outer_filter(ctx, orc):
    rng = np.random.default_rng(77)
    no, ni = 20_000, 15_000
    outer = Chunk([Column(abi.I64, rng.integers(0, 3000, no)), Column(abi.I64, rng.integers(-50, 50, no))])
    innr = Chunk([Column(abi.I64, rng.integers(0, 3000, ni)), Column(abi.I64, rng.integers(-50, 50, ni))])
    keep = []
    conds = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(1, abi.I64), E.Column(3, abi.I64)), E.Constant(0))]
    filt = [E.ScalarFunction("ne", E.Column(1, abi.I64), E.Constant(7))]
    t = [abi.I64, abi.I64]
    cfg = H.join_cfg(t, t, [0], [0], abi.JOIN_LEFT_OUTER, 1, conds, filt, keep)
    assert G.run_join(ctx, cfg, innr, outer, chunk_rows=4096, pull_rows=4096, ordered=True).rows() == orc.hash_join(cfg, innr, outer).rows()


def test_ordered_large_device_join_keeps_probe_order(ctx):
    # 4e6 x 1e6 rows: joined rows carry the probe row number in a payload column — it must be non-decreasing, and inside one
    # probe row the build row numbers must be increasing
    nb, npr = 1_000_000, 4_000_000
    rng = np.random.default_rng(3)
    build = Chunk([Column(abi.I64, rng.integers(0, 400_000, nb)), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, rng.integers(0, 500_000, npr)), Column(abi.I64, np.arange(npr))])
    cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 22, ordered=True)
    p, b = got.columns[1].data, got.columns[3].data
    assert (np.diff(p) >= 0).all()
    same = np.diff(p) == 0
    assert (np.diff(b)[same] > 0).all()
    cnt = np.bincount(build.columns[0].data, minlength=500_000)
    assert got.NumRows() == int(cnt[probe.columns[0].data].sum())


def test_ordered_with_heavily_duplicated_keys(ctx, orc):
    # 3000 inner rows of one key x 40 outer rows of it (match lists far beyond the insertion-sort range) next to ordinary keys
    rng = np.random.default_rng(13)
    ik = np.concatenate([np.full(3000, 7), rng.integers(100, 2000, 5000)])
    rng.shuffle(ik)
    ok = np.concatenate([np.full(40, 7), rng.integers(100, 2000, 3000)])
    rng.shuffle(ok)
    innr = Chunk([Column(abi.I64, ik), Column(abi.I64, np.arange(len(ik)))])
    outer = Chunk([Column(abi.I64, ok), Column(abi.I64, np.arange(len(ok)))])
    cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
    assert G.run_join(ctx, cfg, innr, outer, chunk_rows=1 << 20, pull_rows=1 << 20, ordered=True).rows() == orc.hash_join(cfg, innr, outer).rows()


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
def test_ordered_with_other_conditions_equals_the_merge_join_restatement(ctx, orc, jt, inner):
    # round 6: MergeJoinExec's OtherConditions (merge_join.go:291-299: tryToMatchInners filters the outer row's inner group, a row none of
    # whose candidates passes is a miss) are restated in the oracle — sorted children, the ordered join row for row IN ORDER
    rng = np.random.default_rng(91 + jt)
    no, ni = 30_000, 20_000
    outer = Chunk([Column(abi.I64, np.sort(rng.integers(0, 4000, no))), Column(abi.I64, rng.integers(-50, 50, no), rng.random(no) > 0.05)])
    innr = Chunk([Column(abi.I64, np.sort(rng.integers(0, 4000, ni))), Column(abi.I64, rng.integers(-50, 50, ni), rng.random(ni) > 0.05)])
    keep = []
    conds = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(1, abi.I64), E.Column(3, abi.I64)), E.Constant(0))]
    t = [abi.I64, abi.I64]
    cfg = H.join_cfg(t, t, [0], [0], jt, inner, conds, (), keep)
    want = orc.merge_join(cfg, innr, outer).rows()
    assert len(want) > 10_000
    assert G.run_join(ctx, cfg, innr, outer, chunk_rows=4096, pull_rows=4096, ordered=True).rows() == want
