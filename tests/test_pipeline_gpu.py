"""GPU: device-resident operator pipeline (tinysql_amd/gpu_pipeline.py) on a Q3-shaped plan —
Selection -> HashJoin -> HashJoin -> Projection -> HashAgg with chunks that never leave HBM — against the ORACLE's operators
chained the same way (orc.filter_eval -> orc.hash_join -> orc.hash_join -> orc.expr_eval -> orc.hash_agg; the numpy restatement
of the query in tools/q3.py is checked against that chain too), plus tsq_chunk_compact on its own against numpy boolean indexing."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import gpu_pipeline as GP
from tinysql_amd.chunk import Chunk, Column

from . import helpers as H

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import q3  # noqa: E402

pytestmark = pytest.mark.gpu


def q3_by_the_oracle(orc, customer, orders, lineitem):
    """tools/q3.py's plan, operator by operator, through the oracle (the CPU restatement of the reference executors).
    Returns {orderkey: (orderdate, shippriority, revenue, rows in the group, sum of |price * (1 - discount)|)}."""
    from tinysql_amd import expression as E
    F, Col, K, I, R = E.ScalarFunction, E.Column, E.Constant, abi.I64, abi.F64

    def select(chk, conj):  # SelectionExec: VectorizedFilter, then the selected rows (executor.go:401-438)
        sel, _, _ = orc.filter_eval(E.compile_list(conj), len(conj), chk)
        return Chunk([Column(c.tp, c.data[sel], None if c.notnull is None else c.notnull[sel]) for c in chk.columns])

    cust = select(customer, [F("eq", Col(1, I), K(q3.SEG))])
    ords = select(orders, [F("lt", Col(2, I), K(q3.D))])
    line = select(lineitem, [F("gt", Col(1, I), K(q3.D))])
    j1 = orc.hash_join(H.join_cfg(ords.types(), cust.types(), [1], [0], abi.JOIN_INNER, 1), cust, ords)     # orders probe (left), customer build
    j2 = orc.hash_join(H.join_cfg(line.types(), j1.types(), [0], [0], abi.JOIN_INNER, 1), j1, line)         # lineitem probe (left), j1 build
    rev, _ = orc.expr_eval(E.compile_expr(F("mul", Col(2, R), F("minus", K(1.0), Col(3, R)))), j2)
    proj = Chunk([j2.columns[0], j2.columns[6], j2.columns[7], rev])
    aggs = [(abi.AGG_FIRSTROW, 0, I), (abi.AGG_FIRSTROW, 1, I), (abi.AGG_FIRSTROW, 2, I), (abi.AGG_SUM, 3, R)]
    out = orc.hash_agg(H.agg_cfg(proj.types(), [0, 1, 2], aggs), proj, 4, 4)
    keys = proj.columns[0].data
    order = np.argsort(keys, kind="stable")
    uk, start, cnt = np.unique(keys[order], return_index=True, return_counts=True)
    sabs = np.add.reduceat(np.abs(rev.data[order]), start) if len(keys) else np.zeros(0)
    n_of, abs_of = dict(zip(uk.tolist(), cnt.tolist())), dict(zip(uk.tolist(), sabs.tolist()))
    res = {}
    for k, d, p, s in out.rows():
        assert k not in res
        res[k] = (d, p, s, n_of[k], abs_of[k])
    return res


def sum_tol(n_rows, sum_abs):
    """SUM(double) of a group is order dependent (partial -> final workers); any two orders differ by at most
    2 * n_g * 2^-53 * sum(|v|)."""
    return 2.0 * n_rows * 2.0 ** -53 * sum_abs


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 70001])
def test_chunk_compact_matches_boolean_indexing(ctx, n):
    rng = np.random.default_rng(n)
    chk = Chunk([Column(abi.I64, rng.integers(-9, 9, n), rng.random(n) > 0.2), Column(abi.F64, rng.random(n)),
                 Column(abi.F32, rng.random(n).astype(np.float32), rng.random(n) > 0.5), Column(abi.U64, rng.integers(0, 1 << 62, n).astype(np.uint64))])
    sel = rng.random(n) > 0.4
    dev = GP.DeviceChunk.from_host(ctx, chk)
    out = [GP.DeviceColumn(ctx, c.tp, n) for c in chk.columns]
    flags = ctx.alloc(n + 64)
    try:
        ctx.h2d(flags, sel.astype(np.uint8))
        m = C.c_int64(0)
        oc = (abi.Col * 4)(*[c.col(n) for c in out])
        _lib.check(ctx.lib.tsq_chunk_compact(ctx.h, dev.cols(), 4, n, flags, oc, C.byref(m)), ctx.h)
        assert m.value == int(sel.sum())
        got = GP.DeviceChunk(out, m.value).to_host()
        want = Chunk([Column(c.tp, c.data[sel], None if c.notnull is None else c.notnull[sel]) for c in chk.columns])
        assert H.rows_equal_unordered(got, want)
    finally:
        dev.free()
        for c in out:
            c.free()
        ctx.free(flags)


@pytest.mark.parametrize("classic", [False, True], ids=["flags_into_joins_used_columns", "compacting_selections_all_columns"])
@pytest.mark.parametrize("jit", [abi.JIT_OFF, abi.JIT_FORCE])
def test_q3_shaped_plan_on_device_chunks(ctx, orc, jit, classic):
    customer, orders, lineitem = q3.tables(0.05)  # 7.5e3 / 7.5e4 / 3e5 rows
    want = q3_by_the_oracle(orc, customer, orders, lineitem)
    dev = [GP.DeviceChunk.from_host(ctx, t) for t in (customer, orders, lineitem)]
    try:
        out = GP.drain_device(q3.plan(ctx, *dev, batch_rows=50_000, jit=jit, classic=classic))  # several batches per table
        got = {}
        for c in out:
            for k, d, p, s in c.rows():
                assert k not in got
                got[k] = (d, p, s)
        assert set(got) == set(want) and len(got) > 1000
        for k, (d, p, s, n_g, s_abs) in want.items():
            g = got[k]
            assert g[0] == d and g[1] == p
            assert abs(g[2] - s) <= sum_tol(n_g, s_abs), (k, g[2], s, n_g)
        # tools/q3.py's numpy restatement (the checker of the SF 10 / SF 100 runs) against the same chain
        uk, dates, prios, sums = q3.reference(customer, orders, lineitem)
        assert set(uk.tolist()) == set(want)
        for k, d, p, s in zip(uk.tolist(), dates.tolist(), prios.tolist(), sums.tolist()):
            assert want[k][0] == d and want[k][1] == p and abs(want[k][2] - s) <= sum_tol(want[k][3], want[k][4])
    finally:
        for d in dev:
            d.free()


def test_q3_with_the_market_segment_as_a_varchar_column(ctx, orc):
    # tools/q3.py --string-segment: WHERE c_mktsegment = 'BUILDING' as EQString on a device-resident var-len column; the same seed gives
    # the same segment per customer as the int-coded tables, so the groups must be those of the oracle's chain over the int codes
    customer, orders, lineitem = q3.tables(0.05)
    want = q3_by_the_oracle(orc, customer, orders, lineitem)
    scust = q3.tables(0.05, string_segment=True)[0]
    assert [q3.SEGMENTS.index(v) for v in scust.columns[1].values()] == customer.columns[1].data.tolist()
    dev = [GP.DeviceChunk.from_host(ctx, t) for t in (scust, orders, lineitem)]
    try:
        out = GP.drain_device(q3.plan(ctx, *dev, batch_rows=50_000, string_segment=True))
        got = {k: (d, p, s) for c in out for k, d, p, s in c.rows()}
        assert set(got) == set(want)
        for k, (d, p, s, n_g, s_abs) in want.items():
            assert got[k][0] == d and got[k][1] == p and abs(got[k][2] - s) <= sum_tol(n_g, s_abs)
    finally:
        for d in dev:
            d.free()


@pytest.mark.parametrize("jt", [abi.JOIN_INNER, abi.JOIN_LEFT_OUTER])
def test_device_pipeline_equals_host_chunk_pipeline_with_nulls(ctx, jt):
    # the same plan through device-resident chunks (gpu_pipeline.py) and through 1024-row host chunks (executor.py, itself
    # pinned on the oracle): NULL join keys, NULL filter inputs (a NULL never passes a filter, expression.go:272-276),
    # NULL aggregate arguments, NULL as a group (codec.go:718-719), outer-join NULL padding feeding the aggregate.
    from tinysql_amd import executor as X
    from tinysql_amd import expression as E
    from tinysql_amd.executor import AggFuncDesc

    rng = np.random.default_rng(100 + jt)
    nf, nd = 120_000, 9_000
    fact = Chunk([Column(abi.I64, rng.integers(0, 10_000, nf), rng.random(nf) > 0.05), Column(abi.I64, rng.integers(-50, 50, nf), rng.random(nf) > 0.1),
                  Column(abi.F64, rng.integers(0, 1000, nf) / 8.0, rng.random(nf) > 0.1)])
    dim = Chunk([Column(abi.I64, rng.permutation(12_000)[:nd].astype(np.int64), rng.random(nd) > 0.02), Column(abi.I64, rng.integers(0, 40, nd), rng.random(nd) > 0.2)])
    F, Col, K = E.ScalarFunction, E.Column, E.Constant
    filt = [F("gt", Col(1, abi.I64), K(-40)), F("lt", Col(2, abi.F64), K(120.0))]
    proj = [Col(4, abi.I64), F("mul", Col(2, abi.F64), K(2.0)), Col(1, abi.I64)]           # d.grp, f.x * 2, f.v
    aggs = [AggFuncDesc(abi.AGG_FIRSTROW, 0, abi.I64), AggFuncDesc(abi.AGG_COUNT, -1, abi.I64), AggFuncDesc(abi.AGG_SUM, 2, abi.I64),
            AggFuncDesc(abi.AGG_AVG, 2, abi.I64), AggFuncDesc(abi.AGG_SUM, 1, abi.F64), AggFuncDesc(abi.AGG_MAX, 1, abi.F64)]

    host = X.HashAggExec(ctx, X.ProjectionExec(ctx, X.HashJoinExec(ctx, X.SelectionExec(ctx, X.MockDataSource(ctx, fact), filt), X.MockDataSource(ctx, dim),
                                                               [0], [0], jt, 1), proj), [0], aggs)
    want = {}
    for c in X.drain(host):
        for r in c.rows():
            want[r[0]] = r
    dfact, ddim = GP.DeviceChunk.from_host(ctx, fact), GP.DeviceChunk.from_host(ctx, dim)
    try:
        dev = GP.GpuHashAggExec(ctx, GP.GpuProjectionExec(ctx, GP.GpuHashJoinExec(ctx, GP.GpuSelectionExec(ctx, GP.DeviceTableScan(ctx, dfact, 30_000), filt),
                                                                               GP.DeviceTableScan(ctx, ddim, 4_000), [0], [0], jt, 1), proj), [0], aggs)
        got = {}
        for c in GP.drain_device(dev):
            for r in c.rows():
                assert r[0] not in got
                got[r[0]] = r
    finally:
        dfact.free()
        ddim.free()
    assert set(got) == set(want) and len(got) > 30 and None in got
    for k, w in want.items():
        g = got[k]
        assert g[:4] == w[:4], (k, g, w)                                  # firstrow(key), count, sum(int), avg(int): bit-exact
        for a, b in zip(g[4:], w[4:]):                                    # sum / max of doubles that are multiples of 1/4: exact as well
            assert a == b or (a is not None and b is not None and abs(a - b) <= 1e-9 * max(1.0, abs(b))), (k, g, w)


def test_q3_shaped_plan_with_order_by_revenue_limit_10(ctx, orc):
    # the full Q3 shape: ... GROUP BY ... ORDER BY revenue DESC, o_orderdate LIMIT 10 — the TopN runs on the device chunk of groups
    customer, orders, lineitem = q3.tables(0.05)
    want = q3_by_the_oracle(orc, customer, orders, lineitem)
    dev = [GP.DeviceChunk.from_host(ctx, t) for t in (customer, orders, lineitem)]
    try:
        out = GP.drain_device(q3.plan(ctx, *dev, batch_rows=50_000, topn=10))
        rows = [r for c in out for r in c.rows()]
        # the oracle's groups in the query's order (orc.sort_rows: SortExec restated, stable): revenue DESC, o_orderdate
        ks = sorted(want)
        groups = Chunk([Column(abi.I64, np.array(ks)), Column(abi.I64, np.array([want[k][0] for k in ks])), Column(abi.I64, np.array([want[k][1] for k in ks])),
                        Column(abi.F64, np.array([want[k][2] for k in ks]))])
        top = orc.sort_rows(groups, [3, 1], [True, False]).rows()[:10]
        assert len(rows) == 10
        for r, w in zip(rows, top):
            # the ten largest revenues are far apart compared with the SUM tolerance, so the order itself is pinned
            assert r[0] == w[0] and r[1] == w[1] and r[2] == w[2] and abs(r[3] - w[3]) <= sum_tol(want[w[0]][3], want[w[0]][4])
        assert all(rows[i][3] >= rows[i + 1][3] for i in range(9))
    finally:
        for d in dev:
            d.free()


def test_string_columns_stay_in_hbm_through_the_pipeline(ctx, orc):
    # the Q3 shape with its REAL predicate — c_mktsegment = 'BUILDING' is a string compare — and string columns travelling along:
    # Selection(customer: segment = 'BUILDING') -> HashJoin(orders) with the customer's name as payload -> HashAgg GROUP BY name
    # (MAX of a string, COUNT) -> ORDER BY count DESC, key with the name as payload; var-len columns never leave HBM between the
    # operators (tsq_chunk_compact, tsq_join_peek / tsq_agg_peek / tsq_sort_peek size the device buffers)
    from tinysql_amd import expression as E
    from tinysql_amd.chunk import StrColumn
    from tinysql_amd.executor import AggFuncDesc
    rng = np.random.default_rng(12)
    nc, no = 5000, 60_000
    segs = [b"BUILDING", b"AUTOMOBILE", b"MACHINERY", b"FURNITURE", b"HOUSEHOLD"]
    customer = Chunk([Column(abi.I64, np.arange(nc, dtype=np.int64), None), StrColumn([None if i % 97 == 0 else segs[int(rng.integers(0, 5))] for i in range(nc)]),
                      StrColumn([b"Customer#%09d" % (i % 700) for i in range(nc)])])
    orders = Chunk([Column(abi.I64, np.arange(no, dtype=np.int64), None), Column(abi.I64, rng.integers(0, nc, no), rng.random(no) >= 0.02),
                    StrColumn([None if rng.random() < 0.1 else b"P%d" % int(rng.integers(0, 9)) + b"-" * int(rng.integers(0, 6)) for _ in range(no)])])
    cond = [E.ScalarFunction("eq", E.Column(1, abi.BYTES), E.Constant("BUILDING"))]
    # oracle chain
    keep, _, _ = orc.filter_eval(E.compile_list(cond), 1, customer)
    idx = np.flatnonzero(keep)
    cust = Chunk([Column(abi.I64, customer.columns[0].data[idx], None), StrColumn([customer.columns[1].values()[i] for i in idx]),
                  StrColumn([customer.columns[2].values()[i] for i in idx])])
    j = orc.hash_join(H.join_cfg(orders.types(), cust.types(), [1], [0], abi.JOIN_INNER, 1), cust, orders)  # orders probe (left), customer build
    aggs = [(abi.AGG_FIRSTROW, 5, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_MAX, 2, abi.BYTES), (abi.AGG_MIN, 0, abi.I64)]
    want = orc.hash_agg(H.agg_cfg(j.types(), [5], aggs), j, 4, 4)
    want_sorted = orc.sort_rows(want, [1, 3], [True, False])
    # device pipeline
    dcust, dord = GP.DeviceChunk.from_host(ctx, customer), GP.DeviceChunk.from_host(ctx, orders)
    try:
        sel = GP.GpuSelectionExec(ctx, GP.DeviceTableScan(ctx, dcust, batch_rows=2048), cond)
        join = GP.GpuHashJoinExec(ctx, GP.DeviceTableScan(ctx, dord, batch_rows=16384), sel, [1], [0], abi.JOIN_INNER, 1, pull_rows=1 << 15)
        descs = [AggFuncDesc(f, c, t) for f, c, t in aggs]
        agg = GP.GpuHashAggExec(ctx, join, [5], descs, pull_rows=1 << 12)
        top = GP.GpuSortExec(ctx, agg, [1, 3], [True, False], pull_rows=1 << 12)
        got = [r for chk in GP.drain_device(top) for r in chk.rows()]
    finally:
        dcust.free()
        dord.free()
    assert len(got) == want.NumRows() and len(got) > 100
    assert got == want_sorted.rows()  # (count DESC, min orderkey): a total order — names included, row for row
