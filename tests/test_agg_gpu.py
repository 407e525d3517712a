"""GPU parity: HIP hash aggregation (through the C-ABI) vs the oracle and the reference's KATs.

COUNT / SUM(int) / AVG(int) / MAX / MIN are bit-exact; SUM/AVG(double) are compared with the
re-ordering bound |gpu-ref| <= 2*n_g*2^-53*sum|v_i|; group order is unspecified
(Go map order in the reference), so rows are matched by key.
"""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column, concat
from tinysql_amd.executor import AggFuncDesc, HashAggExec, MockDataSource, drain

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu


def out_types_for(aggs):
    t = []
    for a in aggs:
        mode = a[3] if len(a) > 3 else abi.MODE_COMPLETE
        t += AggFuncDesc(a[0], a[1], a[2], mode).out_types()
    return t


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
def test_aggfunc_kats(ctx, case):
    func = H.AGG_FUNCS[case["func"]]
    if "complete" in case:
        chk, tp = _agg_input(case["type"], 5, trailing_null=True)
        aggs = [(func, 0, tp)]
        out = G.run_agg(ctx, H.agg_cfg([tp], [], aggs), chk, out_types_for(aggs))
        assert out.NumRows() == 1 and H.approx_equal(out.rows()[0][0], case["complete"][1], 0), case["ref"]
        out = G.run_agg(ctx, H.agg_cfg([tp], [], aggs), H.chunk_from_rows([], [tp]), out_types_for(aggs))
        assert out.rows() == [(case["complete"][0],)], case["ref"]
    if "merge" in case and func != abi.AGG_FIRSTROW:
        chk, tp = _agg_input(case["type"], 5)
        paggs = [(func, 0, tp, abi.MODE_PARTIAL1)]
        pt = out_types_for(paggs)
        p1 = G.run_agg(ctx, H.agg_cfg([tp], [], paggs), chk, pt)
        p2 = G.run_agg(ctx, H.agg_cfg([tp], [], paggs), chk.slice(2, 5), pt)
        both = concat([p1, p2], pt)
        if func == abi.AGG_AVG:
            faggs = [(func, 0, tp, abi.MODE_FINAL, 1)]
        else:
            faggs = [(func, 0, pt[0] if func != abi.AGG_COUNT else tp, abi.MODE_FINAL)]
        out = G.run_agg(ctx, H.agg_cfg(pt, [], faggs), both, out_types_for([(func, 0, tp)]))
        assert H.approx_equal(out.rows()[0][0], case["merge"][2], 0), case["ref"]


@pytest.mark.parametrize("case", H.golden("agg_cases.json")["sql"], ids=lambda c: c["ref"][:40])
def test_agg_sql_rows(ctx, case):
    types = [H.TYPES[t] for t in case["types"]]
    chk = H.chunk_from_rows(case["rows"], types)
    aggs = [(H.AGG_FUNCS[f], col, H.TYPES[t]) for f, col, t in case["aggs"]]
    out = G.run_agg(ctx, H.agg_cfg(types, case["group_by"], aggs), chk, out_types_for(aggs))
    assert H.rows_equal_unordered(out, [tuple(r) for r in case["expect"]]), case["ref"]


def group_tols(chk, key_idx, aggs, real_cols):
    """SUM(double) of a group depends on the order of its rows (partial -> final workers); any two orders differ
    by at most 2 * n_g * 2^-53 * sum_g |v_i| — PER GROUP, over the group's non-NULL argument cells.  AVG = that sum / n_g (+ one
    rounding of the division).  Returns {canonical single group key: {output column: tolerance}} for an integer group key."""
    kc = chk.columns[key_idx]
    keys = kc.data.astype(np.int64, copy=False)
    knn = np.ones(len(keys), bool) if kc.notnull is None else kc.notnull
    out = {}
    for c in real_cols:
        func, arg = aggs[c][0], aggs[c][1]
        vc = chk.columns[arg]
        v = np.abs(vc.data.astype(np.float64))
        ok = np.ones(len(v), bool) if vc.notnull is None else vc.notnull
        for sel, null_group in ((ok & knn, False), (ok & ~knn, True)):
            if not sel.any():
                continue
            if null_group:
                groups, n_g, s_g = [None], [int(sel.sum())], [float(v[sel].sum())]
            else:
                uk, inv = np.unique(keys[sel], return_inverse=True)
                groups, n_g, s_g = uk.tolist(), np.bincount(inv).tolist(), np.bincount(inv, weights=v[sel]).tolist()
            for g, n, sa in zip(groups, n_g, s_g):
                t = 2.0 * n * 2.0 ** -53 * sa
                if func == abi.AGG_AVG:
                    t = t / n + (sa / n) * 2.0 ** -52
                out.setdefault((H.canon(g),), {})[c] = t
    return out


def _match_by_key(got, want, key_cols, exact_cols, real_cols, tol):
    """tol: one bound for every group (a float), or group_tols()' per-group bounds"""
    g = {tuple(H.canon(r[c]) for c in key_cols): r for r in got.rows()}
    w = {tuple(H.canon(r[c]) for c in key_cols): r for r in want.rows()}
    assert len(g) == got.NumRows() and set(g) == set(w)
    for k, wr in w.items():
        gr = g[k]
        for c in exact_cols:
            assert H.canon(gr[c]) == H.canon(wr[c]), (k, c, gr, wr)
        for c in real_cols:
            t = tol.get(k, {}).get(c, 0.0) if isinstance(tol, dict) else tol
            assert H.approx_equal(gr[c], wr[c], t), (k, c, gr, wr, t)


def test_agg_random_vs_oracle_single_key(ctx, orc):
    rng = np.random.default_rng(21)
    n = 60000
    k = Column(abi.I64, rng.integers(-500, 500, n), rng.random(n) > 0.02)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-10**6, hi=10**6)
    d = H.random_column(rng, abi.F64, n, 0.1)
    f = Column(abi.F32, rng.integers(-50, 50, n).astype(np.float32), rng.random(n) > 0.1)
    u = H.random_column(rng, abi.U64, n, 0.1)
    chk = Chunk([k, v, d, f, u])
    types = [abi.I64, abi.I64, abi.F64, abi.F32, abi.U64]
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_COUNT, 1, abi.I64), (abi.AGG_SUM, 1, abi.I64),
            (abi.AGG_AVG, 1, abi.I64), (abi.AGG_MAX, 1, abi.I64), (abi.AGG_MIN, 1, abi.I64), (abi.AGG_MAX, 4, abi.U64),
            (abi.AGG_MIN, 4, abi.U64), (abi.AGG_MAX, 2, abi.F64), (abi.AGG_MIN, 3, abi.F32), (abi.AGG_SUM, 2, abi.F64),
            (abi.AGG_AVG, 2, abi.F64), (abi.AGG_SUM, 3, abi.F32)]
    cfg = H.agg_cfg(types, [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1024)
    _match_by_key(got, want, [0], list(range(1, 11)), [11, 12, 13], group_tols(chk, 0, aggs, [11, 12, 13]))


def test_agg_multi_key_growth_and_modes(ctx, orc):
    rng = np.random.default_rng(22)
    n = 80000
    k1 = Column(abi.I64, rng.integers(0, 300, n), rng.random(n) > 0.02)
    k2 = Column(abi.F64, rng.integers(0, 7, n).astype(np.float64) * np.where(rng.random(n) < 0.5, -1.0, 1.0), rng.random(n) > 0.02)
    k3 = Column(abi.U64, rng.integers(0, 4, n).astype(np.uint64))
    v = H.random_column(rng, abi.I64, n, 0.05, lo=-1000, hi=1000)
    chk = Chunk([k1, k2, k3, v])
    types = [abi.I64, abi.F64, abi.U64, abi.I64]
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_FIRSTROW, 2, abi.U64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 3, abi.I64),
            (abi.AGG_MAX, 3, abi.I64)]
    cfg = H.agg_cfg(types, [0, 1, 2], aggs, est_groups=8)  # tiny hint: forces table growth + retry rounds
    want = orc.hash_agg(cfg, chk, 4, 4)
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs))
    # group identity = (k1, |k2| sign-collapsed zero, k3): compare by the order-insensitive multiset of
    # (first_row(k1), first_row(k3), count, sum, max) since k2 itself is not projected
    assert got.NumRows() == want.NumRows()
    assert H.rows_equal_unordered(got, want)
    # Partial1 on two halves -> Final merge == Complete (aggregate.go two-phase protocol)
    paggs = [(abi.AGG_FIRSTROW, 0, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_COUNT, -1, abi.I64, abi.MODE_PARTIAL1),
             (abi.AGG_SUM, 3, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_AVG, 3, abi.I64, abi.MODE_PARTIAL1),
             (abi.AGG_MIN, 3, abi.I64, abi.MODE_PARTIAL1)]
    pt = out_types_for(paggs)
    pcfg = H.agg_cfg(types, [0], paggs)
    p = concat([G.run_agg(ctx, pcfg, chk.slice(0, n // 2), pt), G.run_agg(ctx, pcfg, chk.slice(n // 2, n), pt)], pt)
    faggs = [(abi.AGG_FIRSTROW, 0, abi.I64, abi.MODE_FINAL), (abi.AGG_COUNT, 1, abi.I64, abi.MODE_FINAL),
             (abi.AGG_SUM, 2, abi.I64, abi.MODE_FINAL), (abi.AGG_AVG, 3, abi.I64, abi.MODE_FINAL, 4), (abi.AGG_MIN, 5, abi.I64, abi.MODE_FINAL)]
    fin = G.run_agg(ctx, H.agg_cfg(pt, [0], faggs), p, out_types_for([(a[0], a[1], a[2]) for a in faggs]))
    caggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 3, abi.I64), (abi.AGG_AVG, 3, abi.I64),
             (abi.AGG_MIN, 3, abi.I64)]
    comp = orc.hash_agg(H.agg_cfg(types, [0], caggs), chk, 1, 1)
    assert H.rows_equal_unordered(fin, comp)


def test_agg_float_group_key_zero_signs_and_null_group(ctx, orc):
    chk = H.chunk_from_rows([[0.0, 1], [-0.0, 2], [1.5, 3], [None, 4], [None, 5]], [abi.F64, abi.I64])
    aggs = [(abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    cfg = H.agg_cfg([abi.F64, abi.I64], [0], aggs)
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs))
    assert H.rows_equal_unordered(got, orc.hash_agg(cfg, chk, 1, 1))
    assert H.rows_equal_unordered(got, [(2, 3), (1, 3), (2, 9)])


def test_agg_int_sum_overflow_is_an_error(ctx):
    big = (1 << 63) - 1
    chk = H.chunk_from_rows([[1, big], [1, 1], [2, 5]], [abi.I64, abi.I64])
    aggs = [(abi.AGG_SUM, 1, abi.I64)]
    with pytest.raises(_lib.TsqError) as ei:
        G.run_agg(ctx, H.agg_cfg([abi.I64, abi.I64], [0], aggs), chk, out_types_for(aggs))
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT
    # a transient excursion that ends inside the range is NOT an error on the GPU (128-bit accumulate);
    # the reference reports it only for some worker interleavings (func_sum.go:133-137) — documented.
    chk = H.chunk_from_rows([[1, big], [1, 1], [1, -2]], [abi.I64, abi.I64])
    out = G.run_agg(ctx, H.agg_cfg([abi.I64, abi.I64], [0], aggs), chk, out_types_for(aggs))
    assert out.rows() == [(big - 1,)]


def test_agg_through_executor_interface(ctx, orc):
    rng = np.random.default_rng(23)
    n = 10000
    chk = Chunk([Column(abi.I64, rng.integers(0, 100, n)), Column(abi.F64, rng.random(n))])
    exe = HashAggExec(ctx, MockDataSource(ctx, chk, 1024), [0],
                      [AggFuncDesc(abi.AGG_FIRSTROW, 0, abi.I64), AggFuncDesc(abi.AGG_SUM, 1, abi.F64), AggFuncDesc(abi.AGG_COUNT, -1)],
                      max_chunk_size=32)
    chunks = drain(exe)
    assert all(0 < c.NumRows() <= 32 for c in chunks) and sum(c.NumRows() for c in chunks) == 100
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.F64), (abi.AGG_COUNT, -1, abi.I64)]
    want = orc.hash_agg(H.agg_cfg([abi.I64, abi.F64], [0], aggs), chk, 4, 4)
    got = concat(chunks, [abi.I64, abi.F64, abi.I64])
    _match_by_key(got, want, [0], [2], [1], group_tols(chk, 0, aggs, [1]))


def test_agg_config3_shape_device_resident_property(ctx):
    # BASELINE config[2] shape (SELECT k, SUM(v), COUNT(*) GROUP BY k), scaled to 1e8 rows / 1e6 groups:
    # k = r(i,0) mod G, v int = r(i,1) mod 1000.  Properties: #groups == G, sum(COUNT) == N,
    # sum(SUM) == sum(v) and every (k, SUM, COUNT) equals a numpy bincount restatement.
    lib = ctx.lib
    n, g = 100_000_000, 1_000_000
    k, v = G.DevCol(ctx, abi.I64, n), G.DevCol(ctx, abi.I64, n)
    try:
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=3, col=0, m=g), n, k.data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=3, col=1, m=1000), n, v.data)
        aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64)]
        cfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs, est_groups=g)
        h = C.c_void_p()
        _lib.check(lib.tsq_agg_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
        try:
            _lib.check(lib.tsq_agg_push(h, G.dev_cols([k, v]), 2, n), h)
            _lib.check(lib.tsq_agg_finish(h), h)
            ng = C.c_int64(0)
            _lib.check(lib.tsq_agg_num_groups(h, C.byref(ng)), h)
            assert ng.value == g
            outs = [G.DevCol(ctx, abi.I64, g, with_nulls=True) for _ in range(3)]
            nr, eos = C.c_int64(0), C.c_int32(0)
            _lib.check(lib.tsq_agg_pull(h, G.dev_cols(outs), 3, g, C.byref(nr), C.byref(eos)), h)
            assert nr.value == g
            keys, sums, cnts = [o.to_host().data for o in outs]
            for o in outs:
                o.free()
        finally:
            lib.tsq_agg_destroy(h)
    finally:
        k.free()
        v.free()
    exp_sum, exp_cnt = np.zeros(g, np.int64), np.zeros(g, np.int64)
    step = 1 << 24
    for lo in range(0, n, step):
        i = np.arange(lo, min(n, lo + step), dtype=np.uint64)
        kk = (G.np_gen_r(42, 3, 0, i) % np.uint64(g)).astype(np.int64)
        vv = (G.np_gen_r(42, 3, 1, i) % np.uint64(1000)).astype(np.int64)
        exp_cnt += np.bincount(kk, minlength=g)
        exp_sum += np.bincount(kk, weights=vv, minlength=g).astype(np.int64)
    order = np.argsort(keys)
    assert (keys[order] == np.arange(g)).all()
    assert (cnts[order] == exp_cnt).all() and (sums[order] == exp_sum).all()


def test_sum_int64_running_overflow_divergence_is_pinned(ctx, orc):
    # DESIGN.md "Known divergences": the reference adds row by row and reports ErrOverflow as soon as a RUNNING sum leaves BIGINT
    # (aggfuncs/func_sum.go:133-137, types/overflow.go:33-40) — with one partial worker that is deterministic, and the oracle
    # reproduces it.  The GPU sums exactly in 128 bits and reports overflow iff the FINAL sum of a group leaves BIGINT, so a
    # transient excursion that comes back into range is a value here and an error there.  INTEGRATION.md lists it.
    from oracle.binding import OracleError
    i64max = (1 << 63) - 1
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    cfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs)
    transient = Chunk([Column(abi.I64, np.array([1, 1, 1, 2])), Column(abi.I64, np.array([i64max, 1, -5, 7]))])
    with pytest.raises(OracleError) as ei:
        orc.hash_agg(cfg, transient, 1, 1)
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT
    got = G.run_agg(ctx, cfg, transient, [abi.I64, abi.I64])
    assert sorted(got.rows()) == [(1, i64max - 4), (2, 7)]
    # a final sum outside BIGINT is an error on both sides
    final = Chunk([Column(abi.I64, np.array([1, 1, 2])), Column(abi.I64, np.array([i64max, 5, 7]))])
    with pytest.raises(OracleError):
        orc.hash_agg(cfg, final, 1, 1)
    with pytest.raises(_lib.TsqError) as e2:
        G.run_agg(ctx, cfg, final, [abi.I64, abi.I64])
    assert e2.value.status == abi.ERR_OVERFLOW_BIGINT


@pytest.mark.parametrize("bits", [7, 10])
def test_multi_key_tag_collisions_are_resolved_not_reported(ctx, orc, bits):
    # TSQ_AGG_TAG_BITS truncates the 64-bit tag of the multi-key path: with 2^bits tags for ~7 K groups, 7 to 55 different keys
    # share every tag (a run of that many slots; the walk limit is 256), which is what a real 64-bit collision looks like to the
    # table.  Every group must still come out once, with its own rows.
    import os

    rng = np.random.default_rng(bits)
    n = 120_000
    a = Column(abi.I64, rng.integers(0, 90, n), rng.random(n) > 0.03)
    b = Column(abi.I64, rng.integers(-40, 40, n), rng.random(n) > 0.03)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-1000, hi=1000)
    chk = Chunk([a, b, v])
    types = [abi.I64, abi.I64, abi.I64]
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_FIRSTROW, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 2, abi.I64), (abi.AGG_MAX, 2, abi.I64)]
    cfg = H.agg_cfg(types, [0, 1], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    # (AGG_WIDE_KEYS = 0: the several-column upsert itself is under test — since round 4 integer key columns would otherwise become one
    # composite key for a single-key child aggregate, which has no tags to collide)
    with ctx.knobs(AGG_TAG_BITS=bits, AGG_WIDE_KEYS=0):
        stats = []
        got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 22, stats_out=stats)
        got2 = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=7000)  # many batches: later rows meet earlier groups
    assert stats[0].build_handed_back_rows > n // 2  # most rows really went the long way
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    assert H.rows_equal_unordered(got2, want)
