"""GPU parity: StreamAggExec (tsq_agg_set_stream, csrc/tsq_streamagg.h) through the C-ABI.

The reference has no StreamAggExec body (only the plan name, planner/core/cbo_test.go:200-212), so the checker is the oracle's
HashAggExec restatement on the same rows: the SAME groups and aggregate values (COUNT / SUM(int) / AVG(int) / MAX / MIN bit-exact,
SUM / AVG(double) within the re-ordering bound of SURVEY.md 8d), and — what makes it a stream aggregate — emitted in the ORDER in which
the groups appear in the (key-ordered) input, with FIRST_ROW = the first row of the group.  The reference's aggregate KATs
(tests/golden/agg_cases.json, transcribed from executor/aggfuncs/*_test.go and executor/aggregate_test.go) are replayed in stream mode too.
"""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column, StrColumn, concat
from tinysql_amd.executor import AggFuncDesc, HashAggExec, MockDataSource, StreamAggExec, drain

from . import gpu_helpers as G
from . import helpers as H
from .test_agg_gpu import _agg_input, _match_by_key, group_tols, out_types_for

pytestmark = pytest.mark.gpu


def _sorted_by(chk, key_cols):
    """the rows of chk ordered by the key columns (NULL first, then by value; stable) — what a SortExec below the operator delivers"""
    rows = chk.rows()
    def k(r):
        return tuple((0, 0) if r[c] is None else (1, (r[c] + 0.0) if isinstance(r[c], float) else r[c]) for c in key_cols)
    order = sorted(range(len(rows)), key=lambda i: k(rows[i]))
    return chk.take(np.asarray(order, dtype=np.int64)) if hasattr(chk, "take") else H.chunk_from_rows([list(rows[i]) for i in order], chk.types())


def _first_appearance_order(chk, key_cols):
    seen, order = set(), []
    for r in chk.rows():
        k = tuple(H.canon(r[c]) for c in key_cols)
        if k not in seen:
            seen.add(k)
            order.append(k)
    return order


@pytest.mark.parametrize("case", H.golden("agg_cases.json")["funcs"], ids=lambda c: "%s-%s" % (c["func"], c["type"]))
def test_stream_aggfunc_kats_without_group_by(ctx, case):
    func = H.AGG_FUNCS[case["func"]]
    if "complete" in case:
        chk, tp = _agg_input(case["type"], 5, trailing_null=True)
        aggs = [(func, 0, tp)]
        out = G.run_agg(ctx, H.agg_cfg([tp], [], aggs), chk, out_types_for(aggs), stream=True)
        assert out.NumRows() == 1 and H.approx_equal(out.rows()[0][0], case["complete"][1], 0), case["ref"]
        out = G.run_agg(ctx, H.agg_cfg([tp], [], aggs), H.chunk_from_rows([], [tp]), out_types_for(aggs), stream=True)
        assert out.rows() == [(case["complete"][0],)], case["ref"]  # the empty-input default row (aggregate.go:572-574)
    if "merge" in case and func != abi.AGG_FIRSTROW:
        chk, tp = _agg_input(case["type"], 5)
        paggs = [(func, 0, tp, abi.MODE_PARTIAL1)]
        pt = out_types_for(paggs)
        p1 = G.run_agg(ctx, H.agg_cfg([tp], [], paggs), chk, pt, stream=True)
        p2 = G.run_agg(ctx, H.agg_cfg([tp], [], paggs), chk.slice(2, 5), pt, stream=True)
        both = concat([p1, p2], pt)
        faggs = [(func, 0, tp, abi.MODE_FINAL, 1)] if func == abi.AGG_AVG else [(func, 0, pt[0] if func != abi.AGG_COUNT else tp, abi.MODE_FINAL)]
        out = G.run_agg(ctx, H.agg_cfg(pt, [], faggs), both, out_types_for([(func, 0, tp)]), stream=True)
        assert H.approx_equal(out.rows()[0][0], case["merge"][2], 0), case["ref"]


@pytest.mark.parametrize("case", H.golden("agg_cases.json")["sql"], ids=lambda c: c["ref"][:40])
def test_stream_agg_sql_rows_on_ordered_input(ctx, orc, case):
    types = [H.TYPES[t] for t in case["types"]]
    chk = _sorted_by(H.chunk_from_rows(case["rows"], types), case["group_by"])
    aggs = [(H.AGG_FUNCS[f], col, H.TYPES[t]) for f, col, t in case["aggs"]]
    cfg = H.agg_cfg(types, case["group_by"], aggs)
    out = G.run_agg(ctx, cfg, chk, out_types_for(aggs), stream=True)
    if not any(a[0] == abi.AGG_FIRSTROW and a[1] not in case["group_by"] for a in aggs):  # (FIRST_ROW of a non-key column depends on the row order)
        assert H.rows_equal_unordered(out, [tuple(r) for r in case["expect"]]), case["ref"]
    assert H.rows_equal_unordered(out, orc.hash_agg(cfg, chk, 1, 1)), case["ref"]


AGGS_1K = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_COUNT, 1, abi.I64), (abi.AGG_SUM, 1, abi.I64),
           (abi.AGG_AVG, 1, abi.I64), (abi.AGG_MAX, 1, abi.I64), (abi.AGG_MIN, 1, abi.I64), (abi.AGG_MAX, 4, abi.U64),
           (abi.AGG_MIN, 4, abi.U64), (abi.AGG_MAX, 2, abi.F64), (abi.AGG_MIN, 3, abi.F32), (abi.AGG_SUM, 2, abi.F64),
           (abi.AGG_AVG, 2, abi.F64), (abi.AGG_SUM, 3, abi.F32)]
TYPES_1K = [abi.I64, abi.I64, abi.F64, abi.F32, abi.U64]


def _chunk_1k(rng, n, nkeys, null_key=0.02):
    keys = np.sort(rng.integers(-nkeys // 2, nkeys // 2 + 1, n))
    knn = np.ones(n, bool)
    nn = int(n * null_key)
    if nn:  # NULL keys first (one run)
        knn[:nn] = False
    k = Column(abi.I64, keys, knn)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-10**6, hi=10**6)
    d = H.random_column(rng, abi.F64, n, 0.1)
    f = Column(abi.F32, rng.integers(-50, 50, n).astype(np.float32), rng.random(n) > 0.1)
    u = H.random_column(rng, abi.U64, n, 0.1)
    return Chunk([k, v, d, f, u])


def _check_stream_vs_oracle(ctx, orc, chk, key_cols, aggs, exact, real, **run_kw):
    cfg = H.agg_cfg(chk.types(), key_cols, aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), stream=True, **run_kw)
    # the key columns come out as FIRST_ROW(key) in output columns [0, len(key_cols))
    kc = list(range(len(key_cols)))
    _match_by_key(got, want, kc, exact, real, group_tols(chk, key_cols[0], aggs, real) if (real and len(key_cols) == 1) else 0.0)
    # ... in the order the groups appear in the input
    assert [tuple(H.canon(r[c]) for c in kc) for r in got.rows()] == _first_appearance_order(chk, key_cols)
    return got


@pytest.mark.parametrize("nkeys,chunk_rows", [(700, 1024), (40000, 1000), (3, 1024)])
def test_stream_agg_random_single_key_vs_oracle(ctx, orc, nkeys, chunk_rows):
    rng = np.random.default_rng(51)
    chk = _chunk_1k(rng, 60000, nkeys)
    _check_stream_vs_oracle(ctx, orc, chk, [0], AGGS_1K, list(range(1, 11)), [11, 12, 13], chunk_rows=chunk_rows)


@pytest.mark.parametrize("lanes", [1, 0])
@pytest.mark.parametrize("aggset", ["sum_count_max", "avg_min_real", "merge_modes"])
def test_stream_agg_long_and_short_runs_per_lane_accumulators(ctx, orc, aggset, lanes):
    """plans with <= 4 reducing aggregates keep the open run's partial results in registers, one per lane (k_sa_update_lanes; knob
    STREAMAGG_LANES = 0: every 64-row step is reduced across the lanes, k_sa_update) — runs from 1 to 9000 rows, so that steps
    without a head, steps that start with one, steps with several, stripes (4096 rows) that end inside a run and NULL arguments all occur"""
    rng = np.random.default_rng(len(aggset) + lanes)
    lens = np.concatenate([rng.integers(1, 4, 3000), rng.integers(50, 200, 300), rng.integers(3000, 9000, 30), rng.integers(1, 70, 2000)])
    rng.shuffle(lens)
    keys = np.repeat(np.arange(len(lens), dtype=np.int64) * 7 - 1000, lens)
    n = len(keys)
    knn = np.ones(n, bool)
    knn[:int(lens[0])] = False  # the first run is the NULL group
    chk = Chunk([Column(abi.I64, keys, knn), H.random_column(rng, abi.I64, n, 0.2, lo=-10**9, hi=10**9), H.random_column(rng, abi.F64, n, 0.1),
                 Column(abi.I64, np.where(np.arange(n) % 2 == 0, 1 << 60, -(1 << 60)) + rng.integers(-1000, 1000, n))])  # (cells beyond 2^56: the 128-bit scans; alternating signs keep every running sum inside BIGINT)
    types = chk.types()
    if aggset == "sum_count_max":
        aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, 1, abi.I64), (abi.AGG_MAX, 2, abi.F64), (abi.AGG_SUM, 3, abi.I64)]
        exact, real = [1, 2, 3], []
    elif aggset == "avg_min_real":
        aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_AVG, 1, abi.I64), (abi.AGG_MIN, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 2, abi.F64)]
        exact, real = [1, 2, 3], [4]
    else:  # Partial1 over the rows, then Final over the partial rows (still ordered by key): COUNT / SUM / AVG merge their partial columns
        aggs = None
    with ctx.knobs(STREAMAGG_LANES=lanes, AGG_BATCH_ROWS=100_000):
        if aggs is not None:
            cfg = H.agg_cfg(types, [0], aggs)
            want = orc.hash_agg(cfg, chk, 4, 4)
            got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), stream=True, chunk_rows=4096)
            if aggset == "sum_count_max":  # SUM(column 3): the 128-bit path (cells up to 2^62): exact too
                exact = exact + [4]
            _match_by_key(got, want, [0], exact, real, group_tols(chk, 0, aggs, real) if real else 0.0)
            assert [H.canon(r[0]) for r in got.rows()] == [k[0] for k in _first_appearance_order(chk, [0])]
        else:
            paggs = [(abi.AGG_FIRSTROW, 0, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_COUNT, 1, abi.I64, abi.MODE_PARTIAL1), (abi.AGG_SUM, 1, abi.I64, abi.MODE_PARTIAL1),
                     (abi.AGG_AVG, 1, abi.I64, abi.MODE_PARTIAL1)]
            pt = out_types_for(paggs)
            half = n // 2
            parts = concat([G.run_agg(ctx, H.agg_cfg(types, [0], paggs), chk.slice(0, half), pt, stream=True),
                            G.run_agg(ctx, H.agg_cfg(types, [0], paggs), chk.slice(half, n), pt, stream=True)], pt)  # (the run cut at `half` yields two partial rows of one key, adjacent)
            faggs = [(abi.AGG_FIRSTROW, 0, abi.I64, abi.MODE_FINAL), (abi.AGG_COUNT, 1, abi.I64, abi.MODE_FINAL), (abi.AGG_SUM, 2, abi.I64, abi.MODE_FINAL),
                     (abi.AGG_AVG, 3, abi.I64, abi.MODE_FINAL, 4)]
            fin = G.run_agg(ctx, H.agg_cfg(pt, [0], faggs), parts, out_types_for([(a[0], a[1], a[2]) for a in faggs]), stream=True)
            caggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, 1, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_AVG, 1, abi.I64)]
            comp = orc.hash_agg(H.agg_cfg(types, [0], caggs), chk, 1, 1)
            assert H.rows_equal_unordered(fin, comp)


def test_stream_agg_groups_continue_across_device_batches(ctx, orc):
    """host chunks are aggregated in device batches (knob: 4096 rows here): a group that straddles two batches is ONE group — the
    open group's key cells stay in the table and row 0 of the next batch is compared with them"""
    rng = np.random.default_rng(52)
    chk = _chunk_1k(rng, 50000, 37)
    ctx.set_knob(abi.KNOB_AGG_BATCH_ROWS, 4096)
    try:
        got = _check_stream_vs_oracle(ctx, orc, chk, [0], AGGS_1K, list(range(1, 11)), [11, 12, 13], chunk_rows=999)
    finally:
        ctx.set_knob(abi.KNOB_AGG_BATCH_ROWS, abi.KNOB_DEFAULT)
    assert got.NumRows() == 39  # 38 key values + the NULL group
    # FIRST_ROW of a NON-key column is the first row of the group (deterministic here, unlike under hash workers)
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_FIRSTROW, 1, abi.I64), (abi.AGG_FIRSTROW, 2, abi.F64)]
    out = G.run_agg(ctx, H.agg_cfg(TYPES_1K, [0], aggs), chk, out_types_for(aggs), stream=True)
    rows, first = chk.rows(), {}
    for r in rows:
        first.setdefault(H.canon(r[0]), (r[0], r[1], r[2]))
    assert out.rows() == [first[k] for k in [t[0] for t in _first_appearance_order(chk, [0])]]


def test_stream_agg_every_row_its_own_group_and_growth_of_the_group_array(ctx, orc):
    n = 200000  # > the 2^16 groups the table is created with: the group array grows by copy, numbers stay
    k = Column(abi.I64, np.arange(n, dtype=np.int64) * 3 - 7)
    v = Column(abi.I64, (np.arange(n, dtype=np.int64) * 2654435761) % 1000)
    chk = Chunk([k, v])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64)]
    out = G.run_agg(ctx, H.agg_cfg([abi.I64, abi.I64], [0], aggs), chk, out_types_for(aggs), stream=True, chunk_rows=65536, pull_rows=65536)
    assert out.NumRows() == n
    assert (out.columns[0].data == k.data).all() and (out.columns[1].data == v.data).all() and (out.columns[2].data == 1).all()


def test_stream_agg_several_key_columns_float_zero_signs_null_runs(ctx, orc):
    rng = np.random.default_rng(53)
    n = 30000
    a = np.sort(rng.integers(0, 40, n))
    b = rng.integers(0, 5, n).astype(np.float64)
    order = np.lexsort((b, a))
    a, b = a[order], b[order]
    b = b * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    b[b == 0] = np.where(rng.random(int((b == 0).sum())) < 0.5, -0.0, 0.0)  # +0.0 and -0.0 are ONE group key (codec.go:713-746 via float.go)
    # sort again by (a, |b| with sign) so equal keys are adjacent: group identity of a real is its memcomparable image
    key_b = np.where(b >= 0, b, b)  # (-0.0 >= 0: with +0.0)
    order = np.lexsort((key_b + 0.0, a))
    a, b = a[order], b[order]
    ann = np.ones(n, bool)
    ann[:500] = False  # the NULL run of the first key column first
    chk = Chunk([Column(abi.I64, a, ann), Column(abi.F64, b), H.random_column(rng, abi.I64, n, 0.05, lo=-1000, hi=1000)])
    chk = _sorted_by(chk, [0, 1])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 2, abi.I64), (abi.AGG_MAX, 2, abi.I64), (abi.AGG_AVG, 2, abi.I64)]
    cfg = H.agg_cfg(chk.types(), [0, 1], aggs)
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), stream=True)
    want = orc.hash_agg(cfg, chk, 4, 4)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    # order: non-decreasing first key, NULLs first
    k0 = [r[0] for r in got.rows()]
    assert k0 == sorted(k0, key=lambda x: (x is not None, x if x is not None else 0))


def test_stream_agg_string_key_and_string_values(ctx, orc):
    rng = np.random.default_rng(54)
    n = 20000
    words = sorted([("w%05d" % int(x)).encode() * (1 + int(x) % 3) for x in rng.integers(0, 900, n)])
    keys = [None] * 300 + words[300:]  # a NULL run first
    vals = [None if rng.random() < 0.1 else ("v%04d" % int(x)).encode() for x in rng.integers(0, 5000, n)]
    chk = Chunk([StrColumn(keys), StrColumn(vals), H.random_column(rng, abi.I64, n, 0.1, lo=-99, hi=99)])
    types = [abi.BYTES, abi.BYTES, abi.I64]
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, 1, abi.BYTES), (abi.AGG_MAX, 1, abi.BYTES), (abi.AGG_MIN, 1, abi.BYTES), (abi.AGG_SUM, 2, abi.I64)]
    cfg = H.agg_cfg(types, [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    for knobs in ({}, {abi.KNOB_AGG_BATCH_ROWS: 2048, abi.KNOB_AGG_HEAP_GC_BYTES: 4096}):  # (second run: many batches, the string heap compacted between them)
        for k, v in knobs.items():
            ctx.set_knob(k, v)
        try:
            got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), stream=True)
        finally:
            for k in knobs:
                ctx.set_knob(k, abi.KNOB_DEFAULT)
        assert H.rows_equal_unordered(got, want)
        assert [H.canon(r[0]) for r in got.rows()] == [k[0] for k in _first_appearance_order(chk, [0])]


def test_stream_agg_device_resident_batches(ctx, orc):
    rng = np.random.default_rng(55)
    chk = _chunk_1k(rng, 300000, 5000)
    aggs = AGGS_1K
    cfg = H.agg_cfg(TYPES_1K, [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    dev = [G.to_device(ctx, c) for c in chk.columns]  # the child's chunks already in HBM (TSQ_COL_DEVICE), pushed in 70 008-row batches
    lib, h = ctx.lib, G.C.c_void_p()
    _lib.check(lib.tsq_agg_create(ctx.h, G.C.byref(cfg), G.C.byref(h)), ctx.h)
    try:
        _lib.check(lib.tsq_agg_set_stream(h, 1), h)
        n, step = chk.NumRows(), 70008  # (a multiple of 8: the bitmap of a batch starts on a byte)
        for off in range(0, n, step):
            m = min(step, n - off)
            cols = G.dev_cols(dev)
            for c in cols:
                c.data = c.data + off * c.elem_size
                if c.null_bitmap:
                    c.null_bitmap = c.null_bitmap + off // 8
                c.length = m
            _lib.check(lib.tsq_agg_push(h, cols, len(dev), m), h)
        _lib.check(lib.tsq_agg_finish(h), h)
        ot = out_types_for(aggs)
        ng = G.C.c_int64(0)
        _lib.check(lib.tsq_agg_num_groups(h, G.C.byref(ng)), h)
        outs = [G.DevCol(ctx, t, ng.value, with_nulls=True) for t in ot]  # device pushes -> device pulls (the hand-off between GPU operators)
        try:
            oc = G.dev_cols(outs)
            got_n, eos = G.C.c_int64(0), G.C.c_int32(0)
            _lib.check(lib.tsq_agg_pull(h, oc, len(ot), ng.value, G.C.byref(got_n), G.C.byref(eos)), h)
            assert got_n.value == ng.value
            got = Chunk([o.to_host() for o in outs])
        finally:
            for o in outs:
                o.free()
    finally:
        lib.tsq_agg_destroy(h)
        for d in dev:
            d.free()
    _match_by_key(got, want, [0], list(range(1, 11)), [11, 12, 13], group_tols(chk, 0, aggs, [11, 12, 13]))
    assert [H.canon(r[0]) for r in got.rows()] == [k[0] for k in _first_appearance_order(chk, [0])]


def test_stream_agg_int_sum_overflow_is_an_error(ctx):
    big = (1 << 63) - 1
    chk = H.chunk_from_rows([[1, big], [1, 1], [2, 5]], [abi.I64, abi.I64])
    aggs = [(abi.AGG_SUM, 1, abi.I64)]
    with pytest.raises(_lib.TsqError) as ei:
        G.run_agg(ctx, H.agg_cfg([abi.I64, abi.I64], [0], aggs), chk, out_types_for(aggs), stream=True)
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT  # func_sum.go:133-137


def test_stream_agg_a_key_that_comes_back_opens_a_new_group(ctx):
    """the contract of a stream aggregate: its child's order.  Unordered input is not an error, it is other groups."""
    chk = H.chunk_from_rows([[1, 10], [1, 20], [2, 1], [1, 5], [None, 7], [None, 8], [2, 2]], [abi.I64, abi.I64])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64)]
    out = G.run_agg(ctx, H.agg_cfg([abi.I64, abi.I64], [0], aggs), chk, out_types_for(aggs), stream=True)
    assert out.rows() == [(1, 30, 2), (2, 1, 1), (1, 5, 1), (None, 15, 2), (2, 2, 1)]
    with pytest.raises(_lib.TsqError):  # the mode is chosen before the first row
        lib, h = ctx.lib, G.C.c_void_p()
        cfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs)
        _lib.check(lib.tsq_agg_create(ctx.h, G.C.byref(cfg), G.C.byref(h)), ctx.h)
        try:
            G.push_chunked(lib.tsq_agg_push, h, chk, 1024)
            _lib.check(lib.tsq_agg_set_stream(h, 1), h)
        finally:
            lib.tsq_agg_destroy(h)


def test_stream_agg_through_the_executor_interface(ctx, orc):
    rng = np.random.default_rng(56)
    n = 30000
    chk = Chunk([Column(abi.I64, rng.integers(0, 100, n), rng.random(n) > 0.03), Column(abi.F64, rng.random(n)), Column(abi.I64, rng.integers(-9, 9, n))])
    funcs = [AggFuncDesc(abi.AGG_FIRSTROW, 0, abi.I64), AggFuncDesc(abi.AGG_COUNT, -1), AggFuncDesc(abi.AGG_SUM, 2, abi.I64), AggFuncDesc(abi.AGG_AVG, 1, abi.F64)]
    # an UNORDERED child: the operator sorts below itself (child_is_ordered=False), groups come out in key order, NULL first
    exe = StreamAggExec(ctx, MockDataSource(ctx, chk, 1024), [0], funcs, child_is_ordered=False)
    ot = [abi.I64, abi.I64, abi.I64, abi.F64]
    got = concat(drain(exe), ot)
    ref = concat(drain(HashAggExec(ctx, MockDataSource(ctx, chk, 1024), [0], funcs)), ot)
    assert got.NumRows() == ref.NumRows() == 101
    keys = [r[0] for r in got.rows()]
    assert keys == [None] + list(range(100))
    g = {r[0]: r for r in got.rows()}
    for r in ref.rows():
        assert g[r[0]][:3] == r[:3] and abs(g[r[0]][3] - r[3]) <= 1e-12
