"""GPU parity of the LDS pre-aggregation path of HashAggExec (tsq_aggfast.h): forced on small inputs so
that the LDS tables, the radix partition with payload cells, the spill / exception / fallback routes and
the partial-group merge are all compared with the oracle; AUTO is checked at C3 scale through
size-independent properties.  Integer aggregates are bit-exact; SUM/AVG(double) within the re-ordering
bound 2*n_g*2^-53*sum_g|v| of every GROUP."""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H
from .test_agg_gpu import _match_by_key, group_tols, out_types_for

pytestmark = pytest.mark.gpu
SENT = np.uint64(0x8080808080808080).astype(np.int64)


def _run(ctx, cfg, chk, aggs, fast, chunk_rows=1 << 22, stats=None):
    return G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=chunk_rows, fast=fast, stats_out=stats)


@pytest.mark.parametrize("case", H.golden("agg_cases.json")["sql"], ids=lambda c: c["ref"][:40])
def test_fast_forced_on_golden_sql_rows(ctx, case):
    types = [H.TYPES[t] for t in case["types"]]
    chk = H.chunk_from_rows(case["rows"], types)
    aggs = [(H.AGG_FUNCS[f], col, H.TYPES[t]) for f, col, t in case["aggs"]]
    out = _run(ctx, H.agg_cfg(types, case["group_by"], aggs), chk, aggs, abi.AGGFAST_FORCE)
    assert H.rows_equal_unordered(out, [tuple(r) for r in case["expect"]]), case["ref"]


AGG_SETS = {
    "c3": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64)],
    "ints": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, 1, abi.I64), (abi.AGG_AVG, 1, abi.I64), (abi.AGG_MAX, 1, abi.I64)],
    "minmax2": [(abi.AGG_MIN, 1, abi.I64), (abi.AGG_MAX, 4, abi.U64), (abi.AGG_MIN, 4, abi.U64), (abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64)],
    "reals": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 2, abi.F64), (abi.AGG_AVG, 2, abi.F64), (abi.AGG_MAX, 2, abi.F64)],
    "f32": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 3, abi.F32), (abi.AGG_MIN, 3, abi.F32), (abi.AGG_COUNT, 3, abi.F32)],
}


@pytest.mark.parametrize("aggset", sorted(AGG_SETS))
@pytest.mark.parametrize("n,groups,est", [(1, 1, 0), (1000, 7, 0), (70001, 900, 0), (70001, 900, 900), (150001, 40000, 40000), (150001, 40000, 0)])
def test_fast_random_vs_oracle(ctx, orc, aggset, n, groups, est):
    # est = 0: unknown cardinality (LDS tables spill what does not fit); est = groups >= 2048: radix partition + LDS
    rng = np.random.default_rng(n + groups)
    k = Column(abi.I64, rng.integers(-groups // 2, groups - groups // 2, n), rng.random(n) > 0.03)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-10**6, hi=10**6)
    d = H.random_column(rng, abi.F64, n, 0.1)
    f = Column(abi.F32, rng.integers(-50, 50, n).astype(np.float32), rng.random(n) > 0.1)
    u = H.random_column(rng, abi.U64, n, 0.1)
    chk = Chunk([k, v, d, f, u])
    types = [abi.I64, abi.I64, abi.F64, abi.F32, abi.U64]
    aggs = AGG_SETS[aggset]
    cfg = H.agg_cfg(types, [0], aggs, est_groups=est)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    got = _run(ctx, cfg, chk, aggs, abi.AGGFAST_FORCE, stats=stats)
    assert stats[0].radix_batches >= 1  # the LDS path really ran
    real_cols = [i for i, a in enumerate(aggs) if a[0] in (abi.AGG_SUM, abi.AGG_AVG) and a[2] in (abi.F64, abi.F32)]
    tol = group_tols(chk, 0, aggs, real_cols)  # per group: 2 n_g 2^-53 sum_g|v|
    exact_cols = [i for i in range(len(aggs)) if i not in real_cols]
    key_out = [i for i, a in enumerate(aggs) if a[0] == abi.AGG_FIRSTROW][0]
    _match_by_key(got, want, [key_out], exact_cols, real_cols, tol)
    # chunk-sized pushes accumulate in staging and reach the same batches
    got2 = _run(ctx, cfg, chk, aggs, abi.AGGFAST_FORCE, chunk_rows=1024)
    _match_by_key(got2, want, [key_out], exact_cols, real_cols, tol)


@pytest.mark.parametrize("kt", [abi.F64, abi.F32, abi.U64])
def test_fast_group_key_types_null_group_zero_signs_sentinel(ctx, orc, kt):
    rng = np.random.default_rng(9)
    n = 50000
    if kt in (abi.F64, abi.F32):
        kv = rng.integers(-6, 6, n).astype(np.float64) * 0.5
        kv[rng.random(n) < 0.2] *= -1.0  # -0.0 and +0.0 share a group (util/codec/float.go:22-30)
        kd = kv.astype(np.float32) if kt == abi.F32 else kv
    else:
        kd = rng.integers(0, 9, n).astype(np.uint64)
        kd[rng.random(n) < 0.3] |= np.uint64(1 << 63)
        kd[rng.random(n) < 0.05] = np.uint64(0x8080808080808080)  # the table's EMPTY sentinel is a legal key
    k = Column(kt, kd, rng.random(n) > 0.05)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-1000, hi=1000)
    chk = Chunk([k, v])
    aggs = [(abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_MIN, 1, abi.I64), (abi.AGG_COUNT, 1, abi.I64)]
    cfg = H.agg_cfg([kt, abi.I64], [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    got = _run(ctx, cfg, chk, aggs, abi.AGGFAST_FORCE)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_fast_int64_sums_are_exact_in_128_bits_and_overflow_is_reported(ctx, orc):
    big = (1 << 62) + 12345
    k = np.array([1, 1, 1, 1, 2, 2, 3], dtype=np.int64)
    v = np.array([big, big, -big, -big, 7, -9, big], dtype=np.int64)  # group 1 leaves int64 transiently, ends at 0
    rep = 3000
    chk = Chunk([Column(abi.I64, np.tile(k, rep)), Column(abi.I64, np.tile(v, rep))])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64)]
    cfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs)
    with pytest.raises(_lib.TsqError) as ei:  # group 3: rep * big does not fit BIGINT (func_sum.go:133-137)
        _run(ctx, cfg, chk, aggs, abi.AGGFAST_FORCE)
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT
    chk2 = Chunk([Column(abi.I64, np.tile(k[:6], rep)), Column(abi.I64, np.tile(v[:6], rep))])
    got = _run(ctx, cfg, chk2, aggs, abi.AGGFAST_FORCE)
    assert H.rows_equal_unordered(got, [(1, 0, 4 * rep), (2, -2 * rep, 2 * rep)])


@pytest.mark.parametrize("est", [0, 5000])
def test_fast_skewed_keys(ctx, orc, est):
    # one hot group (LDS atomics on one slot; with est > 2048 one partition region overflows into the list)
    rng = np.random.default_rng(4)
    n = 400_000
    kd = rng.integers(0, 3000, n)
    kd[rng.random(n) < 0.7] = 42
    chk = Chunk([Column(abi.I64, kd), Column(abi.I64, rng.integers(-5, 6, n))])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_MAX, 1, abi.I64)]
    cfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs, est_groups=est)
    want = orc.hash_agg(cfg, chk, 4, 4)
    got = _run(ctx, cfg, chk, aggs, abi.AGGFAST_FORCE)
    assert H.rows_equal_unordered(got, want)


def _device_agg(ctx, n, groups, vtype, fast, batch=50_000_000):
    """C3-shaped aggregate on device-generated rows; returns (groups, sum of counts, sum of sums, stats)."""
    lib = ctx.lib
    k, v = G.DevCol(ctx, abi.I64, batch), G.DevCol(ctx, vtype, batch)
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, vtype), (abi.AGG_COUNT, -1, abi.I64)]
    cfg = H.agg_cfg([abi.I64, vtype], [0], aggs)
    h = C.c_void_p()
    _lib.check(lib.tsq_agg_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        _lib.check(lib.tsq_agg_set_fast(h, fast), h)
        done = 0
        while done < n:
            m = min(batch, n - done)
            ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=3, col=0, m=groups, start=done), m, k.data)
            if vtype == abi.F64:
                ctx.gen_column(G.gen_spec(abi.GEN_RAND_F64, table=3, col=1, start=done), m, v.data)
            else:
                ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=3, col=1, m=1000, start=done), m, v.data)
            kc, vc = k.col(), v.col()
            kc.length = vc.length = m
            _lib.check(lib.tsq_agg_push(h, (abi.Col * 2)(kc, vc), 2, m), h)
            done += m
        _lib.check(lib.tsq_agg_finish(h), h)
        st = abi.Stats()
        _lib.check(lib.tsq_agg_stats(h, C.byref(st)), h)
        out_t = out_types_for(aggs)
        keys, sums, cnts = [], [], []
        cap = 1 << 20  # device pulls: the cursor stays a multiple of 8
        outs = [G.DevCol(ctx, t, cap, with_nulls=True) for t in out_t]
        try:
            while True:
                oc = (abi.Col * len(out_t))(*[o.col() for o in outs])
                nr, eos = C.c_int64(0), C.c_int32(0)
                _lib.check(lib.tsq_agg_pull(h, oc, len(out_t), cap, C.byref(nr), C.byref(eos)), h)
                if nr.value == 0:
                    break
                cols = [o.to_host() for o in outs]
                keys.append(cols[0].data[:nr.value].copy())
                sums.append(cols[1].data[:nr.value].copy())
                cnts.append(cols[2].data[:nr.value].copy())
        finally:
            for o in outs:
                o.free()
        return np.concatenate(keys), np.concatenate(sums), np.concatenate(cnts), st
    finally:
        lib.tsq_agg_destroy(h)
        k.free()
        v.free()


def test_fast_auto_c3_shape_matches_row_path_and_closed_form(ctx):
    n, groups = 60_000_000, 1_000_000  # C3 shape (1e9 rows / 1e6 groups) at a size the row path finishes quickly too
    k1, s1, c1, st1 = _device_agg(ctx, n, groups, abi.I64, abi.AGGFAST_AUTO)
    k0, s0, c0, st0 = _device_agg(ctx, n, groups, abi.I64, abi.AGGFAST_OFF)
    assert st1.radix_batches >= 1 and st0.radix_batches == 0
    o1, o0 = np.argsort(k1), np.argsort(k0)
    assert np.array_equal(k1[o1], k0[o0]) and np.array_equal(s1[o1], s0[o0]) and np.array_equal(c1[o1], c0[o0])
    assert len(k1) == groups and int(c1.sum()) == n
    # closed form of the total: sum over rows of r(i,1) mod 1000
    tot, step = 0, 1 << 24
    for lo in range(0, n, step):
        i = np.arange(lo, min(n, lo + step), dtype=np.uint64)
        tot += int((G.np_gen_r(42, 3, 1, i) % np.uint64(1000)).sum())
    assert int(s1.sum()) == tot


def test_fast_auto_few_groups_double_sum(ctx):
    n, groups = 30_000_000, 1000  # low cardinality: LDS tables straight from the columns
    k1, s1, c1, st1 = _device_agg(ctx, n, groups, abi.F64, abi.AGGFAST_AUTO)
    k0, s0, c0, st0 = _device_agg(ctx, n, groups, abi.F64, abi.AGGFAST_OFF)
    assert st1.radix_batches >= 1 and len(k1) == groups
    o1, o0 = np.argsort(k1), np.argsort(k0)
    assert np.array_equal(k1[o1], k0[o0]) and np.array_equal(c1[o1], c0[o0])
    # values in [0,1): per group |a-b| <= 2 * n_g * 2^-53 * sum|v| (both sides are re-ordered sums)
    ng = c1[o1].astype(np.float64)
    assert np.all(np.abs(s1[o1] - s0[o0]) <= 4 * ng * 2.0 ** -53 * np.maximum(s1[o1], 1.0))
