"""GPU parity of HashAggExec's PARTITIONED GROUPS mode (round 6, csrc/tsq_aggfast.h K7p): about as many groups as rows — the group table
is a set of LDS-sized sub-tables kept in HBM between the batches; a batch is radix partitioned by the table word, one workgroup per
partition loads a sub-table, lets the partition's rows find or insert their groups, stores it back.  The reference keys its partial
results by the encoded group key and merges them in its final workers (executor/aggregate.go:332-356, 424-457): every aggregate of every
group is compared with the oracle — integer aggregates bit-exact, SUM / AVG(double) within 2 n_g 2^-53 sum_g|v| per group — over
several batches, NULL keys and NULL argument cells (exception rows: the merging way out), sub-tables that fill up (spilled words: a key
has ONE home), the sentinel key, real keys, the composite-key child of a several-column GROUP BY, and the activation by the planner's
estimate and by the key sample.  The mode is asserted through tsq_stats.build_partitioned == 4 (child: dense_flushes == -4)."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H
from .test_agg_gpu import _match_by_key, group_tols, out_types_for

pytestmark = pytest.mark.gpu
FORCE = abi.AGGFAST_FORCE
# (a group's LDS words: COUNT 1, SUM / AVG of integers 2 / 3, of reals 1 / 2, MAX / MIN 1 — at most 5 per plan, tsq_aggfast.h)
AGGS = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_MAX, 2, abi.F64), (abi.AGG_MIN, 1, abi.I64)]
AGGS2 = [(abi.AGG_AVG, 1, abi.I64), (abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_AVG, 2, abi.F64)]


def _check(ctx, orc, chk, aggs, types, est=0, knob=1, chunk_rows=1 << 22, fast=FORCE, want_pg=True, batch_rows=None, keys=(0,)):
    cfg = H.agg_cfg(types, list(keys), aggs, est_groups=est)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    kn = {"AGG_PG": knob}
    if batch_rows:
        kn["AGG_BATCH_ROWS"] = batch_rows
    with ctx.knobs(**kn):
        got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=chunk_rows, fast=fast, stats_out=stats, pull_rows=1 << 16)
    st = stats[0]
    if want_pg is not None:
        assert (st.build_partitioned == 4 or st.dense_flushes == -4) == want_pg, (st.build_partitioned, st.dense_flushes, st.radix_batches)
    real_cols = [i for i, a in enumerate(aggs) if a[0] in (abi.AGG_SUM, abi.AGG_AVG) and a[2] in (abi.F64, abi.F32)]
    if len(keys) == 1 and chk.columns[keys[0]].tp == abi.I64:
        tol = group_tols(chk, keys[0], aggs, real_cols)
        key_out = [i for i, a in enumerate(aggs) if a[0] == abi.AGG_FIRSTROW and a[1] == keys[0]][0]
        _match_by_key(got, want, [key_out], [i for i in range(len(aggs)) if i not in real_cols], real_cols, tol)
    else:
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    return st


def _table(rng, n, ndv, null_key=0.0, null_arg=0.0, spread=1 << 62):
    pool = rng.integers(-spread, spread, ndv)
    k = Column(abi.I64, pool[rng.integers(0, ndv, n)], (rng.random(n) > null_key) if null_key else None)
    v = H.random_column(rng, abi.I64, n, null_arg, lo=-10**9, hi=10**9)
    d = H.random_column(rng, abi.F64, n, null_arg)
    return Chunk([k, v, d]), [abi.I64, abi.I64, abi.F64]


@pytest.mark.parametrize("n,ndv,knob", [(1, 1, 2), (5000, 4000, 2), (200_001, 150_000, 8), (300_000, 300_000, 13), (120_000, 90_000, 6)])
def test_pg_vs_oracle_no_nulls_straight_from_the_slots(ctx, orc, n, ndv, knob):
    # knob v: 2^(v - 2) sub-tables of 4096 (W <= 3) / 2048 slots: one sub-table .. 2^11 partitions; nothing spills at these sizes except
    # in the (120 000, 90 000, 16 sub-tables = 32 Ki slots) case, where two thirds of the keys find their sub-table full
    rng = np.random.default_rng(n + knob)
    chk, types = _table(rng, n, ndv)
    st = _check(ctx, orc, chk, AGGS, types, knob=knob)
    assert st.radix_batches >= 1
    _check(ctx, orc, chk, AGGS2, types, knob=knob)


def test_pg_several_batches_and_chunked_pushes(ctx, orc):
    rng = np.random.default_rng(3)
    chk, types = _table(rng, 400_000, 120_000)
    _check(ctx, orc, chk, AGGS, types, knob=9, batch_rows=65_536, chunk_rows=65_536)  # seven device batches: the keys of a batch meet their groups of the earlier ones
    _check(ctx, orc, chk, AGGS, types, knob=9, batch_rows=65_536, chunk_rows=1024)    # host chunks of tidb_max_chunk_size rows reach the same batches


def test_pg_null_keys_null_arguments_and_the_sentinel_key(ctx, orc):
    # NULL group keys and NULL argument cells are exception rows (the row upsert keeps their exact NULL protocol), the table's EMPTY
    # sentinel is a legal key: the table in HBM then holds groups too and the sub-tables' groups are merged into it at the end
    rng = np.random.default_rng(5)
    chk, types = _table(rng, 150_000, 60_000, null_key=0.03, null_arg=0.1)
    chk.columns[0].data[rng.integers(0, 150_000, 200)] = np.uint64(0x8080808080808080).astype(np.int64)
    _check(ctx, orc, chk, AGGS, types, knob=8)
    _check(ctx, orc, chk, AGGS, types, knob=8, batch_rows=40_000, chunk_rows=40_000)


@pytest.mark.parametrize("kt", [abi.F64, abi.U64])
def test_pg_real_and_unsigned_keys(ctx, orc, kt):
    rng = np.random.default_rng(7)
    n = 100_000
    if kt == abi.F64:
        kd = rng.integers(-40_000, 40_000, n).astype(np.float64) * 0.25
        kd[rng.random(n) < 0.1] *= -1.0
        kd[kd == 0] = 0.0  # (-0.0 and +0.0 share a group, util/codec/float.go:22-30, and FIRST_ROW of that group is either: test_agg_fast_gpu covers the zero signs)
    else:
        kd = rng.integers(0, 1 << 63, 70_000).astype(np.uint64)[rng.integers(0, 70_000, n)] | np.uint64(1 << 63)
    chk = Chunk([Column(kt, kd), H.random_column(rng, abi.I64, n, 0.0, lo=-1000, hi=1000)])
    aggs = [(abi.AGG_FIRSTROW, 0, kt), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_MIN, 1, abi.I64)]
    _check(ctx, orc, chk, aggs, [kt, abi.I64], knob=8)


def test_pg_composite_key_child(ctx, orc):
    # GROUP BY three integer columns whose fields need 42 bits (the shape of Q3's aggregate): the composite-key child takes the mode
    rng = np.random.default_rng(11)
    n = 200_000
    ok = rng.integers(0, 1 << 27, 90_000)
    pick = rng.integers(0, 90_000, n)
    chk = Chunk([Column(abi.I64, ok[pick]), Column(abi.I64, (ok[pick] * 7) % 2400), Column(abi.I64, ok[pick] % 3), H.random_column(rng, abi.F64, n, 0.0)])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_FIRSTROW, 1, abi.I64), (abi.AGG_FIRSTROW, 2, abi.I64), (abi.AGG_SUM, 3, abi.F64)]
    cfg = H.agg_cfg([abi.I64, abi.I64, abi.I64, abi.F64], [0, 1, 2], aggs, est_groups=0)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    with ctx.knobs(AGG_PG=9):
        got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 22, fast=FORCE, stats_out=stats, pull_rows=1 << 16)
    assert stats[0].dense_flushes == -4, (stats[0].build_partitioned, stats[0].dense_flushes)
    g = {r[:3]: r[3] for r in got.rows()}
    w = {r[:3]: r[3] for r in want.rows()}
    assert len(g) == got.NumRows() and set(g) == set(w)
    for key, val in w.items():
        assert H.approx_equal(g[key], val, 1e-9 * max(1.0, abs(val))), (key, g[key], val)


def test_pg_by_the_planners_estimate_and_by_the_key_sample(ctx, orc):
    # AUTO: est_groups beyond what the LDS tables of H mode hold per batch -> the sub-tables; no estimate -> 8192 sampled keys say
    # "about as many groups as rows" (k_pg_sample); few distinct keys -> the other modes, as before
    rng = np.random.default_rng(13)
    n = 1_200_000
    chk, types = _table(rng, n, n)
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    _check(ctx, orc, chk, aggs, types, est=4_000_000, fast=abi.AGGFAST_AUTO)
    # batches much smaller than the state wait (every pass rewrites all the sub-tables) and are aggregated together at the end
    _check(ctx, orc, chk, aggs, types, est=4_000_000, fast=abi.AGGFAST_AUTO, chunk_rows=1 << 20, batch_rows=1 << 20)
    m = 4_500_000  # every key once: the sample finds no duplicate and takes "four times the rows" for the number of keys (H mode holds 3.2e6 groups a batch)
    chk2 = Chunk([Column(abi.I64, rng.permutation(m).astype(np.int64) * 1_000_003 - 7), H.random_column(rng, abi.I64, m, 0.0, lo=-10**9, hi=10**9), H.random_column(rng, abi.F64, m, 0.0)])
    _check(ctx, orc, chk2, aggs, types, est=0, fast=abi.AGGFAST_AUTO, chunk_rows=1 << 23, batch_rows=1 << 23)
    chk3, _ = _table(rng, n, 5000)
    _check(ctx, orc, chk3, aggs, types, est=0, fast=abi.AGGFAST_AUTO, want_pg=False)
    _check(ctx, orc, chk, aggs, types, est=4_000_000, fast=abi.AGGFAST_AUTO, knob=0, want_pg=False)  # the knob switches the mode off
