"""GPU: the packed routes chosen by AUTO (no FORCE anywhere) at the sizes that gate them — 4 Mi-row build sides and probe
batches: several key columns, bit cells, OtherConditions over the materialised batch, the several-column aggregate.  The expected
results are numpy closed forms (the oracle is too slow at these sizes; the same routes are compared with it row by row, FORCED, in
tests/test_join_packed_gpu.py and tests/test_agg_packed_gpu.py)."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H
from .test_agg_gpu import out_types_for

pytestmark = pytest.mark.gpu
NB, NP = (1 << 22) + 4096, (1 << 22) + 12_345  # just above the AUTO thresholds


def _count(ctx, cfg, build, probe):
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 23, count_only=True, stats_out=stats)
    return got, stats[0]


def test_auto_two_key_columns_count(ctx):
    rng = np.random.default_rng(1)
    k = rng.permutation(NB).astype(np.int64)  # unique pairs (k div 3000, k mod 3000)
    build = Chunk([Column(abi.I64, k // 3000), Column(abi.I64, k % 3000)])
    pk = rng.integers(0, 2 * NB, NP)  # half of the probe pairs exist
    pnn = rng.random(NP) > 0.02
    probe = Chunk([Column(abi.I64, pk // 3000, pnn), Column(abi.I64, pk % 3000)])
    cfg = H.join_cfg(probe.types(), build.types(), [0, 1], [0, 1], abi.JOIN_INNER, 1)
    got, st = _count(ctx, cfg, build, probe)
    assert got == int(((pk < NB) & pnn).sum())
    assert st.radix_batches >= 1 and st.packed_key_bits > 0  # (the last, short batch of a push takes the direct route: probe_route is its route)


def test_auto_bit_cells_30_bits(ctx):
    rng = np.random.default_rng(2)
    k = rng.permutation(NB).astype(np.int64) * 200 + 7  # unique, spread over 30 bits (4.2e6 x 200 = 8.4e8): 32 B of bit image per build row
    build = Chunk([Column(abi.I64, k), Column(abi.I64, np.arange(NB))])
    pk = rng.integers(0, NB, NP) * 200 + 7 + (rng.random(NP) < 0.3)  # 30 % moved off the grid
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, np.arange(NP))])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], abi.JOIN_INNER, 1)
    got, st = _count(ctx, cfg, build, probe)
    assert got == int((pk % 200 == 7).sum())
    assert st.radix_batches >= 1 and st.packed_key_bits == 30


def test_auto_other_conditions_over_the_materialised_batch(ctx):
    from oracle import binding as orc_b

    rng = np.random.default_rng(3)
    bk = rng.permutation(NB).astype(np.int64)
    bv = rng.integers(0, 1000, NB)
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, bv)])
    pk = rng.integers(0, NB, NP)
    pv = rng.integers(0, 1000, NP)
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, pv, rng.random(NP) > 0.05)])
    keep = []
    cond = [E.ScalarFunction("lt", E.Column(1, abi.I64), E.Column(3, abi.I64))]  # probe.v < build.v
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], abi.JOIN_INNER, 1, cond, (), keep)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 23, pull_rows=1 << 22, stats_out=stats)
    inv = np.empty(NB, dtype=np.int64)
    inv[bk] = np.arange(NB)
    brow = inv[pk]
    pnn = probe.columns[1].notnull
    passed = pnn & (pv < bv[brow])
    assert got.NumRows() == int(passed.sum())
    assert stats[0].radix_batches >= 1 and stats[0].packed_key_bits > 0
    want = Chunk([Column(abi.I64, pk[passed]), Column(abi.I64, pv[passed]), Column(abi.I64, pk[passed]), Column(abi.I64, bv[brow][passed])])
    assert orc_b.rows_checksum(got) == orc_b.rows_checksum(want)


def test_auto_three_key_aggregate(ctx):
    rng = np.random.default_rng(4)
    n = (1 << 22) + 999
    a, b, c = rng.integers(0, 200, n), rng.integers(-5, 5, n), rng.integers(0, 3, n)
    v = rng.integers(-1000, 1000, n)
    cnn = rng.random(n) > 0.1
    chk = Chunk([Column(abi.I64, a), Column(abi.I64, b), Column(abi.I64, c, cnn), Column(abi.I64, v)])
    types = [abi.I64] * 4
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_FIRSTROW, 1, abi.I64), (abi.AGG_FIRSTROW, 2, abi.I64), (abi.AGG_SUM, 3, abi.I64), (abi.AGG_COUNT, -1, abi.I64)]
    cfg = H.agg_cfg(types, [0, 1, 2], aggs, est_groups=20_000)
    stats = []
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 23, pull_rows=1 << 16, stats_out=stats)
    assert stats[0].packed_key_bits > 0  # AUTO took the several-column packed route
    code = (a * 10 + (b + 5)) * 4 + np.where(cnn, c, 3)
    uk, inv = np.unique(code, return_inverse=True)
    sums = np.bincount(inv, weights=v.astype(np.float64)).astype(np.int64)
    cnts = np.bincount(inv)
    rows = got.rows()
    assert len(rows) == len(uk)
    seen = {}
    for r in rows:
        seen[(r[0] * 10 + (r[1] + 5)) * 4 + (3 if r[2] is None else r[2])] = (r[3], r[4])
    assert len(seen) == len(uk)
    for k_, s_, c_ in zip(uk.tolist(), sums.tolist(), cnts.tolist()):
        assert seen[k_] == (s_, c_), (k_, seen[k_], s_, c_)
