"""GPU parity of the PACKED-KEY pre-aggregation of HashAggExec (csrc/tsq_daagg.h): an integer group key whose values span
few bits travels as a 2-byte entry next to its argument cells and is aggregated in a direct-addressed LDS table.  Forced on
small inputs (every aggregate function, NULL keys and arguments, negative and unsigned keys, a hot key that overflows its
partition's region, later batches with keys outside the range the first batch showed) against the oracle; integer results
are bit-exact, SUM/AVG(double) within 2 n_g 2^-53 sum_g|v| per group."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H
from .test_agg_gpu import _match_by_key, group_tols, out_types_for

pytestmark = pytest.mark.gpu

AGG_SETS = {
    "c3": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64)],
    "c3_double": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 2, abi.F64), (abi.AGG_COUNT, -1, abi.I64)],
    "ints": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, 1, abi.I64), (abi.AGG_AVG, 1, abi.I64), (abi.AGG_MAX, 1, abi.I64)],
    "minmax2": [(abi.AGG_MIN, 1, abi.I64), (abi.AGG_MAX, 4, abi.U64), (abi.AGG_MIN, 4, abi.U64), (abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64)],
    "reals": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 2, abi.F64), (abi.AGG_AVG, 2, abi.F64), (abi.AGG_MAX, 2, abi.F64)],
    "f32": [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 3, abi.F32), (abi.AGG_MIN, 3, abi.F32), (abi.AGG_COUNT, 3, abi.F32)],
    "count_only": [(abi.AGG_COUNT, -1, abi.I64), (abi.AGG_FIRSTROW, 0, abi.I64)],
}


def _chunk(rng, n, lo, hi, kt=abi.I64, key_nulls=0.03):
    kv = rng.integers(lo, hi, n)
    k = Column(kt, kv.astype(np.uint64) if kt == abi.U64 else kv, rng.random(n) > key_nulls)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-10**6, hi=10**6)
    d = H.random_column(rng, abi.F64, n, 0.1)
    f = Column(abi.F32, rng.integers(-50, 50, n).astype(np.float32), rng.random(n) > 0.1)
    u = H.random_column(rng, abi.U64, n, 0.1)
    return Chunk([k, v, d, f, u]), [kt, abi.I64, abi.F64, abi.F32, abi.U64]


def _check(ctx, orc, chk, types, aggs, est, chunk_rows=1 << 22, want_packed=True):
    cfg = H.agg_cfg(types, [0], aggs, est_groups=est)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=chunk_rows, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].radix_batches >= 1
    if want_packed is not None:
        assert (stats[0].packed_key_bits > 0) == want_packed, stats[0].packed_key_bits
    real_cols = [i for i, a in enumerate(aggs) if a[0] in (abi.AGG_SUM, abi.AGG_AVG) and a[2] in (abi.F64, abi.F32)]
    exact_cols = [i for i in range(len(aggs)) if i not in real_cols]
    key_out = [i for i, a in enumerate(aggs) if a[0] == abi.AGG_FIRSTROW][0]
    _match_by_key(got, want, [key_out], exact_cols, real_cols, group_tols(chk, 0, aggs, real_cols))
    return stats[0]


@pytest.mark.parametrize("aggset", sorted(AGG_SETS))
@pytest.mark.parametrize("n,lo,hi", [(5000, -3000, 3000), (70_001, 10**12, 10**12 + 50_000), (300_001, -(1 << 21), 1 << 21)])
def test_packed_agg_random_vs_oracle(ctx, orc, aggset, n, lo, hi):
    rng = np.random.default_rng(n + len(aggset))
    chk, types = _chunk(rng, n, lo, hi)
    st = _check(ctx, orc, chk, types, AGG_SETS[aggset], est=max(4096, (hi - lo) // 2))
    assert st.packed_key_bits >= 14


def test_packed_agg_unsigned_keys_above_2_63(ctx, orc):
    rng = np.random.default_rng(3)
    n = 60_000
    kv = (np.uint64(1 << 63) + rng.integers(0, 20_000, n).astype(np.uint64))
    chk = Chunk([Column(abi.U64, kv, rng.random(n) > 0.02), H.random_column(rng, abi.I64, n, 0.1, lo=-99, hi=99)])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.U64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64)]
    cfg = H.agg_cfg([abi.U64, abi.I64], [0], aggs, est_groups=20_000)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 22, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].packed_key_bits > 0 and H.rows_equal_unordered(got, want)


def test_packed_agg_hot_key_and_later_batches_outside_the_range(ctx, orc, monkeypatch):
    ctx.set_knob(abi.KNOB_AGG_BATCH_ROWS, 1 << 17)
    # batch 1 (device batches of 2^17 rows): keys in [0, 30000) with one key carrying 40 % of
    # the rows (its partition's region overflows: those rows take the row-at-a-time upsert); batch 2 brings keys far outside the range
    # the first batch showed (exception rows); batch 3 is mostly outside (the operator leaves the packed route)
    rng = np.random.default_rng(8)
    n1 = 1 << 17
    k1 = rng.integers(0, 30_000, n1)
    k1[rng.random(n1) < 0.4] = 777
    k2 = np.concatenate([rng.integers(0, 30_000, n1 - 5000), rng.integers(10**9, 10**9 + 50, 5000)])
    k3 = rng.integers(-10**15, -10**15 + 40_000, n1)
    keys = np.concatenate([k1, k2, k3])
    n = len(keys)
    chk = Chunk([Column(abi.I64, keys, rng.random(n) > 0.01), H.random_column(rng, abi.I64, n, 0.05, lo=-1000, hi=1000), H.random_column(rng, abi.F64, n, 0.05)])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 2, abi.F64), (abi.AGG_MIN, 1, abi.I64)]
    cfg = H.agg_cfg([abi.I64, abi.I64, abi.F64], [0], aggs, est_groups=30_000)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=n1, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].packed_key_bits > 0 and stats[0].radix_batches >= 3
    _match_by_key(got, want, [0], [1, 2, 4], [3], group_tols(chk, 0, aggs, [3]))


def test_packed_agg_wide_keys_keep_the_64_bit_route(ctx, orc):
    rng = np.random.default_rng(5)
    chk, types = _chunk(rng, 80_000, -(1 << 40), 1 << 40)
    chk.columns[0] = Column(abi.I64, rng.integers(0, 30_000, 80_000) * (1 << 30))  # 30 K groups spread over 2^45
    _check(ctx, orc, chk, types, AGG_SETS["c3"], est=30_000, want_packed=False)


# ---------------------------------------------------------------- several integer key columns as the fields of one packed word
def _group_tols_multi(chk, key_idxs, aggs, real_cols):
    """group_tols() for a several-column key: {tuple of canonical key cells: {output column: 2 n_g 2^-53 sum_g|v|}}"""
    rows = chk.rows()
    out = {}
    for c in real_cols:
        func, arg = aggs[c][0], aggs[c][1]
        acc = {}
        for r in rows:
            if r[arg] is None:
                continue
            k = tuple(H.canon(r[i]) for i in key_idxs)
            n, s = acc.get(k, (0, 0.0))
            acc[k] = (n + 1, s + abs(float(r[arg])))
        for k, (n, sa) in acc.items():
            t = 2.0 * n * 2.0 ** -53 * sa
            if func == abi.AGG_AVG:
                t = t / n + (sa / n) * 2.0 ** -52
            out.setdefault(k, {})[c] = t
    return out


def _mk_chunk(rng, n, specs):
    """specs: per key column (type, lo, hi, null fraction or None for a column without a bitmap); then v I64, d F64, f F32, u U64"""
    cols, types = [], []
    for tp, lo, hi, nf in specs:
        kv = rng.integers(lo, hi, n)
        cols.append(Column(tp, kv.astype(np.uint64) if tp == abi.U64 else kv, None if nf is None else rng.random(n) >= nf))
        types.append(tp)
    nk = len(specs)
    cols += [H.random_column(rng, abi.I64, n, 0.1, lo=-10**6, hi=10**6), H.random_column(rng, abi.F64, n, 0.1),
             Column(abi.F32, rng.integers(-50, 50, n).astype(np.float32), rng.random(n) > 0.1), H.random_column(rng, abi.U64, n, 0.1)]
    types += [abi.I64, abi.F64, abi.F32, abi.U64]
    return Chunk(cols), types, nk


def _mk_aggs(nk, types, which):
    v, d, f, u = nk, nk + 1, nk + 2, nk + 3
    first = [(abi.AGG_FIRSTROW, k, types[k]) for k in range(nk)]
    sets = {
        "sum_count": [(abi.AGG_SUM, v, abi.I64), (abi.AGG_COUNT, -1, abi.I64)],
        "ints": [(abi.AGG_COUNT, v, abi.I64), (abi.AGG_AVG, v, abi.I64), (abi.AGG_MAX, v, abi.I64)],
        "minmax2": [(abi.AGG_MIN, v, abi.I64), (abi.AGG_MAX, u, abi.U64), (abi.AGG_MIN, u, abi.U64), (abi.AGG_COUNT, -1, abi.I64)],
        "reals": [(abi.AGG_SUM, d, abi.F64), (abi.AGG_AVG, d, abi.F64), (abi.AGG_MAX, d, abi.F64)],
        "f32": [(abi.AGG_SUM, f, abi.F32), (abi.AGG_MIN, f, abi.F32), (abi.AGG_COUNT, f, abi.F32)],
    }
    return first + sets[which]


def _mk_check(ctx, orc, chk, types, nk, aggs, chunk_rows=1 << 22, want_packed=True, min_batches=1):
    cfg = H.agg_cfg(types, list(range(nk)), aggs, est_groups=5000)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=chunk_rows, fast=abi.AGGFAST_FORCE, stats_out=stats)
    if want_packed is not None:
        assert (stats[0].packed_key_bits > 0) == want_packed, stats[0].packed_key_bits
    if want_packed:
        assert stats[0].radix_batches >= min_batches
    real_cols = [i for i, a in enumerate(aggs) if a[0] in (abi.AGG_SUM, abi.AGG_AVG) and a[2] in (abi.F64, abi.F32)]
    exact_cols = [i for i in range(nk, len(aggs)) if i not in real_cols]
    _match_by_key(got, want, list(range(nk)), exact_cols, real_cols, _group_tols_multi(chk, list(range(nk)), aggs, real_cols))
    return stats[0]


MK_SPECS = {
    # 7 + 5 + 2 bits: partitioned route (more than one LDS table of cells)
    "partitioned": [(abi.I64, -50, 50, 0.03), (abi.I64, 10**12, 10**12 + 20, None), (abi.U64, 0, 3, 0.05)],
    # 5 + 3 bits: the word fits one LDS table, no partition pass
    "one_table": [(abi.I64, 0, 30, 0.04), (abi.I64, -3, 3, None)],
    # four columns, one of them constant (a field of width 0)
    "four": [(abi.I64, -5, 5, 0.1), (abi.U64, (1 << 63) + 5, (1 << 63) + 9, None), (abi.I64, 7, 8, None), (abi.I64, 0, 40, 0.02)],
}


@pytest.mark.parametrize("which", ["sum_count", "ints", "minmax2", "reals", "f32"])
@pytest.mark.parametrize("spec", sorted(MK_SPECS))
def test_packed_agg_several_key_columns_vs_oracle(ctx, orc, spec, which):
    rng = np.random.default_rng(len(spec) * 7 + len(which))
    specs = MK_SPECS[spec]
    if spec == "four":  # rng.integers cannot draw above 2^63: the unsigned column is built by hand
        specs = [specs[0], (abi.I64, 5, 9, None), specs[2], specs[3]]
    chk, types, nk = _mk_chunk(rng, 120_001, specs)
    if spec == "four":
        chk.columns[1] = Column(abi.U64, chk.columns[1].data.astype(np.uint64) + np.uint64(1 << 63))
        types[1] = abi.U64
    st = _mk_check(ctx, orc, chk, types, nk, _mk_aggs(nk, types, which))
    assert st.packed_key_bits >= 14


def test_packed_agg_several_key_columns_later_batches_outside_the_fields(ctx, orc, monkeypatch):
    ctx.set_knob(abi.KNOB_AGG_BATCH_ROWS, 1 << 16)
    # batch 1 shows (a in [0, 100), b in [0, 8)) without NULLs in b; batch 2 brings b = NULL (no code: exception rows), a few a
    # outside its field and a = NULL; batch 3 is mostly outside (the operator leaves the packed route for the row upsert)
    rng = np.random.default_rng(12)
    n1 = 1 << 16
    a1, b1 = rng.integers(0, 100, n1), rng.integers(0, 8, n1)
    a2 = np.concatenate([rng.integers(0, 100, n1 - 3000), rng.integers(5000, 5050, 3000)])
    b2 = rng.integers(0, 8, n1)
    a3, b3 = rng.integers(-10**9, -10**9 + 300, n1), rng.integers(0, 8, n1)
    a = np.concatenate([a1, a2, a3])
    b = np.concatenate([b1, b2, b3])
    n = len(a)
    ann = rng.random(n) > 0.02
    bnn = np.ones(n, bool)
    bnn[n1:] = rng.random(n - n1) > 0.05
    chk = Chunk([Column(abi.I64, a, ann), Column(abi.I64, b, bnn), H.random_column(rng, abi.I64, n, 0.05, lo=-1000, hi=1000), H.random_column(rng, abi.F64, n, 0.05)])
    types = [abi.I64, abi.I64, abi.I64, abi.F64]
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_FIRSTROW, 1, abi.I64), (abi.AGG_SUM, 2, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 3, abi.F64),
            (abi.AGG_MIN, 2, abi.I64)]
    cfg = H.agg_cfg(types, [0, 1], aggs, est_groups=1000)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=n1, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].packed_key_bits > 0 and stats[0].radix_batches >= 2
    _match_by_key(got, want, [0, 1], [2, 3, 5], [4], _group_tols_multi(chk, [0, 1], aggs, [4]))


def test_packed_agg_several_key_columns_with_shared_tags(ctx, orc, monkeypatch):
    # TSQ_AGG_TAG_BITS (tests): 2^9 tags for ~3000 groups — distinct keys share a tag, the merge of the packed partial groups
    # compares cells and walks on (k_agg_merge_multi phase 1), as the row upsert does
    ctx.set_knob(abi.KNOB_AGG_TAG_BITS, 9)
    rng = np.random.default_rng(13)
    chk, types, nk = _mk_chunk(rng, 90_000, [(abi.I64, 0, 400, 0.02), (abi.I64, -4, 4, 0.02)])
    st = _mk_check(ctx, orc, chk, types, nk, _mk_aggs(nk, types, "sum_count"))
    assert st.build_handed_back_rows > 0


def test_packed_agg_several_key_columns_too_wide_keeps_the_row_upsert(ctx, orc):
    rng = np.random.default_rng(14)
    chk, types, nk = _mk_chunk(rng, 50_000, [(abi.I64, 0, 1 << 20, None), (abi.I64, 0, 1 << 10, None)])
    chk.columns[0] = Column(abi.I64, rng.integers(0, 50, 50_000) * (1 << 14))
    _mk_check(ctx, orc, chk, types, nk, _mk_aggs(nk, types, "sum_count"), want_packed=False)


# ---------------------------------------------------------------- several integer key columns as ONE 64-bit composite key (round 4)
# Key sets wider than the packed route's 23 bits (Q3's l_orderkey, o_orderdate, o_shippriority: 42 bits) used to take the several-column
# row upsert.  Now the first batch gives every key column a field, the cells of a row become ONE 64-bit number d, and a child aggregate
# with the single key d does the work (tsq_agg.hip: wide_setup / wide_batch); FIRST_ROW(key column) is decoded from d.  Rows with a
# cell outside its field stay in the parent's own several-column table (exceptions).  stats.build_partitioned == 2 says the child ran.
WIDE_SPECS = {
    # Q3-like: a wide surrogate key, a day number, a tiny code; NULLs in two of them
    "q3_like": [(abi.I64, 0, 1 << 28, 0.02), (abi.I64, 8000, 10500, None), (abi.I64, 0, 3, 0.05)],
    # negative and unsigned cells; four columns
    "four_mixed": [(abi.I64, -(1 << 20), 1 << 20, 0.03), (abi.I64, -5, 5, None), (abi.I64, 10**15, 10**15 + 1000, 0.01), (abi.I64, 0, 2, None)],
    # two very wide columns: 30 + 30 bits
    "two_wide": [(abi.I64, -(1 << 29), 1 << 29, None), (abi.I64, 0, 1 << 30, 0.04)],
}


@pytest.mark.parametrize("which", ["sum_count", "ints", "minmax2", "reals", "f32"])
@pytest.mark.parametrize("spec", sorted(WIDE_SPECS))
def test_wide_composite_group_key_vs_oracle(ctx, orc, spec, which):
    rng = np.random.default_rng(len(spec) * 11 + len(which))
    chk, types, nk = _mk_chunk(rng, 150_001, WIDE_SPECS[spec])
    # few distinct values per column so that groups repeat (the ranges stay wide: the values are spread)
    for k in range(nk):
        c = chk.columns[k]
        lo, hi = WIDE_SPECS[spec][k][1], WIDE_SPECS[spec][k][2]
        vals = rng.integers(lo, hi, 40)
        vals[0], vals[1] = lo, hi - 1
        chk.columns[k] = Column(c.tp, vals[rng.integers(0, 40, len(c))], c.notnull)
    st = _mk_check(ctx, orc, chk, types, nk, _mk_aggs(nk, types, which), want_packed=None)
    assert st.build_partitioned == 2 and st.build_handed_back_rows == 0


def test_wide_composite_group_key_later_batches_outside_the_fields_and_other_first_rows(ctx, orc):
    ctx.set_knob(abi.KNOB_AGG_BATCH_ROWS, 1 << 16)
    # batch 1 shows a in [0, 2^26), b in [100, 200); batch 2 brings keys far outside both fields (exception rows: this operator's own
    # several-column table) next to keys inside them; FIRST_ROW of a column that is NOT a key travels through the child like any aggregate
    rng = np.random.default_rng(77)
    n1, n2 = 1 << 16, 90_000
    a1, b1 = rng.integers(0, 1 << 26, 300)[rng.integers(0, 300, n1)], rng.integers(100, 200, n1)
    a2 = np.where(rng.random(n2) < 0.3, rng.integers(1 << 40, (1 << 40) + 50, n2), a1[rng.integers(0, n1, n2)])
    b2 = np.where(rng.random(n2) < 0.2, rng.integers(-10**9, -10**9 + 5, n2), rng.integers(100, 200, n2))
    a, b = np.concatenate([a1, a2]), np.concatenate([b1, b2])
    n = n1 + n2
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-1000, hi=1000)
    chk = Chunk([Column(abi.I64, a, rng.random(n) > 0.02), Column(abi.I64, b), v, Column(abi.I64, a * 3 + b)])  # column 3 depends on the key
    types = [abi.I64] * 4
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_FIRSTROW, 1, abi.I64), (abi.AGG_FIRSTROW, 3, abi.I64), (abi.AGG_SUM, 2, abi.I64), (abi.AGG_COUNT, -1, abi.I64),
            (abi.AGG_AVG, 2, abi.I64)]
    cfg = H.agg_cfg(types, [0, 1], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 20, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].build_partitioned == 2 and stats[0].build_handed_back_rows > 10_000
    # rows with a NULL first key have a NULL a * 3 + b as well?  No: column 3 has no bitmap; FIRST_ROW(col 3) of the NULL-a groups is any
    # row's value — those groups hold several a values.  Compare everything but that column for groups with a NULL key cell.
    _match_by_key(got, want, [0, 1], [3, 4, 5], [], {})
    gd = {(r[0], r[1]): r[2] for r in got.rows()}
    wd = {(r[0], r[1]): r[2] for r in want.rows()}
    assert all(gd[k] == wd[k] for k in wd if k[0] is not None)


@pytest.mark.parametrize("modes", [(abi.MODE_PARTIAL1, abi.MODE_FINAL)])
def test_wide_composite_group_key_partial_then_final(ctx, orc, modes):
    # the distributed plans run HashAgg as partial -> shuffle -> final (parallel.py): both stages take the composite-key route
    rng = np.random.default_rng(5)
    n = 120_000
    k0 = rng.integers(0, 1 << 27, 500)[rng.integers(0, 500, n)]
    k1, k2 = rng.integers(9000, 9400, n), rng.integers(0, 3, n)
    v = rng.random(n)
    chk = Chunk([Column(abi.I64, k0), Column(abi.I64, k1, rng.random(n) > 0.02), Column(abi.I64, k2), Column(abi.F64, v, rng.random(n) > 0.05)])
    t = [abi.I64, abi.I64, abi.I64, abi.F64]
    full = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_FIRSTROW, 1, abi.I64), (abi.AGG_FIRSTROW, 2, abi.I64), (abi.AGG_SUM, 3, abi.F64), (abi.AGG_COUNT, 3, abi.F64)]
    want = orc.hash_agg(H.agg_cfg(t, [0, 1, 2], full), chk, 4, 4)
    paggs = [(f, c, tp, modes[0]) for f, c, tp in full]
    stats = []
    part = G.run_agg(ctx, H.agg_cfg(t, [0, 1, 2], paggs), chk, [abi.I64, abi.I64, abi.I64, abi.F64, abi.I64], chunk_rows=1 << 20, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].build_partitioned == 2
    pt = [abi.I64, abi.I64, abi.I64, abi.F64, abi.I64]
    faggs = [(abi.AGG_FIRSTROW, 0, abi.I64, modes[1]), (abi.AGG_FIRSTROW, 1, abi.I64, modes[1]), (abi.AGG_FIRSTROW, 2, abi.I64, modes[1]),
             (abi.AGG_SUM, 3, abi.F64, modes[1]), (abi.AGG_COUNT, 4, abi.I64, modes[1])]
    stats = []
    got = G.run_agg(ctx, H.agg_cfg(pt, [0, 1, 2], faggs), part, [abi.I64, abi.I64, abi.I64, abi.F64, abi.I64], chunk_rows=1 << 20, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].build_partitioned == 2
    _match_by_key(got, want, [0, 1, 2], [4], [3], _group_tols_multi(chk, [0, 1, 2], full, [3]))


# ---------------------------------------------------------------- round 4: dense partial state + narrow argument cells
def _dense_case(rng, n, key_hi, vmax, neg_frac=0.0):
    k = Column(abi.I64, rng.integers(0, key_hi, n), rng.random(n) > 0.01)
    vv = rng.integers(0, vmax, n)
    if neg_frac:
        vv[rng.random(n) < neg_frac] = -5
    v = Column(abi.I64, vv, rng.random(n) > 0.03)
    return Chunk([k, v]), [abi.I64, abi.I64]


@pytest.mark.parametrize("dense,narrow,key_hi,vmax,bits", [
    (1, 1, 30_000, 60_000, 16), (1, 1, 30_000, 3_000_000_000, 32), (1, 1, 3_000_000, 1000, 16), (1, 1, 3_000_000, 1 << 40, 64),
    (0, 1, 30_000, 60_000, 16), (0, 1, 3_000_000, 1 << 40, 64), (1, 0, 30_000, 3_000_000_000, 32), (1, 0, 3_000_000, 1000, 16),
    (0, 0, 30_000, 60_000, 16), (0, 0, 3_000_000, 1 << 40, 64)])  # (round 5: 10 of the 16 combinations — every (dense, narrow) pair on a small and a large key range)
def test_packed_agg_dense_state_and_narrow_cells(ctx, orc, dense, narrow, key_hi, vmax, bits):
    """30 000 keys: 32 partitions, each split over 8 workgroups (device atomics into the dense state); 3e6 keys: one workgroup per
    partition (plain read-modify-write).  Several device batches, so that the state accumulates across launches."""
    rng = np.random.default_rng(key_hi % 97 + bits)
    n = 700_001
    chk, types = _dense_case(rng, n, key_hi, vmax)
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_MAX, 1, abi.I64), (abi.AGG_MIN, 1, abi.I64)]
    with ctx.knobs(AGG_DENSE=dense, AGG_NARROW_CELLS=narrow, AGG_BATCH_ROWS=1 << 18):
        cfg = H.agg_cfg(types, [0], aggs, est_groups=key_hi)
        want = orc.hash_agg(cfg, chk, 4, 4)
        stats = []
        got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 18, fast=abi.AGGFAST_FORCE, stats_out=stats)
    st = stats[0]
    assert st.packed_key_bits > 0 and st.radix_batches >= 3
    assert st.dense_flushes == (1 if dense else 0)
    assert st.table_slice_bits == (bits if narrow else 64)  # (one travelling argument column: the four aggregates read the same one)
    assert H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("hot", [1, 0])
@pytest.mark.parametrize("vmax", [60_000, 3_000_000_000])
def test_packed_agg_hot_keys_are_absorbed_in_the_partition_kernel(ctx, orc, hot, vmax):
    """round 5: a skewed batch (six keys hold 45 % of the rows, one of them 20 %) — the sampled hot keys are aggregated inside the partition
    kernel's LDS and added to the dense state, the other rows travel as before; the same groups either way (knob DAAGG_HOT = 0: the hot
    rows overflow their regions and take the overflow store).  NULL keys / NULL arguments stay exception rows."""
    rng = np.random.default_rng(vmax % 1000 + hot)
    n = 900_001
    keys = rng.integers(0, 200_000, n)
    r = rng.random(n)
    for i, (lo, hi, k) in enumerate([(0.0, 0.20, 7), (0.20, 0.28, 199_999), (0.28, 0.34, 12_345), (0.34, 0.39, 0), (0.39, 0.43, 65_536), (0.43, 0.45, 100_000)]):
        keys[(r >= lo) & (r < hi)] = k
    chk = Chunk([Column(abi.I64, keys, rng.random(n) > 0.01), Column(abi.I64, rng.integers(0, vmax, n), rng.random(n) > 0.02)])
    aggs = AGG_SETS["c3"]
    with ctx.knobs(AGG_BATCH_ROWS=300_000, DAAGG_HOT=hot):
        cfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs, est_groups=200_000)
        want = orc.hash_agg(cfg, chk, 4, 4)
        stats = []
        got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=100_000, fast=abi.AGGFAST_FORCE, stats_out=stats)
    st = stats[0]
    assert st.packed_key_bits > 0 and st.radix_batches >= 3 and st.dense_flushes == 1
    assert got.NumRows() == want.NumRows() and H.multiset(got) == H.multiset(want)


@pytest.mark.parametrize("vmax,bits", [(60_000, 16), (3_000_000_000, 32), (1 << 40, 64)])
def test_packed_agg_narrow_cells_width_and_values_that_do_not_fit(ctx, orc, vmax, bits):
    """SUM + COUNT(*) (C3's plan: one travelling argument column).  Batch 1 shows values below vmax; batch 2 brings a few larger and
    negative ones (exception rows, exact all the same); batch 3 is mostly large: the operator goes back to full cells."""
    rng = np.random.default_rng(bits)
    nb = 1 << 17
    k = rng.integers(0, 200_000, 3 * nb)
    v = rng.integers(0, vmax, 3 * nb)
    big = np.int64(1 << 50)
    sel2 = nb + rng.choice(nb, 300, replace=False)
    v[sel2[:200]] = big + rng.integers(0, 1000, 200)
    v[sel2[200:]] = -rng.integers(1, 1000, 100)
    v[2 * nb:][rng.random(nb) < 0.5] = big
    chk = Chunk([Column(abi.I64, k, rng.random(3 * nb) > 0.01), Column(abi.I64, v, rng.random(3 * nb) > 0.02)])
    aggs = AGG_SETS["c3"]
    with ctx.knobs(AGG_BATCH_ROWS=nb):
        cfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs, est_groups=200_000)
        want = orc.hash_agg(cfg, chk, 4, 4)
        stats = []
        got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=nb, fast=abi.AGGFAST_FORCE, stats_out=stats)
    st = stats[0]
    assert st.packed_key_bits > 0 and st.radix_batches == 3 and st.dense_flushes == 1
    assert st.table_slice_bits == 64  # (after batch 3; batches 1 and 2 travelled `bits` wide)
    assert H.rows_equal_unordered(got, want)
    # the same plan on values that fit all the way keeps its narrow cells
    chk2 = Chunk([chk.columns[0], Column(abi.I64, rng.integers(0, vmax, 3 * nb), rng.random(3 * nb) > 0.02)])
    with ctx.knobs(AGG_BATCH_ROWS=nb):
        want = orc.hash_agg(cfg, chk2, 4, 4)
        stats = []
        got = G.run_agg(ctx, cfg, chk2, out_types_for(aggs), chunk_rows=nb, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].table_slice_bits == bits and stats[0].dense_flushes == 1
    assert H.rows_equal_unordered(got, want)


def test_packed_agg_dense_state_emptied_between_batches(ctx, orc):
    """AGG_DENSE = v > 1: the state becomes groups of the table before more than v rows went into it (product: 2^31 rows, so that a
    group's lo32 sums cannot wrap) — the groups of earlier flushes and later ones must add up; DOUBLE sums within the group tolerance."""
    rng = np.random.default_rng(77)
    n = 600_000
    chk, types = _chunk(rng, n, 0, 100_000)
    for name in ("c3", "c3_double", "ints", "minmax2"):
        aggs = AGG_SETS[name]
        with ctx.knobs(AGG_DENSE=150_000, AGG_BATCH_ROWS=1 << 16):
            cfg = H.agg_cfg(types, [0], aggs, est_groups=100_000)
            want = orc.hash_agg(cfg, chk, 4, 4)
            stats = []
            got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 16, fast=abi.AGGFAST_FORCE, stats_out=stats)
        assert stats[0].dense_flushes >= 4, stats[0].dense_flushes
        real_cols = [i for i, a in enumerate(aggs) if a[0] in (abi.AGG_SUM, abi.AGG_AVG) and a[2] in (abi.F64, abi.F32)]
        exact_cols = [i for i in range(len(aggs)) if i not in real_cols]
        key_out = [i for i, a in enumerate(aggs) if a[0] == abi.AGG_FIRSTROW][0]
        _match_by_key(got, want, [key_out], exact_cols, real_cols, group_tols(chk, 0, aggs, real_cols))


@pytest.mark.parametrize("sig", [2, 1])
@pytest.mark.parametrize("key_hi", [30_000, 3_000_000])
def test_packed_agg_count_and_sum_in_one_lds_word(ctx, orc, key_hi, sig):
    """TSQ_KNOB_DAAGG_SIG = 2 (the default since round 6; 1: two words, as before): SUM(BIGINT) + COUNT(*) over 2-byte argument cells
    keeps `count << 40 | sum` in ONE LDS word per cell; the fold into the dense state takes it apart.  Same groups as the oracle, with
    workgroups sharing partitions (30 000 keys: device atomics) and owning them (3e6 keys)."""
    rng = np.random.default_rng(key_hi % 89)
    n = 600_001
    chk, types = _dense_case(rng, n, key_hi, 50_000)
    aggs = AGG_SETS["c3"]
    with ctx.knobs(DAAGG_SIG=sig, AGG_BATCH_ROWS=1 << 18):
        cfg = H.agg_cfg(types, [0], aggs, est_groups=key_hi)
        want = orc.hash_agg(cfg, chk, 4, 4)
        stats = []
        got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 18, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].packed_key_bits > 0 and stats[0].dense_flushes == 1 and stats[0].table_slice_bits == 16
    assert H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("name", ["c3", "c3_double", "ints", "minmax2"])
@pytest.mark.parametrize("key_hi", [30_000, 3_000_000])
def test_packed_agg_side_stream_beside_the_next_partition_pass(ctx, orc, name, key_hi):
    """round 6 (TSQ_KNOB_AGG_OVERLAP; tests: v >= 2 = batches of v rows or more): k_agg_da / k_daagg_ovf of batch i run on the operator's side
    stream while batch i + 1 is partitioned into the second store; the fold into the dense state is made of device atomics then (the
    partition kernel adds its hot keys to the same cells).  Ten batches — both stores reused four times —, a skewed key (hot in every
    batch, and overflowing its region where the hot keys are switched off), NULL keys and arguments as exception rows on the first stream:
    the same groups as the oracle and as the one-stream run."""
    rng = np.random.default_rng(key_hi % 101 + len(name))
    n = 1_000_000
    chk, types = _chunk(rng, n, 0, key_hi)
    hot_rows = rng.random(n) < 0.15
    if chk.columns[0].notnull is not None:
        hot_rows &= chk.columns[0].notnull  # (NULL slots keep their zero bytes)
    chk.columns[0].data[hot_rows] = 4242
    aggs = AGG_SETS[name]
    cfg = H.agg_cfg(types, [0], aggs, est_groups=key_hi)
    want = orc.hash_agg(cfg, chk, 4, 4)
    real_cols = [i for i, a in enumerate(aggs) if a[0] in (abi.AGG_SUM, abi.AGG_AVG) and a[2] in (abi.F64, abi.F32)]
    exact_cols = [i for i in range(len(aggs)) if i not in real_cols]
    key_out = [i for i, a in enumerate(aggs) if a[0] == abi.AGG_FIRSTROW][0]
    for overlap, hot in ((2, 1), (2, 0), (0, 1)):
        stats = []
        with ctx.knobs(AGG_OVERLAP=overlap, DAAGG_HOT=hot, AGG_BATCH_ROWS=100_000):
            got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=100_000, fast=abi.AGGFAST_FORCE, stats_out=stats)
        st = stats[0]
        assert st.packed_key_bits > 0 and st.radix_batches == 10 and st.dense_flushes == 1
        assert st.side_stream_batches == (10 if overlap else 0), st.side_stream_batches
        _match_by_key(got, want, [key_out], exact_cols, real_cols, group_tols(chk, 0, aggs, real_cols))


def test_packed_agg_side_stream_then_other_modes_and_flushes(ctx, orc):
    """the side stream is joined wherever the first stream needs what it works on: a dense state emptied every other batch (AGG_DENSE = v),
    and batches whose keys leave the packed range (the operator goes back to the 64-bit H mode, which reuses the first partitioned store)."""
    rng = np.random.default_rng(5)
    n = 800_000
    chk, types = _chunk(rng, n, 0, 150_000)
    aggs = AGG_SETS["c3"]
    cfg = H.agg_cfg(types, [0], aggs, est_groups=150_000)
    want = orc.hash_agg(cfg, chk, 4, 4)
    stats = []
    with ctx.knobs(AGG_OVERLAP=2, AGG_DENSE=150_000, AGG_BATCH_ROWS=1 << 16):
        got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), chunk_rows=1 << 16, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert stats[0].side_stream_batches >= 12 and stats[0].dense_flushes >= 5
    assert H.rows_equal_unordered(got, want)
    # the second half of the input far outside the range the first batches showed
    k2 = chk.columns[0].data.copy()
    k2[n // 2:] = rng.integers(1 << 40, (1 << 40) + 10**9, n - n // 2)
    chk2 = Chunk([Column(abi.I64, k2, chk.columns[0].notnull)] + chk.columns[1:])
    want2 = orc.hash_agg(cfg, chk2, 4, 4)
    stats = []
    with ctx.knobs(AGG_OVERLAP=2, AGG_BATCH_ROWS=100_000):
        got2 = G.run_agg(ctx, cfg, chk2, out_types_for(aggs), chunk_rows=100_000, fast=abi.AGGFAST_FORCE, stats_out=stats)
    assert 3 <= stats[0].side_stream_batches <= 5, stats[0].side_stream_batches
    assert H.rows_equal_unordered(got2, want2)
