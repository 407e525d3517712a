"""GPU parity of GROUP BY through the dictionary of group keys (round 5, csrc/tsq_keydict.h): string keys and key sets of integers and
strings whose cells fit a 32-byte key record get a dense id per key (hash-partitioned records, one workgroup per partition finds or
inserts), a child aggregate groups by the id, and the key columns come back from the dictionary records.  The reference keys its
partial results by the encoded group key (executor/aggregate.go:332-350, util/codec/codec.go:700-760: NULL is a group, '' is another).
The route is FORCED (tsq_agg_set_fast) and asserted through tsq_stats.build_partitioned == 3; everything is compared with the oracle."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column, StrColumn

from . import gpu_helpers as G
from . import helpers as H
from .test_agg_gpu import out_types_for
from .test_agg_string_gpu import _words

pytestmark = pytest.mark.gpu
FORCE = abi.AGGFAST_FORCE


def _run(ctx, cfg, chk, aggs, want_dict=True, **kw):
    stats = []
    got = G.run_agg(ctx, cfg, chk, out_types_for(aggs), fast=FORCE, stats_out=stats, **kw)
    if want_dict is not None:
        assert (stats[0].build_partitioned == 3) == want_dict, stats[0].build_partitioned
    return got, stats[0]


@pytest.mark.parametrize("n,distinct,chunk_rows", [(1, 1, 1024), (37, 5, 1024), (5000, 300, 1 << 20), (200_001, 20_000, 1 << 20), (200_001, 20_000, 50_000),
                                                    (300_000, 299_000, 100_000)])
def test_string_group_key_vs_oracle(ctx, orc, n, distinct, chunk_rows):
    rng = np.random.default_rng(n + distinct)
    k = _words(rng, n, distinct)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-1000, hi=1000)
    d = H.random_column(rng, abi.F64, n, 0.1)
    chk = Chunk([k, v, d])
    types = [abi.BYTES, abi.I64, abi.F64]
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_MIN, 1, abi.I64), (abi.AGG_MAX, 2, abi.F64),
            (abi.AGG_AVG, 1, abi.I64), (abi.AGG_COUNT, 2, abi.F64)]
    cfg = H.agg_cfg(types, [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    got, st = _run(ctx, cfg, chk, aggs, chunk_rows=chunk_rows, pull_rows=4096)
    assert got.NumRows() == want.NumRows()
    assert H.rows_equal_unordered(got, want)
    assert st.build_handed_back_rows == 0  # every cell fits a record: no exception rows


def test_separate_arrays_instead_of_slots(ctx, orc):
    # the scatter pass writes one 64-byte slot per row (record, travelling cells, source row) when <= 3 columns travel; with FOUR argument
    # columns — and under TSQ_KNOB_KEYREC = 2 — records, row ids and cells go to separate arrays: the same groups either way
    rng = np.random.default_rng(77)
    n = 70_000
    k = _words(rng, n, 900, hi=14)
    vs = [H.random_column(rng, abi.I64, n, 0.1, lo=-1000, hi=1000) for _ in range(4)]
    chk = Chunk([k] + vs)
    types = [abi.BYTES] + [abi.I64] * 4
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64)] + [(abi.AGG_SUM, 1 + i, abi.I64) for i in range(4)]
    cfg = H.agg_cfg(types, [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    got, _ = _run(ctx, cfg, chk, aggs, chunk_rows=1 << 20, pull_rows=4096)
    assert H.rows_equal_unordered(got, want)
    aggs3 = aggs[:5]
    cfg3 = H.agg_cfg(types, [0], aggs3)
    want3 = orc.hash_agg(cfg3, chk, 4, 4)
    for knob in (1, 2):
        with ctx.knobs(KEYREC=knob):
            got3, _ = _run(ctx, cfg3, chk, aggs3, chunk_rows=1 << 20, pull_rows=4096)
        assert H.rows_equal_unordered(got3, want3), knob


def test_null_and_empty_string_keys_are_different_groups(ctx, orc):
    k = StrColumn([None, b"", b"", None, b"a", b"a\0", b"a", None])  # a trailing NUL byte is part of the value
    chk = Chunk([k, Column(abi.I64, np.arange(8))])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    cfg = H.agg_cfg([abi.BYTES, abi.I64], [0], aggs)
    got, _ = _run(ctx, cfg, chk, aggs)
    assert H.rows_equal_unordered(got, [(None, 3, 0 + 3 + 7), (b"", 2, 3), (b"a", 2, 10), (b"a\0", 1, 5)])


@pytest.mark.parametrize("keys", [[0, 1], [1, 0], [0, 1, 2], [2, 1]])
def test_key_sets_of_strings_and_integers_vs_oracle(ctx, orc, keys):
    # (string, bigint[, short string]): NULL cells in every key column, negative and huge integers, the same integer as BIGINT UNSIGNED
    rng = np.random.default_rng(len(keys) * 7 + keys[0])
    n = 60_000
    a = _words(rng, n, 40, hi=10)
    bvals = np.array([-(1 << 63), -7, 0, 3, (1 << 62) + 5, (1 << 63) - 1])[rng.integers(0, 6, n)]
    b = Column(abi.I64, bvals, rng.random(n) > 0.05)
    c = _words(rng, n, 6, lo=0, hi=4)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-50, hi=50)
    chk = Chunk([a, b, c, v])
    types = [abi.BYTES, abi.I64, abi.BYTES, abi.I64]
    aggs = [(abi.AGG_FIRSTROW, kc, types[kc]) for kc in keys] + [(abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 3, abi.I64), (abi.AGG_MAX, 3, abi.I64)]
    cfg = H.agg_cfg(types, keys, aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    got, _ = _run(ctx, cfg, chk, aggs, chunk_rows=1 << 20, pull_rows=4096)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_cells_that_do_not_fit_a_record_are_exception_rows(ctx, orc):
    # a third of the keys is longer than a record can hold (30 bytes for one string key): those rows take the several-column upsert into the
    # operator's own table, the others the dictionary; the groups of both come out once each
    rng = np.random.default_rng(5)
    n = 80_000
    short = _words(rng, n, 500, hi=20, null_frac=0.02)
    longs = _words(rng, n, 300, lo=31, hi=70, null_frac=0.0)
    pick = rng.random(n) < 0.33
    sv, lv = short.values(), longs.values()
    k = StrColumn([lv[i] if pick[i] else sv[i] for i in range(n)])
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-1000, hi=1000)
    chk = Chunk([k, v])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    cfg = H.agg_cfg([abi.BYTES, abi.I64], [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    for chunk_rows in (1 << 20, 30_000):
        got, st = _run(ctx, cfg, chk, aggs, chunk_rows=chunk_rows, pull_rows=4096)
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
        assert st.build_handed_back_rows == sum(1 for x in k.values() if x is not None and len(x) > 30)


def test_a_full_partition_of_the_dictionary_hands_rows_back(ctx, orc):
    # est_groups = 1 -> ONE partition (12 288 places): 40 000 distinct keys do not fit; the keys that found no place are exception rows in
    # every batch (a key is either in the dictionary for good or never), every group comes out exactly once
    rng = np.random.default_rng(9)
    n = 150_000
    ids = rng.integers(0, 40_000, n)
    k = StrColumn([b"key-%07d" % i for i in ids])
    v = Column(abi.I64, rng.integers(-9, 9, n))
    chk = Chunk([k, v])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    cfg = H.agg_cfg([abi.BYTES, abi.I64], [0], aggs)
    cfg.est_groups = 1
    want = orc.hash_agg(cfg, chk, 4, 4)
    with ctx.knobs(AGG_BATCH_ROWS=50_000):  # three device batches of 50 000 rows (a first batch of 65 536 rows or more gets 256 partitions at least)
        got, st = _run(ctx, cfg, chk, aggs, chunk_rows=50_000, pull_rows=4096)
    assert got.NumRows() == want.NumRows() == len(np.unique(ids)) and H.rows_equal_unordered(got, want)
    assert st.build_handed_back_rows > 0


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_partition_fills_up_in_a_batch_with_several_rows_per_key(ctx, orc, seed):
    # the batch in which the ONE partition (12 288 places, index 3/4 full: long walks) runs out of places brings every key
    # ~4 times — a reservation that draws no place must not turn back into an empty slot (a key that settled BEHIND it would be looked
    # for at the empty slot by its later rows: the group would live in the dictionary AND in the parent's table and come out twice).
    # Every group exactly once, with its exact count and sum, also for the rows of the next batch.
    rng = np.random.default_rng(seed)
    n = 120_000
    ids = rng.integers(0, 14_000, n)
    k = StrColumn([b"key-%07d" % i for i in ids])
    v = Column(abi.I64, rng.integers(-9, 9, n))
    chk = Chunk([k, v])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    cfg = H.agg_cfg([abi.BYTES, abi.I64], [0], aggs)
    cfg.est_groups = 1
    want = orc.hash_agg(cfg, chk, 4, 4)
    with ctx.knobs(AGG_BATCH_ROWS=60_000):  # (a first batch of 65 536 rows or more gets 256 partitions at least)
        got, st = _run(ctx, cfg, chk, aggs, chunk_rows=60_000, pull_rows=4096)
    assert got.NumRows() == want.NumRows() == len(np.unique(ids)) and H.rows_equal_unordered(got, want)
    assert st.build_handed_back_rows > 0


def test_hot_key_and_many_rows_per_key(ctx, orc):
    # half of the rows carry one key: hundreds of rows bring the same NEW key to a partition at once (each draws a place, one publishes,
    # the others leave tombstones); later batches find it
    rng = np.random.default_rng(2)
    n = 400_000
    ids = np.where(rng.random(n) < 0.5, 7, rng.integers(0, 3000, n))
    k = StrColumn([b"k%d" % i for i in ids])
    v = Column(abi.I64, rng.integers(0, 100, n))
    chk = Chunk([k, v])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    cfg = H.agg_cfg([abi.BYTES, abi.I64], [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    got, st = _run(ctx, cfg, chk, aggs, chunk_rows=100_000, pull_rows=4096)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    assert st.build_handed_back_rows == 0


def test_partial_and_final_modes_through_the_dictionary(ctx, orc):
    # descriptor.go:56-91 Split: Complete == Final(Partial1), both halves on the dictionary route
    rng = np.random.default_rng(21)
    n = 90_000
    k = _words(rng, n, 2000, hi=12)
    v = H.random_column(rng, abi.I64, n, 0.1, lo=-1000, hi=1000)
    chk = Chunk([k, v])
    types = [abi.BYTES, abi.I64]
    complete = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_AVG, 1, abi.I64), (abi.AGG_COUNT, 1, abi.I64)]
    want = orc.hash_agg(H.agg_cfg(types, [0], complete), chk, 1, 1)
    paggs = [(f, c, t, abi.MODE_PARTIAL1) for f, c, t in complete]
    pt = out_types_for(paggs)
    from tinysql_amd.chunk import concat
    parts = [_run(ctx, H.agg_cfg(types, [0], paggs), chk.slice(lo, hi), paggs, chunk_rows=1 << 20)[0] for lo, hi in [(0, 40_000), (40_000, 40_001), (40_001, n)]]
    both = concat(parts, pt)
    # partial columns: key, sum, (count, sum) of AVG, count
    faggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES, abi.MODE_FINAL), (abi.AGG_SUM, 1, abi.I64, abi.MODE_FINAL), (abi.AGG_AVG, 2, abi.I64, abi.MODE_FINAL, 3),
             (abi.AGG_COUNT, 4, abi.I64, abi.MODE_FINAL)]
    got, _ = _run(ctx, H.agg_cfg(pt, [0], faggs), both, faggs, chunk_rows=1 << 20)
    assert H.rows_equal_unordered(got, want)


def test_plans_the_dictionary_does_not_take(ctx, orc):
    # MAX of a string argument (var-len cells do not travel) and a double key keep the several-column upsert, with the same groups
    rng = np.random.default_rng(4)
    n = 20_000
    k = _words(rng, n, 100)
    s = _words(rng, n, 50)
    chk = Chunk([k, s])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_MAX, 1, abi.BYTES)]
    cfg = H.agg_cfg([abi.BYTES, abi.BYTES], [0], aggs)
    got, _ = _run(ctx, cfg, chk, aggs, want_dict=False)
    assert H.rows_equal_unordered(got, orc.hash_agg(cfg, chk, 4, 4))
    with ctx.knobs(KEYREC=0):  # the knob that turns the key-record routes off
        aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64)]
        cfg = H.agg_cfg([abi.BYTES, abi.BYTES], [0], aggs)
        got, _ = _run(ctx, cfg, chk, aggs, want_dict=False)
        assert H.rows_equal_unordered(got, orc.hash_agg(cfg, chk, 4, 4))


def _record_tags(keys):
    """the 18-bit index tag k_kd_assign derives from the key record of a single string key (csrc/tsq_keyrec.h kr_record / kr_hash): flag 2,
    length byte, bytes, zero padding to 32 bytes = four little-endian words through kr_hash; tag = bits 14..31 of the mix"""
    rec = np.zeros((len(keys), 32), np.uint8)
    for i, k in enumerate(keys):
        rec[i, 0], rec[i, 1] = 2, len(k)
        rec[i, 2:2 + len(k)] = np.frombuffer(k, np.uint8)
    w = rec.view("<u8")
    M = np.uint64

    def rotl32(x):
        return (x << M(32)) | (x >> M(32))
    with np.errstate(over="ignore"):  # kr_hash: one multiply per word, a multiply-xorshift finish
        h = (w[:, 0] ^ M(0x6A09E667F3BCC908)) * M(0x9E3779B97F4A7C15)
        for c, k in ((1, 0xBF58476D1CE4E5B9), (2, 0x94D049BB133111EB), (3, 0xD6E8FEB86659FD93)):
            h = (w[:, c] ^ rotl32(h)) * M(k)
        h = (h ^ (h >> M(32))) * M(0xFF51AFD7ED558CCD)
        h = h ^ (h >> M(29))
    return (h >> M(14)) & M(0x3ffff)


def test_keys_whose_tag_is_all_ones(ctx, orc):
    # (all-ones tag, "reserved" place code) is the bit pattern of an EMPTY slot: a slot reserved for such a key looked empty to the next
    # row that brought the key, and the key entered the dictionary twice — 66 groups too many in 1e6 (found by a group count at 1e8 rows).
    # Keys with that tag (searched with the numpy restatement of the record hash above), each brought by thousands of rows at once
    cand = [b"t%d" % i for i in range(1_500_000)]
    tags = _record_tags(cand)
    hot = [cand[i] for i in np.flatnonzero(tags == 0x3ffff)]
    assert len(hot) >= 2
    rng = np.random.default_rng(1)
    n = 300_000
    pool = hot + [b"f%d" % i for i in range(500)]
    pick = np.where(rng.random(n) < 0.8, rng.integers(0, len(hot), n), len(hot) + rng.integers(0, 500, n))
    k = StrColumn([pool[i] for i in pick])
    v = Column(abi.I64, rng.integers(0, 100, n))
    chk = Chunk([k, v])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.BYTES), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    cfg = H.agg_cfg([abi.BYTES, abi.I64], [0], aggs)
    want = orc.hash_agg(cfg, chk, 4, 4)
    for chunk_rows in (1 << 20, 70_000):
        got, _ = _run(ctx, cfg, chk, aggs, chunk_rows=chunk_rows, pull_rows=4096)
        assert got.NumRows() == want.NumRows() == len(np.unique(pick)) and H.rows_equal_unordered(got, want)


def test_integer_key_columns_too_wide_for_the_composite_word(ctx, orc):
    # three BIGINT key columns with 64-bit ranges: 192 bits of fields — neither the packed several-column route (23 bits) nor the composite
    # word (63 bits) holds them; their cells fit a key record (3 x 9 bytes): the dictionary takes the aggregate.  Four such columns do not
    # fit (36 bytes): that plan keeps the several-column upsert
    rng = np.random.default_rng(12)
    n = 120_000
    pool = rng.integers(-(1 << 63), (1 << 63) - 1, (3000, 4), dtype=np.int64)
    pick = rng.integers(0, 3000, n)
    cols = [Column(abi.I64, pool[pick, c], rng.random(n) > 0.03) for c in range(4)]
    v = Column(abi.I64, rng.integers(-100, 100, n))
    for nk, want_dict in ((3, True), (4, False)):
        chk = Chunk(cols[:nk] + [v])
        types = [abi.I64] * (nk + 1)
        aggs = [(abi.AGG_FIRSTROW, c, abi.I64) for c in range(nk)] + [(abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, nk, abi.I64)]
        cfg = H.agg_cfg(types, list(range(nk)), aggs)
        want = orc.hash_agg(cfg, chk, 4, 4)
        got, _ = _run(ctx, cfg, chk, aggs, want_dict=want_dict, chunk_rows=1 << 20, pull_rows=4096)
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


# ------------------------------------------------------------------ the reference's own aggregate vectors through the dictionary
# tests/golden/agg_cases.json "sql" (executor/aggregate_test.go) groups by an integer column; grouping by its decimal text — a string
# key, NULL staying NULL — makes the same groups, and with the fast paths FORCED that key goes through the dictionary of key records
@pytest.mark.parametrize("case", [c for c in H.golden("agg_cases.json")["sql"] if c["group_by"]], ids=lambda c: c["ref"][:40])
def test_golden_aggregate_cases_with_the_group_key_as_a_string(ctx, case):
    types = [H.TYPES[t] for t in case["types"]]
    gk = case["group_by"][0]
    rows = [[(None if v is None else b"%d" % v) if i == gk else v for i, v in enumerate(r)] for r in case["rows"]]
    stypes = [abi.BYTES if i == gk else t for i, t in enumerate(types)]
    chk = H.chunk_from_rows(rows, stypes)
    aggs = [(H.AGG_FUNCS[f], col, abi.BYTES if col == gk else H.TYPES[t]) for f, col, t in case["aggs"]]
    cfg = H.agg_cfg(stypes, case["group_by"], aggs)
    got, _ = _run(ctx, cfg, chk, aggs, want_dict=bool(case["rows"]))
    assert H.rows_equal_unordered(got, [tuple(r) for r in case["expect"]]), case["ref"]
