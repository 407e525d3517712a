"""GPU parity of the materialising packed join with the BUILD SIDE IN LDS (round 6, csrc/tsq_damat.h): a build side without duplicate
keys goes through two partition levels (k_da_partition_cols, then k_dm_split: one workgroup per level-1 partition, S ways, an inner
join drops the probe rows whose word has no bit in the build side's bitmap), the emit kernel ranks a final partition's build rows in
LDS and every probe row makes exactly one output row.  HashJoinExec.Next materialises every joined row (executor/join.go:290-323,
joiner.go:145-410, util/chunk/chunk.go:334-356): inner / left outer / right outer, NULL keys and NULL payload cells on both sides,
probe keys outside the build side's range, OtherConditions, outer-side filters, selected[], several key columns — everything is
compared with the oracle, the variant is asserted through tsq_stats.packed_lds_bits, and TSQ_KNOB_DA_LDS_BUILD forces every S."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu
FORCE, OFF = abi.RADIX_FORCE, abi.RADIX_OFF


def _pay(rng, n, c, null_pay):
    tp = (abi.I64, abi.F64, abi.U64)[c % 3]
    if tp == abi.F64:
        data = rng.random(n)
    elif tp == abi.U64:
        data = rng.integers(0, 1 << 63, n).astype(np.uint64)
    else:
        data = rng.integers(-(1 << 40), 1 << 40, n)
    return Column(tp, data, (rng.random(n) > null_pay) if c % 2 else None)


def _unique_build(rng, n, lo, hi, ncols, null_key=0.02, null_pay=0.1):
    keys = rng.permutation(np.arange(lo, hi))[:n]  # no key twice (NULL keys are never inserted: hash_table.go:161-163)
    return Chunk([Column(abi.I64, keys, rng.random(n) > null_key)] + [_pay(rng, n, c, null_pay) for c in range(1, ncols)])


def _probe(rng, n, lo, hi, ncols, null_key=0.03, null_pay=0.1):
    return Chunk([Column(abi.I64, rng.integers(lo, hi, n), rng.random(n) > null_key)] + [_pay(rng, n, c + 1, null_pay) for c in range(1, ncols)])


def _rows(ctx, cfg, build, probe, want_lds=True, chunk_rows=1 << 22, selected=None, knob=1):
    stats = []
    with ctx.knobs(DA_LDS_BUILD=knob):
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=chunk_rows, pull_rows=4096, radix=FORCE, packing=FORCE, stats_out=stats, selected=selected)
    st = stats[0]
    assert st.probe_route == abi.ROUTE_PACKED, (st.probe_route, st.radix_batches)
    if want_lds is not None:
        assert (st.packed_lds_bits > 0) == want_lds, (st.packed_lds_bits, st.radix_bits)
    return got, st


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_INNER, 0), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("n_probe,np_cols,nb_cols,knob", [(1, 1, 1, 1), (64, 2, 2, 2), (4097, 4, 6, 3), (60_001, 8, 8, 4), (150_003, 3, 2, 5), (33_333, 2, 1, 5), (70_000, 1, 3, 1)])
def test_lds_build_vs_oracle(ctx, orc, jt, inner, n_probe, np_cols, nb_cols, knob):
    rng = np.random.default_rng(13 * n_probe + jt + inner)
    bside = _unique_build(rng, 6000, -4000, 5000, nb_cols)   # 2/3 of the keys of the range: probe rows miss inside the range too
    pside = _probe(rng, n_probe, -5000, 6000, np_cols)       # ... and on both sides of it; NULL keys; NULL payload cells
    left, right = (pside, bside) if inner == 1 else (bside, pside)
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], jt, inner)
    want = orc.hash_join(cfg, bside, pside)
    got, st = _rows(ctx, cfg, bside, pside, knob=knob)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    if knob >= 2:
        assert st.packed_lds_bits == st.radix_bits + (knob - 2)
    got, _ = _rows(ctx, cfg, bside, pside, chunk_rows=1024, knob=knob)  # host chunks of tidb_max_chunk_size rows reach the same batch
    assert H.rows_equal_unordered(got, want)


def test_lds_build_knob_off_and_duplicate_build_keys_take_the_sorted_columns_variant(ctx, orc):
    rng = np.random.default_rng(5)
    pside = _probe(rng, 50_000, -1100, 1200, 2)
    uniq = _unique_build(rng, 1500, -900, 1000, 2)
    cfg = H.join_cfg(pside.types(), uniq.types(), [0], [0], abi.JOIN_LEFT_OUTER, 1)
    want = orc.hash_join(cfg, uniq, pside)
    got, _ = _rows(ctx, cfg, uniq, pside, want_lds=False, knob=0)
    assert H.rows_equal_unordered(got, want)
    got, _ = _rows(ctx, cfg, uniq, pside, want_lds=True)
    assert H.rows_equal_unordered(got, want)
    dups = Chunk([Column(abi.I64, rng.integers(-900, 1000, 3000)), _pay(rng, 3000, 1, 0.1)])  # ~1.6 build rows per key
    want = orc.hash_join(cfg, dups, pside)
    got, _ = _rows(ctx, cfg, dups, pside, want_lds=False)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("jt", [abi.JOIN_INNER, abi.JOIN_LEFT_OUTER])
def test_lds_build_skewed_probe_keys_fall_back(ctx, orc, jt):
    # a hot probe key overflows its level-1 region: the batch is taken by the sorted-columns variant (its overflow list), the next,
    # evenly spread batch of the same join by the LDS variant again
    rng = np.random.default_rng(17 + jt)
    build = _unique_build(rng, 4000, 0, 5000, 2, null_key=0.0)
    n = 70_000
    hot = Chunk([Column(abi.I64, rng.choice(np.array([5, 5, 5, 101, 4000, 77, -3, 50_000], dtype=np.int64), n), rng.random(n) > 0.02), _pay(rng, n, 1, 0.1)])
    cfg = H.join_cfg(hot.types(), build.types(), [0], [0], jt, 1)
    want = orc.hash_join(cfg, build, hot)
    got, st = _rows(ctx, cfg, build, hot, want_lds=False)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want) and st.radix_overflow_rows > 0
    even = _probe(rng, n, -100, 5100, 2)
    nn = lambda c: c.notnull if c.notnull is not None else np.ones(len(c), bool)  # noqa: E731
    both = Chunk([Column(c0.tp, np.concatenate([c0.data, c1.data]), np.concatenate([nn(c0), nn(c1)])) for c0, c1 in zip(hot.columns, even.columns)])
    cfg2 = H.join_cfg(both.types(), build.types(), [0], [0], jt, 1, probe_batch_rows=70_016)  # (rounded up to 64 rows)
    want = orc.hash_join(cfg2, build, both)
    got, st = _rows(ctx, cfg2, build, both, want_lds=True, chunk_rows=70_016)  # two device batches: the last one is (almost) even
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
def test_lds_build_other_conditions_and_outer_filter(ctx, orc, jt, inner):
    # OtherConditions over the joined rows (joiner.go:155-167, 351-378; an outer row whose only candidate fails becomes the padded row:
    # 274-281) and the outer-side filter of an outer join (join.go:328-345: a row that fails it is padded without touching the table)
    rng = np.random.default_rng(23 + jt)
    n = 80_000
    bside = Chunk([Column(abi.I64, rng.permutation(6000), rng.random(6000) > 0.02), Column(abi.I64, rng.integers(-50, 50, 6000), rng.random(6000) > 0.1)])
    pside = Chunk([Column(abi.I64, rng.integers(-500, 6500, n), rng.random(n) > 0.03), Column(abi.I64, rng.integers(-50, 50, n), rng.random(n) > 0.1)])
    left, right = (pside, bside) if inner == 1 else (bside, pside)
    keep = []
    conds = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(1, abi.I64), E.Column(3, abi.I64)), E.Constant(0))]
    filt = [E.ScalarFunction("lt", E.Column(1, abi.I64), E.Constant(30))] if jt != abi.JOIN_INNER else ()  # (over the outer side's own row)
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], jt, inner, conds, filt, keep)
    want = orc.hash_join(cfg, bside, pside)
    got, _ = _rows(ctx, cfg, bside, pside, knob=4)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_lds_build_selected_flags(ctx, orc):
    # an externally evaluated outer-side filter (tsq_join_probe_push's selected[]): a row with flag 0 behaves like a row with a NULL key
    rng = np.random.default_rng(29)
    n = 50_000
    bside = _unique_build(rng, 3000, 0, 4000, 2)
    pside = _probe(rng, n, -100, 4100, 3)
    sel = (rng.random(n) > 0.3).astype(np.uint8)
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = H.join_cfg(pside.types(), bside.types(), [0], [0], jt, 1)
        want = orc.hash_join(cfg, bside, pside, selected=sel)
        got, _ = _rows(ctx, cfg, bside, pside, selected=sel, knob=3)
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_lds_build_two_key_columns(ctx, orc):
    # several integer key columns ride the packed routes as one composite column (k_da_compose): the key columns travel like payload
    rng = np.random.default_rng(31)
    nb, n = 5000, 60_000
    pairs = rng.permutation(100 * 80)[:nb]
    bside = Chunk([Column(abi.I64, pairs // 80 - 30, rng.random(nb) > 0.02), Column(abi.I64, pairs % 80 + 1000), _pay(rng, nb, 1, 0.1)])
    pside = Chunk([Column(abi.I64, rng.integers(-35, 75, n), rng.random(n) > 0.03), Column(abi.I64, rng.integers(995, 1085, n), rng.random(n) > 0.03), _pay(rng, n, 2, 0.1)])
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = H.join_cfg(pside.types(), bside.types(), [0, 1], [0, 1], jt, 1)
        want = orc.hash_join(cfg, bside, pside)
        got, _ = _rows(ctx, cfg, bside, pside, knob=4)
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_lds_build_1e7_checksum_every_split(ctx):
    # at scale (2^23 unique build rows, 3 x 2^22 probe rows, hit ratio 0.5, nullable payloads on both sides): the joined rows'
    # order-independent checksum equals the direct route's (tsq_join_set_checksum) for AUTO's S and for S = 8; inner and left outer
    rng = np.random.default_rng(37)
    nb, n = 1 << 23, 3 * (4 << 20)
    build = Chunk([Column(abi.I64, rng.permutation(nb).astype(np.int64)), Column(abi.I64, rng.integers(0, 1 << 40, nb), rng.random(nb) > 0.03)])
    probe = Chunk([Column(abi.I64, rng.integers(0, 2 * nb, n), rng.random(n) > 0.03), Column(abi.F64, rng.random(n), rng.random(n) > 0.03)])
    from oracle import binding as orc_b
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = H.join_cfg(probe.types(), build.types(), [0], [0], jt, 1)
        c, s, x = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 24, count_only=True, checksum=True, radix=OFF)
        for knob in (1, 5):
            stats = []
            with ctx.knobs(DA_LDS_BUILD=knob):
                got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 24, pull_rows=1 << 20, stats_out=stats)
            assert stats[0].probe_route == abi.ROUTE_PACKED and stats[0].radix_batches == 3 and stats[0].packed_lds_bits > 0
            assert got.NumRows() == c
            assert orc_b.rows_checksum(got) == (s, x), (jt, knob)


@pytest.mark.parametrize("second_push", [False, True])
def test_retained_build_chunk(ctx, orc, second_push):
    # TSQ_COL_RETAIN: the first build chunk is kept where it is (hash_table.go:146-169: PutChunk keeps the chunk it is handed,
    # util/chunk/list.go:96-110) — nullable payload, NULL keys; a second push copies the retained rows into the operator's own storage
    # first.  Host-pulled rows equal the oracle's either way, and the caller's buffers are unchanged afterwards.
    import ctypes as C
    from tinysql_amd import _lib
    rng = np.random.default_rng(41 + second_push)
    nb, n = 40_000, 90_000
    bside = _unique_build(rng, nb, 0, 60_000, 3)
    pside = _probe(rng, n, -500, 60_500, 2)
    cfg = H.join_cfg(pside.types(), bside.types(), [0], [0], abi.JOIN_LEFT_OUTER, 1)
    want = orc.hash_join(cfg, bside, pside)
    cut = 25_000 if second_push else nb
    first, rest = bside.slice(0, cut), bside.slice(cut, nb)
    dcols = [G.to_device(ctx, c) for c in first.columns]
    before = [d.to_host() for d in dcols]
    lib = ctx.lib
    h = C.c_void_p()
    _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        _lib.check(lib.tsq_join_set_radix(h, FORCE), h)
        _lib.check(lib.tsq_join_set_key_packing(h, FORCE), h)
        arr = G.dev_cols(dcols)
        for c in arr:
            c.flags = abi.COL_DEVICE | abi.COL_RETAIN
        _lib.check(lib.tsq_join_build_push(h, arr, len(dcols), cut), h)
        if second_push:
            keep = []
            _lib.check(lib.tsq_join_build_push(h, G.make_cols(rest.columns, keep), len(rest.columns), nb - cut), h)
        _lib.check(lib.tsq_join_build_finish(h), h)
        keep = []
        _lib.check(lib.tsq_join_probe_push(h, G.make_cols(pside.columns, keep), len(pside.columns), n, None), h)
        _lib.check(lib.tsq_join_probe_finish(h), h)
        out_types = pside.types() + bside.types()
        got = []
        while True:
            keep = []
            out, bufs = G.out_buffers(out_types, 8192, keep, None)
            nr, eos = C.c_int64(0), C.c_int32(0)
            _lib.check(lib.tsq_join_pull(h, out, len(out_types), 8192, C.byref(nr), C.byref(eos)), h)
            if nr.value == 0:
                break
            got.append(G.chunk_from_buffers(out_types, bufs, nr.value))
        from tinysql_amd.chunk import concat
        got = concat(got, out_types)
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    finally:
        lib.tsq_join_destroy(h)
    for d, b in zip(dcols, before):
        a = d.to_host()
        assert np.array_equal(a.data, b.data) and ((a.notnull is None) == (b.notnull is None)) and (a.notnull is None or np.array_equal(a.notnull, b.notnull))
        d.free()
