"""GPU parity of the PACKED-KEY route of the COUNT(*) probe (csrc/tsq_dajoin.h): the build side's key range is
reduced once, keys travel as 2-byte entries of a bijective mix of key - kmin and meet one-byte direct-address images in
LDS.  Forced on small inputs so that every edge — ranges of one value and of 2^28, negative keys, unsigned keys above
2^63, mixed signedness, NULL keys, probe keys outside the range, 255 / 256 duplicates, region overflow on both sides —
is compared with the oracle; AUTO is checked at 1e7 rows with a hit ratio below one.

The joined rows of an equi-join depend on key equality only (util/codec/codec.go:363-382); the route must not show.
"""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu

FORCE, OFF = abi.RADIX_FORCE, abi.RADIX_OFF


def _cfg(bt=abi.I64, pt=abi.I64):
    return H.join_cfg([pt, abi.I64], [bt, abi.I64], [0], [0], abi.JOIN_INNER, 1)


def _count(ctx, cfg, build, probe, packing=FORCE, chunk_rows=1 << 22, want_route=None):
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=chunk_rows, count_only=True, radix=FORCE, packing=packing, stats_out=stats)
    if want_route is not None:
        assert stats[0].probe_route == want_route, (stats[0].probe_route, stats[0].packed_key_bits)
    return got


def _tables(bk, pk, bnn=None, pnn=None, bt=abi.I64, pt=abi.I64):
    build = Chunk([Column(bt, bk, bnn), Column(abi.I64, np.arange(len(bk)))])
    probe = Chunk([Column(pt, pk, pnn), Column(abi.I64, np.arange(len(pk)))])
    return build, probe


@pytest.mark.parametrize("case", H.golden("join_cases.json"), ids=lambda c: c["ref"][:48])
def test_packed_forced_on_golden_rows(ctx, case):
    keep = []
    cfg, _, _, build, probe, _, _ = H.lower_join_case(case, keep)
    assert G.run_join(ctx, cfg, build, probe, count_only=True, radix=FORCE, packing=FORCE) == len(case["expect"]), case["ref"]


@pytest.mark.parametrize("n_probe", [1, 63, 64, 65, 1000, 16383, 16384, 16385, 50001, 200_003])
def test_packed_ragged_sizes_dups_nulls_vs_oracle(ctx, orc, n_probe):
    rng = np.random.default_rng(n_probe)
    n_build = 3000
    bk = rng.integers(-400, 500, n_build)
    pk = rng.integers(-450, 600, n_probe)  # some probe keys lie outside the build range on either side
    build, probe = _tables(bk, pk, rng.random(n_build) > 0.1, rng.random(n_probe) > 0.1)
    cfg = _cfg()
    want = orc.hash_join(cfg, build, probe).NumRows()
    assert _count(ctx, cfg, build, probe, want_route=abi.ROUTE_PACKED) == want
    assert _count(ctx, cfg, build, probe, packing=OFF) == want
    assert _count(ctx, cfg, build, probe, chunk_rows=1024) == want  # host chunks through pinned staging reach the same batch


@pytest.mark.parametrize("n_build,span", [(1, 1), (5, 1), (40_000, 40_000), (300_000, 1 << 20), (70_000, (1 << 28) - 1)])
def test_packed_ranges_and_build_overflow(ctx, n_build, span):
    # one build key; a dense range; a range of exactly 2^28 - 1 (b = 28: 4-byte entries, 128 KB images).  A build side of more
    # than ~7 K rows in one tile overflows its regions: those rows reach the images through k_da_build_ovf
    rng = np.random.default_rng(n_build)
    base = -(1 << 40) + 12345
    bk = base + rng.integers(0, span, n_build)
    bk[0], bk[-1] = base, base + span - 1
    pk = base + rng.integers(-span // 8 - 3, span + span // 8 + 3, 150_000)
    build, probe = _tables(bk, pk)
    keys, cnts = np.unique(bk, return_counts=True)
    if cnts.max() > 255:
        pytest.skip("by construction")
    pos = np.searchsorted(keys, pk)
    pos[pos == len(keys)] = 0
    want = int(cnts[pos][keys[pos] == pk].sum())
    assert _count(ctx, _cfg(), build, probe, want_route=abi.ROUTE_PACKED) == want


@pytest.mark.parametrize("bt,pt", [(abi.U64, abi.I64), (abi.I64, abi.U64), (abi.U64, abi.U64), (abi.I64, abi.I64)])
def test_packed_signedness(ctx, orc, bt, pt):
    # same-typed keys compare as 64-bit cells; BIGINT against BIGINT UNSIGNED never matches from 2^63 on (flag 8 vs 9,
    # util/codec/codec.go:219-224) — a build side that holds BOTH small and huge cells has no packable range unless the huge
    # ones are unusable anyway (mixed signedness)
    rng = np.random.default_rng(17)

    def col(n, hi_frac):
        v = rng.integers(0, 2000, n).astype(np.uint64)
        hi = rng.random(n) < hi_frac
        v[hi] = np.uint64(1 << 63) + v[hi]
        return v

    for hi_frac in (0.0, 0.3, 1.0):
        bv, pv = col(5000, hi_frac), col(40_000, 0.3)
        build = Chunk([Column(bt, bv.view(np.int64) if bt == abi.I64 else bv, rng.random(5000) > 0.05), Column(abi.I64, np.arange(5000))])
        probe = Chunk([Column(pt, pv.view(np.int64) if pt == abi.I64 else pv, rng.random(40_000) > 0.05), Column(abi.I64, np.arange(40_000))])
        cfg = _cfg(bt, pt)
        want = orc.hash_join(cfg, build, probe).NumRows()
        assert _count(ctx, cfg, build, probe) == want, (bt, pt, hi_frac)
        assert _count(ctx, cfg, build, probe, packing=OFF) == want


def test_packed_duplicates_255_stay_256_fall_back(ctx):
    rng = np.random.default_rng(2)
    pk = rng.integers(0, 1200, 100_000)
    for dups in (255, 256, 3000):
        bk = np.concatenate([np.full(dups, 2077), np.arange(1000)]).astype(np.int64)  # key 2077: `dups` build rows
        rng.shuffle(bk)
        pk[::7] = 2077
        build, probe = _tables(bk, pk)
        want = int((pk < 1000).sum()) + dups * int((pk == 2077).sum())
        stats = []
        got = G.run_join(ctx, _cfg(), build, probe, chunk_rows=1 << 22, count_only=True, radix=FORCE, packing=FORCE, stats_out=stats)
        assert got == want
        if dups == 255:
            assert stats[0].probe_route == abi.ROUTE_PACKED
        else:
            assert stats[0].probe_route != abi.ROUTE_PACKED  # a cell cannot hold the multiplicity: 64-bit table words keep the join


def test_packed_probe_skew_takes_the_overflow_list(ctx):
    n = 600_000
    build, _ = _tables(np.array([7, 7, 7, 8, 9, 4000], dtype=np.int64), np.zeros(1, dtype=np.int64))
    pk = np.full(n, 7, dtype=np.int64)
    pk[::1000] = 8
    pk[1::1000] = 5000  # outside the range
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, np.arange(n))])
    stats = []
    got = G.run_join(ctx, _cfg(), build, probe, chunk_rows=1 << 22, count_only=True, radix=FORCE, packing=FORCE, stats_out=stats)
    assert got == 3 * (n - 2 * (n // 1000)) + n // 1000
    assert stats[0].probe_route == abi.ROUTE_PACKED and stats[0].radix_overflow_rows > 0


def test_packed_auto_at_1e7_hit_ratio_half(ctx):
    # AUTO: 1e7 unique build keys (a bijection of [0, 1e7) shifted by an offset), probe keys uniform in [0, 2e7): the count
    # is the number of probe keys inside the range — a wrong route cannot produce it by counting rows
    rng = np.random.default_rng(9)
    n = 3 * (4 << 20)  # three full probe batches of the host-push path (probe_batch_rows = 4 Mi)
    off = 5_000_000_000
    bk = off + rng.permutation(n).astype(np.int64)
    pk = off + rng.integers(0, 2 * n, n)
    build, probe = _tables(bk, pk)
    want = int((pk < off + n).sum())
    stats = []
    got = G.run_join(ctx, _cfg(), build, probe, chunk_rows=1 << 24, count_only=True, stats_out=stats)
    assert got == want
    assert stats[0].probe_route == abi.ROUTE_PACKED and stats[0].packed_key_bits == 24 and stats[0].radix_batches == 3
    # sparse keys (range 2^40): AUTO keeps the 64-bit route
    stats = []
    bk2 = bk * 100_003
    got = G.run_join(ctx, _cfg(), Chunk([Column(abi.I64, bk2), build.columns[1]]), Chunk([Column(abi.I64, pk * 100_003), probe.columns[1]]),
                     chunk_rows=1 << 24, count_only=True, stats_out=stats)
    assert got == want and stats[0].probe_route == abi.ROUTE_RADIX_LDS


# ------------------------------------------------------------------ materialising packed route (K4d: pairs through the CSR images)
from tinysql_amd.chunk import StrColumn  # noqa: E402


def _strs(rng, n, maxlen, null_p=0.1):
    return [None if rng.random() < null_p else bytes(rng.integers(97, 123, int(rng.integers(0, maxlen + 1)), dtype=np.uint8)) for _ in range(n)]


def _rows(ctx, cfg, build, probe, chunk_rows=1 << 22, want_route=abi.ROUTE_PACKED, radix=FORCE, packing=FORCE):
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=chunk_rows, pull_rows=4096, radix=radix, packing=packing, stats_out=stats)
    if want_route is not None:
        assert stats[0].probe_route == want_route, (stats[0].probe_route, stats[0].radix_batches)
    return got


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_INNER, 0), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("n_probe", [1, 64, 4097, 60_001])
def test_packed_rows_wide_nullable_vs_oracle(ctx, orc, jt, inner, n_probe):
    # 6-column build side and 4-column probe side with NULLs in keys and payloads, float / double / string payloads, ~3 build rows per
    # key, probe keys on both sides of the build range: none of this fits the 64-bit LDS route's shape, all of it joins on the packed route
    rng = np.random.default_rng(7 * n_probe + jt + inner)
    nb = 5000
    bside = Chunk([Column(abi.F64, rng.random(nb), rng.random(nb) > 0.1), Column(abi.I64, rng.integers(-800, 900, nb), rng.random(nb) > 0.03),
                   Column(abi.I64, rng.integers(-9, 9, nb), rng.random(nb) > 0.2), Column(abi.F32, rng.random(nb).astype(np.float32)),
                   StrColumn(_strs(rng, nb, 12)), Column(abi.U64, rng.integers(0, 1 << 62, nb).astype(np.uint64))])
    pside = Chunk([Column(abi.I64, rng.integers(-1000, 1100, n_probe), rng.random(n_probe) > 0.03), Column(abi.F64, rng.random(n_probe), rng.random(n_probe) > 0.3),
                   StrColumn(_strs(rng, n_probe, 20, 0.2)), Column(abi.I64, np.arange(n_probe))])
    left, right = (pside, bside) if inner == 1 else (bside, pside)
    cfg = H.join_cfg(left.types(), right.types(), [0] if inner == 1 else [1], [1] if inner == 1 else [0], jt, inner)
    want = orc.hash_join(cfg, bside, pside)
    got = _rows(ctx, cfg, bside, pside)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    got = _rows(ctx, cfg, bside, pside, chunk_rows=1024)  # host chunks of tidb_max_chunk_size rows reach the same batch
    assert H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("jt", [abi.JOIN_INNER, abi.JOIN_LEFT_OUTER])
def test_packed_rows_duplicates_overflow_and_misses(ctx, orc, jt):
    # every probe row carries one of three keys (its partition's region overflows: the overflow list's rows come out through
    # k_da_emit_ovf), a key has 200 build rows, and an outer join pads the probe rows whose key is NULL or outside the build range
    rng = np.random.default_rng(31 + jt)
    bk = np.concatenate([np.full(200, 5), np.arange(100, 1100), np.full(3, 40_000)]).astype(np.int64)
    rng.shuffle(bk)
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(len(bk)))])
    n = 70_000
    pk = rng.choice(np.array([5, 5, 5, 101, 40_000, 77, -3, 50_000], dtype=np.int64), n)
    probe = Chunk([Column(abi.I64, pk, rng.random(n) > 0.02), Column(abi.I64, np.arange(n))])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], jt, 1)
    want = orc.hash_join(cfg, build, probe)
    got = _rows(ctx, cfg, build, probe)
    assert got.NumRows() == want.NumRows() > 200 * n // 4 and H.rows_equal_unordered(got, want)


def test_packed_rows_1e7_checksum_vs_count_route(ctx):
    # at scale: 2^23 build rows (a bijection of the key range), 3 x 2^22 probe rows with hit ratio 0.5, a nullable payload (the
    # 64-bit LDS route refuses it): the joined rows' order-independent checksum must equal the direct route's (tsq_join_set_checksum)
    rng = np.random.default_rng(13)
    nb, n = 1 << 23, 3 * (4 << 20)
    build = Chunk([Column(abi.I64, rng.permutation(nb).astype(np.int64)), Column(abi.I64, rng.integers(0, 1 << 40, nb), rng.random(nb) > 0.03)])
    probe = Chunk([Column(abi.I64, rng.integers(0, 2 * nb, n)), Column(abi.I64, np.arange(n))])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], abi.JOIN_INNER, 1)
    c, s, x = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 24, count_only=True, checksum=True, radix=OFF)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 24, pull_rows=1 << 20, stats_out=stats, packing=FORCE)
    assert stats[0].probe_route == abi.ROUTE_PACKED and stats[0].radix_batches == 3
    assert got.NumRows() == c == int((probe.columns[0].data < nb).sum())
    from oracle import binding as orc_b
    assert orc_b.rows_checksum(got) == (s, x)


# ------------------------------------------------------------------ travelling columns (K5f + K4e): 8-byte columns on both sides
def _wide8(rng, n, key_lo, key_hi, ncols, null_key=0.03, null_pay=0.1):
    cols = [Column(abi.I64, rng.integers(key_lo, key_hi, n), rng.random(n) > null_key)]
    for c in range(1, ncols):
        tp = (abi.I64, abi.F64, abi.U64)[c % 3]
        if tp == abi.F64:
            data = rng.random(n)
        elif tp == abi.U64:
            data = rng.integers(0, 1 << 63, n).astype(np.uint64)
        else:
            data = rng.integers(-(1 << 40), 1 << 40, n)
        cols.append(Column(tp, data, (rng.random(n) > null_pay) if c % 2 else None))
    return Chunk(cols)


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_INNER, 0), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("n_probe,np_cols,nb_cols", [(1, 1, 1), (64, 2, 2), (4097, 4, 6), (60_001, 8, 8), (150_003, 3, 2)])
def test_packed_travelling_columns_vs_oracle(ctx, orc, jt, inner, n_probe, np_cols, nb_cols):
    rng = np.random.default_rng(11 * n_probe + jt + inner)
    bside = _wide8(rng, 6000, -900, 1000, nb_cols)      # ~3 build rows per key, NULL keys and NULL payload cells
    pside = _wide8(rng, n_probe, -1100, 1200, np_cols)  # probe keys on both sides of the build range
    left, right = (pside, bside) if inner == 1 else (bside, pside)
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], jt, inner)
    want = orc.hash_join(cfg, bside, pside)
    got = _rows(ctx, cfg, bside, pside)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    got = _rows(ctx, cfg, bside, pside, chunk_rows=1024)
    assert H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("jt", [abi.JOIN_INNER, abi.JOIN_LEFT_OUTER])
def test_packed_travelling_columns_overflow_list_and_misses(ctx, orc, jt):
    # a hot probe key overflows its partition's region: those rows, and the NULL / out-of-range probe rows of an outer join, are the
    # batch's exception rows (pairs + one lane per row); a key with 200 build rows multiplies its probe rows
    rng = np.random.default_rng(41 + jt)
    bk = np.concatenate([np.full(200, 5), np.arange(100, 1100), np.full(3, 40_000)]).astype(np.int64)
    rng.shuffle(bk)
    build = Chunk([Column(abi.I64, bk), Column(abi.F64, rng.random(len(bk)), rng.random(len(bk)) > 0.2)])
    n = 70_000
    pk = rng.choice(np.array([5, 5, 5, 101, 40_000, 77, -3, 50_000], dtype=np.int64), n)
    probe = Chunk([Column(abi.I64, pk, rng.random(n) > 0.02), Column(abi.I64, np.arange(n)), Column(abi.U64, rng.integers(0, 99, n).astype(np.uint64), rng.random(n) > 0.5)])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], jt, 1)
    want = orc.hash_join(cfg, build, probe)
    got = _rows(ctx, cfg, build, probe)
    assert got.NumRows() == want.NumRows() > 200 * n // 4 and H.rows_equal_unordered(got, want)


def test_packed_travelling_columns_auto_1e7_checksum(ctx):
    # AUTO at scale: 2^23 build rows, 3 x 2^22 probe rows (hit ratio 0.5), nullable payloads on both sides, a left outer join:
    # the order-independent checksum of the joined rows equals the direct route's (tsq_join_set_checksum, GEN kernels)
    rng = np.random.default_rng(15)
    nb, n = 1 << 23, 3 * (4 << 20)
    build = Chunk([Column(abi.I64, rng.permutation(nb).astype(np.int64)), Column(abi.I64, rng.integers(0, 1 << 40, nb), rng.random(nb) > 0.03)])
    probe = Chunk([Column(abi.I64, rng.integers(0, 2 * nb, n), rng.random(n) > 0.03), Column(abi.F64, rng.random(n), rng.random(n) > 0.03)])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], abi.JOIN_LEFT_OUTER, 1)
    c, s, x = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 24, count_only=True, checksum=True, radix=OFF)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 24, pull_rows=1 << 20, stats_out=stats)
    assert stats[0].probe_route == abi.ROUTE_PACKED and stats[0].radix_batches == 3
    assert got.NumRows() == c == n
    from oracle import binding as orc_b
    assert orc_b.rows_checksum(got) == (s, x)


# ------------------------------------------------------------------ OtherConditions of an inner join on the travelling-columns route
@pytest.mark.parametrize("inner", [1, 0])
@pytest.mark.parametrize("n_probe", [64, 4097, 90_001])
def test_packed_travelling_columns_other_conditions_vs_oracle(ctx, orc, inner, n_probe):
    # innerJoiner.tryToMatch filters the joined chunk by OtherConditions (joiner.go:351-378): here the batch is materialised first, the
    # conditions run over its rows (NULL cells: an Int-typed NULL drops the row, expression.go:205-279) and the survivors are compacted
    rng = np.random.default_rng(5 * n_probe + inner)
    bside = _wide8(rng, 6000, -900, 1000, 3)   # (k, F64 nullable, U64)
    pside = _wide8(rng, n_probe, -1100, 1200, 4)  # (k, F64 nullable, U64, I64 nullable)
    left, right = (pside, bside) if inner == 1 else (bside, pside)
    nl = len(left.types())
    lf, rf = E.Column(1, abi.F64), E.Column(nl + 1, abi.F64)
    i3 = E.Column(3 if inner == 1 else nl + 3, abi.I64)
    conds = [E.ScalarFunction("lt", lf, rf), E.ScalarFunction("gt", i3, E.Constant(-(1 << 39)))]
    keep = []
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], abi.JOIN_INNER, inner, conds, (), keep)
    want = orc.hash_join(cfg, bside, pside)
    got = _rows(ctx, cfg, bside, pside)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    if n_probe > 4000:
        assert 0 < want.NumRows()
    # a condition nothing passes: an empty result, not an empty batch
    cfg0 = H.join_cfg(left.types(), right.types(), [0], [0], abi.JOIN_INNER, inner, [E.ScalarFunction("lt", lf, E.Constant(-1.0))], (), keep)
    assert _rows(ctx, cfg0, bside, pside).NumRows() == 0


def test_packed_travelling_columns_condition_error_is_the_direct_routes(ctx):
    # an overflow inside a condition: the batch is dropped and goes through the direct route, which reports the error of the
    # first probe row in order (types.ErrOverflow), exactly as without the packed route
    rng = np.random.default_rng(3)
    n = 20_000
    build = Chunk([Column(abi.I64, np.arange(5000)), Column(abi.I64, rng.integers(1, 100, 5000))])
    pv = rng.integers(0, 100, n)
    pv[12_345] = (1 << 63) - 1
    probe = Chunk([Column(abi.I64, rng.integers(0, 5000, n)), Column(abi.I64, pv)])
    cond = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(1, abi.I64), E.Column(3, abi.I64)), E.Constant(-1))]
    keep = []
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], abi.JOIN_INNER, 1, cond, (), keep)
    with pytest.raises(_lib.TsqError) as ei:
        _rows(ctx, cfg, build, probe, want_route=None)
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT
    pv[12_345] = 5  # without the poisoned cell the same join runs on the packed route
    probe = Chunk([Column(abi.I64, probe.columns[0].data), Column(abi.I64, pv)])
    assert _rows(ctx, cfg, build, probe).NumRows() == n


# ------------------------------------------------------------------ bit cells: unique build keys spanning 29..31 bits (COUNT(*) route)
@pytest.mark.parametrize("span_bits,n_build", [(29, 120_000), (30, 50_000), (31, 300_000)])
def test_packed_bit_cells_unique_build_side(ctx, span_bits, n_build):
    # one BIT per cell (k_da_build_bits): 2^b / 8 bytes of images, 4-byte entries.  Probe keys: hits, misses inside the range, keys
    # outside it on both sides, NULLs; a hot probe key overflows its partition's region (the overflow list tests bits in HBM)
    rng = np.random.default_rng(span_bits)
    span = (1 << span_bits) - 5
    base = -(1 << 50) + 99
    bk = base + np.unique(np.concatenate([rng.integers(0, span, n_build), np.array([0, span - 1])]))
    rng.shuffle(bk)
    n = 400_000
    pk = np.where(rng.random(n) < 0.5, bk[rng.integers(0, len(bk), n)], base + rng.integers(-span // 8, span + span // 8, n))
    pk[::5] = bk[3]  # the hot key
    pnn = rng.random(n) > 0.03
    build, probe = _tables(bk, pk, None, pnn)
    want = int(np.isin(pk[pnn], bk).sum())
    stats = []
    got = G.run_join(ctx, _cfg(), build, probe, chunk_rows=1 << 22, count_only=True, radix=FORCE, packing=FORCE, stats_out=stats)
    assert got == want
    assert stats[0].probe_route == abi.ROUTE_PACKED and stats[0].packed_key_bits == span_bits and stats[0].radix_overflow_rows > 0
    assert _count(ctx, _cfg(), build, probe, chunk_rows=4096) == want


def test_packed_bit_cells_need_a_unique_build_side(ctx):
    # one duplicate among keys that span 30 bits: a bit cannot hold a multiplicity, the join keeps 64-bit table words (a materialising
    # join over such a range never takes the packed route: bit cells carry no ranks)
    rng = np.random.default_rng(31)
    bk = np.unique(rng.integers(0, 1 << 30, 60_000))
    bk = np.concatenate([bk, bk[:1]])
    pk = np.concatenate([bk[rng.integers(0, len(bk), 100_000)], rng.integers(0, 1 << 30, 100_000)])
    build, probe = _tables(bk, pk)
    keys, cnts = np.unique(bk, return_counts=True)
    pos = np.searchsorted(keys, pk)
    pos[pos == len(keys)] = 0
    want = int(cnts[pos][keys[pos] == pk].sum())
    stats = []
    got = G.run_join(ctx, _cfg(), build, probe, chunk_rows=1 << 22, count_only=True, radix=FORCE, packing=FORCE, stats_out=stats)
    assert got == want and stats[0].probe_route != abi.ROUTE_PACKED
    rows = G.run_join(ctx, _cfg(), build, probe.slice(0, 5000), chunk_rows=1 << 22, radix=FORCE, packing=FORCE)
    assert rows.NumRows() == int(cnts[pos[:5000]][keys[pos[:5000]] == pk[:5000]].sum())


# ------------------------------------------------------------------ several integer key columns: one composite key column (k_da_compose)
def _mk_side(rng, n, fields, n_pay, null_key=0.03):
    """fields: per key column (type, lo, hi); then n_pay 8-byte payload columns (alternating nullable)"""
    cols = []
    for tp, lo, hi in fields:
        v = rng.integers(lo, hi, n)
        cols.append(Column(tp, v.astype(np.uint64) if tp == abi.U64 else v, rng.random(n) > null_key if null_key else None))
    for c in range(n_pay):
        tp = (abi.F64, abi.I64, abi.U64)[c % 3]
        data = rng.random(n) if tp == abi.F64 else (rng.integers(0, 1 << 40, n).astype(np.uint64) if tp == abi.U64 else rng.integers(-(1 << 40), 1 << 40, n))
        cols.append(Column(tp, data, (rng.random(n) > 0.1) if c % 2 == 0 else None))
    return Chunk(cols)


@pytest.mark.parametrize("nk", [2, 3, 4])
@pytest.mark.parametrize("n_probe", [1, 4097, 120_001])
def test_packed_several_key_columns_count_vs_oracle(ctx, orc, nk, n_probe):
    # COUNT(*) over a join on (k1, .., kn): NULL cells, probe cells outside the build side's fields, duplicates of whole key tuples
    rng = np.random.default_rng(nk * 1000 + n_probe)
    fields = [(abi.I64, -40, 41), (abi.I64, 10**12, 10**12 + 9), (abi.U64, 0, 5), (abi.I64, -3, 4)][:nk]
    wider = [(tp, lo - 3, hi + 3) for tp, lo, hi in fields]
    build = _mk_side(rng, 5000, fields, 1)
    probe = _mk_side(rng, n_probe, wider, 1)
    keys = list(range(nk))
    cfg = H.join_cfg(probe.types(), build.types(), keys, keys, abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe).NumRows()
    assert _count(ctx, cfg, build, probe, want_route=abi.ROUTE_PACKED) == want
    assert _count(ctx, cfg, build, probe, packing=OFF) == want
    assert _count(ctx, cfg, build, probe, chunk_rows=1024) == want


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_INNER, 0), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("n_probe", [64, 60_001])
def test_packed_several_key_columns_rows_vs_oracle(ctx, orc, jt, inner, n_probe):
    # the same on the travelling-columns route: the key columns are ordinary payload there (they travel / are sorted by word)
    rng = np.random.default_rng(17 * n_probe + jt + inner)
    fields = [(abi.I64, -30, 31), (abi.U64, 5, 14), (abi.I64, 0, 3)]
    build = _mk_side(rng, 4000, fields, 2)
    probe = _mk_side(rng, n_probe, [(tp, lo - 2, hi + 2) for tp, lo, hi in fields], 3)
    left, right = (probe, build) if inner == 1 else (build, probe)
    cfg = H.join_cfg(left.types(), right.types(), [0, 1, 2], [0, 1, 2], jt, inner)
    want = orc.hash_join(cfg, build, probe)
    got = _rows(ctx, cfg, build, probe)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_packed_several_key_columns_mixed_signedness_and_too_wide(ctx, orc):
    rng = np.random.default_rng(77)
    # BIGINT against BIGINT UNSIGNED in the second key column: cells >= 2^63 never match (codec.go:219-224)
    n = 6000
    b2 = rng.integers(0, 50, n).astype(np.uint64)
    b2[::9] += np.uint64(1 << 63)
    build = Chunk([Column(abi.I64, rng.integers(0, 60, n)), Column(abi.U64, b2), Column(abi.I64, np.arange(n))])
    p2 = rng.integers(0, 50, 50_000)
    p2[::11] = -5  # the same bits as a huge unsigned cell would need: still no match
    probe = Chunk([Column(abi.I64, rng.integers(0, 60, 50_000)), Column(abi.I64, p2), Column(abi.I64, np.arange(50_000))])
    cfg = H.join_cfg(probe.types(), build.types(), [0, 1], [0, 1], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe).NumRows()
    assert _count(ctx, cfg, build, probe, want_route=abi.ROUTE_PACKED) == want
    # fields that add up to more than 28 bits (32 here): COUNT(*) goes through the composite-key child join (round 4) — off the direct
    # route; the same join with key packing off must agree
    build = Chunk([Column(abi.I64, rng.integers(0, 1 << 20, n)), Column(abi.I64, rng.integers(0, 1 << 12, n))])
    probe = Chunk([Column(abi.I64, build.columns[0].data[rng.integers(0, n, 30_000)]), Column(abi.I64, rng.integers(0, 1 << 12, 30_000))])
    cfg = H.join_cfg(probe.types(), build.types(), [0, 1], [0, 1], abi.JOIN_INNER, 1)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, count_only=True, radix=FORCE, packing=FORCE, stats_out=stats)
    assert got == orc.hash_join(cfg, build, probe).NumRows() and stats[0].probe_route != abi.ROUTE_DIRECT
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, count_only=True, radix=FORCE, packing=OFF, stats_out=stats)
    assert got == orc.hash_join(cfg, build, probe).NumRows()


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("unique", [True, False])
def test_packed_travelling_columns_outer_join_conditions(ctx, orc, jt, inner, unique):
    # leftOuterJoiner / rightOuterJoiner with OtherConditions (joiner.go:220-344): an outer row whose candidates all fail the
    # conditions is padded with NULLs.  With a UNIQUE build side there is one candidate at most: the packed route materialises the
    # batch and un-matches the failed rows (k_outer_unmatch); with duplicates (round 5) the candidates of an outer row are consecutive
    # rows of the batch: failed candidates go, and an outer row that lost them all keeps its first row, padded (k_outer_segments) —
    # also for the rows of a hot probe key, which take the overflow list
    rng = np.random.default_rng(23 + jt + unique)
    nb = 4000
    bk = rng.permutation(6000)[:nb] - 1000 if unique else rng.integers(-900, 1000, nb)
    bside = Chunk([Column(abi.I64, bk, None if unique else rng.random(nb) > 0.03), Column(abi.F64, rng.random(nb), rng.random(nb) > 0.1),
                   Column(abi.I64, rng.integers(0, 100, nb))])
    n = 50_001
    pk = rng.integers(-1200, 5200, n)
    if not unique:
        pk[::3] = bk[11]  # a hot probe key with several build rows: its partition's region overflows
    pside = Chunk([Column(abi.I64, pk, rng.random(n) > 0.03), Column(abi.F64, rng.random(n), rng.random(n) > 0.2),
                   Column(abi.I64, rng.integers(0, 100, n))])
    left, right = (pside, bside) if inner == 1 else (bside, pside)
    nl = 3
    conds = [E.ScalarFunction("lt", E.Column(1, abi.F64), E.Column(nl + 1, abi.F64)), E.ScalarFunction("ne", E.Column(2, abi.I64), E.Column(nl + 2, abi.I64))]
    keep = []
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], jt, inner, conds, (), keep)
    want = orc.hash_join(cfg, bside, pside)
    got = _rows(ctx, cfg, bside, pside, want_route=abi.ROUTE_PACKED)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    if unique:
        assert got.NumRows() == n


# ------------------------------------------------------------------ hot probe keys through BOTH partition kernels, at the tile shapes that differ
# k_da_partition2 instantiated for 4-byte entries failed the bit-cell test with a hot probe key at the end of round 3 and was
# reverted without an explanation.  These cases run both kernels (TSQ_KNOB_DA_PARTITION: 1 = one 1024-thread workgroup per CU, 2 = two of 512)
# over every entry width (2-byte entries; 4-byte entries of a 28-bit byte-cell range; 4-byte entries of bit cells) at the tile shapes whose code
# paths differ: exactly one full tile, a partial last tile, a NULL bitmap (no tile takes the 16-byte-load path), and more tiles than workgroups
# with the hot key in the FIRST tiles only (the workgroup's overflow flag is sticky across its tiles).
T_TILE = 16384


def _hot_case(rng, span_bits, n, nulls, hot_first_only):
    span = (1 << span_bits) - 3
    bk = np.unique(np.concatenate([rng.integers(0, span, 50_000), np.array([0, span - 1])])) + 1000
    rng.shuffle(bk)
    pk = np.where(rng.random(n) < 0.5, bk[rng.integers(0, len(bk), n)], 1000 + rng.integers(-span // 8, span + span // 8, n))
    hot = slice(0, min(n, 40 * T_TILE)) if hot_first_only else slice(0, n)
    sel = np.zeros(n, bool)
    sel[hot] = rng.random(len(pk[hot])) < 0.6
    pk[sel] = bk[7]
    pnn = rng.random(n) > 0.02 if nulls else None
    want = int(np.isin(pk if pnn is None else pk[pnn], bk).sum())
    return bk, pk, pnn, want


@pytest.mark.parametrize("variant", [1, 2], ids=["one_wg_per_cu", "two_wg_per_cu"])
@pytest.mark.parametrize("span_bits", [20, 28, 30], ids=["u16_entries", "u32_byte_cells", "u32_bit_cells"])
@pytest.mark.parametrize("shape", ["one_full_tile", "partial_last_tile", "null_bitmap", "sticky_overflow_flag"])
def test_packed_hot_probe_key_both_partition_kernels(ctx, variant, span_bits, shape):
    rng = np.random.default_rng(span_bits * 10 + variant)
    n = {"one_full_tile": T_TILE, "partial_last_tile": 3 * T_TILE + 1, "null_bitmap": 5 * T_TILE + 100, "sticky_overflow_flag": 600 * T_TILE + 77}[shape]
    bk, pk, pnn, want = _hot_case(rng, span_bits, n, shape == "null_bitmap", shape == "sticky_overflow_flag")
    build, probe = _tables(bk, pk, None, pnn)
    cfg = _cfg()
    cfg.probe_batch_rows = 1 << 24  # ONE device batch: 600 tiles for at most 512 workgroups in the last shape
    with ctx.knobs(DA_PARTITION=variant):
        stats = []
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 24, count_only=True, radix=FORCE, packing=FORCE, stats_out=stats)
    assert stats[0].probe_route == abi.ROUTE_PACKED and stats[0].packed_key_bits == span_bits
    assert got == want, (got, want, stats[0].radix_overflow_rows)
    if shape != "one_full_tile":
        assert stats[0].radix_overflow_rows > 0


# ------------------------------------------------------------------ division-by-zero WARNINGS of join conditions reach the host (tsq_stats)
# The reference evaluates OtherConditions over the joined chunk (joiner.go:351-378 -> VecEvalBool) and every division by zero appends a
# warning to the statement context (expression/errors.go:65-77: NULL result + ErrDivisionByZero warning).  The shim needs the COUNT
# (tsq_stats.div_by_zero_warnings) to call handleDivisionByZeroError that many times.  Expected value: the oracle's VecEvalBool over the
# oracle's condition-less join of the same tables — one evaluation per candidate pair, as in the reference.
@pytest.mark.parametrize("route", ["direct", "packed", "packed_duplicate_build_keys"])
@pytest.mark.parametrize("jt", [abi.JOIN_INNER, abi.JOIN_LEFT_OUTER])
def test_division_by_zero_warnings_of_join_conditions_are_counted(ctx, orc, route, jt):
    rng = np.random.default_rng(77)
    nb, npr = 40_000, 90_000
    bk = rng.permutation(nb).astype(np.int64)  # unique build keys ...
    if route == "packed_duplicate_build_keys":  # ... or ~2 rows per key: every candidate of an outer row is evaluated (and warns) once
        bk = rng.integers(0, nb // 2, nb).astype(np.int64)
        route = "packed"
    bv = rng.integers(0, 4, nb).astype(np.float64)  # a quarter of the divisors is zero (DIV is real-only, builtin_arithmetic.go:435-444)
    pk = rng.integers(-5000, nb + 5000, npr).astype(np.int64)
    pv = rng.integers(-50, 50, npr).astype(np.float64)
    build = Chunk([Column(abi.I64, bk), Column(abi.F64, bv, rng.random(nb) > 0.05)])
    probe = Chunk([Column(abi.I64, pk, rng.random(npr) > 0.03), Column(abi.F64, pv, rng.random(npr) > 0.05)])
    t = [abi.I64, abi.F64]
    keep = []
    # probe.v / build.v > 1  (joined row = probe columns 0..1, build columns 2..3)
    conds = [E.ScalarFunction("gt", E.ScalarFunction("div", E.Column(1, abi.F64), E.Column(3, abi.F64)), E.Constant(1.0))]
    cfg = H.join_cfg(t, t, [0], [0], jt, 1, conds, (), keep)
    plain = H.join_cfg(t, t, [0], [0], abi.JOIN_INNER, 1)
    candidates = orc.hash_join(plain, build, probe)  # every (probe row, build row) pair the conditions are evaluated on
    _, _, want_w = orc.filter_eval(E.compile_list(conds), 1, candidates)
    assert want_w > 1000
    want = orc.hash_join(cfg, build, probe)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, stats_out=stats,
                     radix=FORCE if route == "packed" else abi.RADIX_OFF, packing=FORCE if route == "packed" else abi.RADIX_OFF)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    assert stats[0].probe_route == (abi.ROUTE_PACKED if route == "packed" else abi.ROUTE_DIRECT)
    assert stats[0].div_by_zero_warnings == want_w, (stats[0].div_by_zero_warnings, want_w)


# ------------------------------------------------------------------ selected[] on the packed routes (round 4)
# tsq_join_probe_push(selected): an externally evaluated outer-side filter — a row with selected == 0 goes to onMissMatch (join.go:344-345):
# dropped by an inner join, NULL-padded by an outer join.  The packed kernels treat such a row like a row with a NULL key; the device
# pipeline uses it to hand a Selection's flags to the join without compacting the chunk (gpu_pipeline.GpuSelectionExec(compact=False)).
@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("route", ["travelling_columns", "pairs"])
def test_packed_routes_take_a_selected_vector(ctx, orc, jt, inner, route):
    rng = np.random.default_rng(91 + jt)
    nb, npr = 30_000, 70_001
    bk = rng.integers(0, 25_000, nb).astype(np.int64)  # duplicates
    pk = rng.integers(-3000, 28_000, npr).astype(np.int64)
    build = Chunk([Column(abi.I64, bk, rng.random(nb) > 0.03), Column(abi.I64, rng.integers(-9, 9, nb), rng.random(nb) > 0.1)])
    probe = Chunk([Column(abi.I64, pk, rng.random(npr) > 0.03), Column(abi.F64, rng.random(npr)), Column(abi.I64, np.arange(npr))])
    sel = (rng.random(npr) > 0.4).astype(np.uint8)
    left, right = (probe, build) if inner == 1 else (build, probe)
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], jt, inner)
    want = orc.hash_join(cfg, build, probe, selected=sel)
    with ctx.knobs(PACKED_EMIT_PAIRS=1, DA_PAIRS_BELOW_PERMILLE=(1001 if route == "pairs" else 0)):
        stats = []
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, selected=sel, stats_out=stats, radix=FORCE, packing=FORCE)
    assert stats[0].probe_route == abi.ROUTE_PACKED
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    if jt == abi.JOIN_INNER:
        c = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, selected=sel, count_only=True, radix=FORCE, packing=FORCE)
        assert c == want.NumRows()


# ------------------------------------------------------------------ pairs route on BIT cells: a unique build side with a 28..30-bit key range (round 4)
# The primary-key side of a PK-FK join is unique and its key range is often wider than the 27 bits byte cells + 2-byte entries take
# (TPC-H SF 100: 1.5e8 order keys).  One bit per cell + popcount ranks give (probe row, build row) pairs with 4-byte entries
# (tsq_dajoin.h: k_da_build_rows_bits, k_da_emit_pairs_bits); the columns follow through the pairs.
@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("span_bits", [28, 29, 30])
def test_packed_bit_cell_pairs_unique_wide_build_side(ctx, orc, jt, inner, span_bits):
    rng = np.random.default_rng(span_bits * 3 + jt)
    span = (1 << span_bits) - 7
    base = -(1 << 40) + 11
    bk = base + np.unique(np.concatenate([rng.integers(0, span, 60_000), np.array([0, span - 1])]))
    rng.shuffle(bk)
    nb, n = len(bk), 150_001
    pk = np.where(rng.random(n) < 0.5, bk[rng.integers(0, nb, n)], base + rng.integers(-span // 8, span + span // 8, n))
    pk[::7] = bk[5]  # a hot probe key: its partition's region overflows (the overflow list emits through the images in HBM)
    build = Chunk([Column(abi.I64, bk), Column(abi.F64, rng.random(nb), rng.random(nb) > 0.1), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, pk, rng.random(n) > 0.03), Column(abi.I64, rng.integers(-9, 9, n), rng.random(n) > 0.1)])
    sel = (rng.random(n) > 0.2).astype(np.uint8)
    left, right = (probe, build) if inner == 1 else (build, probe)
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], jt, inner)
    for selected in (None, sel):
        want = orc.hash_join(cfg, build, probe, selected=selected)
        stats = []
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, selected=selected, stats_out=stats, radix=FORCE, packing=FORCE)
        assert stats[0].probe_route == abi.ROUTE_PACKED and stats[0].packed_key_bits == span_bits and stats[0].radix_overflow_rows > 0
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_packed_bit_cell_pairs_need_a_unique_build_side(ctx, orc):
    # one duplicate among keys that span 28 bits: 4-byte entries against byte cells have no materialising route — the direct route keeps the join
    rng = np.random.default_rng(8)
    bk = np.unique(rng.integers(0, (1 << 28) - 1, 50_000))
    bk = np.concatenate([bk, bk[:1], np.array([(1 << 28) - 2])])
    pk = np.concatenate([bk[rng.integers(0, len(bk), 40_000)], rng.integers(0, 1 << 28, 40_000)])
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(len(bk)))])
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, np.arange(len(pk)))])
    t = [abi.I64, abi.I64]
    cfg = H.join_cfg(t, t, [0], [0], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, stats_out=stats, radix=FORCE, packing=FORCE)
    assert stats[0].probe_route != abi.ROUTE_PACKED  # (the 64-bit LDS route or the direct one)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


# ------------------------------------------------------------------ several integer key columns of 29..63 bits: COUNT(*) through a composite-key child join
@pytest.mark.parametrize("shape", ["two_wide", "three_mixed", "four"])
@pytest.mark.parametrize("n_probe", [1, 4097, 150_001])
def test_wide_several_key_columns_count_vs_oracle(ctx, orc, shape, n_probe):
    # the fields of the key columns add up to more than the packed route's 28 bits: the composite (exact up to 63 bits) is the ONE key of a
    # child join, which takes a single-key route (tsq_join.hip: wide_prepare).  NULL cells, probe cells outside the fields, duplicate tuples,
    # BIGINT UNSIGNED against BIGINT.
    rng = np.random.default_rng(len(shape) * 100 + n_probe)
    fields = {"two_wide": [(abi.I64, -(1 << 21), 1 << 21), (abi.I64, 10**12, 10**12 + (1 << 20))],
              "three_mixed": [(abi.I64, 0, 1 << 24), (abi.U64, 0, 3000), (abi.I64, -100, 100)],
              "four": [(abi.I64, 0, 1 << 14), (abi.I64, 0, 1 << 14), (abi.I64, -(1 << 13), 1 << 13), (abi.I64, 5, 9)]}[shape]
    nk = len(fields)
    build = _mk_side(rng, 40_000, fields, 1)
    # probe tuples: half drawn from the build side's tuples (so that the join is not empty), half random around the fields
    wider = [(tp, lo - 3 if tp != abi.U64 else lo, hi + 3) for tp, lo, hi in fields]
    probe = _mk_side(rng, n_probe, wider, 1)
    take = rng.integers(0, build.NumRows(), n_probe)
    use = rng.random(n_probe) < 0.5
    for k in range(nk):
        bc, pc = build.columns[k], probe.columns[k]
        data = np.where(use, bc.data[take], pc.data)
        nn = np.where(use, bc.notnull[take] if bc.notnull is not None else True, pc.notnull if pc.notnull is not None else True)
        probe.columns[k] = Column(pc.tp, data.astype(pc.data.dtype), nn.astype(bool))
    keys = list(range(nk))
    cfg = H.join_cfg(probe.types(), build.types(), keys, keys, abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe).NumRows()
    if n_probe > 1:
        assert want > 0
    stats_route = _count(ctx, cfg, build, probe)
    assert stats_route == want
    assert _count(ctx, cfg, build, probe, packing=OFF) == want
    assert _count(ctx, cfg, build, probe, chunk_rows=1024) == want


# ------------------------------------------------------------------ outer-side filters on the packed routes (round 5)
# join.go:328-345: the outer side's filter is evaluated over the probe chunk (VectorizedFilter -> selected[]); a row that fails it goes
# to onMissMatch — NULL-padded — without touching the table.  The library evaluates the filter into flags (k_outer_filter_flags) and
# the packed kernels take them like an external selected[] vector; before round 5 every join with an outer filter took the direct route.
@pytest.mark.parametrize("jt,inner", [(abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("route", ["travelling_columns", "pairs"])
@pytest.mark.parametrize("with_selected", [False, True])
def test_packed_routes_take_the_outer_side_filter(ctx, orc, jt, inner, route, with_selected):
    rng = np.random.default_rng(131 + jt + with_selected)
    nb, npr = 30_000, 70_001
    bk = rng.integers(0, 25_000, nb).astype(np.int64)  # duplicates
    pk = rng.integers(-3000, 28_000, npr).astype(np.int64)
    build = Chunk([Column(abi.I64, bk, rng.random(nb) > 0.03), Column(abi.I64, rng.integers(-9, 9, nb), rng.random(nb) > 0.1)])
    probe = Chunk([Column(abi.I64, pk, rng.random(npr) > 0.03), Column(abi.F64, rng.random(npr), rng.random(npr) > 0.1), Column(abi.F64, rng.integers(0, 4, npr).astype(np.float64))])
    sel = (rng.random(npr) > 0.3).astype(np.uint8) if with_selected else None
    left, right = (probe, build) if inner == 1 else (build, probe)
    # probe.f > 0.4 AND 10.0 / probe.c > 3.0: NULL cells and divisions by zero fail the filter (VecEvalBool: NULL is false) and warn
    filters = [E.ScalarFunction("gt", E.Column(1, abi.F64), E.Constant(0.4)),
               E.ScalarFunction("gt", E.ScalarFunction("div", E.Constant(10.0), E.Column(2, abi.F64)), E.Constant(3.0))]
    keep = []
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], jt, inner, (), filters, keep)
    want = orc.hash_join(cfg, build, probe, selected=sel)
    with ctx.knobs(PACKED_EMIT_PAIRS=1, DA_PAIRS_BELOW_PERMILLE=(1001 if route == "pairs" else 0)):
        stats = []
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, selected=sel, stats_out=stats, radix=FORCE, packing=FORCE)
    assert stats[0].probe_route == abi.ROUTE_PACKED
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    # the same warnings as the direct route counts
    stats_d = []
    got_d = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, selected=sel, stats_out=stats_d, radix=abi.RADIX_OFF, packing=abi.RADIX_OFF)
    assert stats_d[0].probe_route == abi.ROUTE_DIRECT and got_d.NumRows() == want.NumRows()
    assert stats[0].div_by_zero_warnings == stats_d[0].div_by_zero_warnings > 0


def test_an_outer_side_filter_that_raises_an_error_is_the_direct_routes(ctx, orc):
    # BIGINT overflow in the filter: the error the statement reports depends on the row order -> the batch is the direct route's
    rng = np.random.default_rng(7)
    nb, npr = 5000, 20_000
    build = Chunk([Column(abi.I64, rng.permutation(nb)), Column(abi.I64, np.arange(nb))])
    pv = rng.integers(0, 100, npr)
    pv[777] = (1 << 63) - 1
    probe = Chunk([Column(abi.I64, rng.integers(0, nb, npr)), Column(abi.I64, pv)])
    filters = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(1, abi.I64), E.Constant(1)), E.Constant(10))]
    keep = []
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], abi.JOIN_LEFT_OUTER, 1, (), filters, keep)
    with pytest.raises(Exception) as ei:
        G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, radix=FORCE, packing=FORCE)
    assert "overflow" in str(ei.value).lower() or "range" in str(ei.value).lower()


@pytest.mark.parametrize("nk,wide", [(2, False), (3, False), (4, False), (2, True)])
def test_packed_several_key_columns_composed_in_the_partition_kernel(ctx, orc, nk, wide):
    """round 6: COUNT(*) over several integer key columns WITHOUT NULL bitmaps — k_da_partition2<.., MK> reads the columns itself and makes
    the composite in registers (no k_da_compose, no composite column in HBM).  Full tiles and a partly filled last one, probe cells outside
    the build side's fields, BIGINT cells against BIGINT UNSIGNED ones (cells >= 2^63 never match), 2-byte and 4-byte entries (fields of
    27 bits); the same counts from the oracle, from the kernel that reads a composed column (knob DA_PARTITION = 1) and with packing off,
    and one kernel launch fewer per batch."""
    rng = np.random.default_rng(nk * 7 + wide)
    if wide:
        fields = [(abi.I64, -5000, 5000), (abi.I64, 10**12, 10**12 + 9000)]  # 14 + 14 bits: entries of 17 bits
        nb = 200_000
    else:
        fields = [(abi.I64, -40, 41), (abi.I64, 10**12, 10**12 + 9), (abi.U64, 0, 5), (abi.I64, -3, 4)][:nk]
        nb = 5000
    build = _mk_side(rng, nb, fields, 1, null_key=0.03)
    n_probe = 16384 * 3 + 77
    probe = _mk_side(rng, n_probe, [(abi.I64 if tp == abi.U64 else tp, lo - 3, hi + 3) for tp, lo, hi in fields], 1, null_key=0)
    keys = list(range(nk))
    cfg = H.join_cfg(probe.types(), build.types(), keys, keys, abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe).NumRows()
    assert want > 0
    launches = {}
    for variant in (0, 1):
        stats = []
        with ctx.knobs(DA_PARTITION=variant):
            got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, count_only=True, radix=FORCE, packing=FORCE, stats_out=stats)
        assert stats[0].probe_route == abi.ROUTE_PACKED and got == want, (variant, got, want)
        launches[variant] = stats[0].kernel_launches
    assert launches[0] == launches[1] - 1, launches  # (no k_da_compose)
    assert _count(ctx, cfg, build, probe, packing=OFF) == want
    # NULL key cells on the probe side: the batch is composed as before (a bitmap is not read by the fused kernel), same count
    probe2 = _mk_side(rng, n_probe, [(abi.I64 if tp == abi.U64 else tp, lo - 3, hi + 3) for tp, lo, hi in fields], 1, null_key=0.05)
    assert _count(ctx, cfg, build, probe2, want_route=abi.ROUTE_PACKED) == orc.hash_join(cfg, build, probe2).NumRows()
