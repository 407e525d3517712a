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
