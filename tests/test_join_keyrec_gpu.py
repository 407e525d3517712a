"""GPU parity: the KEY-RECORD route (csrc/tsq_keyrec.h, round 5) — COUNT(*) of an inner join on several key columns / string keys,
partitioned by a mix of a 32-byte record of the key cells and matched in LDS — against the oracle's HashJoinExec restatement (count of
its joined rows) and against the direct route.  The route is FORCED and asserted through tsq_stats.probe_route;
key equality is the reference's codec.EqualChunkRow (util/codec/codec.go:363-382): same flag, same bytes, cell by cell — a string
never equals a number, "a" never equals "a\\0", an UNSIGNED cell above 2^63 never equals a negative BIGINT, float32 1.0 equals double 1.0.
The key shape of the reference's own join benchmark, keyIdx {0, 1} = (bigint, varstring) (executor/benchmark_test.go:352-360), is the
first case."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column, StrColumn

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu
FORCE, OFF = abi.RADIX_FORCE, abi.RADIX_OFF


def _count(ctx, cfg, build, probe, radix=FORCE, chunk_rows=1 << 22, knobs=None):
    stats = []
    with ctx.knobs(**(knobs or {})):
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=chunk_rows, count_only=True, radix=radix, stats_out=stats)
    return got, stats[0]


def _strs(rng, n, pool, null_frac=0.03, max_len=12):
    words = [bytes(rng.integers(97, 123, int(rng.integers(0, max_len + 1)), dtype=np.uint8)) for _ in range(pool)]
    ids = rng.integers(0, pool, n)
    return StrColumn([None if rng.random() < null_frac else words[i] for i in ids.tolist()])


def test_bigint_and_varstring_keys_the_reference_benchmark_shape(ctx, orc):
    rng = np.random.default_rng(101)
    nb, npr = 60_000, 90_000
    build = Chunk([Column(abi.I64, rng.integers(0, 300, nb), rng.random(nb) > 0.02), _strs(rng, nb, 400, max_len=16), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, rng.integers(0, 330, npr), rng.random(npr) > 0.02), _strs(rng, npr, 440, max_len=16), Column(abi.F64, rng.random(npr))])
    # the two sides draw their strings from different pools: equal words only by chance of the generator -> reseed the probe pool from the build's
    words = [w for w in build.columns[1].values() if w is not None]
    pw = [None if rng.random() < 0.03 else (words[int(i)] if rng.random() < 0.7 else b"zz" + words[int(i)][:10]) for i in rng.integers(0, len(words), npr)]
    probe = Chunk([probe.columns[0], StrColumn(pw), probe.columns[2]])
    cfg = H.join_cfg(probe.types(), build.types(), [0, 1], [0, 1], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe).NumRows()
    got, st = _count(ctx, cfg, build, probe)
    assert st.probe_route == abi.ROUTE_KEYREC and got == want > 1000
    # the direct route says the same (knob KEYREC = 0)
    got0, st0 = _count(ctx, cfg, build, probe, knobs={"KEYREC": 0})
    assert st0.probe_route == abi.ROUTE_DIRECT and got0 == want
    # ... and so do several probe batches through one handle (the build side's records are made once)
    cfg3 = H.join_cfg(probe.types(), build.types(), [0, 1], [0, 1], abi.JOIN_INNER, 1, probe_batch_rows=25_000)
    got3, st3 = _count(ctx, cfg3, build, probe, chunk_rows=5_000)
    assert st3.probe_route == abi.ROUTE_KEYREC and st3.radix_batches >= 3 and got3 == want


@pytest.mark.parametrize("shape", ["string", "string,string", "u64,i64", "f32,f64,string"])
def test_key_shapes_and_the_flag_rules_of_EqualChunkRow(ctx, orc, shape):
    rng = np.random.default_rng(len(shape))
    nb, npr = (4_000, 6_000) if shape.startswith("string") else (20_000, 30_000)  # (nine distinct words: the oracle materialises nb x npr / 9 joined rows)

    def col(kind, n, side):
        if kind == "string":
            base = [b"", b"a", b"a\x00", b"ab", b"abcdefgh", b"abcdefghi", b"abcdefgX", b"k" * 12, None]
            return StrColumn([base[i] for i in rng.integers(0, len(base), n)])
        if kind == "u64":  # the build side's cells above 2^63 (flag 9) must not meet the probe side's negative BIGINTs with the same bits (flag 8)
            v = rng.integers(0, 50, n).astype(np.uint64)
            v[rng.random(n) < 0.3] += np.uint64(1 << 63)
            return Column(abi.U64, v)
        if kind == "i64":
            return Column(abi.I64, rng.integers(-5, 5, n), rng.random(n) > 0.05)
        if kind == "f32":
            return Column(abi.F32, rng.integers(0, 6, n).astype(np.float32) * 0.5)
        return Column(abi.F64, rng.integers(0, 6, n).astype(np.float64) * 0.5)
    kinds = shape.split(",")
    build = Chunk([col(k, nb, 0) for k in kinds] + [Column(abi.I64, np.arange(nb))])
    if shape == "u64,i64":  # probe: the SAME bit patterns as BIGINT (signed) in the first column -> only cells below 2^63 may match
        pv = build.columns[0].data[rng.integers(0, nb, npr)].view(np.int64)
        probe = Chunk([Column(abi.I64, pv), col("i64", npr, 1), Column(abi.I64, np.arange(npr))])
    elif shape == "f32,f64,string":  # float32 1.5 joins double 1.5 (codec.go:288-289: float32 is widened before it is hashed / compared)
        probe = Chunk([col("f64", npr, 1), col("f32", npr, 1), col("string", npr, 1), Column(abi.I64, np.arange(npr))])
    else:
        probe = Chunk([col(k, npr, 1) for k in kinds] + [Column(abi.I64, np.arange(npr))])
    keys = list(range(len(kinds)))
    cfg = H.join_cfg(probe.types(), build.types(), keys, keys, abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe).NumRows()
    got, st = _count(ctx, cfg, build, probe, knobs={"DA_MIN_BUILD_ROWS": 1 << 40})  # (keep the integer shapes away from the packed route: this test is about the records)
    # few distinct keys with thousands of rows each: a partition may not fit LDS -> another route (the direct one for string shapes), same count
    if "string" in shape:
        assert st.probe_route in (abi.ROUTE_KEYREC, abi.ROUTE_DIRECT)
    assert got == want > 0
    got0, _ = _count(ctx, cfg, build, probe, knobs={"DA_MIN_BUILD_ROWS": 1 << 40, "KEYREC": 0})
    assert got0 == want


def test_records_that_do_not_fit_and_partitions_that_do_not_fit(ctx, orc):
    rng = np.random.default_rng(7)
    nb, npr = 30_000, 40_000
    ids_b, ids_p = rng.integers(0, 5000, nb), rng.integers(0, 6000, npr)
    short = lambda i: b"k%05d" % i  # noqa: E731
    # (1) a PROBE row whose cells need more than 32 bytes cannot equal any build row: dropped, the route stays
    build = Chunk([Column(abi.I64, ids_b % 7), StrColumn([short(i) for i in ids_b.tolist()]), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, ids_p % 7), StrColumn([short(i) if i % 5 else short(i) * 8 for i in ids_p.tolist()]), Column(abi.I64, np.arange(npr))])
    cfg = H.join_cfg(probe.types(), build.types(), [0, 1], [0, 1], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe).NumRows()
    got, st = _count(ctx, cfg, build, probe)
    assert st.probe_route == abi.ROUTE_KEYREC and got == want > 0
    assert st.keyrec_digests == 0
    # (2) a BUILD row that does not fit (round 6): the string cells enter the records as (length, digest), every candidate is compared byte for byte
    build2 = Chunk([build.columns[0], StrColumn([short(i) if i % 1000 else short(i) * 8 for i in ids_b.tolist()]), build.columns[2]])
    want2 = orc.hash_join(cfg, build2, probe).NumRows()
    got2, st2 = _count(ctx, cfg, build2, probe)
    assert st2.probe_route == abi.ROUTE_KEYREC and st2.keyrec_digests == 1 and got2 == want2 > want
    # (3) one key with 18 000 build rows: its partition exceeds the index in LDS (12 288 records) -> direct route, exact
    hot = np.where(rng.random(nb) < 0.6, 4242, ids_b)
    build3 = Chunk([Column(abi.I64, hot % 7), StrColumn([short(i) for i in hot.tolist()]), build.columns[2]])
    want3 = orc.hash_join(cfg, build3, probe).NumRows()
    got3, st3 = _count(ctx, cfg, build3, probe)
    assert st3.probe_route == abi.ROUTE_DIRECT and got3 == want3
    # (4) empty sides and all-NULL keys
    none = Chunk([Column(abi.I64, np.zeros(100, np.int64), np.zeros(100, bool)), StrColumn([b"x"] * 100), Column(abi.I64, np.arange(100))])
    assert _count(ctx, cfg, none, probe)[0] == 0 and _count(ctx, cfg, build, none)[0] == 0


def test_auto_takes_the_route_at_scale_and_counts_like_numpy(ctx):
    """1e6 x 1.5e6 rows on (bigint, 16-byte varstring), AUTO (no forcing): every build key (k, 's%015d' % k) once, probe keys uniform over twice
    the build domain -> the joined rows are the probe rows with k < N_b (numpy).  The 1e7 x 1e7 form is bench.py's
    two_key_bigint_string_count."""
    rng = np.random.default_rng(5)
    nb, npr = 1_000_000, 1_500_000

    def table(keys):
        return Chunk([Column(abi.I64, keys), StrColumn([b"s%015d" % k for k in keys.tolist()]), Column(abi.I64, np.arange(len(keys)))])
    bk = rng.permutation(nb).astype(np.int64)
    pk = rng.integers(0, 2 * nb, npr)
    build, probe = table(bk), table(pk)
    cfg = H.join_cfg(probe.types(), build.types(), [0, 1], [0, 1], abi.JOIN_INNER, 1)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, count_only=True, stats_out=stats)
    assert stats[0].probe_route == abi.ROUTE_KEYREC
    assert got == int(np.count_nonzero(pk < nb))


def test_materialising_join_on_bigint_and_varstring_keys(ctx, orc):
    """the same records, joined ROWS: (probe row, build row) pairs out of the LDS index, then the usual column gather — the reference's
    BenchmarkHashJoinExec materialises its joined chunks too (benchmark_test.go:352-360).  String keys, string payload, NULLs, duplicate
    build keys; two probe batches through one handle."""
    rng = np.random.default_rng(202)
    nb, npr = 8_000, 20_000
    words = [b"w%03d" % i for i in range(120)] + [b"", b"a", b"a\x00"]
    bw = [None if rng.random() < 0.03 else words[int(i)] for i in rng.integers(0, len(words), nb)]
    pw = [None if rng.random() < 0.03 else words[int(i)] for i in rng.integers(0, len(words), npr)]
    build = Chunk([Column(abi.I64, rng.integers(0, 40, nb), rng.random(nb) > 0.02), StrColumn(bw), Column(abi.F64, rng.random(nb), rng.random(nb) > 0.1),
                   StrColumn([None if i % 11 == 0 else b"pay%d" % i for i in range(nb)])])
    probe = Chunk([Column(abi.I64, rng.integers(0, 44, npr), rng.random(npr) > 0.02), StrColumn(pw), Column(abi.I64, np.arange(npr))])
    cfg = H.join_cfg(probe.types(), build.types(), [0, 1], [0, 1], abi.JOIN_INNER, 1, probe_batch_rows=12_000)
    want = orc.hash_join(cfg, build, probe)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=4096, pull_rows=4096, radix=FORCE, stats_out=stats)
    assert stats[0].probe_route == abi.ROUTE_KEYREC and stats[0].radix_batches >= 2
    assert got.NumRows() == want.NumRows() > 10_000 and H.rows_equal_unordered(got, want)
    # the build side as the LEFT child (inner_child 0): output = left || right = build || probe columns
    cfg0 = H.join_cfg(build.types(), probe.types(), [0, 1], [0, 1], abi.JOIN_INNER, 0)
    want0 = orc.hash_join(cfg0, build, probe)
    stats = []
    got0 = G.run_join(ctx, cfg0, build, probe, chunk_rows=1 << 20, radix=FORCE, stats_out=stats)
    assert stats[0].probe_route == abi.ROUTE_KEYREC and H.rows_equal_unordered(got0, want0)


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("with_selected,with_filter", [(False, False), (True, False), (False, True), (True, True)])
def test_outer_joins_on_key_records(ctx, orc, jt, inner, with_selected, with_filter):
    """leftOuterJoiner / rightOuterJoiner (joiner.go:220-344) on (bigint, varstring) keys: an outer row without a joined build row comes out
    once, NULL-padded — a probe record that met no equal build record (a pair with the MISS build row), and the outer rows that have no key
    record at all: a NULL key cell, cells that do not fit 32 bytes, selected == 0 (join.go:344), an outer-side filter that said no
    (evaluated into flags by the library, k_outer_filter_flags)."""
    rng = np.random.default_rng(303 + jt + 2 * with_selected + with_filter)
    nb, npr = 9_000, 30_001
    words = [b"w%03d" % i for i in range(150)] + [b"", b"a", b"a\x00", b"L" * 40]
    bw = [None if rng.random() < 0.03 else words[int(i)] for i in rng.integers(0, len(words), nb)]
    pw = [None if rng.random() < 0.03 else words[int(i)] for i in rng.integers(0, len(words), npr)]
    build = Chunk([Column(abi.I64, rng.integers(0, 40, nb), rng.random(nb) > 0.02), StrColumn(bw), Column(abi.F64, rng.random(nb), rng.random(nb) > 0.1)])
    probe = Chunk([Column(abi.I64, rng.integers(0, 50, npr), rng.random(npr) > 0.02), StrColumn(pw), Column(abi.F64, rng.random(npr), rng.random(npr) > 0.1)])
    # (a 40-byte word on the BUILD side would switch the route off: the build side keeps short words only)
    build = Chunk([build.columns[0], StrColumn([None if w is not None and len(w) > 30 else w for w in bw]), build.columns[2]])
    sel = (rng.random(npr) > 0.3).astype(np.uint8) if with_selected else None
    left, right = (probe, build) if inner == 1 else (build, probe)
    from tinysql_amd import expression as E
    filters = [E.ScalarFunction("gt", E.Column(2, abi.F64), E.Constant(0.25))] if with_filter else ()
    keep = []
    cfg = H.join_cfg(left.types(), right.types(), [0, 1], [0, 1], jt, inner, (), filters, keep, probe_batch_rows=16_000)
    want = orc.hash_join(cfg, build, probe, selected=sel)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 20, pull_rows=4096, selected=sel, radix=FORCE, stats_out=stats)
    assert stats[0].probe_route == abi.ROUTE_KEYREC, stats[0].probe_route
    assert got.NumRows() == want.NumRows() >= npr and H.rows_equal_unordered(got, want)


# ------------------------------------------------------------------ the reference's own join vectors on the key-record route
# tests/golden/join_cases.json (transcribed from executor/join_test.go) joins on ONE integer column.  Joining on (k, k') with k' a copy
# of k appended to both sides is the same join — the same rows match, a NULL k is a NULL k' — and its key no longer packs into one
# word, so with the radix routes FORCED the join takes the key records (joins with OtherConditions keep the direct route).
@pytest.mark.parametrize("case", H.golden("join_cases.json"), ids=lambda c: c["ref"][:48])
def test_golden_join_cases_on_two_key_columns(ctx, case):
    from tinysql_amd.chunk import concat
    keep = []
    cfg1, left, right, _, _, conds, filt = H.lower_join_case(case, keep)
    lk, rk = case["left_keys"][0], case["right_keys"][0]

    def widen(chk, kc):
        c = chk.columns[kc]
        return Chunk(list(chk.columns) + [Column(c.tp, c.data.copy(), None if c.notnull is None else c.notnull.copy())])
    left2, right2 = widen(left, lk), widen(right, rk)
    nl, nr = len(left.columns), len(right.columns)
    # conditions address the joined row left || right: the right side's columns moved one place to the right
    def shift(e):
        if isinstance(e, E.Column):
            return E.Column(e.index + 1, e.tp) if e.index >= nl else e
        if isinstance(e, E.ScalarFunction):
            return E.ScalarFunction(e.name, *[shift(a) for a in e.args])
        return e
    inner = case["inner_child"]
    cfg = H.join_cfg(left2.types(), right2.types(), [lk, nl], [rk, nr], H.JOIN_TYPES[case["type"]], inner, [shift(c) for c in conds], filt, keep)
    build, probe = (right2, left2) if inner == 1 else (left2, right2)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, radix=FORCE, stats_out=stats)
    want = []
    for row in case["expect"]:
        l, r = list(row[:nl]), list(row[nl:])
        want.append(tuple(l + [l[lk]] + r + [r[rk]]))
    assert H.rows_equal_unordered(got, want), case["ref"]
    if not conds and (len(case["left"]) and len(case["right"])):
        assert stats[0].probe_route == abi.ROUTE_KEYREC, (case["ref"], stats[0].probe_route)


# ------------------------------------------------------------------ long string keys (round 6): digest records + byte-for-byte verification
def _long_words(rng, pool, lo=40, hi=5000):
    # lengths 40..5000 (most of them short, one in ten long); families of words that share their length AND all bytes but the last one
    words = []
    while len(words) < pool:
        n = int(rng.integers(lo, 200)) if rng.random() < 0.9 else int(rng.integers(200, hi + 1))
        w = bytes(rng.integers(97, 123, n, dtype=np.uint8))
        words.append(w)
        if rng.random() < 0.2:
            words.append(w[:-1] + bytes([(w[-1] - 97 + 1) % 26 + 97]))  # the same length, the same first n - 1 bytes
    return words[:pool]


def test_long_string_keys_1e6_rows_vs_oracle(ctx, orc):
    # the reference's benchmark keys on a 5 KiB varstring (executor/benchmark_test.go:328-360): cells that do not fit a 32-byte record enter it
    # as (length, digest) and every candidate match is compared byte for byte (codec.EqualChunkRow, codec.go:363-382).  1e6 probe rows x
    # 2e5 build rows, keys of 40..5000 bytes, (bigint, varstring) and varstring alone; COUNT(*) and the joined rows' checksum
    rng = np.random.default_rng(1001)
    nb, npr, pool = 200_000, 1_000_000, 150_000
    words = _long_words(rng, pool)
    bw = [None if rng.random() < 0.02 else words[i] for i in rng.integers(0, pool, nb).tolist()]
    pid = rng.integers(0, int(pool * 1.3), npr)
    extra = [b"q" + w for w in words[:int(pool * 0.3)]]  # probe-only words: 23 % of the probe rows find no build row
    allw = words + extra
    pw = [None if rng.random() < 0.02 else allw[i] for i in pid.tolist()]
    build = Chunk([Column(abi.I64, rng.integers(0, 3, nb)), StrColumn(bw), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, rng.integers(0, 3, npr)), StrColumn(pw), Column(abi.F64, rng.random(npr))])
    for keys in ([0, 1], [1]):
        cfg = H.join_cfg(probe.types(), build.types(), keys, keys, abi.JOIN_INNER, 1)
        want = orc.hash_join(cfg, build, probe)
        got, st = _count(ctx, cfg, build, probe)
        assert st.probe_route == abi.ROUTE_KEYREC and st.keyrec_digests == 1 and got == want.NumRows() > 100_000, (st.probe_route, got, want.NumRows())
        got0, st0 = _count(ctx, cfg, build, probe, radix=OFF)  # the direct route says the same
        assert st0.probe_route == abi.ROUTE_DIRECT and got0 == got
        stats = []
        rows = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 16, radix=FORCE, stats_out=stats)  # materialised on the key-record route
        assert stats[0].probe_route == abi.ROUTE_KEYREC and stats[0].keyrec_digests == 1 and rows.NumRows() == want.NumRows()
        # the fixed-width cells through the oracle's row checksum (the build row's id is one of them); the string cells: a joined row
        # carries the same word on both sides, and it is the word of that build row
        fixed = lambda ch: Chunk([c for c in ch.columns if not isinstance(c, StrColumn)])  # noqa: E731
        assert orc.rows_checksum(fixed(rows)) == orc.rows_checksum(fixed(want))
        gp, gb = rows.columns[1].values(), rows.columns[4].values()
        assert gp == gb and gb == [bw[i] for i in rows.columns[5].data.tolist()]


@pytest.mark.parametrize("jt", [abi.JOIN_INNER, abi.JOIN_LEFT_OUTER])
def test_long_string_keys_rows_small(ctx, orc, jt):
    # row-for-row at a size Python compares: equal-length words that differ in their last byte only, NULL keys, the empty string, an
    # outer join's padded rows; several probe batches against one set of build records
    rng = np.random.default_rng(77 + jt)
    words = _long_words(rng, 300, lo=33, hi=900) + [b"", b"x" * 33, b"x" * 32 + b"y"]
    nb, npr = 2_000, 9_000
    build = Chunk([StrColumn([None if rng.random() < 0.03 else words[i] for i in rng.integers(0, 200, nb).tolist()]), Column(abi.I64, np.arange(nb))])
    probe = Chunk([StrColumn([None if rng.random() < 0.03 else words[i] for i in rng.integers(0, len(words), npr).tolist()]), Column(abi.I64, np.arange(npr)),
                   Column(abi.F64, rng.random(npr), rng.random(npr) > 0.1)])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], jt, 1, probe_batch_rows=4_096)
    want = orc.hash_join(cfg, build, probe)
    stats = []
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=4_096, pull_rows=1 << 14, radix=FORCE, stats_out=stats)
    assert stats[0].probe_route == abi.ROUTE_KEYREC and stats[0].radix_batches >= 3
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("jt", [abi.JOIN_INNER, abi.JOIN_LEFT_OUTER])
def test_long_string_keys_when_every_digest_of_a_length_collides(ctx, orc, jt):
    # knob KEYREC = 3: a digest that only carries the length — all 40 words of a length are candidates of one another, so the byte comparison
    # decides nearly every match (and the materialising form's notes say "not every candidate passed": its emit launch compares again).
    # COUNT(*) and rows, several batches, NULL keys
    rng = np.random.default_rng(5 + jt)
    words = [bytes(rng.integers(97, 123, n, dtype=np.uint8)) for n in (40, 41, 64, 200, 1000) for _ in range(40)]
    nb, npr = 3_000, 12_000
    build = Chunk([StrColumn([None if rng.random() < 0.03 else words[i] for i in rng.integers(0, 150, nb).tolist()]), Column(abi.I64, np.arange(nb))])
    probe = Chunk([StrColumn([None if rng.random() < 0.03 else words[i] for i in rng.integers(0, len(words), npr).tolist()]), Column(abi.I64, np.arange(npr))])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], jt, 1, probe_batch_rows=5_000)
    want = orc.hash_join(cfg, build, probe)
    with ctx.knobs(KEYREC=3):
        stats = []
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=5_000, pull_rows=1 << 14, radix=FORCE, stats_out=stats)
        assert stats[0].probe_route == abi.ROUTE_KEYREC and stats[0].keyrec_digests == 1 and stats[0].radix_batches >= 3
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
        if jt == abi.JOIN_INNER:
            c, st = _count(ctx, cfg, build, probe, chunk_rows=5_000)
            assert st.probe_route == abi.ROUTE_KEYREC and c == want.NumRows()
