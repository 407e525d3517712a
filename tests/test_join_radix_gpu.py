"""GPU parity of the radix-partitioned COUNT(*) probe (tsq_radix.h) — forced on small inputs so that
every edge of the partition / ordered-queue / deferred-spill / overflow machinery is compared with
the oracle, and run at the BASELINE size through a size-independent property.

The joined rows of an equi-join do not depend on how probe rows are dispatched to workers
(executor/join.go:160-231), so radix FORCE, OFF and the oracle must agree bit for bit on the count.
"""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu

SENT = np.uint64(0x8080808080808080).astype(np.int64)  # the table's EMPTY sentinel is a legal key
SENT_PRE = H.unmix64(0x8080808080808080)               # ... and so is the key whose TABLE WORD mix64(key) is the sentinel


def _cfg(bt=abi.I64, pt=abi.I64):
    return H.join_cfg([pt, abi.I64], [bt, abi.I64], [0], [0], abi.JOIN_INNER, 1)


def _count(ctx, cfg, build, probe, radix, chunk_rows=1 << 22, stats=None):
    return G.run_join(ctx, cfg, build, probe, chunk_rows=chunk_rows, count_only=True, radix=radix, stats_out=stats)


@pytest.mark.parametrize("case", H.golden("join_cases.json"), ids=lambda c: c["ref"][:48])
def test_radix_forced_on_golden_rows(ctx, case):
    keep = []
    cfg, _, _, build, probe, _, _ = H.lower_join_case(case, keep)
    assert G.run_join(ctx, cfg, build, probe, count_only=True, radix=abi.RADIX_FORCE) == len(case["expect"]), case["ref"]


@pytest.mark.parametrize("n_probe", [1, 63, 64, 65, 1000, 16383, 16384, 16385, 50001])
def test_radix_ragged_sizes_dups_nulls_vs_oracle(ctx, orc, n_probe):
    rng = np.random.default_rng(n_probe)
    n_build = 3000
    bk = rng.integers(0, 900, n_build)  # ~3.3 duplicates per key: buckets overflow, spill chains
    pk = rng.integers(-50, 1000, n_probe)
    build = Chunk([Column(abi.I64, bk, rng.random(n_build) > 0.1), Column(abi.I64, rng.integers(0, 9, n_build))])
    probe = Chunk([Column(abi.I64, pk, rng.random(n_probe) > 0.1), Column(abi.I64, rng.integers(0, 9, n_probe))])
    cfg = _cfg()
    want = orc.hash_join(cfg, build, probe).NumRows()
    stats = []
    assert _count(ctx, cfg, build, probe, abi.RADIX_FORCE, stats=stats) == want
    assert stats[0].radix_batches >= 1
    assert _count(ctx, cfg, build, probe, abi.RADIX_OFF) == want
    # pushes of tidb_max_chunk_size rows accumulate in pinned staging and reach the same batch
    assert _count(ctx, cfg, build, probe, abi.RADIX_FORCE, chunk_rows=1024) == want


def test_radix_sentinel_key_and_long_chains(ctx, orc):
    rng = np.random.default_rng(5)
    bk = np.concatenate([np.full(7, SENT), np.full(5, SENT_PRE), np.full(40, 12345), rng.integers(0, 50, 500)]).astype(np.int64)
    pk = np.concatenate([np.full(11, SENT), np.full(13, SENT_PRE), np.full(9, 12345), rng.integers(0, 60, 3000)]).astype(np.int64)
    rng.shuffle(bk)
    rng.shuffle(pk)
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(len(bk)))])
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, np.arange(len(pk)))])
    cfg = _cfg()
    want = orc.hash_join(cfg, build, probe).NumRows()
    assert want >= 7 * 11 + 5 * 13 + 40 * 9
    assert _count(ctx, cfg, build, probe, abi.RADIX_FORCE) == want
    assert _count(ctx, cfg, build, probe, abi.RADIX_OFF) == want


@pytest.mark.parametrize("bt,pt", [(abi.F64, abi.F64), (abi.F32, abi.F64), (abi.U64, abi.I64), (abi.I64, abi.U64), (abi.U64, abi.U64)])
def test_radix_key_types(ctx, orc, bt, pt):
    # float keys compare through their float64 image, int keys of mixed signedness never match above 2^63
    # (util/codec/codec.go:212-240)
    rng = np.random.default_rng(11)

    def col(t, n):
        if t in (abi.F64, abi.F32):
            v = rng.integers(-20, 20, n).astype(np.float64) * 0.5
            return Column(t, v.astype(np.float32) if t == abi.F32 else v, rng.random(n) > 0.05)
        v = rng.integers(0, 40, n).astype(np.uint64)
        v[rng.random(n) < 0.3] |= np.uint64(1 << 63)
        return Column(t, v.view(np.int64) if t == abi.I64 else v, rng.random(n) > 0.05)

    build = Chunk([col(bt, 700), Column(abi.I64, np.arange(700))])
    probe = Chunk([col(pt, 20000), Column(abi.I64, np.arange(20000))])
    cfg = _cfg(bt, pt)
    want = orc.hash_join(cfg, build, probe).NumRows()
    assert _count(ctx, cfg, build, probe, abi.RADIX_FORCE) == want
    assert _count(ctx, cfg, build, probe, abi.RADIX_OFF) == want


def test_radix_skew_takes_the_overflow_list(ctx):
    # every probe row carries the same key: one region overflows, the rest of the rows go through
    # the overflow list and are probed by the plain kernel; the count stays exact.
    n = 600_000
    build = Chunk([Column(abi.I64, np.array([7, 7, 7, 8, 9], dtype=np.int64)), Column(abi.I64, np.arange(5))])
    pk = np.full(n, 7, dtype=np.int64)
    pk[::1000] = 8
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, np.arange(n))])
    stats = []
    got = _count(ctx, _cfg(), build, probe, abi.RADIX_FORCE, stats=stats)
    assert got == 3 * (n - n // 1000) + n // 1000
    assert stats[0].radix_overflow_rows > 0


def test_radix_zipf_keys_property(ctx):
    # heavy-tailed probe keys against unique build keys 0..99999: count = #probe keys inside the range
    rng = np.random.default_rng(3)
    n = 2_000_000
    pk = (rng.zipf(1.3, n) - 1).astype(np.int64)
    build = Chunk([Column(abi.I64, rng.permutation(100_000).astype(np.int64)), Column(abi.I64, np.arange(100_000))])
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, np.arange(n))])
    want = int((pk < 100_000).sum())
    assert _count(ctx, _cfg(), build, probe, abi.RADIX_FORCE) == want
    assert _count(ctx, _cfg(), build, probe, abi.RADIX_OFF) == want


def _device_count(ctx, n_build, n_probe, hit_mod, radix, steps=1):
    lib = ctx.lib
    bk, pk = G.DevCol(ctx, abi.I64, n_build), G.DevCol(ctx, abi.I64, n_probe)
    try:
        ctx.gen_column(G.gen_spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=n_build), n_build, bk.data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=0, m=hit_mod), n_probe, pk.data)
        cfg = H.join_cfg([abi.I64], [abi.I64], [0], [0], abi.JOIN_INNER, 1)
        h = C.c_void_p()
        _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
        try:
            _lib.check(lib.tsq_join_set_radix(h, radix), h)
            _lib.check(lib.tsq_join_build_push(h, G.dev_cols([bk]), 1, n_build), h)
            _lib.check(lib.tsq_join_build_finish(h), h)
            _lib.check(lib.tsq_join_set_count_only(h, 1), h)
            for _ in range(steps):
                _lib.check(lib.tsq_join_probe_push(h, G.dev_cols([pk]), 1, n_probe, None), h)
            _lib.check(lib.tsq_join_probe_finish(h), h)
            c = C.c_int64(0)
            _lib.check(lib.tsq_join_count(h, C.byref(c)), h)
            st = abi.Stats()
            _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
            return c.value, st
        finally:
            lib.tsq_join_destroy(h)
    finally:
        bk.free()
        pk.free()


def _expected_hits(n_build, n_probe, hit_mod):
    cnt, step = 0, 1 << 24
    for lo in range(0, n_probe, step):
        i = np.arange(lo, min(n_probe, lo + step), dtype=np.uint64)
        cnt += int(((G.np_gen_r(42, 1, 0, i) % np.uint64(hit_mod)) < np.uint64(n_build)).sum())
    return cnt


def test_radix_auto_engages_on_large_batches_and_matches_direct_probe(ctx):
    nb, npr, mod = 10_000_000, 30_000_000, 12_500_000  # BASELINE configs[1] shape (1e7 build), hit ratio 0.8
    want = _expected_hits(nb, npr, mod)
    got, st = _device_count(ctx, nb, npr, mod, abi.RADIX_AUTO, steps=2)
    assert got == 2 * want and st.radix_batches == 2 and st.radix_overflow_rows == 0 and st.radix_bits >= 6
    got, st = _device_count(ctx, nb, npr, mod, abi.RADIX_OFF)
    assert got == want and st.radix_batches == 0


def test_radix_full_size_1e8_by_1e8_property(ctx):
    n = 100_000_000  # BASELINE headline size: every probe key lies in [0, n) and joins exactly once
    got, st = _device_count(ctx, n, n, n, abi.RADIX_AUTO)
    assert got == n and st.radix_batches == 1 and st.radix_bits >= 10 and st.radix_overflow_rows == 0
    # round 4: a build side the packed routes serve never builds its 64-bit table (tsq_join_build_finish leaves it to the first batch
    # that needs it); with the knob off the table is built as before: 2^13+ LDS-sized slices, no retry
    assert st.probe_route == abi.ROUTE_PACKED and st.table_buckets == 0
    with ctx.knobs(LAZY_TABLE=0):
        got, st = _device_count(ctx, n, n, n, abi.RADIX_AUTO)
    assert got == n and st.table_slice_bits >= 13 and st.build_slice_retries == 0 and st.table_buckets > 0


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
def test_radix_forced_leaves_every_other_probe_kernel_exact(ctx, orc, jt, inner):
    # with the radix strategy forced, the materialising / outer-join / checksum probes (which are not eligible and keep
    # the direct kernels) must still equal the oracle row for row on a table with full buckets and long spill chains.
    rng = np.random.default_rng(17 + jt)
    nb, npr = 40_000, 30_000
    bk = rng.integers(0, 9000, nb)           # ~4.4 duplicates per key: full buckets, late list, spill chains
    bk[:300] = 77                            # one key with 300 duplicates
    bk[300:320] = SENT
    bk[320:330] = SENT_PRE
    pk = rng.integers(-100, 9500, npr)
    pk[:50] = SENT
    pk[50:80] = SENT_PRE
    left = Chunk([Column(abi.I64, pk, rng.random(npr) > 0.05), Column(abi.I64, rng.integers(0, 99, npr))])
    right = Chunk([Column(abi.I64, bk, rng.random(nb) > 0.05), Column(abi.I64, rng.integers(0, 99, nb))])
    t = [abi.I64, abi.I64]
    cfg = H.join_cfg(t, t, [0], [0], jt, inner)
    build, probe = (right, left) if inner == 1 else (left, right)
    want = orc.hash_join(cfg, build, probe)
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, radix=abi.RADIX_FORCE)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    c, s, x = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, count_only=True, checksum=True, radix=abi.RADIX_FORCE)
    assert c == want.NumRows() and (s, x) == orc.rows_checksum(want)


# ---------------------------------------------------------------- LDS probe (tsq_ldsprobe.h) on sliced tables
def _env(ctx, **kw):
    """the knobs of tsq_ctx_set_knob (include/tsq.h) for the duration of a with-block"""
    return ctx.knobs(**kw)


def _sliced_inputs(seed, nb, npr):
    rng = np.random.default_rng(seed)
    bk = rng.integers(-(1 << 62), 1 << 62, nb)          # full-width keys: every table slice gets rows
    bk[: nb // 4] = rng.integers(0, nb // 16, nb // 4)  # ... and a quarter of the rows carries ~4 duplicates per key
    bk[:700] = 4242                                     # one key with 700 duplicates: a chain of ~90 full buckets that wraps
    bk[700:712] = SENT_PRE
    bk[712:720] = SENT
    pk = np.concatenate([rng.choice(bk, npr // 2), rng.integers(-(1 << 62), 1 << 62, npr - npr // 2)])
    pk[:40] = SENT_PRE
    pk[40:60] = 4242
    rng.shuffle(pk)
    build = Chunk([Column(abi.I64, bk, rng.random(nb) > 0.03), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, pk, rng.random(npr) > 0.03), Column(abi.I64, np.arange(npr))])
    return build, probe


@pytest.mark.parametrize("knobs", [{}, {"LDS_NF_MAX": 1}, {"LDS_NF_MAX": 3, "RADIX_PB_MAX": 4}, {"LDS_NF_MAX": 2, "RADIX_PB_MAX": 3},
                                   {"RADIX_KERNEL_L2": 1}, {"TABLE_LF_PERMILLE": 500}, {"TABLE_LF_PERMILLE": 650}],
                         ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()) or "default")
def test_lds_probe_on_a_sliced_table_vs_oracle(ctx, orc, knobs):
    # 300 K build rows -> partitioned build, ~100 table slices; the knobs force several images per partition (S > 1, a last
    # image with fewer slices), the L2 route over the same sliced table, and other load factors
    build, probe = _sliced_inputs(23, 300_000, 1_200_000)
    cfg = _cfg()
    want = orc.hash_join(cfg, build, probe).NumRows()
    with _env(ctx, **knobs):
        stats = []
        assert _count(ctx, cfg, build, probe, abi.RADIX_FORCE, stats=stats) == want
        st = stats[0]
        assert st.radix_batches == 1 and st.build_partitioned == 1 and st.table_slice_bits >= 5 and st.build_slice_retries == 0
    assert _count(ctx, cfg, build, probe, abi.RADIX_OFF) == want


def test_sliced_table_serves_the_materialising_and_outer_probes(ctx, orc):
    build, probe = _sliced_inputs(29, 120_000, 90_000)
    t = [abi.I64, abi.I64]
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = H.join_cfg(t, t, [0], [0], jt, 1)
        want = orc.hash_join(cfg, build, probe)
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, radix=abi.RADIX_FORCE)
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
        c, s, x = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, count_only=True, checksum=True)
        assert c == want.NumRows() and (s, x) == orc.rows_checksum(want)


def test_overfull_slice_rebuilds_the_table_as_one_slice(ctx, orc):
    # 6000 duplicates of one key cannot live in one table slice (~4000 slots): the build notices, rebuilds unsliced, stays exact
    rng = np.random.default_rng(31)
    nb, npr = 200_000, 400_000
    bk = rng.integers(0, 1 << 40, nb)
    bk[:6000] = 99
    pk = rng.choice(bk, npr)
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, np.arange(npr))])
    cfg = _cfg()
    want = orc.hash_join(cfg, build, probe).NumRows()
    for radix in (abi.RADIX_FORCE, abi.RADIX_AUTO):
        stats = []
        assert _count(ctx, cfg, build, probe, radix, stats=stats) == want
        assert stats[0].build_slice_retries == 1 and stats[0].table_slice_bits == 0


def test_heavily_duplicated_build_key_takes_the_chained_table(ctx, orc):
    # d duplicates of one key cost O(d^2 / 8) bucket reads in an open-addressing multimap; rowHashMap.Put is O(1) (hash_table.go:247-256).
    # Round 3: when the multimap's bounded walks give up, the table is rebuilt with ONE slot per distinct key + a chain of its rows
    # (rowHashMap's entry list) — any multiplicity, quickly, the same joined rows.
    import time
    nb = 1_500_000
    bk = np.zeros(nb, dtype=np.int64)  # key 0: a million build rows
    bk[: nb // 3] = np.arange(nb // 3)
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(nb))])
    pk = np.array([0, 1, 2, 0, 499_999, 500_000, 7, -1, 0], dtype=np.int64)
    probe = Chunk([Column(abi.I64, pk, np.array([1, 1, 1, 1, 1, 1, 1, 1, 0], dtype=bool)), Column(abi.I64, np.arange(len(pk)))])
    t0 = time.time()
    stats = []
    got = _count(ctx, _cfg(), build, probe, abi.RADIX_AUTO, stats=stats)
    zeros = nb - nb // 3 + 1
    assert got == 2 * zeros + 4 and time.time() - t0 < 30
    assert stats[0].build_slice_retries >= 1
    # the joined rows themselves, inner and left outer, against the oracle on a smaller heavy key (40 K duplicates)
    rng = np.random.default_rng(4)
    bk2 = np.concatenate([np.full(40_000, 77), np.arange(5000)]).astype(np.int64)
    rng.shuffle(bk2)
    build2 = Chunk([Column(abi.I64, bk2), Column(abi.F64, rng.random(len(bk2)), rng.random(len(bk2)) > 0.1)])
    probe2 = Chunk([Column(abi.I64, np.array([77, 3, 77, 9999, 4999], dtype=np.int64), np.array([1, 1, 0, 1, 1], dtype=bool)), Column(abi.I64, np.arange(5))])
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = H.join_cfg(probe2.types(), build2.types(), [0], [0], jt, 1)
        want = orc.hash_join(cfg, build2, probe2)
        got2 = G.run_join(ctx, cfg, build2, probe2, chunk_rows=1 << 22, pull_rows=8192)
        assert got2.NumRows() == want.NumRows() and H.rows_equal_unordered(got2, want)
        c, s, x = G.run_join(ctx, cfg, build2, probe2, chunk_rows=1 << 22, count_only=True, checksum=True)
        assert c == want.NumRows() and (s, x) == orc.rows_checksum(want)


# ---------------------------------------------------------------- materialising radix path (k_lds_probe_count MODE 1 / 2)
def _emit_inputs(seed, nb, npr, kt=abi.I64, n_pay_b=1, n_pay_p=1, pay_t=abi.I64, key_pos=0):
    rng = np.random.default_rng(seed)
    bk = rng.integers(-(1 << 62), 1 << 62, nb)
    bk[: nb // 4] = rng.integers(0, nb // 16, nb // 4)  # ~4 duplicates per key on a quarter of the rows
    bk[:300] = 4242
    bk[300:305] = SENT_PRE
    pk = np.concatenate([rng.choice(bk, npr // 2), rng.integers(-(1 << 62), 1 << 62, npr - npr // 2)])
    pk[:7] = SENT_PRE
    rng.shuffle(pk)
    if kt == abi.U64:
        bk, pk = bk.astype(np.uint64), pk.astype(np.uint64)

    def pay(n):
        return Column(pay_t, rng.random(n) if pay_t == abi.F64 else rng.integers(-(1 << 40), 1 << 40, n))

    bcols = [pay(nb) for _ in range(n_pay_b)]
    pcols = [pay(npr) for _ in range(n_pay_p)]
    bcols.insert(min(key_pos, len(bcols)), Column(kt, bk))
    pcols.insert(min(key_pos, len(pcols)), Column(kt, pk))
    return Chunk(bcols), Chunk(pcols), min(key_pos, n_pay_b), min(key_pos, n_pay_p)


@pytest.mark.parametrize("shape", [dict(), dict(n_pay_b=2, n_pay_p=2, key_pos=1), dict(n_pay_b=0, n_pay_p=1), dict(n_pay_b=1, n_pay_p=0), dict(kt=abi.U64, pay_t=abi.F64, key_pos=2),
                                   dict(knobs={"LDS_NF_MAX": 1}), dict(knobs={"LDS_NF_MAX": 3, "RADIX_PB_MAX": 4}, n_pay_b=2)],
                         ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()) or "k,v x k,v")
def test_materialising_radix_join_vs_oracle(ctx, orc, shape):
    shape = dict(shape)
    knobs = shape.pop("knobs", {})
    build, probe, bkc, pkc = _emit_inputs(41, 250_000, 900_000, **shape)
    cfg = H.join_cfg(probe.types(), build.types(), [pkc], [bkc], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe)
    with _env(ctx, **knobs):
        got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, radix=abi.RADIX_FORCE)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    off = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, radix=abi.RADIX_OFF)
    assert H.rows_equal_unordered(off, want)


def test_materialising_radix_join_skewed_probe_keys_use_the_overflow_list(ctx, orc):
    n = 600_000
    rng = np.random.default_rng(43)
    bk = rng.permutation(200_000).astype(np.int64)
    bk[:3] = 7  # three build rows with the hot key
    pk = np.full(n, 7, dtype=np.int64)
    pk[::1000] = 8
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(len(bk)))])
    probe = Chunk([Column(abi.I64, pk), Column(abi.I64, np.arange(n))])
    t = [abi.I64, abi.I64]
    cfg = H.join_cfg(t, t, [0], [0], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe)
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, radix=abi.RADIX_FORCE)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_materialising_radix_join_full_size_property(ctx):
    # 1e8 x 1e8 (k, v) x (k, v), every probe row joins exactly once: row count, the key columns agree, and the order-independent
    # checksum of the four output columns equals the direct path's (GPU vs GPU at full size; both are pinned on the oracle above)
    n = 50_000_000
    lib = ctx.lib
    cols = [G.DevCol(ctx, abi.I64, n) for _ in range(4)]
    outs = [G.DevCol(ctx, abi.I64, n, with_nulls=True) for _ in range(4)]
    try:
        ctx.gen_column(G.gen_spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=n), n, cols[0].data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=2, col=1, m=1 << 30), n, cols[1].data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=0, m=n), n, cols[2].data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=1, m=1 << 30), n, cols[3].data)
        cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
        sums = {}
        for radix in (abi.RADIX_AUTO, abi.RADIX_OFF):
            h = C.c_void_p()
            _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
            try:
                _lib.check(lib.tsq_join_set_radix(h, radix), h)
                _lib.check(lib.tsq_join_build_push(h, G.dev_cols(cols[:2]), 2, n), h)
                _lib.check(lib.tsq_join_build_finish(h), h)
                _lib.check(lib.tsq_join_probe_push(h, G.dev_cols(cols[2:]), 2, n, None), h)
                _lib.check(lib.tsq_join_probe_finish(h), h)
                total, acc = 0, [0, 0, 0, 0]
                while True:
                    m, eos = C.c_int64(0), C.c_int32(0)
                    _lib.check(lib.tsq_join_pull(h, G.dev_cols(outs), 4, n, C.byref(m), C.byref(eos)), h)
                    if m.value == 0:
                        break
                    got = [np.empty(m.value, np.int64) for _ in range(4)]
                    for g, o in zip(got, outs):
                        ctx.d2h(g, o.data)
                    assert (got[0] == got[2]).all()  # probe key == build key on every joined row
                    with np.errstate(over="ignore"):
                        mix = (got[0].view(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (got[1].view(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) ^ got[3].view(np.uint64)
                        acc[0] += int(mix.sum(dtype=np.uint64))
                        acc[1] ^= int(np.bitwise_xor.reduce(mix))
                    total += m.value
                st = abi.Stats()
                _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
                sums[radix] = (total, acc[0] & ((1 << 64) - 1), acc[1], st.radix_batches)
            finally:
                lib.tsq_join_destroy(h)
        assert sums[abi.RADIX_AUTO][0] == n and sums[abi.RADIX_AUTO][:3] == sums[abi.RADIX_OFF][:3]
        assert sums[abi.RADIX_AUTO][3] >= 1 and sums[abi.RADIX_OFF][3] == 0
    finally:
        for d in cols + outs:
            d.free()
