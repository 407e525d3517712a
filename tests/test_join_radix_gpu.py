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
    assert st.table_slice_bits >= 13 and st.build_slice_retries == 0


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
def _env(**kw):
    import contextlib
    import os

    @contextlib.contextmanager
    def cm():
        old = {k: os.environ.get(k) for k in kw}
        try:
            for k, v in kw.items():
                os.environ[k] = str(v)
            yield
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return cm()


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


@pytest.mark.parametrize("knobs", [{}, {"TSQ_LDS_NF_MAX": 1}, {"TSQ_LDS_NF_MAX": 3, "TSQ_RADIX_PB_MAX": 4}, {"TSQ_LDS_NF_MAX": 2, "TSQ_RADIX_PB_MAX": 3},
                                   {"TSQ_RADIX_KERNEL": "l2"}, {"TSQ_TABLE_LF": 0.5}, {"TSQ_TABLE_LF": 0.65}],
                         ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()) or "default")
def test_lds_probe_on_a_sliced_table_vs_oracle(ctx, orc, knobs):
    # 300 K build rows -> partitioned build, ~100 table slices; the knobs force several images per partition (S > 1, a last
    # image with fewer slices), the L2 route over the same sliced table, and other load factors
    build, probe = _sliced_inputs(23, 300_000, 1_200_000)
    cfg = _cfg()
    want = orc.hash_join(cfg, build, probe).NumRows()
    with _env(**knobs):
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


def test_heavily_duplicated_build_key_is_refused_quickly(ctx):
    # d duplicates of one key cost O(d^2 / 8) bucket reads in an open-addressing multimap; rowHashMap.Put is O(1)
    # (hash_table.go:247-256).  Beyond ~16 K duplicates the build gives the operator back to Go instead of running for minutes.
    import time
    nb = 1_500_000
    bk = np.zeros(nb, dtype=np.int64)
    bk[: nb // 3] = np.arange(nb // 3)
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, np.arange(10, dtype=np.int64)), Column(abi.I64, np.arange(10))])
    t0 = time.time()
    with pytest.raises(_lib.TsqError) as ei:
        _count(ctx, _cfg(), build, probe, abi.RADIX_AUTO)
    assert ei.value.status == abi.ERR_UNSUPPORTED and time.time() - t0 < 20
