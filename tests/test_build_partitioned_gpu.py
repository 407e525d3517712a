"""GPU parity of the partitioned build (tsq_buildpart.h): the join table assembled slice by slice in LDS must behave
exactly like the one hashRowContainer.PutChunk builds row by row (executor/hash_table.go:146-169) — same joined rows for
NULL keys (never inserted, :161-163), duplicate chains that fill buckets and run into the next slice, the sentinel key
word, skewed keys that overflow a partition region, every key type, and at the BASELINE size."""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu

SENT = np.uint64(0x8080808080808080).astype(np.int64)


def _join(ctx, cfg, build, probe, radix, stats=None, **kw):
    return G.run_join(ctx, cfg, build, probe, chunk_rows=1 << 22, pull_rows=1 << 20, radix=radix, stats_out=stats, **kw)


@pytest.mark.parametrize("nb", [65_536, 70_001, 300_000, 1_000_003])
def test_partitioned_build_rows_equal_oracle(ctx, orc, nb):
    rng = np.random.default_rng(nb)
    npr = 50_000
    bk = rng.integers(0, nb // 3, nb)        # ~3 duplicates per key: full buckets, chains into the next bucket / slice
    bk[:500] = 4242                          # one key with 500 duplicates: its chain crosses slice ends
    bk[500:530] = SENT
    pk = rng.integers(-100, nb // 3 + 100, npr)
    pk[:20] = 4242
    pk[20:30] = SENT
    build = Chunk([Column(abi.I64, bk, rng.random(nb) > 0.05), Column(abi.I64, rng.integers(0, 1 << 40, nb))])
    probe = Chunk([Column(abi.I64, pk, rng.random(npr) > 0.05), Column(abi.I64, np.arange(npr))])
    cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe)
    got = _join(ctx, cfg, build, probe, abi.RADIX_FORCE)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)
    stats = []
    c, s, x = _join(ctx, cfg, build, probe, abi.RADIX_FORCE, stats=stats, count_only=True, checksum=True)
    assert stats[0].build_partitioned == 1
    assert stats[0].build_rows_inserted == int(build.columns[0].notnull.sum())
    assert c == want.NumRows() and (s, x) == orc.rows_checksum(want)
    stats = []
    assert _join(ctx, cfg, build, probe, abi.RADIX_OFF, stats=stats, count_only=True, checksum=True) == (c, s, x)
    assert stats[0].build_partitioned == 0


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
def test_partitioned_build_under_outer_joins(ctx, orc, jt, inner):
    rng = np.random.default_rng(jt)
    nb, npr = 120_000, 40_000
    outer = Chunk([Column(abi.I64, rng.integers(0, 80_000, npr), rng.random(npr) > 0.1), Column(abi.F64, rng.random(npr))])
    innr = Chunk([Column(abi.I64, rng.integers(0, 60_000, nb), rng.random(nb) > 0.1), Column(abi.F64, rng.random(nb), rng.random(nb) > 0.3)])
    t = [abi.I64, abi.F64]
    cfg = H.join_cfg(t, t, [0], [0], jt, inner)
    want = orc.hash_join(cfg, innr, outer)
    got = _join(ctx, cfg, innr, outer, abi.RADIX_FORCE)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("bt,pt", [(abi.F64, abi.F64), (abi.F32, abi.F64), (abi.U64, abi.I64), (abi.I64, abi.U64)])
def test_partitioned_build_key_types(ctx, orc, bt, pt):
    rng = np.random.default_rng(bt * 7 + pt)
    nb, npr = 100_000, 30_000

    def keys(tp, n):
        if tp in (abi.F32, abi.F64):
            v = rng.integers(-20_000, 20_000, n).astype(np.float64) / 2
            v[::97] = -0.0
            return v.astype(np.float32) if tp == abi.F32 else v
        if tp == abi.U64:
            v = rng.integers(0, 30_000, n).astype(np.uint64)
            v[::50] |= np.uint64(1 << 63)  # values above MaxInt64 never match a signed column (codec.go:212-240)
            return v
        return rng.integers(-15_000, 30_000, n)

    build = Chunk([Column(bt, keys(bt, nb)), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(pt, keys(pt, npr)), Column(abi.I64, np.arange(npr))])
    cfg = H.join_cfg([pt, abi.I64], [bt, abi.I64], [0], [0], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe)
    got = _join(ctx, cfg, build, probe, abi.RADIX_FORCE)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_partitioned_build_skewed_keys_rebuild_unsliced(ctx):
    # 2.5 % of the build rows share one key: its table slice (~4000 slots) cannot hold 5 000 duplicates, the slice image
    # raises the fail flag and the table is rebuilt as ONE slice by k_build_insert — same rows, same checksum.
    nb = 200_000
    bk = np.arange(nb, dtype=np.int64) + 100
    bk[::40] = 7
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, np.array([7, 100, 101, 5, 7], dtype=np.int64)), Column(abi.I64, np.arange(5))])
    cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
    stats = []
    c, s, x = _join(ctx, cfg, build, probe, abi.RADIX_FORCE, stats=stats, count_only=True, checksum=True)
    n7 = nb // 40
    assert c == 2 * n7 + 1
    assert stats[0].build_slice_retries == 1 and stats[0].table_slice_bits == 0 and stats[0].build_rows_inserted == nb
    assert _join(ctx, cfg, build, probe, abi.RADIX_OFF, count_only=True, checksum=True) == (c, s, x)
    got = _join(ctx, cfg, build, Chunk([Column(abi.I64, np.array([7], dtype=np.int64)), Column(abi.I64, np.zeros(1, np.int64))]), abi.RADIX_FORCE)
    assert got.NumRows() == n7 and sorted(got.columns[3].data.tolist()) == sorted(np.nonzero(bk == 7)[0].tolist())


def test_build_with_one_key_on_most_rows_takes_the_chained_table(ctx):
    # 90 % of the build rows share one key: in the open-addressing multimap every insert would scan the key's whole run (O(d^2 / 8)
    # bucket reads).  rowHashMap.Put is O(1) (hash_table.go:247-256) — and since round 3 so is the build here: when the bounded walks
    # give up (sliced, then one slice) the table is rebuilt with one slot per distinct key + row chains, with either build strategy
    nb = 200_000
    bk = np.full(nb, 7, dtype=np.int64)
    bk[::10] = np.arange(nb // 10) + 100
    build = Chunk([Column(abi.I64, bk), Column(abi.I64, np.arange(nb))])
    probe = Chunk([Column(abi.I64, np.array([7, 100, 99, 7], dtype=np.int64)), Column(abi.I64, np.arange(4))])
    cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
    for radix in (abi.RADIX_FORCE, abi.RADIX_OFF):
        assert _join(ctx, cfg, build, probe, radix, count_only=True) == 2 * (nb - nb // 10) + 1


def _device_join_checksum(ctx, n_build, n_probe, radix):
    lib = ctx.lib
    cols = [G.DevCol(ctx, abi.I64, n_build), G.DevCol(ctx, abi.I64, n_build), G.DevCol(ctx, abi.I64, n_probe), G.DevCol(ctx, abi.I64, n_probe)]
    try:
        ctx.gen_column(G.gen_spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=n_build), n_build, cols[0].data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=2, col=1, m=1 << 30), n_build, cols[1].data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=0, m=n_build + n_build // 4), n_probe, cols[2].data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=1, m=1 << 30), n_probe, cols[3].data)
        cfg = H.join_cfg([abi.I64, abi.I64], [abi.I64, abi.I64], [0], [0], abi.JOIN_INNER, 1)
        h = C.c_void_p()
        _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
        try:
            _lib.check(lib.tsq_join_set_radix(h, radix), h)
            _lib.check(lib.tsq_join_build_push(h, G.dev_cols(cols[:2]), 2, n_build), h)
            _lib.check(lib.tsq_join_build_finish(h), h)
            _lib.check(lib.tsq_join_set_count_only(h, 1), h)
            _lib.check(lib.tsq_join_set_checksum(h, 1), h)
            _lib.check(lib.tsq_join_probe_push(h, G.dev_cols(cols[2:]), 2, n_probe, None), h)
            _lib.check(lib.tsq_join_probe_finish(h), h)
            c, s, x = C.c_int64(0), C.c_uint64(0), C.c_uint64(0)
            _lib.check(lib.tsq_join_count(h, C.byref(c)), h)
            _lib.check(lib.tsq_join_checksum(h, C.byref(s), C.byref(x)), h)
            st = abi.Stats()
            _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
            return c.value, s.value, x.value, st
        finally:
            lib.tsq_join_destroy(h)
    finally:
        for c in cols:
            c.free()


def test_partitioned_build_full_size_checksum_equals_row_build(ctx):
    # 1e8 build rows (BASELINE configs[0]): the row checksum of the joined rows (Σ and ⊕ of rowhash over all four output
    # columns, so every build row id stored in the table matters) must not depend on how the table was built.
    n = 100_000_000
    c1, s1, x1, st1 = _device_join_checksum(ctx, n, 20_000_000, abi.RADIX_AUTO)
    c0, s0, x0, st0 = _device_join_checksum(ctx, n, 20_000_000, abi.RADIX_OFF)
    assert st1.build_partitioned == 1 and st0.build_partitioned == 0
    assert st1.build_rows_inserted == st0.build_rows_inserted == n
    assert (c1, s1, x1) == (c0, s0, x0) and c1 > 15_000_000
