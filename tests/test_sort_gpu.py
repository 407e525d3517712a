"""GPU parity of tsq_sort_* (ORDER BY / TopN, SURVEY.md §8 f rank 3) against the oracle's restatement of SortExec /
TopNExec (executor/sort.go:27-318 over util/chunk/compare.go:27-103).  Both are stable, so on inputs without NaN the
ordered rows must be identical row for row; the reference's own ORDER BY results (union_scan_test.go:33-36) go through the
executor mirror."""
import ctypes as C
import math

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import executor as X
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu


def _t(rows):
    cols = list(zip(*rows))
    return Chunk([Column(abi.I64, np.array(c, dtype=np.int64)) for c in cols])


def _rows(chunks):
    out = []
    for c in chunks:
        out += c.rows()
    return out


def test_reference_order_by_rows_through_the_executors(ctx):
    t = _t([(1, 5), (2, 3), (3, 4), (4, 8), (6, 8), (7, 6)])  # union_scan_test.go:31
    q = lambda by, desc: _rows(X.drain(X.SortExec(ctx, X.MockDataSource(ctx, t), by, desc)))
    assert q([0], [True]) == [(7, 6), (6, 8), (4, 8), (3, 4), (2, 3), (1, 5)]                   # :33
    assert q([1, 0], [False, False]) == [(2, 3), (3, 4), (1, 5), (7, 6), (4, 8), (6, 8)]        # :34
    assert q([1, 0], [True, True]) == [(6, 8), (4, 8), (7, 6), (1, 5), (3, 4), (2, 3)]          # :35
    # distsql_test.go:158 shape: order by b desc limit 2,1 -> the third row of the descending order
    top = X.TopNExec(ctx, X.MockDataSource(ctx, t), [1, 0], [True, True], offset=2, count=1)
    assert _rows(X.drain(top)) == [(7, 6)]
    assert _rows(X.drain(X.TopNExec(ctx, X.MockDataSource(ctx, t), [0], [False], offset=4, count=100))) == [(6, 8), (7, 6)]
    assert _rows(X.drain(X.TopNExec(ctx, X.MockDataSource(ctx, t), [0], [False], offset=9, count=5))) == []
    empty = Chunk([Column(abi.I64, np.zeros(0, np.int64))] * 2)
    assert _rows(X.drain(X.SortExec(ctx, X.MockDataSource(ctx, empty), [0], [False]))) == []


def _rand(rng, n, tp, null_p, small):
    if tp == abi.I64:
        v = rng.integers(-50, 50, n) if small else rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64)
    elif tp == abi.U64:
        v = rng.integers(0, 100, n).astype(np.uint64) if small else rng.integers(0, (1 << 64) - 1, n, dtype=np.uint64)
        if small:
            v[::7] |= np.uint64(1 << 63)
    elif tp == abi.F64:
        v = rng.integers(-40, 40, n) / 4.0 if small else np.ldexp(rng.random(n) - 0.5, rng.integers(-300, 300, n))
        v[::11] = -0.0
        v[::13] = 0.0
        v[1::97] = math.inf
        v[2::97] = -math.inf
    else:
        v = (rng.integers(-40, 40, n) / 4.0).astype(np.float32)
    return Column(tp, v, rng.random(n) >= null_p if null_p else None)


def _same_rows(a, b):
    assert a.NumRows() == b.NumRows()
    if a.NumRows() == 0:
        return
    for ca, cb in zip(a.columns, b.columns):
        na = np.ones(len(ca), bool) if ca.notnull is None else ca.notnull
        nb = np.ones(len(cb), bool) if cb.notnull is None else cb.notnull
        assert (na == nb).all()
        assert (ca.data.view(np.uint8).reshape(len(ca), -1)[na] == cb.data.view(np.uint8).reshape(len(cb), -1)[nb]).all()


@pytest.mark.parametrize("n", [1, 63, 64, 4095, 4096, 4097, 30_011])
@pytest.mark.parametrize("keys", [([0], [False]), ([1], [True]), ([2, 0], [False, True]), ([3, 1, 0], [True, False, False]), ([1, 2, 3, 0], [False] * 4)])
def test_random_tables_equal_the_stable_oracle_row_for_row(ctx, orc, n, keys):
    rng = np.random.default_rng(n * 31 + len(keys[0]))
    chk = Chunk([_rand(rng, n, abi.I64, 0.1, True), _rand(rng, n, abi.F64, 0.1, True), _rand(rng, n, abi.U64, 0.0, True),
                 _rand(rng, n, abi.F32, 0.2, True), Column(abi.I64, np.arange(n))])
    want = orc.sort_rows(chk, *keys)
    _same_rows(G.run_sort(ctx, chk, *keys, chunk_rows=1024, pull_rows=1000), want)


@pytest.mark.parametrize("tp", [abi.I64, abi.U64, abi.F64])
@pytest.mark.parametrize("desc", [False, True])
def test_full_range_keys_all_eight_digit_passes(ctx, orc, tp, desc):
    rng = np.random.default_rng(tp * 2 + desc)
    n = 50_000
    chk = Chunk([_rand(rng, n, tp, 0.05, False), Column(abi.I64, np.arange(n))])
    stats = []
    got = G.run_sort(ctx, chk, [0], [desc], chunk_rows=1 << 20, pull_rows=1 << 20, stats_out=stats)
    _same_rows(got, orc.sort_rows(chk, [0], [desc]))
    assert stats[0]["passes"] == 9 and stats[0]["rows"] == n   # 8 digits + the NULL flag


def test_small_range_keys_skip_the_constant_digits(ctx, orc):
    rng = np.random.default_rng(4)
    n = 100_000
    chk = Chunk([Column(abi.I64, rng.integers(0, 2500, n)), Column(abi.I64, np.arange(n))])  # day numbers: two non-trivial bytes
    stats = []
    got = G.run_sort(ctx, chk, [0], [False], chunk_rows=1 << 20, pull_rows=1 << 20, stats_out=stats)
    _same_rows(got, orc.sort_rows(chk, [0], [False]))
    assert stats[0]["passes"] == 2 and stats[0]["passes_skipped"] == 6


def test_topn_offsets_and_limits(ctx, orc):
    rng = np.random.default_rng(6)
    n = 20_000
    chk = Chunk([_rand(rng, n, abi.I64, 0.1, True), _rand(rng, n, abi.F64, 0.0, True), Column(abi.I64, np.arange(n))])
    full = orc.sort_rows(chk, [0, 1], [True, False])
    for off, cnt in [(0, 1), (0, 10), (5, 1000), (19_990, 100), (20_000, 5), (0, 0), (123, 4096)]:
        got = G.run_sort(ctx, chk, [0, 1], [True, False], offset=off, count=cnt, pull_rows=512)
        _same_rows(got, full.slice(min(off, n), min(n, off + cnt)))


def test_nan_keys_sort_after_every_number(ctx):
    # CompareFloat64 answers "greater" whenever a NaN is involved (types/compare.go:104-112): no total order in the
    # reference; here NaNs form one run after +inf (ASC) / before everything (DESC), NULLs stay outside as usual
    v = np.array([1.0, math.nan, -math.inf, 3.0, math.nan, math.inf, 0.0])
    chk = Chunk([Column(abi.F64, v, np.array([1, 1, 1, 0, 1, 1, 1], bool)), Column(abi.I64, np.arange(7))])
    asc = G.run_sort(ctx, chk, [0], [False]).columns[1].data.tolist()
    assert asc == [3, 2, 6, 0, 5, 1, 4]
    assert G.run_sort(ctx, chk, [0], [True]).columns[1].data.tolist() == [1, 4, 5, 0, 6, 2, 3]


def test_device_resident_input_and_output(ctx, orc):
    rng = np.random.default_rng(9)
    n = 300_000
    chk = Chunk([_rand(rng, n, abi.I64, 0.05, False), _rand(rng, n, abi.F64, 0.05, False), Column(abi.I64, np.arange(n))])
    ins = [G.to_device(ctx, c) for c in chk.columns]
    outs = [G.DevCol(ctx, c.tp, n, with_nulls=True) for c in chk.columns]
    cfg = abi.SortCfg()
    cfg.n_cols, cfg.n_keys, cfg.limit_offset, cfg.limit_count = 3, 2, 0, -1
    for i, t in enumerate(chk.types()):
        cfg.col_types[i] = t
    cfg.key_col[0], cfg.key_desc[0], cfg.key_col[1], cfg.key_desc[1] = 1, 1, 0, 0
    h = C.c_void_p()
    _lib.check(ctx.lib.tsq_sort_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        _lib.check(ctx.lib.tsq_sort_push(h, G.dev_cols(ins), 3, n), h)
        _lib.check(ctx.lib.tsq_sort_finish(h), h)
        m, eos = C.c_int64(0), C.c_int32(0)
        _lib.check(ctx.lib.tsq_sort_pull(h, G.dev_cols(outs), 3, n, C.byref(m), C.byref(eos)), h)
        assert m.value == n
        _same_rows(Chunk([o.to_host() for o in outs]), orc.sort_rows(chk, [1, 0], [True, False]))
    finally:
        ctx.lib.tsq_sort_destroy(h)
        for d in ins + outs:
            d.free()


def test_full_size_sortedness_and_permutation_property(ctx):
    # 5e7 rows: the output keys are non-decreasing, the payload is a permutation of the row ids, and every row still carries
    # its own key (payload = f(key) checked through the generator)
    n = 50_000_000
    k, v = G.DevCol(ctx, abi.I64, n), G.DevCol(ctx, abi.I64, n)
    ok_, ov = G.DevCol(ctx, abi.I64, n, with_nulls=True), G.DevCol(ctx, abi.I64, n, with_nulls=True)
    try:
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=5, col=0, m=1 << 40), n, k.data)
        ctx.gen_column(G.gen_spec(abi.GEN_SEQ), n, v.data)
        cfg = abi.SortCfg()
        cfg.n_cols, cfg.n_keys, cfg.limit_offset, cfg.limit_count = 2, 1, 0, -1
        cfg.col_types[0] = cfg.col_types[1] = abi.I64
        h = C.c_void_p()
        _lib.check(ctx.lib.tsq_sort_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
        try:
            _lib.check(ctx.lib.tsq_sort_push(h, G.dev_cols([k, v]), 2, n), h)
            _lib.check(ctx.lib.tsq_sort_finish(h), h)
            m, eos = C.c_int64(0), C.c_int32(0)
            _lib.check(ctx.lib.tsq_sort_pull(h, G.dev_cols([ok_, ov]), 2, n, C.byref(m), C.byref(eos)), h)
            assert m.value == n
        finally:
            ctx.lib.tsq_sort_destroy(h)
        keys, ids = ok_.to_host().data, ov.to_host().data
        assert (np.diff(keys) >= 0).all()
        assert (np.sort(ids) == np.arange(n)).all()
        assert (keys == (G.np_gen_r(42, 5, 0, ids.astype(np.uint64)) % np.uint64(1 << 40)).astype(np.int64)).all()
    finally:
        for d in (k, v, ok_, ov):
            d.free()


@pytest.mark.parametrize("case", ["asc_wide", "desc_nulls", "ties_two_keys", "nulls_cover_the_limit"])
def test_topn_radix_select_on_large_inputs(ctx, orc, case):
    # n >= 1 Mi rows and Offset + Count <= n / 16: the K-th key of the first ORDER BY item is found by radix select and only
    # the candidates (ties at the threshold included, in input order) are sorted; the rows must equal the stable oracle's
    rng = np.random.default_rng(len(case))
    n = 1_500_000
    if case == "asc_wide":
        chk, keys, off, cnt = Chunk([_rand(rng, n, abi.I64, 0.0, False), Column(abi.I64, np.arange(n))]), ([0], [False]), 10, 1000
    elif case == "desc_nulls":
        chk, keys, off, cnt = Chunk([_rand(rng, n, abi.F64, 0.05, False), Column(abi.I64, np.arange(n))]), ([0], [True]), 0, 5000
    elif case == "ties_two_keys":   # 40 distinct first keys: the threshold bucket holds ~37 K rows, the second key decides
        chk = Chunk([Column(abi.I64, rng.integers(0, 40, n)), Column(abi.F64, rng.integers(0, 1000, n) / 8.0), Column(abi.I64, np.arange(n))])
        keys, off, cnt = ([0, 1], [False, True]), 100, 2000
    else:                           # ASC with 20 % NULLs: the first 30 000 rows of the order are all NULL keys (ties, input order)
        chk, keys, off, cnt = Chunk([_rand(rng, n, abi.I64, 0.2, False), Column(abi.I64, np.arange(n))]), ([0], [False]), 0, 30_000
    stats = []
    got = G.run_sort(ctx, chk, *keys, chunk_rows=1 << 20, pull_rows=1 << 16, offset=off, count=cnt, stats_out=stats)
    want = orc.sort_rows(chk, *keys).slice(off, off + cnt)
    _same_rows(got, want)
    assert got.NumRows() == cnt and stats[0]["rows"] < n // 2


def test_sort_call_sequence_and_cancel_contract(ctx):
    # create -> push* -> finish -> pull* -> destroy; cancel from anywhere; misuse is an error, not a crash
    lib = ctx.lib
    cfg = abi.SortCfg()
    cfg.n_cols, cfg.n_keys, cfg.limit_offset, cfg.limit_count = 1, 1, 0, -1
    cfg.col_types[0] = abi.I64
    bad = abi.SortCfg()
    bad.n_cols, bad.n_keys, bad.limit_count = 1, 1, -1
    bad.col_types[0] = 9  # not a column type
    h = C.c_void_p()
    assert lib.tsq_sort_create(ctx.h, C.byref(bad), C.byref(h)) != abi.OK
    bad.col_types[0], bad.key_col[0] = abi.I64, 3
    assert lib.tsq_sort_create(ctx.h, C.byref(bad), C.byref(h)) == abi.ERR_INVALID
    _lib.check(lib.tsq_sort_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        chk = Chunk([Column(abi.I64, np.array([3, 1, 2]))])
        keep = []
        from tinysql_amd.chunk import make_cols, out_buffers
        cols = make_cols(chk.columns, keep)
        out, bufs = out_buffers([abi.I64], 8, keep)
        n, eos = C.c_int64(0), C.c_int32(0)
        assert lib.tsq_sort_pull(h, out, 1, 8, C.byref(n), C.byref(eos)) == abi.ERR_INVALID           # pull before finish
        _lib.check(lib.tsq_sort_push(h, cols, 1, 3), h)
        assert lib.tsq_sort_push(h, cols, 2, 3) == abi.ERR_INVALID                                    # wrong column count
        _lib.check(lib.tsq_sort_finish(h), h)
        assert lib.tsq_sort_push(h, cols, 1, 3) == abi.ERR_INVALID                                    # push after finish
        _lib.check(lib.tsq_sort_pull(h, out, 1, 8, C.byref(n), C.byref(eos)), h)
        assert n.value == 3 and bufs[0][0][:3].tolist() == [1, 2, 3]
        _lib.check(lib.tsq_sort_pull(h, out, 1, 8, C.byref(n), C.byref(eos)), h)
        assert n.value == 0 and eos.value == 1                                                         # idempotent end of stream
        _lib.check(lib.tsq_sort_cancel(h), h)
        assert lib.tsq_sort_pull(h, out, 1, 8, C.byref(n), C.byref(eos)) == abi.ERR_CANCELLED
    finally:
        lib.tsq_sort_destroy(h)


# ------------------------------------------------------------------------------------------------ var-len payload columns (round 2)
def _string_table(rng, n, long_cells=False):
    from tinysql_amd.chunk import StrColumn
    names = [None if rng.random() < 0.1 else (b"" if rng.random() < 0.1 else b"name-%06d" % i) for i in range(n)]
    if long_cells:  # a 5 KiB payload like the reference's join benchmark: one cell per wave in the copy kernel
        notes = [None if rng.random() < 0.2 else bytes([97 + i % 26]) * int(rng.integers(1, 3) * 2500) for i in range(n)]
    else:
        notes = [bytes(rng.integers(0, 256, int(rng.integers(0, 20)), dtype=np.uint8)) for _ in range(n)]
    return Chunk([StrColumn(names), Column(abi.I64, rng.integers(0, max(n // 7, 2), n), rng.random(n) >= 0.1), StrColumn(notes),
                  Column(abi.F64, np.round(rng.standard_normal(n), 1), rng.random(n) >= 0.1)])


@pytest.mark.parametrize("n,long_cells", [(1, False), (1000, False), (70_000, False), (900, True)])
def test_string_payload_columns_follow_their_rows(ctx, orc, n, long_cells):
    # ORDER BY k, f DESC over (name varchar, k, note blob, f): the var-len cells travel with their rows (Chunk.AppendRow), NULL and
    # empty strings stay what they are; equal keys keep input order (the stable oracle), so the rows are identical one by one
    rng = np.random.default_rng(n)
    t = _string_table(rng, n, long_cells)
    want = orc.sort_rows(t, [1, 3], [False, True])
    got = G.run_sort(ctx, t, [1, 3], [False, True], chunk_rows=1024, pull_rows=1000)
    assert got.rows() == want.rows()


def test_string_payload_topn_window_and_the_executor_mirror(ctx, orc):
    from tinysql_amd import executor as X
    rng = np.random.default_rng(4)
    t = _string_table(rng, 30_000)
    want = orc.sort_rows(t, [3, 1], [False, False]).rows()
    got = G.run_sort(ctx, t, [3, 1], [False, False], chunk_rows=4096, pull_rows=333, offset=1234, count=5000)
    assert got.rows() == want[1234:6234]
    chunks = X.drain(X.TopNExec(ctx, X.MockDataSource(ctx, t), [3, 1], [False, False], 10, 2000))
    assert [r for c in chunks for r in c.rows()] == want[10:2010] and all(c.NumRows() <= 1024 for c in chunks)


def test_string_payload_device_resident(ctx, orc):
    # device-resident push of a var-len column (offsets + data in HBM) and device-resident pull sized by tsq_sort_peek
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(8)
    n = 50_000
    t = _string_table(rng, n)
    want = orc.sort_rows(t, [1], [True])
    lib = ctx.lib
    cfg = abi.SortCfg()
    cfg.n_cols, cfg.n_keys, cfg.limit_offset, cfg.limit_count, cfg.max_chunk_size = 2, 1, 0, -1, 1024
    cfg.col_types[0], cfg.col_types[1], cfg.key_col[0], cfg.key_desc[0] = abi.I64, abi.BYTES, 0, 1
    names, keys = t.columns[0], t.columns[1]
    nbytes = int(names.offsets[-1])
    dk, dkb = ctx.alloc(8 * n + 64), ctx.alloc(n // 8 + 64)
    dd, do, db = ctx.alloc(nbytes + 64), ctx.alloc(8 * (n + 1) + 64), ctx.alloc(n // 8 + 64)
    od, oo, ob, okd, okb = ctx.alloc(nbytes + 64), ctx.alloc(8 * (n + 1) + 64), ctx.alloc(n // 8 + 64), ctx.alloc(8 * n + 64), ctx.alloc(n // 8 + 64)
    h = C.c_void_p()
    _lib.check(lib.tsq_sort_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        ctx.h2d(dk, np.ascontiguousarray(keys.data))
        ctx.h2d(dkb, np.packbits(keys.notnull, bitorder="little"))
        ctx.h2d(dd, names.data[:max(nbytes, 1)].copy())
        ctx.h2d(do, names.offsets)
        ctx.h2d(db, np.packbits(names.notnull, bitorder="little"))
        cols = (abi.Col * 2)()
        cols[0].data, cols[0].null_bitmap, cols[0].length, cols[0].elem_size, cols[0].type, cols[0].flags = dk, dkb, n, 8, abi.I64, abi.COL_DEVICE
        cols[1].data, cols[1].offsets, cols[1].null_bitmap, cols[1].length, cols[1].elem_size, cols[1].type, cols[1].flags = dd, do, db, n, -1, abi.BYTES, abi.COL_DEVICE
        _lib.check(lib.tsq_sort_push(h, cols, 2, n), h)
        _lib.check(lib.tsq_sort_finish(h), h)
        vb, nr = (C.c_int64 * 2)(), C.c_int64(0)
        _lib.check(lib.tsq_sort_peek(h, n, C.byref(nr), vb, 2), h)
        assert nr.value == n and vb[0] == 0 and vb[1] == nbytes
        out = (abi.Col * 2)()
        out[0].data, out[0].null_bitmap, out[0].length, out[0].elem_size, out[0].type, out[0].flags = okd, okb, n, 8, abi.I64, abi.COL_DEVICE
        out[1].data, out[1].offsets, out[1].null_bitmap, out[1].length, out[1].elem_size, out[1].type, out[1].flags = od, oo, ob, n, -1, abi.BYTES, abi.COL_DEVICE
        m, eos = C.c_int64(0), C.c_int32(0)
        _lib.check(lib.tsq_sort_pull(h, out, 2, n, C.byref(m), C.byref(eos)), h)
        assert m.value == n
        offs, data, bits = np.zeros(n + 1, np.int64), np.zeros(max(nbytes, 1), np.uint8), np.zeros(n // 8 + 8, np.uint8)
        ctx.d2h(offs, oo)
        ctx.d2h(data, od)
        ctx.d2h(bits, ob)
        nn = np.unpackbits(bits, bitorder="little")[:n].astype(bool)
        got = [bytes(data[offs[r]:offs[r + 1]]) if nn[r] else None for r in range(n)]
        assert offs[0] == 0 and offs[n] == nbytes and got == want.columns[0].values()
    finally:
        lib.tsq_sort_destroy(h)
        for p in (dk, dkb, dd, do, db, od, oo, ob, okd, okb):
            ctx.free(p)


# ------------------------------------------------------------------------------------------------ string ORDER BY items (round 2)
def _string_keys(rng, n, kind):
    if kind == "short":      # many ties, shared prefixes, the empty string
        return [None if rng.random() < 0.08 else bytes(rng.integers(97, 100, int(rng.integers(0, 5)), dtype=np.uint8)) for _ in range(n)]
    if kind == "binary":     # every byte value, zero bytes at the end (a string and the same string + "\0" differ)
        return [None if rng.random() < 0.05 else bytes(rng.integers(0, 256, int(rng.integers(0, 12)), dtype=np.uint8)) + b"\0" * int(rng.integers(0, 3)) for _ in range(n)]
    if kind == "long":       # three 8-byte chunks and more, differences late in the string
        return [b"customer#%09d-%s" % (int(rng.integers(0, 50)), bytes(rng.integers(97, 123, int(rng.integers(0, 9)), dtype=np.uint8))) for _ in range(n)]
    return [b"" for _ in range(n)]  # "empty": only the length image, and it is constant


@pytest.mark.parametrize("n", [1, 64, 4097, 30_011])
@pytest.mark.parametrize("kind", ["short", "binary", "long", "empty"])
@pytest.mark.parametrize("desc", [False, True])
def test_order_by_a_string_column_equals_the_stable_oracle(ctx, orc, n, kind, desc):
    # cmpString (compare.go:71-77): bytes, then the shorter first; NULL before every string; Desc negates.  The string key runs as
    # a sequence of stable radix sorts (length, then 8-byte chunks last to first), so equal strings keep input order like the oracle.
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(n * 7 + len(kind) + desc)
    t = Chunk([StrColumn(_string_keys(rng, n, kind)), Column(abi.I64, np.arange(n, dtype=np.int64))])
    want = orc.sort_rows(t, [0], [desc])
    got = G.run_sort(ctx, t, [0], [desc], chunk_rows=1024, pull_rows=1000)
    assert got.rows() == want.rows()


@pytest.mark.parametrize("keys", [([1, 0], [False, False]), ([0, 1], [True, False]), ([2, 0, 3], [False, True, True]), ([0, 2], [False, False])])
def test_string_and_number_order_by_items_mixed(ctx, orc, keys):
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(len(keys[0]) + keys[0][0])
    n = 20_000
    t = Chunk([StrColumn(_string_keys(rng, n, "short")), Column(abi.I64, rng.integers(0, 40, n), rng.random(n) >= 0.1), StrColumn(_string_keys(rng, n, "long")),
               Column(abi.F64, np.round(rng.standard_normal(n), 1), rng.random(n) >= 0.1)])
    want = orc.sort_rows(t, keys[0], keys[1])
    got = G.run_sort(ctx, t, keys[0], keys[1], chunk_rows=4096, pull_rows=777)
    assert got.rows() == want.rows()


def test_string_key_topn_and_the_executor_mirror(ctx, orc):
    # TopN on a string first key (small inputs take the full sort, large ones select on the first image: next test); a
    # string SECOND key behind a selected first key sorts only the candidates (n >= 2^20 rows so that the select runs)
    from tinysql_amd import executor as X
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(12)
    n = 40_000
    t = Chunk([StrColumn(_string_keys(rng, n, "long")), Column(abi.I64, rng.integers(0, 1000, n))])
    want = orc.sort_rows(t, [0, 1], [True, False]).rows()
    got = G.run_sort(ctx, t, [0, 1], [True, False], chunk_rows=4096, pull_rows=500, offset=100, count=3000)
    assert got.rows() == want[100:3100]
    chunks = X.drain(X.TopNExec(ctx, X.MockDataSource(ctx, t), [0, 1], [True, False], 5, 700))
    assert [r for c in chunks for r in c.rows()] == want[5:705]
    n = (1 << 20) + 17
    k = rng.integers(0, 1 << 40, n)
    k[rng.integers(0, n, 5000)] = 3  # ties on the first key inside the window: the string key orders them
    t = Chunk([Column(abi.I64, k), StrColumn([b"s%03d" % int(x) for x in rng.integers(0, 300, n)])])
    stats = []
    got = G.run_sort(ctx, t, [0, 1], [False, False], chunk_rows=1 << 18, pull_rows=4096, offset=0, count=3000, stats_out=stats)
    assert got.rows() == orc.sort_rows(t, [0, 1], [False, False]).rows()[:3000]
    assert stats[0]["rows"] < n // 2


@pytest.mark.parametrize("desc", [False, True])
def test_topn_radix_select_on_a_string_first_key(ctx, orc, desc):
    # n >= 2^20 and a small window: the candidates are selected on the string's first eight bytes (ties on that image included),
    # then sorted on the whole string; long shared prefixes make the first image a weak filter, short ones a sharp one
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(31 + desc)
    n = (1 << 20) + 333
    vals = rng.integers(0, 1 << 40, n)
    names = [None if v % 1013 == 0 else b"%010x-%d" % (int(v), i % 7) for i, v in enumerate(vals.tolist())]
    t = Chunk([StrColumn(names), Column(abi.I64, np.arange(n, dtype=np.int64))])
    stats = []
    got = G.run_sort(ctx, t, [0, 1], [desc, False], chunk_rows=1 << 18, pull_rows=4096, offset=17, count=2000, stats_out=stats)
    want = orc.sort_rows(t, [0, 1], [desc, False]).rows()[17:2017]
    assert got.rows() == want
    assert stats[0]["rows"] < n // 4  # only the candidates went through the radix passes
