"""GPU parity: HIP hash join (through the C-ABI) vs the oracle and the reference's golden rows.

Join output order is unspecified in the reference (probe workers interleave, join.go:233-239), so
results are compared as multisets; integer keys / payloads are bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import expression as E
from tinysql_amd.chunk import Chunk, Column
from tinysql_amd.executor import HashJoinExec, MockDataSource, drain

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", H.golden("join_cases.json"), ids=lambda c: c["ref"][:48])
def test_join_golden_rows(ctx, case):
    keep = []
    cfg, left, right, build, probe, _, _ = H.lower_join_case(case, keep)
    out = G.run_join(ctx, cfg, build, probe)
    assert H.rows_equal_unordered(out, [tuple(r) for r in case["expect"]]), case["ref"]
    # COUNT(*) fast path agrees with the materialised row count
    assert G.run_join(ctx, cfg, build, probe, count_only=True) == len(case["expect"])


@pytest.mark.parametrize("case", H.golden("join_cases.json")[:6], ids=lambda c: c["ref"][:48])
def test_join_golden_through_executor_interface(ctx, case):
    # Open / Next (<=max_chunk_size rows, empty chunk = EOS, idempotent after EOS) / Close
    keep = []
    cfg, left, right, _, _, conds, filt = H.lower_join_case(case, keep)
    exe = HashJoinExec(ctx, MockDataSource(ctx, left, 2), MockDataSource(ctx, right, 2), case["left_keys"], case["right_keys"],
                       H.JOIN_TYPES[case["type"]], case["inner_child"], conds, filt, max_chunk_size=4)
    chunks = drain(exe)
    assert all(0 < c.NumRows() <= 4 for c in chunks)
    rows = [r for c in chunks for r in c.rows()]
    assert H.rows_equal_unordered(rows, [tuple(r) for r in case["expect"]])
    exe.Open()
    while exe.Next().NumRows():
        pass
    assert exe.Next().NumRows() == 0  # idempotent after EOS (aggregate.go:565-567 convention)
    exe.Close()


def _rand_table(rng, n, key_hi, types, key_null=0.05):
    """column 0 = join key in [0, key_hi) (duplicates + some NULLs), the rest random payloads with 10% NULLs."""
    cols = [Column(abi.I64, rng.integers(0, key_hi, n), rng.random(n) >= key_null if key_null else None)]
    for t in types[1:]:
        if t == abi.I64:
            cols.append(H.random_column(rng, t, n, 0.1, lo=-1000, hi=1000))
        else:
            cols.append(H.random_column(rng, t, n, 0.1))
    return Chunk(cols)


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_INNER, 0), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
def test_join_random_vs_oracle(ctx, orc, jt, inner):
    rng = np.random.default_rng(100 + jt * 2 + inner)
    lt = [abi.I64, abi.I64, abi.F64]
    rt = [abi.I64, abi.F32, abi.U64, abi.I64]
    left = _rand_table(rng, 30011, 9000, lt)     # ~3.3 rows per key: duplicate chains
    right = _rand_table(rng, 20029, 9000, rt)
    cfg = H.join_cfg(lt, rt, [0], [0], jt, inner)
    build, probe = (right, left) if inner == 1 else (left, right)
    want = orc.hash_join(cfg, build, probe)
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1024, pull_rows=1024)
    assert got.NumRows() == want.NumRows()
    assert H.rows_equal_unordered(got, want)
    # count-only + fused checksum == oracle checksum of its materialised rows
    c, s, x = G.run_join(ctx, cfg, build, probe, count_only=True, checksum=True)
    assert c == want.NumRows()
    assert (s, x) == orc.rows_checksum(want)


def test_join_other_conditions_and_outer_filter_vs_oracle(ctx, orc):
    rng = np.random.default_rng(7)
    lt, rt = [abi.I64, abi.I64], [abi.I64, abi.I64, abi.F64]
    left, right = _rand_table(rng, 8000, 1500, lt), _rand_table(rng, 6000, 1500, rt)
    keep = []
    # ON l.k = r.k AND l.v + r.v > 0 AND r.d < 0.0 ; outer side filter l.v != 3
    conds = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(1, abi.I64), E.Column(3, abi.I64)), E.Constant(0)),
             E.ScalarFunction("lt", E.Column(4, abi.F64), E.Constant(0.0))]
    filt = [E.ScalarFunction("ne", E.Column(1, abi.I64), E.Constant(3))]
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = H.join_cfg(lt, rt, [0], [0], jt, 1, conds, filt, keep)
        want = orc.hash_join(cfg, right, left)
        got = G.run_join(ctx, cfg, right, left)
        assert H.rows_equal_unordered(got, want)
        assert G.run_join(ctx, cfg, right, left, count_only=True) == want.NumRows()


def test_join_external_selected_vector(ctx, orc):
    rng = np.random.default_rng(8)
    t = [abi.I64, abi.I64]
    left, right = _rand_table(rng, 5000, 700, t), _rand_table(rng, 5000, 700, t)
    sel = rng.random(5000) < 0.5
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = H.join_cfg(t, t, [0], [0], jt, 1)
        want = orc.hash_join(cfg, right, left, selected=sel)
        got = G.run_join(ctx, cfg, right, left, selected=sel)
        assert H.rows_equal_unordered(got, want)


def test_join_multi_column_keys(ctx, orc):
    rng = np.random.default_rng(9)
    n = 12000
    mk = lambda: Chunk([Column(abi.I64, rng.integers(0, 40, n), rng.random(n) > 0.03),
                        Column(abi.F64, rng.integers(0, 5, n).astype(np.float64), rng.random(n) > 0.03),
                        Column(abi.U64, rng.integers(0, 3, n).astype(np.uint64)),
                        Column(abi.I64, rng.integers(-9, 9, n))])
    left, right = mk(), mk()
    t = [abi.I64, abi.F64, abi.U64, abi.I64]
    for jt, inner in ((abi.JOIN_INNER, 1), (abi.JOIN_RIGHT_OUTER, 0)):
        cfg = H.join_cfg(t, t, [0, 1, 2], [0, 1, 2], jt, inner)
        build, probe = (right, left) if inner == 1 else (left, right)
        want = orc.hash_join(cfg, build, probe)
        got = G.run_join(ctx, cfg, build, probe)
        assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_join_signed_unsigned_and_float_key_classes(ctx, orc):
    # codec.go:219-224: uint64 >= 2^63 carries flag 9 and never equals an int64 with the same bits;
    # float32 keys are widened; an int key never equals a float key.
    big = (1 << 64) - 1
    u = H.chunk_from_rows([[1], [big], [5], [1 << 63], [None]], [abi.U64])
    s = H.chunk_from_rows([[1], [-1], [5], [-(1 << 63)], [None]], [abi.I64])
    for (lt, rt, l, r) in ((abi.U64, abi.I64, u, s), (abi.I64, abi.U64, s, u), (abi.U64, abi.U64, u, u), (abi.I64, abi.I64, s, s)):
        cfg = H.join_cfg([lt], [rt], [0], [0], abi.JOIN_INNER, 1)
        assert H.rows_equal_unordered(G.run_join(ctx, cfg, r, l), orc.hash_join(cfg, r, l))
    f32 = H.chunk_from_rows([[1.0], [1.5], [0.1]], [abi.F32])
    f64 = H.chunk_from_rows([[1.0], [1.5], [0.1], [-0.0], [0.0]], [abi.F64])
    cfg = H.join_cfg([abi.F32], [abi.F64], [0], [0], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, f64, f32)
    assert want.NumRows() == 2  # 0.1f widened != 0.1
    assert H.rows_equal_unordered(G.run_join(ctx, cfg, f64, f32), want)
    cfg = H.join_cfg([abi.F64], [abi.F64], [0], [0], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, f64, f64)
    assert want.NumRows() == 5  # -0.0 and 0.0 are different join keys (raw bytes)
    assert H.rows_equal_unordered(G.run_join(ctx, cfg, f64, f64), want)
    i = H.chunk_from_rows([[1]], [abi.I64])
    cfg = H.join_cfg([abi.I64], [abi.F64], [0], [0], abi.JOIN_INNER, 1)
    assert G.run_join(ctx, cfg, f64, i).NumRows() == orc.hash_join(cfg, f64, i).NumRows() == 0


def test_join_key_equal_to_the_empty_sentinel(ctx, orc):
    # the table's EMPTY marker is an ordinary int64 value for the user: side list keeps it joinable
    sent = 0x8080808080808080 - (1 << 64)
    rows_b = [[sent, i] for i in range(3)] + [[7, 100], [sent + 1, 5]]
    rows_p = [[sent, -1], [7, -2], [sent, -3], [8, -4]]
    t = [abi.I64, abi.I64]
    b, p = H.chunk_from_rows(rows_b, t), H.chunk_from_rows(rows_p, t)
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT_OUTER):
        cfg = H.join_cfg(t, t, [0], [0], jt, 1)
        want = orc.hash_join(cfg, b, p)
        assert H.rows_equal_unordered(G.run_join(ctx, cfg, b, p), want)
        assert G.run_join(ctx, cfg, b, p, count_only=True) == want.NumRows()
    # many sentinel rows (> initial side-list capacity)
    n = 3000
    b = Chunk([Column(abi.I64, np.full(n, sent, np.int64)), Column(abi.I64, np.arange(n))])
    p = H.chunk_from_rows([[sent, 1]], t)
    cfg = H.join_cfg(t, t, [0], [0], abi.JOIN_INNER, 1)
    assert G.run_join(ctx, cfg, b, p, count_only=True) == n


def test_join_empty_and_ragged_inputs(ctx, orc):
    t = [abi.I64, abi.I64]
    empty = H.chunk_from_rows([], t)
    some = H.chunk_from_rows([[1, 2], [3, 4]], t)
    for jt, b, p in ((abi.JOIN_INNER, empty, some), (abi.JOIN_INNER, some, empty), (abi.JOIN_LEFT_OUTER, empty, some),
                     (abi.JOIN_INNER, empty, empty)):
        cfg = H.join_cfg(t, t, [0], [0], jt, 1)
        assert H.rows_equal_unordered(G.run_join(ctx, cfg, b, p), orc.hash_join(cfg, b, p))
    # ragged pushes: chunk sizes that do not divide anything, staging boundaries, tiny probe batches
    rng = np.random.default_rng(3)
    left, right = _rand_table(rng, 7777, 500, t), _rand_table(rng, 3333, 500, t)
    cfg = H.join_cfg(t, t, [0], [0], abi.JOIN_LEFT_OUTER, 1, probe_batch_rows=640)
    want = orc.hash_join(cfg, right, left)
    for cr, pr in ((1, 7), (37, 1024), (1000, 33)):
        if cr == 1:
            l2, r2 = left.slice(0, 300), right.slice(0, 200)
            w2 = orc.hash_join(cfg, r2, l2)
            assert H.rows_equal_unordered(G.run_join(ctx, cfg, r2, l2, chunk_rows=cr, pull_rows=pr), w2)
        else:
            assert H.rows_equal_unordered(G.run_join(ctx, cfg, right, left, chunk_rows=cr, pull_rows=pr), want)


def test_join_config1_count_star_1e5(ctx, orc):
    # BASELINE config[0]: SELECT count(*) FROM t1 JOIN t2 ON t1.k=t2.k, two 1e5-row int64 tables (J-seq)
    n = 100000
    t1 = Chunk([Column(abi.I64, np.arange(n)), Column(abi.I64, np.arange(n))])
    t2 = Chunk([Column(abi.I64, np.arange(n)), Column(abi.I64, np.arange(n))])
    cfg = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [0], abi.JOIN_INNER, 1)
    cnt, s, x = G.run_join(ctx, cfg, t2, t1, count_only=True, checksum=True)
    ocnt, _, _, os_, ox = orc.hash_join_timed(cfg, t2, t1, 4)
    assert cnt == ocnt == n and (s, x) == (os_, ox)


def test_join_errors_are_loud(ctx):
    lib = ctx.lib
    cfg = H.join_cfg([abi.I64], [7], [0], [0], abi.JOIN_INNER, 1)
    h = C.c_void_p()
    assert lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)) == abi.ERR_INVALID  # no such column type (var-len columns are accepted since round 2)
    cfg = H.join_cfg([abi.I64], [abi.I64], [0], [0], abi.JOIN_LEFT_OUTER, 0)
    assert lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)) == abi.ERR_UNSUPPORTED
    # overflow inside an OtherCondition surfaces as types.ErrOverflow
    keep = []
    cond = [E.ScalarFunction("gt", E.ScalarFunction("plus", E.Column(0, abi.I64), E.Column(1, abi.I64)), E.Constant(0))]
    cfg = H.join_cfg([abi.I64], [abi.I64], [0], [0], abi.JOIN_INNER, 1, cond, (), keep)
    big = H.chunk_from_rows([[(1 << 63) - 1]], [abi.I64])
    with pytest.raises(_lib.TsqError) as ei:
        G.run_join(ctx, cfg, big, big)
    assert ei.value.status == abi.ERR_OVERFLOW_BIGINT
    # cancel is honoured
    cfg = H.join_cfg([abi.I64], [abi.I64], [0], [0], abi.JOIN_INNER, 1)
    _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    lib.tsq_join_cancel(h)
    assert lib.tsq_join_build_finish(h) == abi.ERR_CANCELLED
    lib.tsq_join_destroy(h)


def _device_join(ctx, n_build, n_probe, hit_mod, materialise_check=True):
    """J-uniq-shuffled on device: build k = affine bijection of [0,n_build), v_b = splitmix(k^salt);
    probe k = r(i,0) mod hit_mod, v_p = r(i,1).  Returns (count, sum, xor, probe keys on host or None)."""
    lib = ctx.lib
    bk, bv = G.DevCol(ctx, abi.I64, n_build), G.DevCol(ctx, abi.I64, n_build)
    pk, pv = G.DevCol(ctx, abi.I64, n_probe), G.DevCol(ctx, abi.I64, n_probe)
    try:
        ctx.gen_column(G.gen_spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=n_build), n_build, bk.data)
        ctx.gen_column(G.gen_spec(abi.GEN_HASH_OF_COL, table=2, b=0xABCDEF), n_build, bv.data, src=bk.data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=0, m=hit_mod), n_probe, pk.data)
        ctx.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=1, m=1 << 62), n_probe, pv.data)
        cfg = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [0], abi.JOIN_INNER, 1)
        h = C.c_void_p()
        _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
        try:
            _lib.check(lib.tsq_join_build_push(h, G.dev_cols([bk, bv]), 2, n_build), h)
            _lib.check(lib.tsq_join_build_finish(h), h)
            _lib.check(lib.tsq_join_set_count_only(h, 1), h)
            _lib.check(lib.tsq_join_set_checksum(h, 1), h)
            _lib.check(lib.tsq_join_probe_push(h, G.dev_cols([pk, pv]), 2, n_probe, None), h)
            _lib.check(lib.tsq_join_probe_finish(h), h)
            c, s, x = C.c_int64(0), C.c_uint64(0), C.c_uint64(0)
            _lib.check(lib.tsq_join_count(h, C.byref(c)), h)
            _lib.check(lib.tsq_join_checksum(h, C.byref(s), C.byref(x)), h)
        finally:
            lib.tsq_join_destroy(h)
        return c.value, s.value, x.value
    finally:
        for d in (bk, bv, pk, pv):
            d.free()


def _expected_device_join(n_build, n_probe, hit_mod):
    """size-independent property: build keys are a bijection of [0,n_build) and v_b is a function of k,
    so every probe row with k < n_build joins exactly once and the joined row is computable from the
    probe row alone (numpy, chunked)."""
    cnt, s, x = 0, np.uint64(0), np.uint64(0)
    step = 1 << 24
    with np.errstate(over="ignore"):
        for lo in range(0, n_probe, step):
            i = np.arange(lo, min(n_probe, lo + step), dtype=np.uint64)
            k = G.np_gen_r(42, 1, 0, i) % np.uint64(hit_mod)
            v = G.np_gen_r(42, 1, 1, i) % np.uint64(1 << 62)
            m = k < np.uint64(n_build)
            k, v = k[m], v[m]
            vb = G.np_splitmix64(k ^ np.uint64(0xABCDEF))
            h = G.np_rowhash([k, v, k, vb])
            cnt += int(m.sum())
            s = s + h.sum(dtype=np.uint64)
            x = x ^ np.bitwise_xor.reduce(h) if len(h) else x
    return cnt, int(s), int(x)


def test_join_device_resident_1e6_vs_oracle_and_property(ctx, orc):
    nb, npr, mod = 1_000_000, 3_000_000, 1_250_000  # hit ratio 0.8
    got = _device_join(ctx, nb, npr, mod)
    assert got == _expected_device_join(nb, npr, mod)
    # and the oracle agrees on a regenerated 1e5-row prefix of the same generators
    nb2, np2, mod2 = 100_000, 200_000, 125_000
    bk, _ = orc.gen_column(G.gen_spec(abi.GEN_AFFINE, table=2, a=2654435761, b=12345, m=nb2), nb2)
    bv, _ = orc.gen_column(G.gen_spec(abi.GEN_HASH_OF_COL, table=2, b=0xABCDEF), nb2, src=bk)
    pk, _ = orc.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=0, m=mod2), np2)
    pv, _ = orc.gen_column(G.gen_spec(abi.GEN_RAND_MOD, table=1, col=1, m=1 << 62), np2)
    build = Chunk([Column(abi.I64, bk.view(np.int64)), Column(abi.I64, bv.view(np.int64))])
    probe = Chunk([Column(abi.I64, pk.view(np.int64)), Column(abi.I64, pv.view(np.int64))])
    cfg = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [0], abi.JOIN_INNER, 1)
    ocnt, _, _, osum, oxor = orc.hash_join_timed(cfg, build, probe, 4)
    assert _device_join(ctx, nb2, np2, mod2) == (ocnt, osum, oxor) == _expected_device_join(nb2, np2, mod2)


def test_join_full_size_1e8_by_1e8_property(ctx):
    # BASELINE headline size (1e8 ⋈ 1e8, single MI355X, build side resident in HBM): count and
    # order-independent checksum must equal the closed-form expectation.
    n = 100_000_000
    assert _device_join(ctx, n, n, n) == _expected_device_join(n, n, n)


@pytest.mark.parametrize("jt", [abi.JOIN_INNER, abi.JOIN_LEFT_OUTER])
def test_host_chunks_overlapped_copies_same_rows_as_one_stream(ctx, orc, jt):
    # round 6 (TSQ_KNOB_HOST_OVERLAP, default 1): the D2H copies of a result batch run on the operator's copy stream beside the next batch,
    # a flush waits for its H2D copies only, a pull before probe_finish may answer "no rows yet".  Many small batches in flight, ragged
    # chunk and pull sizes, NULL keys and payloads: the rows are the oracle's, with the knob on and off
    rng = np.random.default_rng(61)
    t = [abi.I64, abi.I64]
    left, right = _rand_table(rng, 60_000, 9000, t), _rand_table(rng, 20_000, 9000, t)
    cfg = H.join_cfg(t, t, [0], [0], jt, 1, probe_batch_rows=4096)
    want = orc.hash_join(cfg, right, left)
    for knob in (1, 0):
        ctx.set_knob(abi.KNOB_HOST_OVERLAP, knob)
        try:
            for cr, pr in ((1024, 1024), (333, 4000), (5000, 100)):
                assert H.rows_equal_unordered(G.run_join(ctx, cfg, right, left, chunk_rows=cr, pull_rows=pr), want)
        finally:
            ctx.set_knob(abi.KNOB_HOST_OVERLAP)


def test_host_pull_borrows_the_pinned_result_batch(ctx, orc):
    # TSQ_COL_BORROW on HOST output columns (round 6): tsq_join_pull sets data / null_bitmap to point into the operator's pinned result batch
    # (valid until the next pull) instead of copying into the caller's chunk; same rows as the oracle's, NULL payloads included
    rng = np.random.default_rng(62)
    t = [abi.I64, abi.I64]
    left, right = _rand_table(rng, 30_000, 5000, t), _rand_table(rng, 10_000, 5000, t)
    cfg = H.join_cfg(t, t, [0], [0], abi.JOIN_LEFT_OUTER, 1, probe_batch_rows=8192)
    want = orc.hash_join(cfg, right, left)
    lib = ctx.lib
    h = C.c_void_p()
    _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        G.push_chunked(lib.tsq_join_build_push, h, right, 1024)
        _lib.check(lib.tsq_join_build_finish(h), h)
        got = [[] for _ in range(4)]
        gotnn = [[] for _ in range(4)]

        def pull_all():
            while True:
                oc = (abi.Col * 4)()
                for i in range(4):
                    oc[i].flags = abi.COL_BORROW
                n, eos = C.c_int64(0), C.c_int32(0)
                _lib.check(lib.tsq_join_pull(h, oc, 4, 1024, C.byref(n), C.byref(eos)), h)
                if n.value == 0:
                    return
                for i in range(4):
                    assert oc[i].data and oc[i].length == n.value
                    got[i].append(np.ctypeslib.as_array(C.cast(oc[i].data, C.POINTER(C.c_int64)), (n.value,)).copy())
                    if oc[i].null_bitmap:
                        bits = np.ctypeslib.as_array(C.cast(oc[i].null_bitmap, C.POINTER(C.c_uint8)), ((n.value + 7) // 8,)).copy()
                        gotnn[i].append(np.unpackbits(bits, bitorder="little")[:n.value].astype(bool))
                    else:
                        gotnn[i].append(np.ones(n.value, bool))
        for lo in range(0, left.NumRows(), 1000):
            part = left.slice(lo, min(left.NumRows(), lo + 1000))
            keep = []
            _lib.check(lib.tsq_join_probe_push(h, G.make_cols(part.columns, keep), 2, part.NumRows(), None), h)
            pull_all()
        _lib.check(lib.tsq_join_probe_finish(h), h)
        pull_all()
        out = Chunk([Column(abi.I64, np.concatenate(got[i]), np.concatenate(gotnn[i])) for i in range(4)])
        assert H.rows_equal_unordered(out, want)
    finally:
        lib.tsq_join_destroy(h)


def test_outer_join_conditions_over_a_hot_build_key(ctx, orc):
    # k_outer_segments walked an outer row's whole candidate segment with one lane.  A build key with 20 000 duplicates (the
    # packed route hands such a side back at 256): outer rows of that key whose condition passes for exactly one candidate — the first, one
    # in the middle, the very last — or for none (the padded row), next to ordinary short segments.  Lengths around the 32-row hand-over
    # to the wave and around its 64-row steps too.
    rng = np.random.default_rng(71)
    t = [abi.I64, abi.I64]
    for dup in (20_000, 31, 32, 33, 95, 96, 97, 160):
        bk = np.concatenate([np.full(dup, 7, np.int64), rng.integers(100, 400, 3000)])
        bv = np.concatenate([np.arange(dup, dtype=np.int64), rng.integers(0, 50, 3000)])
        right = Chunk([Column(abi.I64, bk), Column(abi.I64, bv)])
        # l.v + r.v == dup - 1 ... for key 7: l.v = dup - 1 matches r.v = 0 (first), l.v = 0 matches r.v = dup - 1 (last), l.v = -5 matches none
        lk = np.concatenate([np.full(6, 7, np.int64), rng.integers(100, 400, 2000)])
        lv = np.concatenate([np.array([dup - 1, 0, dup // 2, -5, dup + 7, 1], np.int64), rng.integers(0, 50, 2000)])
        perm = rng.permutation(len(lk))
        left = Chunk([Column(abi.I64, lk[perm]), Column(abi.I64, lv[perm])])
        keep = []
        conds = [E.ScalarFunction("eq", E.ScalarFunction("plus", E.Column(1, abi.I64), E.Column(3, abi.I64)), E.Constant(dup - 1))]
        for jt in (abi.JOIN_LEFT_OUTER, abi.JOIN_INNER):
            cfg = H.join_cfg(t, t, [0], [0], jt, 1, conds, (), keep)
            want = orc.hash_join(cfg, right, left)
            got = G.run_join(ctx, cfg, right, left)
            assert H.rows_equal_unordered(got, want), (dup, jt)
