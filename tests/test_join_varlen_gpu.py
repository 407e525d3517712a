"""GPU: var-len (TSQ_BYTES, string) PAYLOAD columns through tsq_join_* — the reference's own join benchmark schema is
(bigint key, ..., varstring payload) (executor/benchmark_test.go:328-407) — against the oracle's join (Chunk.AppendRow of var-len
cells, util/chunk/chunk.go:334-356): host chunks of tidb_max_chunk_size rows through pinned staging, big pushes, inner and outer
joins (NULL padding of a var-len column), NULL and empty strings, long payloads (one wave per cell), duplicates on the build side."""
import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column, StrColumn

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu


def strs(rng, n, maxlen, null_p=0.1):
    out = []
    for _ in range(n):
        if rng.random() < null_p:
            out.append(None)
        else:
            out.append(bytes(rng.integers(0, 256, int(rng.integers(0, maxlen + 1)), dtype=np.uint8)))
    return out


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1), (abi.JOIN_RIGHT_OUTER, 0)])
@pytest.mark.parametrize("chunk_rows", [1024, 1 << 22])
def test_join_with_string_payloads_vs_oracle(ctx, orc, jt, inner, chunk_rows):
    rng = np.random.default_rng(300 + jt + chunk_rows % 7)
    nl, nr = 9000, 6000
    left = Chunk([Column(abi.I64, rng.integers(0, 4000, nl), rng.random(nl) > 0.05), StrColumn(strs(rng, nl, 24)), Column(abi.F64, rng.random(nl))])
    right = Chunk([StrColumn(strs(rng, nr, 12)), Column(abi.I64, rng.integers(0, 4000, nr), rng.random(nr) > 0.05), StrColumn(strs(rng, nr, 40, 0.3))])
    cfg = H.join_cfg(left.types(), right.types(), [0], [1], jt, inner)
    build, probe = (right, left) if inner == 1 else (left, right)
    want = orc.hash_join(cfg, build, probe)
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=chunk_rows, pull_rows=777)
    assert got.NumRows() == want.NumRows() and H.rows_equal_unordered(got, want)


def test_benchmark_schema_bigint_key_and_5k_payload(ctx, orc):
    # executor/benchmark_test.go:328: columns (bigint, ..., varstring of 5 KiB); one wave copies one cell
    rng = np.random.default_rng(11)
    n = 3000
    pay = [bytes(rng.integers(0, 256, 5 << 10, dtype=np.uint8)) for _ in range(40)]
    build = Chunk([Column(abi.I64, rng.permutation(n).astype(np.int64)), StrColumn([pay[i % 40] + bytes([i % 251]) for i in range(n)])])
    probe = Chunk([Column(abi.I64, rng.integers(0, n + 500, 2 * n)), StrColumn([pay[(7 * i) % 40][: 100 + i % 900] for i in range(2 * n)])])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe)
    got = G.run_join(ctx, cfg, build, probe, chunk_rows=1024, pull_rows=500)
    assert got.NumRows() == want.NumRows() > n and H.rows_equal_unordered(got, want)


@pytest.mark.parametrize("jt,inner", [(abi.JOIN_INNER, 1), (abi.JOIN_LEFT_OUTER, 1)])
def test_string_join_keys_vs_oracle(ctx, orc, jt, inner):
    # a string key cell is (compactBytesFlag, bytes) (util/codec/codec.go:233-235); equality = bytes.Equal (:363-382).  Single string key,
    # and (string, bigint) composite key; NULL keys never join; keys that differ only in length / in a late byte / that share 8-byte prefixes
    rng = np.random.default_rng(21 + jt)
    words = [b"", b"a", b"ab", b"abcdefgh", b"abcdefghi", b"abcdefgh\x00", b"abcdefgX", b"zz" * 20, b"zz" * 20 + b"!", None]
    nl, nr = 2400, 1500  # (round 5: ~3.7e5 joined rows with string cells per case instead of 1.5e6 — Python tuples of them were 18 s per case)
    lk = [words[i] for i in rng.integers(0, len(words), nl)]
    rk = [words[i] for i in rng.integers(0, len(words) - 3, nr)] + []
    left = Chunk([StrColumn(lk), Column(abi.I64, rng.integers(0, 4, nl)), Column(abi.I64, np.arange(nl))])
    right = Chunk([StrColumn(rk), Column(abi.I64, rng.integers(0, 4, nr)), StrColumn(strs(rng, nr, 6))])
    for lkeys, rkeys in (([0], [0]), ([0, 1], [0, 1])):
        cfg = H.join_cfg(left.types(), right.types(), lkeys, rkeys, jt, inner)
        want = orc.hash_join(cfg, right, left)
        got = G.run_join(ctx, cfg, right, left, chunk_rows=1024, pull_rows=4096)
        assert got.NumRows() == want.NumRows() > nl and H.rows_equal_unordered(got, want)
        assert G.run_join(ctx, cfg, right, left, chunk_rows=1 << 20, count_only=True) == want.NumRows()


def test_string_key_against_number_key_never_matches(ctx, orc):
    left = Chunk([StrColumn(["1", "2", None]), Column(abi.I64, np.arange(3))])
    right = Chunk([Column(abi.I64, np.array([1, 2, 3])), Column(abi.I64, np.arange(3))])
    cfg = H.join_cfg(left.types(), right.types(), [0], [0], abi.JOIN_LEFT_OUTER, 1)
    want = orc.hash_join(cfg, right, left)
    got = G.run_join(ctx, cfg, right, left)
    assert got.NumRows() == want.NumRows() == 3 and H.rows_equal_unordered(got, want)


def test_count_and_empty_inputs_with_string_columns(ctx, orc):
    rng = np.random.default_rng(5)
    build = Chunk([Column(abi.I64, np.arange(100)), StrColumn(strs(rng, 100, 8))])
    probe = Chunk([Column(abi.I64, rng.integers(0, 150, 1000)), StrColumn(strs(rng, 1000, 8))])
    cfg = H.join_cfg(probe.types(), build.types(), [0], [0], abi.JOIN_INNER, 1)
    want = orc.hash_join(cfg, build, probe)
    assert G.run_join(ctx, cfg, build, probe, count_only=True) == want.NumRows()
    empty = Chunk([Column(abi.I64, np.zeros(0, np.int64)), StrColumn([])])
    assert G.run_join(ctx, cfg, empty, probe).NumRows() == 0
    assert G.run_join(ctx, cfg, build, empty).NumRows() == 0


@pytest.mark.parametrize("n,maxlen", [(1, 5), (64, 9), (70001, 12), (3000, 700)])
def test_chunk_compact_with_a_string_column(ctx, n, maxlen):
    # SelectionExec's copy of the selected rows / Column.CopyReconstruct of a var-len column (executor.go:401-438, column.go:504-552)
    import ctypes as C
    rng = np.random.default_rng(n)
    vals = strs(rng, n, maxlen, 0.2)
    sc = StrColumn(vals)
    ints = Column(abi.I64, rng.integers(-9, 9, n), rng.random(n) > 0.3)
    sel = rng.random(n) > 0.45
    bufs = []

    def dev(arr):
        arr = np.ascontiguousarray(arr)
        p = ctx.alloc(max(arr.nbytes, 8) + 64)
        ctx.h2d(p, arr)
        bufs.append(p)
        return p

    try:
        cin = (abi.Col * 2)()
        cin[0].data, cin[0].offsets, cin[0].null_bitmap = dev(sc.data), dev(sc.offsets), dev(sc.bitmap())
        cin[0].length, cin[0].elem_size, cin[0].type, cin[0].flags = n, -1, abi.BYTES, abi.COL_DEVICE
        cin[1].data, cin[1].null_bitmap = dev(ints.data), dev(ints.bitmap())
        cin[1].length, cin[1].elem_size, cin[1].type, cin[1].flags = n, 8, abi.I64, abi.COL_DEVICE
        cout = (abi.Col * 2)()
        cout[0].data, cout[0].offsets, cout[0].null_bitmap = dev(np.zeros(len(sc.data) + 8, np.uint8)), dev(np.zeros(n + 1, np.int64)), dev(np.zeros(n // 8 + 8, np.uint8))
        cout[0].length, cout[0].elem_size, cout[0].type, cout[0].flags = n, -1, abi.BYTES, abi.COL_DEVICE
        cout[1].data, cout[1].null_bitmap = dev(np.zeros(n, np.int64)), dev(np.zeros(n // 8 + 8, np.uint8))
        cout[1].length, cout[1].elem_size, cout[1].type, cout[1].flags = n, 8, abi.I64, abi.COL_DEVICE
        flags = dev(sel.astype(np.uint8))
        m = C.c_int64(0)
        _lib.check(ctx.lib.tsq_chunk_compact(ctx.h, cin, 2, n, flags, cout, C.byref(m)), ctx.h)
        k = m.value
        assert k == int(sel.sum())
        offs, data, bm = np.zeros(k + 1, np.int64), np.zeros(len(sc.data) + 8, np.uint8), np.zeros(k // 8 + 8, np.uint8)
        iv, ibm = np.zeros(max(k, 1), np.int64), np.zeros(k // 8 + 8, np.uint8)
        ctx.d2h(offs, cout[0].offsets)
        ctx.d2h(data, cout[0].data)
        ctx.d2h(bm, cout[0].null_bitmap)
        ctx.d2h(iv, cout[1].data)
        ctx.d2h(ibm, cout[1].null_bitmap)
        nn = np.unpackbits(bm, bitorder="little")[:k].astype(bool)
        inn = np.unpackbits(ibm, bitorder="little")[:k].astype(bool)
        raw = data.tobytes()
        import collections
        got = collections.Counter((H.canon(raw[offs[i]:offs[i + 1]] if nn[i] else None), H.canon(int(iv[i]) if inn[i] else None)) for i in range(k))
        iw = ints.values()
        want = collections.Counter((H.canon(vals[i]), H.canon(iw[i])) for i in range(n) if sel[i])
        assert got == want  # (the dense rows keep the input order up to a permutation inside 256-row tiles)
    finally:
        for p in bufs:
            ctx.free(p)
