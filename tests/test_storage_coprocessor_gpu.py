"""GPU parity of the storage side's coprocessor path (SURVEY.md §8 f rank 4) through the C-ABI:
  * tsq_rowkeys_decode / tsq_rowkeys_encode against the oracle's restatement of tablecodec (pinned on tablecodec_test.go): sizes
    around the tile and wave boundaries, host and device placement at every pointer alignment, keys cut by offsets (the length
    test of DecodeRowKey), the first bad key in scan order, the argument contract, a 1e7-key round trip;
  * the executor chain of store/mockstore/mocktikv (tableScanExec -> selectionExec -> hashAggExec | topNExec | limitExec ->
    fillUpData4SelectResponse) against the oracle's row-at-a-time restatement of the same executors: orc.cop_hash_agg (aggregate.go
    + expression/aggregation, pinned on aggregation_test.go), orc.filter_eval (evalBool), orc.sort_rows (topNSorter.Less), and
    byte-identical 64-row response chunks."""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import coprocessor as cop
from tinysql_amd import distsql
from tinysql_amd import expression as E
from tinysql_amd import rowcodec as RC
from tinysql_amd import tablecodec
from tinysql_amd.chunk import Chunk, Column

from . import helpers as H

pytestmark = pytest.mark.gpu
EDGE = [0, 1, -1, 2, 255, 256, -256, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, -(1 << 32), (1 << 63) - 1, -(1 << 63), 0x0102030405060708, -0x0102030405060708]


def handles_for(rng, n):
    h = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64) >> rng.integers(0, 64, n)
    h[:min(n, len(EDGE))] = np.array(EDGE, dtype=np.int64)[:min(n, len(EDGE))]
    return h


def oracle_keys(orc, table_id, handles):
    return b"".join(orc.encode_row_key(table_id, int(h)) for h in handles)


# ------------------------------------------------------------------------------------------------ record keys
def test_reference_keys(ctx, orc):
    # tablecodec_test.go:42-53, 111-135
    assert tablecodec.EncodeRowKeyWithHandle(ctx, 1, 2) == orc.encode_row_key(1, 2)
    assert tablecodec.DecodeRowKey(ctx, orc.encode_row_key(1, 2)) == 2
    key = tablecodec.EncodeRowKeyWithHandle(ctx, 55, (1 << 32) - 1)
    assert key == orc.encode_row_key(55, (1 << 32) - 1)
    assert tablecodec.DecodeRecordKey(ctx, key) == (55, (1 << 32) - 1)
    for bad in (b"", b"abcdefghijklmnopqrstuvwxyz", b"abcdefghijklmnopqrs", key[:18], key + b"\x00", b"t" + key[1:9] + b"_i" + key[11:]):
        with pytest.raises(_lib.TsqError) as e:
            tablecodec.DecodeRecordKey(ctx, bad)
        assert e.value.status == abi.ERR_INVALID and e.value.message == "invalid key"
        with pytest.raises(ValueError):
            orc.decode_row_key(bad)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4097, 100_000])
def test_keys_against_the_oracle(ctx, orc, n):
    rng = np.random.default_rng(n)
    h = handles_for(rng, n)
    tid = int(EDGE[n % len(EDGE)])
    keys = tablecodec.EncodeRowKeysWithHandles(ctx, tid, h)
    sample = np.unique(np.concatenate([np.arange(min(n, 40)), rng.integers(0, max(n, 1), 40) % max(n, 1)]))[:n] if n else []
    for i in sample:
        assert bytes(keys[19 * i:19 * i + 19]) == orc.encode_row_key(tid, int(h[i]))
    if n <= 5000:
        assert bytes(keys) == oracle_keys(orc, tid, h)
    got, tids = tablecodec.DecodeRowKeys(ctx, keys, want_table_ids=True)
    assert (got == h).all() and (tids == tid).all()


@pytest.mark.parametrize("phase", [0, 1, 7, 8, 15])
def test_device_resident_keys_at_every_pointer_alignment(ctx, orc, phase):
    rng = np.random.default_rng(90 + phase)
    n = 5000
    h = handles_for(rng, n)
    want = np.frombuffer(oracle_keys(orc, 7, h), dtype=np.uint8)
    guard = 64
    dk = ctx.alloc(19 * n + 2 * guard + 64)
    dh = ctx.alloc(8 * n + 64)
    dh2 = ctx.alloc(8 * n + 64)
    dt = ctx.alloc(8 * n + 64)
    try:
        ctx.memset(dk, 0xEE, 19 * n + 2 * guard + 64)
        ctx.h2d(dh, h)
        kp = dk + guard + phase
        _lib.check(ctx.lib.tsq_rowkeys_encode(ctx.h, 7, C.c_void_p(dh), n, abi.COL_DEVICE, C.c_void_p(kp)), ctx.h)
        ctx.sync()
        back = np.zeros(19 * n + 2 * guard + 64, np.uint8)
        ctx.d2h(back, dk)
        assert (back[guard + phase:guard + phase + 19 * n] == want).all()
        assert (back[:guard + phase] == 0xEE).all() and (back[guard + phase + 19 * n:] == 0xEE).all()  # the neighbours' bytes are untouched
        got = C.c_int64(0)
        _lib.check(ctx.lib.tsq_rowkeys_decode(ctx.h, C.c_void_p(kp), 19 * n, None, n, abi.COL_DEVICE, C.c_void_p(dh2), C.c_void_p(dt), C.byref(got)), ctx.h)
        hh, tt = np.zeros(n, np.int64), np.zeros(n, np.int64)
        ctx.d2h(hh, dh2)
        ctx.d2h(tt, dt)
        assert got.value == n and (hh == h).all() and (tt == 7).all()
    finally:
        for p in (dk, dh, dh2, dt):
            ctx.free(p)


def test_first_bad_key_in_scan_order(ctx, orc):
    rng = np.random.default_rng(5)
    n = 3000
    h = handles_for(rng, n)
    keys = bytearray(oracle_keys(orc, 9, h))
    for bad, (at, byte) in {2500: (0, ord("x")), 1300: (9, ord("-")), 1301: (10, ord("i"))}.items():
        keys[19 * bad + at] = byte
    raw = np.frombuffer(bytes(keys), dtype=np.uint8)
    out = np.full(n, -7, np.int64)
    got = C.c_int64(0)
    st = ctx.lib.tsq_rowkeys_decode(ctx.h, raw.ctypes.data_as(C.c_void_p), raw.size, None, n, 0, out.ctypes.data_as(C.c_void_p), None, C.byref(got))
    assert st == abi.ERR_INVALID and _lib.last_error(ctx.h) == "invalid key" and got.value == 1300
    assert (out[:1300] == h[:1300]).all() and (out[1300:] == -7).all()  # the rows before the bad key were handed on, nothing after
    # keys cut by offsets: an index key (longer) inside the range fails DecodeRowKey's length test
    parts = [orc.encode_row_key(4, i) for i in range(2100)]
    parts[1999] += b"\x00"
    offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])])
    with pytest.raises(_lib.TsqError) as e:
        tablecodec.DecodeRowKeys(ctx, b"".join(parts), key_offsets=offs)
    assert e.value.message == "invalid key"
    good = [orc.encode_row_key(4, i * 5 - 3) for i in range(2100)]
    assert (tablecodec.DecodeRowKeys(ctx, b"".join(good), key_offsets=np.arange(2101) * 19) == np.arange(2100) * 5 - 3).all()


def test_key_argument_contract(ctx):
    got = C.c_int64(5)
    lib = ctx.lib
    buf = np.zeros(64, np.uint8)
    p = buf.ctypes.data_as(C.c_void_p)
    assert lib.tsq_rowkeys_decode(ctx.h, p, 38, None, 2, 0, None, None, C.byref(got)) == abi.ERR_INVALID and got.value == 0  # no handle buffer
    assert lib.tsq_rowkeys_decode(ctx.h, p, 37, None, 2, 0, p, None, C.byref(got)) == abi.ERR_INVALID                        # n_bytes != 19 n
    assert lib.tsq_rowkeys_decode(ctx.h, p, 38, None, 2, 0, p, None, None) == abi.ERR_INVALID
    assert lib.tsq_rowkeys_decode(ctx.h, None, 0, None, 0, 0, None, None, C.byref(got)) == abi.OK and got.value == 0         # an empty scan
    assert lib.tsq_rowkeys_encode(ctx.h, 1, None, 3, 0, p) == abi.ERR_INVALID
    assert lib.tsq_rowkeys_encode(ctx.h, 1, None, 0, 0, None) == abi.OK


def test_full_size_key_round_trip_property(ctx):
    # 1e7 keys: encode -> decode is the identity, and the keys order like their handles (EncodeInt is memcomparable)
    n = 10_000_000
    rng = np.random.default_rng(3)
    h = np.sort(rng.integers(-(1 << 62), 1 << 62, n))
    keys = tablecodec.EncodeRowKeysWithHandles(ctx, 12, h)
    assert (tablecodec.DecodeRowKeys(ctx, keys) == h).all()
    k = keys.reshape(n, 19)
    assert (k[:, 0] == ord("t")).all() and (k[:, 9] == ord("_")).all() and (k[:, 10] == ord("r")).all()
    be = k[:, 11:19].astype(np.uint64)
    hv = np.zeros(n, np.uint64)
    for i in range(8):
        hv = (hv << np.uint64(8)) | be[:, i]
    assert (np.diff(hv.astype(np.float64)) >= 0).all() and (hv[1:] >= hv[:-1]).all()


# ------------------------------------------------------------------------------------------------ the executor chain
COLS = [RC.ColInfo(1, RC.TypeLonglong), RC.ColInfo(2, RC.TypeLonglong), RC.ColInfo(3, RC.TypeDouble), RC.ColInfo(-1, RC.TypeLonglong, 0, True)]
TYPES = [abi.I64, abi.I64, abi.F64, abi.I64]


def make_table(orc, rng, n, groups=500, null_frac=0.05, unique_v=False):
    """KV pairs of a table (k int, v int, f double) with an int handle: keys in handle order (a scan returns them sorted)."""
    handles = np.sort(rng.choice(np.arange(-5 * n, 5 * n), n, replace=False)).astype(np.int64)
    v = rng.permutation(n).astype(np.int64) - n // 2 if unique_v else rng.integers(-99, 99, n)
    table = Chunk([Column(abi.I64, rng.integers(0, groups, n), rng.random(n) >= null_frac), Column(abi.I64, v, None if unique_v else rng.random(n) >= null_frac),
                   Column(abi.F64, np.round(rng.standard_normal(n) * 100, 2), rng.random(n) >= null_frac)])
    vals, offs = orc.rowcodec_encode(table, [1, 2, 3])
    keys = oracle_keys(orc, 41, handles)
    scanned = Chunk(table.columns + [Column(abi.I64, handles, None)])  # what tableScanExec hands on: the three columns + the handle
    return (keys, vals, offs), scanned


def select(chunk, keep):
    from tinysql_amd.chunk import StrColumn
    cols = []
    for c in chunk.columns:
        if c.tp == abi.BYTES:
            vals = c.values()
            idx = range(len(vals))[keep] if isinstance(keep, slice) else np.flatnonzero(keep)
            cols.append(StrColumn([vals[i] for i in idx]))
        else:
            cols.append(Column(c.tp, c.data[keep], None if c.notnull is None else c.notnull[keep]))
    return Chunk(cols)


def read_back(ctx, resp, types):
    """the SQL side reads the response chunks back (readRowsData): host rows"""
    assert resp.Error is None, resp.Error
    back = distsql.SelectResult(ctx, resp.Chunks, types)
    rows = []
    while True:
        c = back.Next(1024)
        if c.NumRows() == 0:
            return rows
        rows += c.rows()


def chunks_of(orc, chunk):
    """fillUpData4SelectResponse by the oracle: EncodeValue of every row, 64 rows per chunk"""
    n = chunk.NumRows()
    return [bytes(orc.encode_rows(select(chunk, slice(lo, min(lo + 64, n))))) for lo in range(0, n, 64)]


def test_scan_selection_aggregate(ctx, orc):
    # SELECT COUNT(*), SUM(v), AVG(v), MAX(f), MIN(v), COUNT(f), k FROM t WHERE v > -50 AND f < 120.5 GROUP BY k, pushed down
    rng = np.random.default_rng(11)
    pairs, scanned = make_table(orc, rng, 70_000)
    conds = [E.ScalarFunction("gt", E.Column(1, abi.I64), E.Constant(-50)), E.ScalarFunction("lt", E.Column(2, abi.F64), E.Constant(120.5))]
    funcs = [(abi.AGG_COUNT, -1), (abi.AGG_SUM, 1), (abi.AGG_AVG, 1), (abi.AGG_MAX, 2), (abi.AGG_MIN, 1), (abi.AGG_COUNT, 2)]
    dag = [("TableScan", COLS), ("Selection", conds), ("Aggregation", funcs, [0])]
    out_types = [abi.I64, abi.I64, abi.I64, abi.I64, abi.F64, abi.I64, abi.I64, abi.I64]
    resp = cop.handleCopDAGRequest(ctx, dag, list(range(8)), pairs)
    got = read_back(ctx, resp, out_types)
    assert resp.OutputCounts == [70_000]
    keep, _, _ = orc.filter_eval(E.compile_list(conds), 2, scanned)
    cfg = H.agg_cfg(TYPES, [0], [(f, c, TYPES[c] if c >= 0 else abi.I64) for f, c in funcs])
    want = orc.cop_hash_agg(cfg, select(scanned, keep))
    assert len(got) == want.NumRows() == 501 and H.rows_equal_unordered(got, want)
    assert all(len(c) > 0 for c in resp.Chunks) and len(resp.Chunks) == (501 + 63) // 64


def test_scan_aggregate_without_group_by_and_sum_of_doubles(ctx, orc):
    rng = np.random.default_rng(12)
    pairs, scanned = make_table(orc, rng, 30_000)
    funcs = [(abi.AGG_COUNT, 1), (abi.AGG_SUM, 2), (abi.AGG_AVG, 2), (abi.AGG_MAX, 3), (abi.AGG_MIN, 3)]
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", COLS), ("Aggregation", funcs, [])], list(range(6)), pairs)
    got = read_back(ctx, resp, [abi.I64, abi.F64, abi.I64, abi.F64, abi.I64, abi.I64])
    cfg = H.agg_cfg(TYPES, [], [(f, c, TYPES[c]) for f, c in funcs])
    want = orc.cop_hash_agg(cfg, scanned).rows()
    assert len(got) == 1 and len(want) == 1
    g, w = got[0], want[0]
    assert (g[0], g[2], g[4], g[5]) == (w[0], w[2], w[4], w[5])  # counts and the handle range: exact
    f = scanned.columns[2]
    bound = 2 * g[2] * 2.0 ** -53 * float(np.abs(f.data[f.notnull]).sum())  # 2 n eps sum|v| for a reordered double sum
    assert abs(g[1] - w[1]) <= bound and abs(g[3] - w[3]) <= bound


def test_scan_selection_topn(ctx, orc):
    # ORDER BY v DESC, handle LIMIT 300 below a selection; v is unique, so the order is total and the rows must be identical
    rng = np.random.default_rng(13)
    pairs, scanned = make_table(orc, rng, 50_000, unique_v=True)
    conds = [E.ScalarFunction("ne", E.Column(0, abi.I64), E.Constant(7))]
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", COLS), ("Selection", conds), ("TopN", [1, 3], [True, False], 300)], [1, 3, 0, 2], pairs)
    keep, _, _ = orc.filter_eval(E.compile_list(conds), 1, scanned)
    src = select(scanned, keep)
    want = select(orc.sort_rows(src, [1, 3], [True, False]), slice(0, 300))
    want = Chunk([want.columns[i] for i in (1, 3, 0, 2)])
    assert resp.Error is None and resp.Chunks == chunks_of(orc, want)  # same rows, same order, same 64-row cut, byte for byte
    # ties on the ORDER BY key: which of the tied rows survive is the heap's business in the reference; the KEYS are decided
    pairs, scanned = make_table(orc, rng, 20_000, groups=40)
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", COLS), ("TopN", [0], [False], 1000)], [0], pairs)
    got = read_back(ctx, resp, [abi.I64])
    want = select(orc.sort_rows(scanned, [0], [False]), slice(0, 1000))
    assert [r[0] for r in got] == [r[0] for r in want.rows()]


@pytest.mark.parametrize("limit", [0, 1, 64, 65, 1000, 10 ** 9])
def test_scan_limit_and_the_64_row_cut(ctx, orc, limit):
    rng = np.random.default_rng(14)
    pairs, scanned = make_table(orc, rng, 3000)
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", COLS), ("Limit", limit)], [3, 0, 1, 2], pairs)
    n = min(limit, 3000)
    want = select(Chunk([scanned.columns[i] for i in (3, 0, 1, 2)]), slice(0, n))
    assert resp.Error is None and resp.Chunks == chunks_of(orc, want)
    assert len(resp.Chunks) == (n + 63) // 64


def test_the_cut_does_not_depend_on_the_batches_rows_arrive_in(ctx, orc):
    # a selection hands on one compacted chunk per scan batch: their sizes are arbitrary, the response still changes chunk every 64 rows
    rng = np.random.default_rng(15)
    pairs, scanned = make_table(orc, rng, 10_000, null_frac=0.0)
    conds = [E.ScalarFunction("lt", E.Column(1, abi.I64), E.Constant(30))]
    scan = cop.tableScanExec(ctx, COLS, *pairs, batch_rows=1000)
    e = cop.selectionExec(ctx, scan, conds)
    chunks, rows, sizes = [], 0, []
    e.Open()
    try:
        while True:
            chk = e.Next()
            if chk.NumRows() == 0:
                break
            sizes.append(chk.NumRows())
            chunks, rows = cop.fillUpData4SelectResponse(ctx, chk, [0, 1, 2, 3], chunks, rows)
    finally:
        e.Close()
    keep, _, _ = orc.filter_eval(E.compile_list(conds), 1, scanned)
    assert len(sizes) == 10 and any(s % 64 for s in sizes) and rows == int(keep.sum())
    assert chunks == chunks_of(orc, select(scanned, keep))


def test_limit_above_a_selection_takes_the_first_rows_in_scan_order(ctx, orc):
    # selectionExec hands rows on in the order it got them (executor.go:360-390), limitExec keeps the first `limit` (:472-507):
    # tsq_chunk_compact must preserve the row order across waves and workgroups
    rng = np.random.default_rng(17)
    pairs, scanned = make_table(orc, rng, 200_000)
    conds = [E.ScalarFunction("ge", E.Column(1, abi.I64), E.Constant(0))]
    keep, _, _ = orc.filter_eval(E.compile_list(conds), 1, scanned)
    sel = select(scanned, keep)
    for limit in (100, 70_001):
        resp = cop.handleCopDAGRequest(ctx, [("TableScan", COLS), ("Selection", conds), ("Limit", limit)], [3, 1], pairs)
        want = select(Chunk([sel.columns[3], sel.columns[1]]), slice(0, limit))
        assert resp.Error is None and resp.Chunks == chunks_of(orc, want)


def test_errors_become_the_response_error(ctx, orc):
    rng = np.random.default_rng(16)
    (keys, vals, offs), _ = make_table(orc, rng, 2000)
    bad = bytearray(keys)
    bad[19 * 700 + 10] = ord("i")  # an index key in the scanned range
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", COLS), ("Limit", 10)], [0], (bytes(bad), vals, offs))
    assert resp.Error == "invalid key" and resp.Chunks == []
    cut = np.array(vals, dtype=np.uint8).copy()
    cut[offs[900]] = 127  # not the codec version
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", COLS), ("Aggregation", [(abi.AGG_COUNT, -1)], [])], [0], (keys, cut, offs))
    assert resp.Error == "invalid codec version" and resp.Chunks == []
    with pytest.raises(_lib.TsqError):
        cop.buildDAG(ctx, [("IndexScan",)], (keys, vals, offs))


def test_running_sum_overflow_divergence_is_pinned(ctx, orc):
    # the reference's executor is sequential: SUM(int64) fails on the first RUNNING sum outside BIGINT (calculateSum -> AddInt64),
    # so [max, 1, -5] is an error and [max, -5, 1] is not.  The GPU adds exactly in 128 bits and checks the final sum: both orders
    # succeed with max - 4 (DESIGN.md "Known, documented divergences").  A final sum outside BIGINT is an error on both sides.
    m = (1 << 63) - 1
    for vals, oracle_fails, gpu_fails in (([m, 1, -5], True, False), ([m, -5, 1], False, False), ([m, 1, 1], True, True)):
        table = Chunk([Column(abi.I64, np.zeros(3, np.int64), None), Column(abi.I64, np.array(vals, dtype=np.int64), None), Column(abi.F64, np.zeros(3), None)])
        v, o = orc.rowcodec_encode(table, [1, 2, 3])
        pairs = (oracle_keys(orc, 1, [1, 2, 3]), v, o)
        resp = cop.handleCopDAGRequest(ctx, [("TableScan", COLS), ("Aggregation", [(abi.AGG_SUM, 1)], [])], [0], pairs)
        cfg = H.agg_cfg(TYPES, [], [(abi.AGG_SUM, 1, abi.I64)])
        scanned = Chunk(table.columns + [Column(abi.I64, np.array([1, 2, 3]), None)])
        if oracle_fails:
            with pytest.raises(orc.OracleError):
                orc.cop_hash_agg(cfg, scanned)
        else:
            assert orc.cop_hash_agg(cfg, scanned).rows() == [(m - 4,)]
        if gpu_fails:
            assert resp.Error is not None and resp.Chunks == []
        else:
            assert read_back(ctx, resp, [abi.I64]) == [(m - 4,)]


# ------------------------------------------------------------------------------------------------ string columns through the chain
def make_string_table(orc, rng, n):
    from tinysql_amd.chunk import StrColumn
    handles = np.arange(n, dtype=np.int64) * 3 - n
    seg = [None if rng.random() < 0.05 else [b"BUILDING", b"AUTOMOBILE", b"MACHINERY", b"", b"HOUSEHOLD"][int(rng.integers(0, 5))] for _ in range(n)]
    note = [None if rng.random() < 0.1 else b"note-%d-" % i + b"x" * int(rng.integers(0, 30)) for i in range(n)]
    table = Chunk([Column(abi.I64, rng.integers(0, 50, n), rng.random(n) >= 0.05), StrColumn(seg), StrColumn(note)])
    vals, offs = orc.rowcodec_encode(table, [1, 2, 3])
    return (oracle_keys(orc, 7, handles), vals, offs), Chunk(table.columns + [Column(abi.I64, handles, None)])


SCOLS = [RC.ColInfo(1, RC.TypeLonglong), RC.ColInfo(2, RC.TypeVarchar), RC.ColInfo(3, RC.TypeBlob), RC.ColInfo(-1, RC.TypeLonglong, 0, True)]
STYPES = [abi.I64, abi.BYTES, abi.BYTES, abi.I64]


def test_string_columns_scan_selection_limit_response(ctx, orc):
    # WHERE seg = 'BUILDING' AND k < 40 over a table with varchar / blob columns, LIMIT 500: string cells stay in HBM from the stored
    # rows to the response bytes (compact-bytes datums); the rows come back in scan order
    rng = np.random.default_rng(21)
    pairs, scanned = make_string_table(orc, rng, 30_000)
    conds = [E.ScalarFunction("eq", E.Column(1, abi.BYTES), E.Constant("BUILDING")), E.ScalarFunction("lt", E.Column(0, abi.I64), E.Constant(40))]
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", SCOLS), ("Selection", conds), ("Limit", 500)], [3, 1, 2, 0], pairs)
    keep, _, _ = orc.filter_eval(E.compile_list(conds), 2, scanned)
    sel = select(scanned, keep)
    want = select(Chunk([sel.columns[i] for i in (3, 1, 2, 0)]), slice(0, 500))
    assert resp.Error is None and resp.Chunks == chunks_of(orc, want)
    back = distsql.decode_chunks(ctx, resp.Chunks, [abi.I64, abi.BYTES, abi.BYTES, abi.I64])
    assert back.rows() == want.rows()


def test_string_group_keys_and_topn_payload_through_the_chain(ctx, orc):
    rng = np.random.default_rng(22)
    pairs, scanned = make_string_table(orc, rng, 20_000)
    # GROUP BY seg: count(*), max(note), sum(k) — the group-by value (a string) follows the partial results
    funcs = [(abi.AGG_COUNT, -1), (abi.AGG_MAX, 2), (abi.AGG_SUM, 0)]
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", SCOLS), ("Aggregation", funcs, [1])], [0, 1, 2, 3], pairs)
    got = distsql.decode_chunks(ctx, resp.Chunks, [abi.I64, abi.BYTES, abi.I64, abi.BYTES])
    cfg = H.agg_cfg(STYPES, [1], [(f, c, STYPES[c] if c >= 0 else abi.I64) for f, c in funcs])
    assert resp.Error is None and H.rows_equal_unordered(got, orc.cop_hash_agg(cfg, scanned))
    # ORDER BY handle DESC LIMIT 100 with both string columns as payload
    resp = cop.handleCopDAGRequest(ctx, [("TableScan", SCOLS), ("TopN", [3], [True], 100)], [3, 2, 1], pairs)
    want = select(orc.sort_rows(scanned, [3], [True]), slice(0, 100))
    assert resp.Error is None and resp.Chunks == chunks_of(orc, Chunk([want.columns[i] for i in (3, 2, 1)]))


def test_table_scan_with_a_bit_column(ctx, orc):
    # a BIT(12) column comes out of tableScanExec as its two-byte binary literal (decoder.go:229-231) and travels like any string column
    rng = np.random.default_rng(21)
    n = 3000
    handles = np.arange(n, dtype=np.int64) * 2 - 100
    flags = rng.integers(0, 1 << 12, n).astype(np.uint64)
    table = Chunk([Column(abi.U64, flags, rng.random(n) >= 0.1), Column(abi.I64, rng.integers(-9, 9, n))])
    vals, offs = orc.rowcodec_encode(table, [1, 2])
    cols = [RC.ColInfo(1, RC.TypeBit, Flen=12), RC.ColInfo(2, RC.TypeLonglong), RC.ColInfo(-1, RC.TypeLonglong, 0, True)]
    scan = cop.tableScanExec(ctx, cols, oracle_keys(orc, 41, handles), vals, offs, batch_rows=1024)
    from tinysql_amd.gpu_pipeline import drain_device
    got = [r for c in drain_device(scan) for r in c.rows()]
    nn = table.columns[0].notnull
    want = [(int(f).to_bytes(2, "big") if ok else None, v, h) for f, ok, v, h in zip(flags.tolist(), nn.tolist(), table.columns[1].values(), handles.tolist())]
    assert got == want


def test_chunk_encoded_response_of_the_same_plan(ctx, orc):
    # the plan of test_string_columns_scan_selection_limit_response answering in the chunk wire format (one chunk.Codec buffer per
    # batch of the DAG's output): every buffer equals chunk.Codec.Encode of those rows (the oracle's restatement) and chunk.Decoder
    # gives the rows back — the same rows the datum-row response carries
    from tinysql_amd import chunk_codec as CC
    rng = np.random.default_rng(21)
    pairs, scanned = make_string_table(orc, rng, 30_000)
    conds = [E.ScalarFunction("eq", E.Column(1, abi.BYTES), E.Constant("BUILDING")), E.ScalarFunction("lt", E.Column(0, abi.I64), E.Constant(40))]
    plan = [("TableScan", SCOLS), ("Selection", conds)]
    resp = cop.handleCopDAGRequest(ctx, plan, [3, 1, 2, 0], pairs, encodeType="chunk")
    keep, _, _ = orc.filter_eval(E.compile_list(conds), 2, scanned)
    sel = select(scanned, keep)
    want = Chunk([sel.columns[i] for i in (3, 1, 2, 0)])
    types = [abi.I64, abi.BYTES, abi.BYTES, abi.I64]
    assert resp.Error is None and len(resp.Chunks) >= 1
    dec = CC.Decoder(ctx, CC.WireChunk(types), types)
    rows, at = [], 0
    for buf in resp.Chunks:
        dec.Reset(buf)
        n = dec.RemainedRows()
        assert buf == orc.WireChunk.from_chunk(select(want, slice(at, at + n))).encode()  # byte for byte Codec.Encode of these rows
        while not dec.IsFinished():
            chk = CC.WireChunk(types, 1024)
            dec.Decode(chk)
            rows += chk.to_chunk().rows()
        at += n
    dec.Close()
    assert rows == want.rows() and at == want.NumRows()
