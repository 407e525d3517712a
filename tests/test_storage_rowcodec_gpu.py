"""GPU parity of tsq_rowcodec_decode (stored rows, rowcodec v2 -> chunk columns; SURVEY.md §8 f rank 4) against the oracle's
restatement of the loop around rowcodec.ChunkDecoder.DecodeToChunk (util/rowcodec/decoder.go:158-238): the reference's own test
rows (rowcodec_test.go), bit-exact values and NULL flags on random scans, tile / wave boundary sizes, large column ids and rows
too wide for the LDS tile, every alignment of a device-resident `values` pointer, host and device placement, the first error in
scan order with the reference's message, the argument contract, a table scan feeding the GPU selection + partial hash
aggregate (the mocktikv executor chain, store/mockstore/mocktikv/executor.go / aggregate.go), and a full-size round trip."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import rowcodec as RC
from tinysql_amd.chunk import Chunk, Column, chunk_from_buffers, out_buffers

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu
NP = {abi.I64: np.int64, abi.U64: np.uint64, abi.F64: np.float64, abi.F32: np.float32}
EDGE = np.array([0, 1, -1, 127, -128, 128, -129, 32767, -32768, 32768, (1 << 31) - 1, -(1 << 31), 1 << 31, (1 << 63) - 1, -(1 << 63)])


def _col(tp, vals, notnull=None):
    return Column(tp, np.array(vals, dtype=NP[tp]), None if notnull is None else np.array(notnull, dtype=bool))


def _scan(rng, n, null_frac=0.2):
    return Chunk([
        Column(abi.I64, np.where(rng.random(n) < 0.5, rng.choice(EDGE, n), rng.integers(-(1 << 62), 1 << 62, n)), rng.random(n) >= null_frac),
        Column(abi.U64, (rng.integers(0, 1 << 62, n) >> rng.integers(0, 62, n)).astype(np.uint64), rng.random(n) >= null_frac),
        Column(abi.F64, rng.standard_normal(n) * 1e6, rng.random(n) >= null_frac),
        Column(abi.F32, rng.standard_normal(n).astype(np.float32), rng.random(n) >= null_frac),
    ])


def _same(a, b):
    assert a.NumRows() == b.NumRows() and a.types() == b.types()
    for ca, cb in zip(a.columns, b.columns):
        na = np.ones(len(ca), bool) if ca.notnull is None else ca.notnull
        nb = np.ones(len(cb), bool) if cb.notnull is None else cb.notnull
        assert (na == nb).all()
        assert (ca.data.view(np.uint8).reshape(len(ca), -1)[na] == cb.data.view(np.uint8).reshape(len(cb), -1)[nb]).all()
        assert not ca.data.view(np.uint8).reshape(len(ca), -1)[~na].any()   # a NULL slot holds zero bytes (column.go:150-158)


# the scan schema of the random tests: ids 7 (int), 2 (uint), 200 (double), 31 (float) in the rows; requested in another order,
# with the handle, an absent column and an absent column with a default
IDS = [7, 2, 200, 31]
COLS = [RC.ColInfo(200, RC.TypeDouble), RC.ColInfo(-1, RC.TypeLonglong, 0, True), RC.ColInfo(7, RC.TypeLonglong), RC.ColInfo(31, RC.TypeFloat),
        RC.ColInfo(2, RC.TypeLonglong, RC.UnsignedFlag), RC.ColInfo(99, RC.TypeLong), RC.ColInfo(98, RC.TypeShort)]
SPECS = [(200, abi.F64), (-1, abi.I64, abi.RC_HANDLE), (7, abi.I64), (31, abi.F32), (2, abi.U64), (99, abi.I64), (98, abi.I64, abi.RC_HAS_DEFAULT, 5)]


def _decoder(ctx):
    return RC.NewChunkDecoder(ctx, COLS, -1, lambda i: 5 if COLS[i].ID == 98 else None)


def test_reference_rows_of_rowcodec_test(ctx, orc):
    # TestDecodeRowWithHandle (rowcodec_test.go:49-163): the handle comes from the key; signed and unsigned handle column
    b, o = orc.rowcodec_encode(Chunk([_col(abi.I64, [1])]), [10])
    for flag in (0, RC.UnsignedFlag):
        d = RC.NewChunkDecoder(ctx, [RC.ColInfo(-1, RC.TypeLonglong, flag, True), RC.ColInfo(10, RC.TypeLonglong)], -1)
        assert d.DecodeToChunk(b, o, [10000]).rows() == [(10000, 1)]
    # TestTypesNewRowCodec (:165-328): fixed-width types next to NULL columns and a string; small ids, a large id, a 65536-byte value
    chk = Chunk([_col(abi.I64, [1]), _col(abi.U64, [1]), _col(abi.F64, [2.0]), _col(abi.I64, [1999]), _col(abi.I64, [0], [False]),
                 _col(abi.I64, [0], [False]), _col(abi.I64, [0], [False]), _col(abi.F32, [6.0])])
    for first_id, pad in ((1, 3), (300, 3), (1, 65536)):
        ids = [first_id, 22, 3, 12, 11, 2, 100, 116]
        cols = [RC.ColInfo(first_id, RC.TypeLonglong), RC.ColInfo(22, RC.TypeShort, RC.UnsignedFlag), RC.ColInfo(3, RC.TypeDouble), RC.ColInfo(12, RC.TypeYear),
                RC.ColInfo(11, RC.TypeLonglong), RC.ColInfo(2, RC.TypeLonglong), RC.ColInfo(100, RC.TypeLonglong), RC.ColInfo(116, RC.TypeFloat)]
        b, o = orc.rowcodec_encode(chk, ids, 24, [pad])
        assert RC.NewChunkDecoder(ctx, cols).DecodeToChunk(b, o).rows() == [(1, 1, 2.0, 1999, None, None, None, 6.0)]
    # TestNilAndDefault (:330-438): the absent column takes its default, a stored NULL stays NULL
    b, o = orc.rowcodec_encode(Chunk([_col(abi.I64, [1])]), [1])
    cols = [RC.ColInfo(1, RC.TypeLonglong), RC.ColInfo(2, RC.TypeLonglong, RC.UnsignedFlag)]
    assert RC.NewChunkDecoder(ctx, cols, -1, lambda i: 9 if i == 1 else None).DecodeToChunk(b, o).rows() == [(1, 9)]
    assert RC.NewChunkDecoder(ctx, cols).DecodeToChunk(b, o).rows() == [(1, None)]
    b, o = orc.rowcodec_encode(Chunk([_col(abi.I64, [1]), _col(abi.U64, [0], [False])]), [1, 2])
    assert RC.NewChunkDecoder(ctx, cols, -1, lambda i: 9 if i == 1 else None).DecodeToChunk(b, o).rows() == [(1, None)]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 1000, 4097, 100_000, 1_000_000])
def test_random_scans_against_the_oracle(ctx, orc, n):
    rng = np.random.default_rng(n)
    chk = _scan(rng, n)
    handles = rng.integers(-(1 << 62), 1 << 62, n)
    b, o = orc.rowcodec_encode(chk, IDS)
    st, want = orc.rowcodec_decode(b, o, handles, SPECS)
    assert st == 0
    _same(_decoder(ctx).DecodeToChunk(b, o, handles), want)
    if n in (257, 100_000):  # the plain kernel as well
        with ctx.knobs(ROWCODEC_PIPELINE=0):
            _same(_decoder(ctx).DecodeToChunk(b, o, handles), want)


def test_large_ids_and_rows_wider_than_the_lds_tile(ctx, orc):
    rng = np.random.default_rng(3)
    n = 3000
    chk = _scan(rng, n)
    pad = np.where(np.arange(n) % 97 == 5, 70000, rng.integers(0, 40, n))  # some tiles exceed the 48 KB LDS budget
    b, o = orc.rowcodec_encode(chk, [7, 300, 200, 31], 24, pad)
    cols = [RC.ColInfo(300, RC.TypeLonglong, RC.UnsignedFlag), RC.ColInfo(7, RC.TypeLonglong), RC.ColInfo(200, RC.TypeDouble), RC.ColInfo(31, RC.TypeFloat)]
    st, want = orc.rowcodec_decode(b, o, None, [(300, abi.U64), (7, abi.I64), (200, abi.F64), (31, abi.F32)])
    assert st == 0
    _same(RC.NewChunkDecoder(ctx, cols).DecodeToChunk(b, o), want)


def _decode_device(ctx, b, o, handles, specs, n, phase=0):
    """values / offsets / handles and the output columns resident in HBM; the values start `phase` bytes into their buffer."""
    types = [s[1] for s in specs]
    dbytes, doffs = ctx.alloc(b.size + phase + 64), ctx.alloc(o.nbytes + 64)
    dh = ctx.alloc(8 * max(n, 1) + 64) if handles is not None else None
    outs = [G.DevCol(ctx, t, n, with_nulls=True) for t in types]
    try:
        if b.size:
            ctx.h2d(dbytes + phase, b)
        ctx.h2d(doffs, o)
        if handles is not None:
            ctx.h2d(dh, np.ascontiguousarray(handles, dtype=np.int64))
        m = C.c_int64(0)
        st = ctx.lib.tsq_rowcodec_decode(ctx.h, C.c_void_p(dbytes + phase), b.size, C.c_void_p(doffs), C.c_void_p(dh) if dh else None, n, abi.COL_DEVICE,
                                         len(specs), orc_cols(specs), G.dev_cols(outs), C.byref(m))
        got = Chunk([c.to_host() for c in outs])
        return st, m.value, Chunk([Column(c.tp, c.data[:m.value], None if c.notnull is None else c.notnull[:m.value]) for c in got.columns])
    finally:
        ctx.free(dbytes)
        ctx.free(doffs)
        if dh:
            ctx.free(dh)
        for c in outs:
            c.free()


def orc_cols(specs):
    arr = (abi.RowcodecCol * len(specs))()
    for i, sp in enumerate(specs):
        arr[i].col_id, arr[i].type = sp[0], sp[1]
        arr[i].flags = sp[2] if len(sp) > 2 else 0
        arr[i].def_bits = sp[3] if len(sp) > 3 else 0
    return arr


@pytest.mark.parametrize("phase", [0, 1, 7, 8, 15])
def test_device_resident_scan_at_every_pointer_alignment(ctx, orc, phase):
    rng = np.random.default_rng(40 + phase)
    n = 20_000
    chk = _scan(rng, n, null_frac=0.1)
    handles = rng.integers(-(1 << 62), 1 << 62, n)
    b, o = orc.rowcodec_encode(chk, IDS)
    st, want = orc.rowcodec_decode(b, o, handles, SPECS)
    gst, m, got = _decode_device(ctx, b, o, handles, SPECS, n, phase)
    assert st == 0 and gst == abi.OK and m == n
    _same(got, want)


MSG = {1: "invalid codec version", 2: "malformed row", 3: "insufficient bytes to decode value"}


@pytest.mark.parametrize("case", ["version", "short_float", "cut_header", "cut_value", "odd_int", "empty_value"])
def test_first_error_in_scan_order(ctx, orc, case):
    rng = np.random.default_rng(9)
    n = 5000
    chk = _scan(rng, n, null_frac=0.0)
    b, o = orc.rowcodec_encode(chk, IDS)
    specs = [(7, abi.I64), (2, abi.U64), (200, abi.F64), (31, abi.F32)]
    b = b.copy()
    at = 3517
    rows = [b[o[r]:o[r + 1]] for r in range(n)]
    if case == "version":
        b[o[at]] = 1
        b[o[at + 100]] = 1
    elif case == "short_float":
        specs = [(7, abi.F64)] + specs[1:]  # an int column read as a real: the first row whose int has fewer than 8 bytes fails
    elif case == "cut_header":
        rows[at] = rows[at][:9]
    elif case == "cut_value":
        rows[at] = rows[at][:-3]
    elif case == "odd_int":
        rows[at] = np.array([128, 0, 1, 0, 0, 0, 7, 3, 0, 1, 2, 3], dtype=np.uint8)
        specs = specs[:1]
    else:
        rows[at] = rows[at][:0]
    if case not in ("version", "short_float"):
        b = np.concatenate(rows)
        o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    st, want = orc.rowcodec_decode(b, o, None, specs)
    assert st != 0 and (case == "short_float" or want.NumRows() == at)
    types = [s[1] for s in specs]
    keep = []
    out, bufs = out_buffers(types, n, keep)
    m = C.c_int64(-1)
    gst = ctx.lib.tsq_rowcodec_decode(ctx.h, b.ctypes.data_as(C.c_void_p), b.size, o.ctypes.data_as(C.c_void_p), None, n, 0, len(specs), orc_cols(specs), out,
                                      C.byref(m))
    assert gst == abi.ERR_INVALID and _lib.last_error(ctx.h) == MSG[st] and m.value == want.NumRows()
    _same(chunk_from_buffers(types, bufs, m.value), want)
    # the device-resident form reports the same row
    gst, m2, got = _decode_device(ctx, b, o, None, specs, n)
    assert gst == abi.ERR_INVALID and m2 == want.NumRows()
    _same(got, want)


def test_argument_contract(ctx, orc):
    lib = ctx.lib
    b, o = orc.rowcodec_encode(Chunk([_col(abi.I64, [5, 6, 7])]), [1])
    keep = []
    out, bufs = out_buffers([abi.I64], 8, keep)
    n = C.c_int64(-1)
    pb, po = b.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p)
    one = orc_cols([(1, abi.I64)])
    # an empty scan batch is not an error
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, po, None, 0, 0, 1, one, out, C.byref(n)) == abi.OK and n.value == 0
    # misuse: NULL outputs / offsets, negative sizes, too many columns, var-len column type, a handle column without handles
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, po, None, 3, 0, 1, one, out, None) == abi.ERR_INVALID
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, None, None, 3, 0, 1, one, out, C.byref(n)) == abi.ERR_INVALID
    assert lib.tsq_rowcodec_decode(ctx.h, pb, -1, po, None, 3, 0, 1, one, out, C.byref(n)) == abi.ERR_INVALID
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, po, None, 3, 0, 17, one, out, C.byref(n)) == abi.ERR_UNSUPPORTED
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, po, None, 3, 0, 1, orc_cols([(1, abi.BYTES)]), out, C.byref(n)) == abi.ERR_INVALID  # a var-len column without offsets[]
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, po, None, 3, 0, 1, orc_cols([(1, abi.BYTES, abi.RC_HAS_DEFAULT)]), out, C.byref(n)) == abi.ERR_INVALID
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, po, None, 3, 0, 1, orc_cols([(1, 9)]), out, C.byref(n)) == abi.ERR_INVALID  # no such column type
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, po, None, 3, 0, 1, orc_cols([(1, abi.I64, abi.RC_HANDLE)]), out, C.byref(n)) == abi.ERR_INVALID
    # offsets that run past n_bytes are caught per row, not read
    bad = o.copy()
    bad[3] = b.size + 100
    assert lib.tsq_rowcodec_decode(ctx.h, pb, b.size, bad.ctypes.data_as(C.c_void_p), None, 3, 0, 1, one, out, C.byref(n)) == abi.ERR_INVALID
    assert _lib.last_error(ctx.h) == "malformed row"
    # the mirror refuses a column type the decoder does not know (decodeColToChunk's default branch, decoder.go:232-233)
    with pytest.raises(_lib.TsqError) as ei:
        RC.NewChunkDecoder(ctx, [RC.ColInfo(1, 246)]).DecodeToChunk(b, o)
    assert ei.value.status == abi.ERR_UNSUPPORTED
    # and the normal call still works on the same context afterwards
    _lib.check(lib.tsq_rowcodec_decode(ctx.h, pb, b.size, po, None, 3, 0, 1, one, out, C.byref(n)), ctx.h)
    assert n.value == 3 and bufs[0][0][:3].tolist() == [5, 6, 7]


def test_table_scan_feeds_selection_and_partial_aggregate(ctx, orc):
    # the mocktikv chain tableScan -> selection -> hashAgg (executor.go:124-196, :368-390, aggregate.go:78-169) with the scan
    # decoded on the GPU: SELECT k, COUNT(*), SUM(v) FROM t WHERE v > 0 GROUP BY k over stored rows, against the oracle run
    # on the oracle-decoded columns
    from tinysql_amd import expression as E
    rng = np.random.default_rng(77)
    n = 60_000
    chk = Chunk([Column(abi.I64, rng.integers(0, 500, n), rng.random(n) > 0.05), Column(abi.I64, rng.integers(-99, 99, n), rng.random(n) > 0.05)])
    b, o = orc.rowcodec_encode(chk, [1, 2])
    cols = [RC.ColInfo(1, RC.TypeLonglong), RC.ColInfo(2, RC.TypeLonglong)]
    scan = RC.NewChunkDecoder(ctx, cols).DecodeToChunk(b, o)
    st, want_scan = orc.rowcodec_decode(b, o, None, [(1, abi.I64), (2, abi.I64)])
    assert st == 0
    _same(scan, want_scan)
    cond = [E.ScalarFunction("gt", E.Column(1, abi.I64), E.Constant(0))]
    ce = E.CompiledExpr(ctx, cond)
    try:
        keep = np.asarray(ce.VectorizedFilter(scan), dtype=bool)
    finally:
        ce.close()
    v = chk.columns[1]
    assert (keep == (v.notnull & (v.data > 0))).all()
    filtered = Chunk([Column(c.tp, c.data[keep], None if c.notnull is None else c.notnull[keep]) for c in scan.columns])
    aggs = [(abi.AGG_FIRSTROW, 0, abi.I64), (abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64)]
    acfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs)
    want_f = Chunk([Column(c.tp, c.data[keep], None if c.notnull is None else c.notnull[keep]) for c in want_scan.columns])
    assert H.rows_equal_unordered(G.run_agg(ctx, acfg, filtered, [abi.I64] * 3), orc.hash_agg(acfg, want_f, 4, 4))


def test_full_size_round_trip_property(ctx):
    # 1e7 stored rows x 5 columns (numpy encoder pinned on the oracle's in tests/test_oracle_rowcodec_golden.py):
    # decode(encode(table)) == table, column by column
    spec = importlib.util.spec_from_file_location("bench_rowcodec", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_rowcodec.py"))
    br = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(br)
    n, piece = 10_000_000, 2_500_000
    rng = np.random.default_rng(99)
    cols = [RC.ColInfo(i, RC.TypeDouble if t == abi.F64 else RC.TypeLonglong) for i, t in zip(br.IDS, br.TYPES)]
    dec = RC.NewChunkDecoder(ctx, cols)
    for lo in range(0, n, piece):
        table = br.make_scan(rng, piece)
        b, o = br.encode_rows_v2(table, br.IDS)
        got = dec.DecodeToChunk(b, o)
        assert got.NumRows() == piece
        for a, t in zip(got.columns, table):
            assert a.notnull is None or a.notnull.all()
            assert (a.data.view(np.uint64) == np.ascontiguousarray(t).view(np.uint64)).all()


# ---- waves whose rows share one layout resolve the column positions once (rc_rows_lds); TSQ_ROWCODEC_FAST_LAYOUT=0 switches the
# vote off, so both paths of the kernel are compared with the oracle on the same scans
def _uniform_scan(rng, n, null_col=None):
    cols = []
    for j in range(3):
        v = np.where(rng.random(n) < 0.5, rng.choice(EDGE, n), rng.integers(-(1 << 62), 1 << 62, n))
        cols.append(Column(abi.I64, v, np.zeros(n, bool) if j == null_col else None))
    cols.append(Column(abi.U64, (rng.integers(0, 1 << 62, n) >> rng.integers(0, 62, n)).astype(np.uint64)))
    cols.append(Column(abi.F64, rng.standard_normal(n) * 1e6))
    cols.append(Column(abi.F32, rng.standard_normal(n).astype(np.float32)))
    return Chunk(cols)


U_IDS = [9, 3, 17, 4, 200, 31]
U_SPECS = [(200, abi.F64), (-1, abi.I64, abi.RC_HANDLE), (9, abi.I64), (3, abi.I64), (31, abi.F32), (4, abi.U64), (17, abi.I64), (99, abi.I64),
           (98, abi.I64, abi.RC_HAS_DEFAULT, 5), (1 << 40, abi.I64), (-7, abi.I64)]


def _both_kernel_paths(ctx, b, o, handles, specs, want_st, want):
    n = len(o) - 1
    for knob, pipe in ((1, 1), (0, 1), (1, 0), (0, 0)):  # ROWCODEC_PIPELINE = 0: the plain (not software-pipelined) kernel
        with ctx.knobs(ROWCODEC_FAST_LAYOUT=knob, ROWCODEC_PIPELINE=pipe):
            gst, m, got = _decode_device(ctx, b, o, handles, specs, n)
        assert (gst == abi.OK) == (want_st == 0) and m == want.NumRows(), (knob, pipe)
        if want_st:
            assert _lib.last_error(ctx.h) == MSG[want_st]
        _same(got, want)


@pytest.mark.parametrize("n", [1, 64, 65, 1000, 100_000])
def test_shared_layout_scans(ctx, orc, n):
    rng = np.random.default_rng(500 + n)
    chk = _uniform_scan(rng, n, null_col=1)
    handles = rng.integers(-(1 << 62), 1 << 62, n)
    b, o = orc.rowcodec_encode(chk, U_IDS)
    st, want = orc.rowcodec_decode(b, o, handles, U_SPECS)
    assert st == 0
    _both_kernel_paths(ctx, b, o, handles, U_SPECS, st, want)


def test_mixed_layouts_and_signature_limits(ctx, orc):
    rng = np.random.default_rng(77)
    n = 6400
    nn = np.ones(n, bool)
    nn[rng.integers(0, n, 20)] = False  # a few rows with one NULL: their waves take the general path
    chk = Chunk([Column(abi.I64, rng.integers(-9999, 9999, n), nn), Column(abi.F64, rng.random(n)), Column(abi.I64, rng.integers(0, 9, n))])
    b1, o1 = orc.rowcodec_encode(chk, [1, 2, 3])
    other = Chunk([Column(abi.I64, rng.integers(-9999, 9999, 640)), Column(abi.F64, rng.random(640))])
    b2, o2 = orc.rowcodec_encode(other, [1, 2])
    rows = [b1[o1[r]:o1[r + 1]] for r in range(n)]
    rows[3200:3840] = [b2[o2[r]:o2[r + 1]] for r in range(640)]  # another table shape in the middle of the scan
    b = np.concatenate(rows)
    o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    specs = [(3, abi.I64), (1, abi.I64), (2, abi.F64)]
    st, want = orc.rowcodec_decode(b, o, None, specs)
    _both_kernel_paths(ctx, b, o, None, specs, st, want)
    for k in (8, 9):  # eight ids fit the signature, nine do not
        wide = Chunk([Column(abi.I64, rng.integers(-300, 300, 3000)) for _ in range(k)])
        ids = list(range(10, 10 + k))
        b, o = orc.rowcodec_encode(wide, ids)
        specs = [(i, abi.I64) for i in reversed(ids)] + [(5, abi.I64)]
        st, want = orc.rowcodec_decode(b, o, None, specs)
        _both_kernel_paths(ctx, b, o, None, specs, st, want)


@pytest.mark.parametrize("case", ["cut_value", "short_float", "version_first_lane", "version_other_lane"])
def test_errors_inside_shared_layout_waves(ctx, orc, case):
    rng = np.random.default_rng(31)
    n = 7000
    chk = _uniform_scan(rng, n)
    b, o = orc.rowcodec_encode(chk, U_IDS)
    specs = [(9, abi.I64), (3, abi.I64), (17, abi.I64), (4, abi.U64), (200, abi.F64), (31, abi.F32)]
    rows = [b[o[r]:o[r + 1]].copy() for r in range(n)]
    at = 4549
    if case == "cut_value":
        rows[at] = rows[at][:-5]
    elif case == "short_float":
        specs = [(9, abi.F64)] + specs[1:]
    elif case == "version_first_lane":
        at = 4544  # lane 0 of its wave
        rows[at][0] = 7
    else:
        rows[at][0] = 7
    b = np.concatenate(rows)
    o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    st, want = orc.rowcodec_decode(b, o, None, specs)
    assert st != 0 and (case == "short_float" or want.NumRows() == at)
    _both_kernel_paths(ctx, b, o, None, specs, st, want)


# ------------------------------------------------------------------------------------------------ string columns (round 2)
def _string_scan(rng, n, long_cells=False):
    from tinysql_amd.chunk import StrColumn
    words = [None if rng.random() < 0.15 else bytes(rng.integers(0, 256, int(rng.integers(0, 24)), dtype=np.uint8)) for _ in range(n)]
    if long_cells:  # the reference's benchmark payload is 5 KiB (executor/benchmark_test.go:328): one cell per wave in the copy kernel
        notes = [None if rng.random() < 0.1 else bytes([65 + i % 26]) * int(rng.integers(0, 3) * 2600) for i in range(n)]
    else:
        notes = [None if rng.random() < 0.1 else (b"" if rng.random() < 0.2 else b"n%05d" % i) for i in range(n)]
    return Chunk([Column(abi.I64, rng.integers(-(1 << 40), 1 << 40, n), rng.random(n) >= 0.2), StrColumn(words),
                  Column(abi.F64, rng.standard_normal(n), rng.random(n) >= 0.2), StrColumn(notes)])


STR_COLS = [RC.ColInfo(4, RC.TypeVarchar), RC.ColInfo(1, RC.TypeLonglong), RC.ColInfo(-1, RC.TypeLonglong, 0, True), RC.ColInfo(2, RC.TypeBlob),
            RC.ColInfo(3, RC.TypeDouble), RC.ColInfo(50, RC.TypeVarString)]
STR_SPECS = [(4, abi.BYTES), (1, abi.I64), (-1, abi.I64, abi.RC_HANDLE), (2, abi.BYTES), (3, abi.F64), (50, abi.BYTES)]


@pytest.mark.parametrize("n,long_cells", [(1, False), (64, False), (257, False), (5000, False), (100_000, False), (700, True)])
def test_string_columns_against_the_oracle(ctx, orc, n, long_cells):
    # varchar / blob cells are the value's bytes as they are (chk.AppendBytes, decoder.go:226-228); a NULL id or an absent column
    # is a NULL cell, an empty value an empty string
    rng = np.random.default_rng(n + 5)
    chk = _string_scan(rng, n, long_cells)
    handles = rng.integers(-(1 << 62), 1 << 62, n)
    b, o = orc.rowcodec_encode(chk, [1, 2, 3, 4])
    st, want = orc.rowcodec_decode_chunk(b, o, handles, STR_SPECS)
    got = RC.NewChunkDecoder(ctx, STR_COLS, -1).DecodeToChunk(b, o, handles)
    assert st == 0 and got.NumRows() == n and got.rows() == want.rows()
    assert all(v is None for v in got.columns[5].values())  # id 50 is in no row


def test_string_columns_device_resident_and_rows_before_an_error(ctx, orc):
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(31)
    n = 20_000
    chk = _string_scan(rng, n)
    b, o = orc.rowcodec_encode(chk, [1, 2, 3, 4])
    specs = [(2, abi.BYTES), (1, abi.I64), (4, abi.BYTES)]
    for damage in (None, 12_345):
        raw = b.copy()
        if damage is not None:
            raw[o[damage]] = 127  # "invalid codec version" at that row: the rows before it are handed over, strings included
        st, want = orc.rowcodec_decode_chunk(raw, o, None, specs)
        dbytes, doffs = ctx.alloc(raw.size + 64 + 7), ctx.alloc(o.nbytes + 64)
        d1, o1, b1 = ctx.alloc(raw.size + 64), ctx.alloc(8 * (n + 1) + 64), ctx.alloc(n // 8 + 64)
        d2, b2 = ctx.alloc(8 * n + 64), ctx.alloc(n // 8 + 64)
        d3, o3, b3 = ctx.alloc(raw.size + 64), ctx.alloc(8 * (n + 1) + 64), ctx.alloc(n // 8 + 64)
        try:
            ctx.h2d(dbytes + 7, raw)  # the values start 7 bytes into their buffer
            ctx.h2d(doffs, o)
            out = (abi.Col * 3)()
            for i, (d, of, bm, tp) in enumerate(((d1, o1, b1, abi.BYTES), (d2, None, b2, abi.I64), (d3, o3, b3, abi.BYTES))):
                out[i].data, out[i].offsets, out[i].null_bitmap, out[i].length, out[i].type, out[i].flags = d, of, bm, n, tp, abi.COL_DEVICE
                out[i].elem_size = -1 if tp == abi.BYTES else 8
            m = C.c_int64(0)
            gst = ctx.lib.tsq_rowcodec_decode(ctx.h, C.c_void_p(dbytes + 7), raw.size, C.c_void_p(doffs), None, n, abi.COL_DEVICE, 3, orc_cols(specs), out, C.byref(m))
            rows = m.value
            assert rows == want.NumRows() == (n if damage is None else damage)
            assert (gst == abi.OK) == (damage is None) and (damage is None or _lib.last_error(ctx.h) == "invalid codec version")
            cols = []
            for d, of, bm, tp in ((d1, o1, b1, abi.BYTES), (d2, None, b2, abi.I64), (d3, o3, b3, abi.BYTES)):
                bits = np.zeros(n // 8 + 8, np.uint8)
                ctx.d2h(bits, bm)
                nn = np.unpackbits(bits, bitorder="little")[:rows].astype(bool)
                if tp == abi.BYTES:
                    offs = np.zeros(rows + 1, np.int64)
                    ctx.d2h(offs, of)
                    data = np.zeros(max(int(offs[rows]), 1), np.uint8)
                    ctx.d2h(data, d)
                    assert offs[0] == 0 and (np.diff(offs) >= 0).all()
                    cols.append(StrColumn([bytes(data[offs[r]:offs[r + 1]]) if nn[r] else None for r in range(rows)]))
                else:
                    v = np.zeros(max(rows, 1), np.int64)
                    ctx.d2h(v, d)
                    cols.append(Column(tp, v[:rows], nn))
            assert Chunk(cols).rows() == want.rows()
        finally:
            for p in (dbytes, doffs, d1, o1, b1, d2, b2, d3, o3, b3):
                ctx.free(p)


@pytest.mark.parametrize("n", [1, 700, 30_000])
def test_default_strings_for_columns_the_rows_lack(ctx, orc, n):
    # ALTER TABLE ... ADD COLUMN note varchar DEFAULT 'n/a' after the rows were written: the rows lack the column and the scan hands
    # out the default (defDatum -> chk.AppendDatum, decoder.go:186-194); a column that IS in the row (NULL or not) ignores its default
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(n)
    words = [None if rng.random() < 0.3 else bytes(rng.integers(97, 123, int(rng.integers(0, 12)), dtype=np.uint8)) for _ in range(n)]
    chk = Chunk([Column(abi.I64, rng.integers(0, 9, n)), StrColumn(words)])
    b, o = orc.rowcodec_encode(chk, [1, 2])
    defaults = {0: b"present-so-unused", 1: b"n/a", 2: b"", 3: b"x" * 100}
    cols = [RC.ColInfo(2, RC.TypeVarchar), RC.ColInfo(7, RC.TypeVarchar), RC.ColInfo(8, RC.TypeBlob), RC.ColInfo(9, RC.TypeVarString), RC.ColInfo(1, RC.TypeLonglong), RC.ColInfo(10, RC.TypeVarchar)]
    got = RC.NewChunkDecoder(ctx, cols, defDatum=lambda i: defaults.get(i)).DecodeToChunk(b, o)
    specs = [(2, abi.BYTES, abi.RC_HAS_DEFAULT, defaults[0]), (7, abi.BYTES, abi.RC_HAS_DEFAULT, defaults[1]), (8, abi.BYTES, abi.RC_HAS_DEFAULT, defaults[2]),
             (9, abi.BYTES, abi.RC_HAS_DEFAULT, defaults[3]), (1, abi.I64), (10, abi.BYTES)]
    st, want = orc.rowcodec_decode_chunk(b, o, None, specs)
    assert st == 0 and got.rows() == want.rows()
    assert got.columns[0].values() == words and got.columns[1].values() == [b"n/a"] * n and got.columns[2].values() == [b""] * n and got.columns[5].values() == [None] * n


@pytest.mark.parametrize("flen", [1, 8, 9, 24, 40, 64])
def test_bit_columns_become_binary_literals(ctx, orc, flen):
    # TypeBit (decoder.go:229-231): the stored unsigned int becomes a literal of (Flen + 7) / 8 big-endian bytes
    # (types.NewBinaryLiteralFromUint, binary_literal.go:57-69); NULLs, rows that lack the column (a default literal / NULL)
    rng = np.random.default_rng(flen)
    n = 5000
    vals = rng.integers(0, 1 << min(flen, 62), n).astype(np.uint64)
    if flen == 64:
        vals = vals * np.uint64(4) + np.uint64(3)
    chk = Chunk([Column(abi.U64, vals, rng.random(n) >= 0.2), Column(abi.I64, np.arange(n))])
    b, o = orc.rowcodec_encode(chk, [3, 1])
    bsz = (flen + 7) >> 3
    lit = bytes(range(1, bsz + 1))
    cols = [RC.ColInfo(3, RC.TypeBit, Flen=flen), RC.ColInfo(1, RC.TypeLonglong), RC.ColInfo(9, RC.TypeBit, Flen=flen), RC.ColInfo(10, RC.TypeBit, Flen=flen)]
    got = RC.NewChunkDecoder(ctx, cols, defDatum=lambda i: lit if i == 2 else None).DecodeToChunk(b, o)
    bit = abi.RC_BIT | (bsz << 8)
    st, want = orc.rowcodec_decode_chunk(b, o, None, [(3, abi.BYTES, bit), (1, abi.I64), (9, abi.BYTES, bit | abi.RC_HAS_DEFAULT, lit), (10, abi.BYTES, bit)])
    assert st == 0 and got.rows() == want.rows()
    nn = chk.columns[0].notnull
    assert got.columns[0].values() == [int(v).to_bytes(8, "big")[8 - bsz:] if ok else None for v, ok in zip(vals.tolist(), nn.tolist())]
    assert got.columns[2].values() == [lit] * n and got.columns[3].values() == [None] * n
