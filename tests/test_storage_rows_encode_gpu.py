"""GPU parity of tsq_rows_encode (chunk rows -> the RowsData bytes of a coprocessor response; SURVEY.md §8 f rank 4) against the
oracle's restatement of codec.EncodeValue / EncodeKey (oracle/codec_rows.cpp, pinned on codec_test.go): byte-identical output at
tile-boundary sizes, both value forms and a per-column mix, host and device placement at every alignment of the output pointer,
the row boundaries (64-row response chunks), the too-small-buffer contract, the argument contract, the round trip through
tsq_rows_decode, and the coprocessor chain scan -> selection -> partial aggregate -> response bytes.

First hardware run: round 2 (gpurun_out/r2i, 22 tests green on the first run; profiles/r02_pytest_gpu.txt)."""
import ctypes as C
import os

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import distsql
from tinysql_amd.chunk import Chunk, Column, make_cols

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu


def rand_chunk(rng, n, null_p=0.2):
    iv = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64) >> rng.integers(0, 64, n)
    iv[:min(n, 8)] = np.array([0, -1, 1, 63, -64, 64, (1 << 63) - 1, -(1 << 63)])[:min(n, 8)]
    uv = (rng.integers(0, (1 << 64) - 1, n, dtype=np.uint64) >> rng.integers(0, 64, n).astype(np.uint64)).astype(np.uint64)
    fv = np.ldexp(rng.random(n) - 0.5, rng.integers(-60, 60, n))
    fv[:min(n, 6)] = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan])[:min(n, 6)]
    f32 = (rng.random(n) * 100 - 50).astype(np.float32)
    nn = (lambda: rng.random(n) >= null_p) if null_p else (lambda: None)
    return Chunk([Column(abi.I64, iv, nn()), Column(abi.U64, uv, nn()), Column(abi.F64, fv, nn()), Column(abi.F32, f32, nn())])


@pytest.mark.parametrize("comparable", [False, True])
@pytest.mark.parametrize("n", [1, 7, 255, 256, 257, 3000, 200_000])
def test_bytes_equal_the_oracle(ctx, orc, n, comparable):
    rng = np.random.default_rng(n)
    chk = rand_chunk(rng, n)
    want = orc.encode_rows(chk, comparable)
    got, offs = distsql.encode_rows(ctx, chk, comparable_cols=range(4) if comparable else ())
    assert got.size == want.size and (got == want).all()
    assert offs[0] == 0 and offs[n] == want.size and (np.diff(offs) > 0).all()
    chunks = distsql.response_chunks(got, offs)
    assert len(chunks) == (n + 63) // 64 and b"".join(chunks) == bytes(want)


def test_mixed_forms_round_trip_through_the_decoder(ctx, orc):
    rng = np.random.default_rng(8)
    n = 50_000
    chk = rand_chunk(rng, n)
    got, offs = distsql.encode_rows(ctx, chk, comparable_cols=(0,))
    dec, used = distsql.decode_rows(ctx, got, chk.types(), n)
    assert used == got.size and bytes(orc.encode_rows(dec, False)) == bytes(orc.encode_rows(chk, False))


@pytest.mark.parametrize("phase", [0, 1, 7, 8, 15])
def test_device_resident_columns_and_output_at_every_alignment(ctx, orc, phase):
    rng = np.random.default_rng(40 + phase)
    n = 30_000
    chk = rand_chunk(rng, n, null_p=0.1)
    want = orc.encode_rows(chk, False)
    dcols = [G.to_device(ctx, c) for c in chk.columns]
    dout, doffs = ctx.alloc(want.size + phase + 64), ctx.alloc(8 * (n + 1) + 64)
    try:
        ctx.memset(dout, 0xEE, want.size + phase + 64)
        m = C.c_int64(0)
        _lib.check(ctx.lib.tsq_rows_encode(ctx.h, G.dev_cols(dcols), 4, None, n, C.c_void_p(dout + phase), want.size, abi.COL_DEVICE, C.c_void_p(doffs),
                                           C.byref(m)), ctx.h)
        assert m.value == want.size
        raw = np.zeros(want.size + phase + 64, np.uint8)
        ctx.d2h(raw, dout)
        assert (raw[:phase] == 0xEE).all() and (raw[phase + want.size:] == 0xEE).all() and (raw[phase:phase + want.size] == want).all()
        offs = np.zeros(n + 1, np.int64)
        ctx.d2h(offs, doffs)
        assert offs[0] == 0 and offs[n] == want.size and (np.diff(offs) > 0).all()
    finally:
        ctx.free(dout)
        ctx.free(doffs)
        for c in dcols:
            c.free()


def test_contracts(ctx, orc):
    lib = ctx.lib
    chk = Chunk([Column(abi.I64, np.array([5, 6, 7]))])
    keep = []
    cols = make_cols(chk.columns, keep)
    out = np.full(64, 0xEE, np.uint8)
    po = out.ctypes.data_as(C.c_void_p)
    m = C.c_int64(-1)
    # nothing to encode is not an error
    assert lib.tsq_rows_encode(ctx.h, cols, 1, None, 0, po, 64, 0, None, C.byref(m)) == abi.OK and m.value == 0
    # too small: the size is reported and nothing is written
    assert lib.tsq_rows_encode(ctx.h, cols, 1, None, 3, po, 5, 0, None, C.byref(m)) == abi.ERR_INVALID and m.value == 6 and (out == 0xEE).all()
    # misuse
    assert lib.tsq_rows_encode(ctx.h, cols, 1, None, 3, po, 64, 0, None, None) == abi.ERR_INVALID
    assert lib.tsq_rows_encode(ctx.h, cols, 17, None, 3, po, 64, 0, None, C.byref(m)) == abi.ERR_UNSUPPORTED
    assert lib.tsq_rows_encode(ctx.h, cols, 1, None, 4, po, 64, 0, None, C.byref(m)) == abi.ERR_INVALID  # column shorter than nrows
    cols[0].type = abi.BYTES
    assert lib.tsq_rows_encode(ctx.h, cols, 1, None, 3, po, 64, 0, None, C.byref(m)) == abi.ERR_INVALID  # a var-len column without offsets
    cols[0].type = abi.I64
    _lib.check(lib.tsq_rows_encode(ctx.h, cols, 1, None, 3, po, 64, 0, None, C.byref(m)), ctx.h)
    assert m.value == 6 and bytes(out[:6]) == b"\x08\x0a\x08\x0c\x08\x0e" and (out[6:] == 0xEE).all()


def test_coprocessor_chain_scan_selection_partial_aggregate_response(ctx, orc):
    # tableScan -> selection -> hashAgg -> response chunks (store/mockstore/mocktikv/executor.go, aggregate.go:78-116,
    # cop_handler_dag.go:49-83): SELECT COUNT(*), SUM(v), k FROM t WHERE v > 0 GROUP BY k pushed down; every step on the GPU
    from tinysql_amd import expression as E
    from tinysql_amd import rowcodec as RC
    rng = np.random.default_rng(77)
    n = 60_000
    table = Chunk([Column(abi.I64, rng.integers(0, 500, n), rng.random(n) > 0.05), Column(abi.I64, rng.integers(-99, 99, n), rng.random(n) > 0.05)])
    vals, offs = orc.rowcodec_encode(table, [1, 2])
    scan = RC.NewChunkDecoder(ctx, [RC.ColInfo(1, RC.TypeLonglong), RC.ColInfo(2, RC.TypeLonglong)]).DecodeToChunk(vals, offs)
    ce = E.CompiledExpr(ctx, [E.ScalarFunction("gt", E.Column(1, abi.I64), E.Constant(0))])
    try:
        keep = np.asarray(ce.VectorizedFilter(scan), dtype=bool)
    finally:
        ce.close()
    filtered = Chunk([Column(c.tp, c.data[keep], None if c.notnull is None else c.notnull[keep]) for c in scan.columns])
    # partial results first, group-by values last (aggregate.go:96-113)
    aggs = [(abi.AGG_COUNT, -1, abi.I64), (abi.AGG_SUM, 1, abi.I64), (abi.AGG_FIRSTROW, 0, abi.I64)]
    acfg = H.agg_cfg([abi.I64, abi.I64], [0], aggs)
    partial = G.run_agg(ctx, acfg, filtered, [abi.I64] * 3)
    raw, roffs = distsql.encode_rows(ctx, partial)
    chunks = distsql.response_chunks(raw, roffs)
    # the SQL side reads the response back (readRowsData) and must see the oracle's groups
    back = distsql.SelectResult(ctx, chunks, [abi.I64] * 3)
    rows = []
    while True:
        c = back.Next(1024)
        if c.NumRows() == 0:
            break
        rows += c.rows()
    v = table.columns[1]
    want_keep = v.notnull & (v.data > 0)
    want_in = Chunk([Column(c.tp, c.data[want_keep], None if c.notnull is None else c.notnull[want_keep]) for c in table.columns])
    assert H.rows_equal_unordered(rows, orc.hash_agg(acfg, want_in, 4, 4))
    assert bytes(raw) == bytes(orc.encode_rows(partial))


# ------------------------------------------------------------------------------------------------ string columns (round 2)
@pytest.mark.parametrize("n,long_every", [(1, 0), (257, 0), (5000, 0), (100_000, 0), (700, 50)])
def test_string_columns_as_compact_bytes(ctx, orc, n, long_every):
    # a var-len cell = compactBytesFlag + varint(len) + the bytes (codec.go:101-109, bytes.go:141-148); tiles with 70 000-byte cells
    # bypass the LDS image; the response decodes back to the rows (tsq_rows_decode_chunks) and equals the oracle byte for byte
    from .test_hostsim_encode import string_chunk
    rng = np.random.default_rng(n)
    chk = string_chunk(rng, n, long_every)
    want = orc.encode_rows(chk)
    raw, offs = distsql.encode_rows(ctx, chk)
    assert bytes(raw) == bytes(want) and offs[0] == 0 and offs[-1] == len(want)
    chunks = distsql.response_chunks(raw, offs)
    assert len(chunks) == (n + 63) // 64
    back = distsql.decode_chunks(ctx, chunks, chk.types())
    assert back.rows() == chk.rows()


def test_string_response_size_can_be_asked_first(ctx, orc):
    from .test_hostsim_encode import string_chunk
    from tinysql_amd.chunk import make_cols
    rng = np.random.default_rng(3)
    chk = string_chunk(rng, 1000)
    want = orc.encode_rows(chk)
    keep = []
    cols = make_cols(chk.columns, keep)
    m = C.c_int64(0)
    assert ctx.lib.tsq_rows_encode(ctx.h, cols, 4, None, 1000, None, 0, 0, None, C.byref(m)) == abi.ERR_INVALID and m.value == len(want)
    out = np.full(m.value + 8, 0xEE, np.uint8)
    _lib.check(ctx.lib.tsq_rows_encode(ctx.h, cols, 4, None, 1000, out.ctypes.data_as(C.c_void_p), m.value, 0, None, C.byref(m)), ctx.h)
    assert bytes(out[:m.value]) == bytes(want) and (out[m.value:] == 0xEE).all()


@pytest.mark.parametrize("n,long_every", [(1, 0), (5000, 0), (700, 50)])
def test_string_columns_in_the_memcomparable_form(ctx, orc, n, long_every):
    # EncodeKey of a var-len cell = bytesFlag + groups of 8 bytes with their markers (codec.go:86-91, bytes.go:35-67) — what an index
    # key holds; equals the oracle byte for byte, and the single-stream decoder reads the rows back
    from .test_hostsim_encode import string_chunk
    rng = np.random.default_rng(50 + n)
    chk = string_chunk(rng, n, long_every)
    want = orc.encode_rows(chk, comparable=True)
    raw, offs = distsql.encode_rows(ctx, chk, comparable_cols=range(4))
    assert bytes(raw) == bytes(want) and offs[0] == 0 and offs[-1] == len(want)
    back = distsql.decode_chunks(ctx, distsql.response_chunks(raw, offs), chk.types())
    assert back.rows() == chk.rows()
