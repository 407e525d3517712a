"""GPU parity of tsq_rows_decode (coprocessor-response rows -> columns, SURVEY.md §8 f rank 2) against the oracle's
restatement of readRowsData + Decoder.DecodeOne (distsql/select_result.go:139-155, util/codec/codec.go:623-690):
bit-exact values and NULL flags, the reference's errors decided by the first offending value in stream order, the
chunk-capacity / remainder contract, both value encodings, byte strings whose payload bytes look like flag bytes (what the
speculative parser has to get right), tile / sub-block boundary sizes, host and device placement, and a full-size round
trip."""
import ctypes as C
import math

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import distsql
from tinysql_amd.chunk import Chunk, Column

from . import gpu_helpers as G
from . import helpers as H

pytestmark = pytest.mark.gpu
MSG = {1: "invalid encoded key", 2: "insufficient bytes to decode value", 3: "value larger than 64 bits", 4: "invalid encoded key flag"}


def _rand_chunk(rng, n, types, null_p=0.2, wide=True):
    cols = []
    for t in types:
        if t == abi.I64:
            v = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64) >> rng.integers(0, 63, n) if wide else rng.integers(-1000, 1000, n)
            v = v.astype(np.int64)
        elif t == abi.U64:
            v = (rng.integers(0, (1 << 64) - 1, n, dtype=np.uint64) >> rng.integers(0, 64, n).astype(np.uint64)).astype(np.uint64)
        elif t == abi.F64:
            v = np.ldexp(rng.random(n) - 0.5, rng.integers(-60, 60, n))
        else:
            v = (rng.random(n) * 100 - 50).astype(np.float32)
        cols.append(Column(t, v, rng.random(n) >= null_p if null_p else None))
    return Chunk(cols)


def _same(a, b):
    assert a.NumRows() == b.NumRows() and a.types() == b.types()
    for ca, cb in zip(a.columns, b.columns):
        na = np.ones(len(ca), bool) if ca.notnull is None else ca.notnull
        nb = np.ones(len(cb), bool) if cb.notnull is None else cb.notnull
        assert (na == nb).all()
        assert (ca.data.view(np.uint8).reshape(len(ca), -1)[na] == cb.data.view(np.uint8).reshape(len(cb), -1)[nb]).all()
        assert not ca.data.view(np.uint8).reshape(len(ca), -1)[~na].any()   # a NULL slot holds zero bytes (column.go:150-158)


def test_reference_row_of_TestDecodeOneToChunk(ctx, orc):
    types = [abi.I64] * 7 + [abi.U64, abi.F32, abi.F64, abi.I64]
    vals = [None, 1, 1, 1, 1, -1, 1, 1, 1.0, 1.0, 1]
    cols = []
    for t, v in zip(types, vals):
        dt = {abi.I64: np.int64, abi.U64: np.uint64, abi.F64: np.float64, abi.F32: np.float32}[t]
        cols.append(Column(t, np.array([0 if v is None else v] * 3, dtype=dt), None if v is not None else np.zeros(3, bool)))
    raw = orc.encode_rows(Chunk(cols))
    got, used = distsql.decode_rows(ctx, raw, types, 32)
    assert used == raw.size and [list(r) for r in got.rows()] == [vals] * 3


@pytest.mark.parametrize("comparable", [False, True])
@pytest.mark.parametrize("n", [1, 2, 63, 700, 2730, 2731, 5461, 40_000])
def test_random_chunks_equal_the_oracle_bit_for_bit(ctx, orc, n, comparable):
    # n around 2731 / 5461: the byte string ends near a 16 KB tile boundary with 6-byte average rows
    rng = np.random.default_rng(n * 2 + comparable)
    types = [abi.I64, abi.F64, abi.U64, abi.F32, abi.I64]
    chk = _rand_chunk(rng, n, types)
    raw = orc.encode_rows(chk, comparable)
    st, want, used_w = orc.decode_rows(raw, types, n + 5)
    got, used = distsql.decode_rows(ctx, raw, types, n + 5)
    assert st == 0 and used == used_w == raw.size
    _same(got, want)
    _same(got, chk)  # Column() keeps NULL slots zeroed, like AppendNull (column.go:150-158)


@pytest.mark.parametrize("cut", [-1, 0, 1, 5])
def test_every_byte_length_around_the_tile_and_subblock_boundaries(ctx, orc, cut):
    # one int column of small values: 2 bytes per row, so rows straddle every 64-byte / 16 KB boundary in all phases when a
    # prefix of odd length (a 9-byte float row) is put in front
    rng = np.random.default_rng(7 + cut)
    for lead in range(0, 3):
        chk = Chunk([Column(abi.I64, rng.integers(-60, 60, 16384 + 40 + cut))])
        head = orc.encode_rows(Chunk([Column(abi.F64, rng.random(lead))]))
        body = orc.encode_rows(chk)
        raw = np.concatenate([head, body])
        if lead:  # the leading floats are rows of the same single column (DecodeOne appends by flag, codec.go:623-690)
            st, want, _ = orc.decode_rows(raw, [abi.I64], 1 << 20)
        else:
            st, want, _ = orc.decode_rows(raw, [abi.I64], 1 << 20)
        got, used = distsql.decode_rows(ctx, raw, [abi.I64], 1 << 20)
        assert st == 0 and used == raw.size
        _same(got, want)


def test_payload_bytes_that_look_like_flags(ctx, orc):
    # 8-byte payloads made only of the bytes 0x00 0x03 0x05 0x08 0x09 0x80 0xff: every wrong entry offset parses "valid" values
    rng = np.random.default_rng(3)
    n = 30_000
    alphabet = np.array([0x00, 0x03, 0x05, 0x08, 0x09, 0x80, 0xFF, 0x01], dtype=np.uint8)
    u = alphabet[rng.integers(0, 8, (n, 8))].copy().view(np.uint64).reshape(n)
    chk = Chunk([Column(abi.U64, u), Column(abi.I64, u.view(np.int64), rng.random(n) > 0.3), Column(abi.F64, rng.random(n))])
    for comparable in (False, True):
        raw = orc.encode_rows(chk, comparable)
        st, want, _ = orc.decode_rows(raw, chk.types(), n)
        got, used = distsql.decode_rows(ctx, raw, chk.types(), n)
        assert st == 0 and used == raw.size
        _same(got, want)


def test_chunk_capacity_and_remainder_contract(ctx, orc):
    # select_result.go:139-155: decode until the chunk is full, keep the rest of RowsData for the next Next
    rng = np.random.default_rng(11)
    types = [abi.I64, abi.F64]
    chk = _rand_chunk(rng, 10_000, types)
    raw = orc.encode_rows(chk)
    for cap in (1, 1024, 4097):
        st, want, used_w = orc.decode_rows(raw, types, cap)
        got, used = distsql.decode_rows(ctx, raw, types, cap)
        assert st == 0 and used == used_w and got.NumRows() == cap
        _same(got, want)
    # SelectResult.Next over three response chunks, 1024 rows per call, equals the whole table
    parts = [orc.encode_rows(chk.slice(0, 3000)), np.zeros(0, np.uint8), orc.encode_rows(chk.slice(3000, 3001)), orc.encode_rows(chk.slice(3001, 10_000))]
    sr = distsql.SelectResult(ctx, parts, types)
    rows = []
    while True:
        c = sr.Next(1024)
        if c.NumRows() == 0:
            break
        assert c.NumRows() <= 1024
        rows += c.rows()
    assert rows == orc.decode_rows(raw, types, 10_000)[1].rows()


def _status_of(ctx, raw, types, cap=64):
    try:
        got, used = distsql.decode_rows(ctx, np.frombuffer(raw, np.uint8) if isinstance(raw, bytes) else raw, types, cap)
        return 0, got.NumRows(), ""
    except _lib.TsqError as e:
        return e.status, None, e.message


def test_errors_are_the_reference_errors_in_stream_order(ctx, orc):
    ok = orc.encode_rows(Chunk([Column(abi.I64, np.array([1000])), Column(abi.I64, np.array([7]))]))
    t2 = [abi.I64, abi.I64]
    cases = [(ok[:-2], t2), (ok[:2], t2), (b"\x03\x80\x00", [abi.I64]), (b"\x08" + b"\xff" * 10 + b"\x01", [abi.I64]), (b"\x08" + b"\xff" * 9 + b"\x02", [abi.I64]),
             (b"\x07\x00", [abi.I64]), (bytes(np.concatenate([ok, ok[:1]])), t2)]
    for raw, types in cases:
        raw = np.frombuffer(raw, np.uint8) if isinstance(raw, bytes) else raw
        st, _, _ = orc.decode_rows(raw, types, 64)
        gs, _, msg = _status_of(ctx, raw, types)
        assert st in MSG and gs == abi.ERR_INVALID and msg == MSG[st], (bytes(raw), st, msg)
    # a bytes datum where the schema says number: the chunk decoder's message (round 5: tsq_rows_decode takes var-len columns itself)
    gs, _, msg = _status_of(ctx, b"\x02\x02ab", [abi.I64])
    assert gs == abi.ERR_INVALID and msg == "datum kind does not match the column type"
    # the FIRST offending value decides: a bad flag in row 5000 wins over a cut varint at the very end, and the rows before
    # it are delivered (DecodeOne has appended them by then)
    rng = np.random.default_rng(5)
    chk = Chunk([Column(abi.I64, rng.integers(-5, 5, 20_000)), Column(abi.F64, rng.random(20_000))])
    raw = orc.encode_rows(chk).copy()
    raw[5000 * 11] = 0x07          # rows are 2 + 9 bytes: the flag of row 5000's int
    raw = np.concatenate([raw, np.frombuffer(b"\x08\xff", np.uint8)])
    st, want, _ = orc.decode_rows(raw, chk.types(), 1 << 20)
    assert st == 4 and want.NumRows() == 5000
    keep = []
    from tinysql_amd.chunk import chunk_from_buffers, out_buffers
    out, bufs = out_buffers(chk.types(), 1 << 20, keep)
    n, used = C.c_int64(0), C.c_int64(0)
    tp = (C.c_int32 * 2)(*chk.types())
    s = ctx.lib.tsq_rows_decode(ctx.h, raw.ctypes.data_as(C.c_void_p), raw.size, 0, 2, tp, out, 1 << 20, C.byref(n), C.byref(used))
    assert s == abi.ERR_INVALID and n.value == 5000 and _lib.last_error(ctx.h) == MSG[4]
    _same(chunk_from_buffers(chk.types(), bufs, 5000), want)
    # ... but an error beyond the requested rows is never reached (the reference stops when the chunk is full)
    got, used = distsql.decode_rows(ctx, raw, chk.types(), 4000)
    assert got.NumRows() == 4000 and used == 4000 * 11


def test_device_resident_bytes_and_columns(ctx, orc):
    rng = np.random.default_rng(21)
    types = [abi.I64, abi.I64, abi.F64]
    n = 200_000
    chk = _rand_chunk(rng, n, types, null_p=0.1)
    raw = orc.encode_rows(chk)
    dbytes = ctx.alloc(raw.size + 64)
    outs = [G.DevCol(ctx, t, n, with_nulls=True) for t in types]
    try:
        ctx.h2d(dbytes, raw)
        oc = G.dev_cols(outs)
        tp = (C.c_int32 * 3)(*types)
        m, used = C.c_int64(0), C.c_int64(0)
        _lib.check(ctx.lib.tsq_rows_decode(ctx.h, C.c_void_p(dbytes), raw.size, abi.COL_DEVICE, 3, tp, oc, n, C.byref(m), C.byref(used)), ctx.h)
        assert m.value == n and used.value == raw.size
        got = Chunk([o.to_host() for o in outs])
        _same(got, orc.decode_rows(raw, types, n)[1])
    finally:
        ctx.free(dbytes)
        for o in outs:
            o.free()


def test_full_size_round_trip_property(ctx, orc):
    # 2e7 rows x (int64 key, int64 value, double): decode(encode(table)) == table, compared through the order-independent
    # row checksum the join tests use (Σ and ⊕ of rowhash) — the oracle only ENCODES here (its decoder takes ~1 s per 1e7 values)
    n = 20_000_000
    rng = np.random.default_rng(99)
    chk = Chunk([Column(abi.I64, rng.integers(0, 1 << 40, n)), Column(abi.I64, rng.integers(-1000, 1000, n)), Column(abi.F64, rng.random(n))])
    raw = orc.encode_rows(chk)
    got, used = distsql.decode_rows(ctx, raw, chk.types(), n)
    assert used == raw.size and got.NumRows() == n
    for a, b in zip(got.columns, chk.columns):
        assert a.notnull is None or a.notnull.all()
        assert (a.data.view(np.uint64) == b.data.view(np.uint64)).all()


def test_decode_argument_contract(ctx, orc):
    lib = ctx.lib
    from tinysql_amd.chunk import out_buffers
    keep = []
    out, bufs = out_buffers([abi.I64], 8, keep)
    n, used = C.c_int64(-1), C.c_int64(-1)
    tp = (C.c_int32 * 1)(abi.I64)
    raw = orc.encode_rows(Chunk([Column(abi.I64, np.array([5, 6, 7]))]))
    p = raw.ctypes.data_as(C.c_void_p)
    # nothing to do is not an error: empty response, or a chunk that wants no rows (select_result.go:139-142)
    assert lib.tsq_rows_decode(ctx.h, p, 0, 0, 1, tp, out, 8, C.byref(n), C.byref(used)) == abi.OK and (n.value, used.value) == (0, 0)
    assert lib.tsq_rows_decode(ctx.h, p, raw.size, 0, 1, tp, out, 0, C.byref(n), C.byref(used)) == abi.OK and (n.value, used.value) == (0, 0)
    # misuse: NULL outputs, negative sizes, too many columns, var-len column type
    assert lib.tsq_rows_decode(ctx.h, p, raw.size, 0, 1, tp, out, 8, None, C.byref(used)) == abi.ERR_INVALID
    assert lib.tsq_rows_decode(ctx.h, p, -1, 0, 1, tp, out, 8, C.byref(n), C.byref(used)) == abi.ERR_INVALID
    assert lib.tsq_rows_decode(ctx.h, p, raw.size, 0, 17, tp, out, 8, C.byref(n), C.byref(used)) == abi.ERR_UNSUPPORTED
    assert lib.tsq_rows_decode(ctx.h, p, raw.size, 0, 1, (C.c_int32 * 1)(4), out, 8, C.byref(n), C.byref(used)) == abi.ERR_INVALID  # a var-len column without offsets
    # and the normal call still works on the same context afterwards
    _lib.check(lib.tsq_rows_decode(ctx.h, p, raw.size, 0, 1, tp, out, 8, C.byref(n), C.byref(used)), ctx.h)
    assert n.value == 3 and used.value == raw.size and bufs[0][0][:3].tolist() == [5, 6, 7]


# ---------------------------------------------------------------- round 5: var-len columns in the single-stream decoder (codec.go:670-690)
def _decode_varlen(ctx, raw, types, cap_rows, device=False):
    from tinysql_amd.chunk import chunk_from_buffers, out_buffers
    nbytes = len(raw)
    raw = np.ascontiguousarray(np.frombuffer(bytes(raw), np.uint8)) if nbytes else np.zeros(1, np.uint8)
    keep = []
    out, bufs = out_buffers(types, max(cap_rows, 1), keep, var_bytes=[max(nbytes, 1) if t == abi.BYTES else 0 for t in types])
    tp = (C.c_int32 * len(types))(*types)
    n, used = C.c_int64(0), C.c_int64(0)
    ptr, flags = raw.ctypes.data_as(C.c_void_p), 0
    dev = None
    if device:
        dev = ctx.alloc(raw.size + 64)
        ctx.h2d(dev, raw)
        ptr, flags = C.c_void_p(dev), abi.COL_DEVICE
    try:
        st = ctx.lib.tsq_rows_decode(ctx.h, ptr, nbytes, flags, len(types), tp, out, cap_rows, C.byref(n), C.byref(used))
    finally:
        if dev:
            ctx.free(dev)
    return st, chunk_from_buffers(types, bufs, n.value), used.value


def _varlen_table(rng, n):
    from tinysql_amd.chunk import StrColumn
    words = [None if rng.random() < 0.15 else bytes(rng.integers(0, 256, int(rng.integers(0, 60)), dtype=np.uint8)) for _ in range(n)]
    notes = [None if rng.random() < 0.1 else (b"" if rng.random() < 0.2 else b"n%d" % i) for i in range(n)]
    return Chunk([Column(abi.I64, rng.integers(-(1 << 40), 1 << 40, n), rng.random(n) > 0.2), StrColumn(words), Column(abi.F64, rng.standard_normal(n), rng.random(n) > 0.2),
                  StrColumn(notes), Column(abi.U64, rng.integers(0, 1 << 62, n).astype(np.uint64))])


@pytest.mark.parametrize("n", [1, 63, 64, 65, 5000])
def test_single_stream_with_bytes_datums_equals_the_oracle(ctx, orc, n):
    rng = np.random.default_rng(600 + n)
    t = _varlen_table(rng, n)
    raw = bytes(orc.encode_rows(t))  # ONE byte string, no chunk boundaries: compact-bytes datums between the numbers
    for device in (False, True):
        st, got, used = _decode_varlen(ctx, raw, t.types(), n + 10, device)
        assert st == abi.OK and used == len(raw) and got.rows() == t.rows()
    # the chunk-capacity contract of select_result.go:153: cap_rows rows, the remainder stays with the caller
    cap = max(1, n // 3)
    st, got, used = _decode_varlen(ctx, raw, t.types(), cap)
    assert st == abi.OK and got.rows() == t.slice(0, cap).rows() and used == len(bytes(orc.encode_rows(t.slice(0, cap))))
    st, rest, used2 = _decode_varlen(ctx, raw[used:], t.types(), n)
    assert st == abi.OK and rest.rows() == t.slice(cap, n).rows() and used + used2 == len(raw)


def test_single_stream_with_bytes_datums_reports_the_first_error_in_stream_order(ctx, orc):
    rng = np.random.default_rng(601)
    t = _varlen_table(rng, 300)
    raw = bytearray(bytes(orc.encode_rows(t)))
    at = len(bytes(orc.encode_rows(t.slice(0, 200))))
    raw[at] = 0x7E  # the first flag of row 200: no such flag (codec.go:683)
    st, got, used = _decode_varlen(ctx, bytes(raw), t.types(), 1000)
    assert st == abi.ERR_INVALID and _lib.last_error(ctx.h) == "invalid encoded key flag" and used == 0 and got.rows() == t.slice(0, 200).rows()
    # a string whose declared length runs past the end of the stream (bytes.go:156-158)
    cut = bytes(orc.encode_rows(t))[:at + 3]
    st, got, used = _decode_varlen(ctx, cut, t.types(), 1000)
    assert st == abi.ERR_INVALID and got.NumRows() == 200 and _lib.last_error(ctx.h) in ("insufficient bytes to decode value", "invalid encoded key")
    # a memcomparable bytes datum (flag 1, the EncodeKey form) decodes too: bytes.go:35-118
    from tinysql_amd.chunk import StrColumn
    key = Chunk([StrColumn([b"abcdefgh-long-key", b"", None, b"12345678"]), Column(abi.I64, np.array([1, 2, 3, 4]))])
    rawk = bytes(orc.encode_rows(key, comparable=True))
    st, got, used = _decode_varlen(ctx, rawk, key.types(), 10)
    assert st == abi.OK and got.rows() == key.rows() and used == len(rawk)
