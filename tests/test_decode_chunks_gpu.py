"""GPU parity of tsq_rows_decode_chunks (the tipb.Chunks of a coprocessor response -> chunk columns, var-len columns included; SURVEY.md
§8 f rank 2) against the oracle's restatement of selectResult.readRowsData + Decoder.DecodeOne with compact-bytes datums
(distsql/select_result.go:102-155, util/codec/codec.go:623-690, bytes.go:150-160): random responses with strings, NULLs and empty
strings at several chunk sizes, host and device placement at odd pointer alignments, long cells, every error of the reference at
its position in the stream, the capacity contract, and the SelectResult mirror handing rows out in pieces."""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import distsql
from tinysql_amd.chunk import Chunk, Column, StrColumn

from .test_hostsim_decode_chunks import response, table

pytestmark = pytest.mark.gpu
MSG = {1: "invalid encoded key", 2: "insufficient bytes to decode value", 3: "value larger than 64 bits", 4: "invalid encoded key flag",
       6: "datum kind does not match the column type", 7: "invalid marker byte", 8: "invalid padding byte"}


def chunks_of(data, offs):
    return [bytes(data[int(offs[i]):int(offs[i + 1])]) for i in range(len(offs) - 1)]


@pytest.mark.parametrize("n,per", [(1, 64), (64, 64), (65, 64), (5000, 64), (5000, 7), (200_000, 64), (400, 1)])
def test_responses_against_the_oracle(ctx, orc, n, per):
    rng = np.random.default_rng(n + per)
    t = table(rng, n, long_strings=(n == 400))
    data, offs = response(t, per)
    st, want = orc.decode_rows_chunks(data, offs, t.types())
    got = distsql.decode_chunks(ctx, chunks_of(data, offs), t.types(), cap_rows=n)
    assert st == 0 and got.NumRows() == n and got.rows() == want.rows()


@pytest.mark.parametrize("n,per,long_strings", [(1, 64, False), (5000, 64, False), (5000, 7, False), (400, 1, True), (600, 64, True)])
def test_comparable_responses_memcomparable_strings(ctx, orc, n, per, long_strings):
    # every value in its EncodeKey form (what an index scan answers with): ints flag 3, uints flag 4, strings bytesFlag + groups of
    # 8 bytes + marker (util/codec/bytes.go:35-67) -> DecodeOne's DecodeBytes (codec.go:662-668, bytes.go:69-118)
    rng = np.random.default_rng(n * 3 + per)
    t = table(rng, n, long_strings=long_strings)
    if n == 600:  # long cells: the byte copy runs one cell per wave
        t = Chunk([t.columns[0], StrColumn([None if i % 11 == 0 else bytes([65 + i % 26]) * (700 + i % 9) for i in range(n)])])
    parts = [bytes(orc.encode_rows(t.slice(lo, min(lo + per, n)), comparable=True)) for lo in range(0, n, per)]
    data, offs = b"".join(parts), np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    st, want = orc.decode_rows_chunks(data, offs, t.types())
    got = distsql.decode_chunks(ctx, chunks_of(data, offs), t.types(), cap_rows=n)
    assert st == 0 and got.NumRows() == n and got.rows() == want.rows() == t.rows()


def test_reference_bytes_codec_vectors_on_the_gpu(ctx, orc):
    # util/codec/bytes_test.go:33-47, 68-78
    from . import test_hostsim_decode_chunks as T
    for dec, enc in T.BYTES_CODEC:
        assert distsql.decode_chunks(ctx, [bytes([1] + enc)], [abi.BYTES]).rows() == [(bytes(dec),)]
    for enc in T.BYTES_CODEC_ERR:
        data = bytes([1] + enc)
        st, _ = orc.decode_rows_chunks(data, [0, len(data)], [abi.BYTES])
        with pytest.raises(_lib.TsqError) as e:
            distsql.decode_chunks(ctx, [data], [abi.BYTES])
        assert e.value.status == abi.ERR_INVALID and MSG[st] in str(e.value)


def test_long_cells_one_per_wave(ctx, orc):
    # a 5 KiB payload like the reference's join benchmark (executor/benchmark_test.go:328): the byte copy runs one cell per wave
    rng = np.random.default_rng(9)
    n = 600
    t = Chunk([Column(abi.I64, rng.integers(0, 100, n), None), StrColumn([None if i % 11 == 0 else bytes([65 + i % 26]) * (5000 + i % 7) for i in range(n)])])
    data, offs = response(t, 64)
    got = distsql.decode_chunks(ctx, chunks_of(data, offs), t.types())
    assert got.rows() == t.rows()


@pytest.mark.parametrize("phase", [0, 1, 5])
def test_device_resident_response_and_columns(ctx, orc, phase):
    rng = np.random.default_rng(20 + phase)
    n = 30_000
    t = table(rng, n)
    data, offs = response(t, 64)
    raw = np.frombuffer(data, dtype=np.uint8)
    types = t.types()
    st, want = orc.decode_rows_chunks(data, offs, types)
    dd, do = ctx.alloc(raw.size + 64 + phase), ctx.alloc(offs.nbytes + 64)
    bufs = []
    out = (abi.Col * len(types))()
    try:
        ctx.h2d(dd + phase, raw)
        ctx.h2d(do, offs)
        for c, tp in enumerate(types):
            es = 4 if tp == abi.F32 else 8
            d = ctx.alloc((raw.size if tp == abi.BYTES else n * es) + 64)
            bm = ctx.alloc(n // 8 + 64)
            of = ctx.alloc(8 * (n + 1) + 64) if tp == abi.BYTES else None
            bufs.append((d, bm, of))
            out[c].data, out[c].null_bitmap, out[c].offsets, out[c].length, out[c].type, out[c].flags = d, bm, of, n, tp, abi.COL_DEVICE
            out[c].elem_size = -1 if tp == abi.BYTES else es
        tpa = (C.c_int32 * len(types))(*types)
        m = C.c_int64(0)
        _lib.check(ctx.lib.tsq_rows_decode_chunks(ctx.h, C.c_void_p(dd + phase), raw.size, C.c_void_p(do), len(offs) - 1, abi.COL_DEVICE, len(types), tpa, out, n,
                                                  C.byref(m)), ctx.h)
        assert m.value == n
        cols = []
        for (d, bm, of), tp in zip(bufs, types):
            bits = np.zeros(n // 8 + 8, np.uint8)
            ctx.d2h(bits, bm)
            nn = np.unpackbits(bits, bitorder="little")[:n].astype(bool)
            if tp == abi.BYTES:
                o = np.zeros(n + 1, np.int64)
                ctx.d2h(o, of)
                by = np.zeros(max(int(o[n]), 1), np.uint8)
                ctx.d2h(by, d)
                cols.append(StrColumn([bytes(by[o[r]:o[r + 1]]) if nn[r] else None for r in range(n)]))
            else:
                v = np.zeros(n, {abi.I64: np.int64, abi.U64: np.uint64, abi.F64: np.float64, abi.F32: np.float32}[tp])
                ctx.d2h(v, d)
                v[~nn] = 0
                cols.append(Column(tp, v, nn))
        assert Chunk(cols).rows() == want.rows()
    finally:
        ctx.free(dd)
        ctx.free(do)
        for d, bm, of in bufs:
            ctx.free(d)
            ctx.free(bm)
            if of:
                ctx.free(of)


@pytest.mark.parametrize("case", ["bad_flag", "bytes_flag", "cut_int", "cut_varint", "long_varint", "cut_string", "negative_length", "row_cut", "kind", "kind_reverse", "offsets"])
def test_first_error_in_stream_order(ctx, orc, case):
    # the same damaged responses the CPU walk of the kernels is checked on (tests/test_hostsim_decode_chunks.py)
    from . import test_hostsim_decode_chunks as T
    rng = np.random.default_rng(11)
    types = [abi.I64, abi.BYTES, abi.F64]
    t = Chunk([Column(abi.I64, rng.integers(-5, 5, 200), None), StrColumn([b"s%03d" % i for i in range(200)]), Column(abi.F64, rng.random(200), None)])
    data, offs = response(t, 20)
    at = int(offs[3]) + 5 * 17
    b = bytearray(data)
    offs = offs.copy()
    if case == "bad_flag":
        b[at] = 7
    elif case == "bytes_flag":
        b[at + 2] = 1
    elif case in ("cut_int", "cut_varint"):
        b[int(offs[4]) - 9:int(offs[4])] = b"\x03\x00\x00" if case == "cut_int" else b"\x08\x80\x80"
        offs[4:] -= 6
    elif case == "long_varint":
        b[at:at + 2] = b"\x08\x80"
        b[at + 2:at + 2] = b"\x80" * 10
        offs[4:] += 10
    elif case == "cut_string":
        b[at + 3] = 0xC8
    elif case == "negative_length":
        b[at + 3] = 1
    elif case == "row_cut":
        del b[int(offs[4]) - 9:int(offs[4])]
        offs[4:] -= 9
    elif case == "kind":
        b[at + 2:at + 8] = b"\x08\x02\x08\x02\x08\x02"
    elif case == "kind_reverse":
        types = [abi.I64, abi.I64, abi.F64]
    else:
        offs[4] = offs[3] - 1
    data = bytes(b)
    st, want = orc.decode_rows_chunks(data, offs, types)
    assert st != 0
    raw = np.frombuffer(data, dtype=np.uint8)
    from tinysql_amd.chunk import chunk_from_buffers, out_buffers
    keep = []
    out, bufs = out_buffers(types, 256, keep, var_bytes=[raw.size if tp == abi.BYTES else 0 for tp in types])
    tpa = (C.c_int32 * 3)(*types)
    m = C.c_int64(0)
    gst = ctx.lib.tsq_rows_decode_chunks(ctx.h, raw.ctypes.data_as(C.c_void_p), raw.size, offs.ctypes.data_as(C.c_void_p), len(offs) - 1, 0, 3, tpa, out, 256, C.byref(m))
    assert gst == abi.ERR_INVALID and _lib.last_error(ctx.h) == MSG[st]
    assert m.value == want.NumRows() and chunk_from_buffers(types, bufs, m.value).rows() == want.rows()  # the rows before the error were handed over


def test_capacity_contract_and_the_select_result_mirror(ctx, orc):
    rng = np.random.default_rng(5)
    t = table(rng, 1000)
    data, offs = response(t, 100)  # chunks of 100 rows: more than the 64 the mirror assumes
    chunks = chunks_of(data, offs)
    assert distsql.decode_chunks(ctx, chunks, t.types()).rows() == t.rows()  # first call too small (640 rows), repeated with the reported size
    sr = distsql.SelectResult(ctx, chunks, t.types())
    rows, sizes = [], []
    while True:
        c = sr.Next(333)
        if c.NumRows() == 0:
            break
        sizes.append(c.NumRows())
        rows += c.rows()
    assert sizes == [333, 333, 333, 1] and rows == t.rows()
    # misuse
    lib = ctx.lib
    m = C.c_int64(0)
    raw = np.frombuffer(data, dtype=np.uint8)
    from tinysql_amd.chunk import out_buffers
    keep = []
    out, _ = out_buffers([abi.I64], 8, keep)
    tp = (C.c_int32 * 1)(abi.BYTES)
    assert lib.tsq_rows_decode_chunks(ctx.h, raw.ctypes.data_as(C.c_void_p), raw.size, offs.ctypes.data_as(C.c_void_p), len(offs) - 1, 0, 1, tp, out, 8, C.byref(m)) == abi.ERR_INVALID
    assert lib.tsq_rows_decode_chunks(ctx.h, raw.ctypes.data_as(C.c_void_p), raw.size, None, 3, 0, 1, tp, out, 8, C.byref(m)) == abi.ERR_INVALID
    tp[0] = abi.I64
    assert lib.tsq_rows_decode_chunks(ctx.h, None, 0, None, 0, 0, 1, tp, out, 8, C.byref(m)) == abi.OK and m.value == 0
