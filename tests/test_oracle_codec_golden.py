"""The oracle's restatement of the coprocessor-response row codec (oracle/codec_rows.cpp) pinned on the reference's own
tests: util/codec/codec_test.go TestNumberCodec (:210-304), TestFloatCodec (:385-442), TestValueSizeOfSignedInt /
UnsignedInt (:771-809: closed-form lengths of the varint forms), TestDecodeOneToChunk (:600-668, fixed-width datums) and
the Go standard library's documented varint bytes.  CPU only."""
import math
import struct

import numpy as np
import pytest

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column

I64_TABLE = [-(1 << 63), -(1 << 31), -(1 << 15), -(1 << 7), 0, (1 << 7) - 1, (1 << 15) - 1, (1 << 31) - 1, (1 << 63) - 1, (1 << 47) - 1, -(1 << 47),
             (1 << 23) - 1, -(1 << 23), (1 << 55) - 1, -(1 << 55), 1, -1]                                        # codec_test.go:212-230
U64_TABLE = [0, 255, 65535, (1 << 32) - 1, (1 << 64) - 1, (1 << 24) - 1, (1 << 48) - 1, (1 << 56) - 1, 1, 32767, 127, (1 << 31) - 1, (1 << 63) - 1]  # :254-268
F64_TABLE = [-1.0, 0.0, 1.0, 1.7976931348623157e308, 3.4028234663852886e38, 1.401298464324817e-45, 5e-324, -math.inf, math.inf]  # :387-397


def _col(tp, vals):
    dt = {abi.I64: np.int64, abi.U64: np.uint64, abi.F64: np.float64, abi.F32: np.float32}[tp]
    return Column(tp, np.array(vals, dtype=dt))


@pytest.mark.parametrize("comparable", [False, True])
def test_number_codec_tables_round_trip(comparable):
    # TestNumberCodec: DecodeInt(EncodeInt(v)) == v and DecodeVarint(EncodeVarint(v)) == v (same for the unsigned forms)
    chk = Chunk([_col(abi.I64, I64_TABLE)])
    st, got, used = orc.decode_rows(orc.encode_rows(chk, comparable), [abi.I64], 100)
    assert st == 0 and got.columns[0].data.tolist() == I64_TABLE
    chk = Chunk([_col(abi.U64, U64_TABLE)])
    raw = orc.encode_rows(chk, comparable)
    st, got, used = orc.decode_rows(raw, [abi.U64], 100)
    assert st == 0 and used == raw.size and got.columns[0].data.tolist() == U64_TABLE
    if comparable:  # EncodeInt / EncodeUint: flag + 8 big-endian bytes, ints with the sign bit flipped (number.go:24-42)
        assert raw.size == 9 * len(U64_TABLE) and bytes(raw[:9]) == b"\x04" + bytes(8) and bytes(raw[9:18]) == b"\x04" + bytes(7) + b"\xff"
        raw = orc.encode_rows(Chunk([_col(abi.I64, [0, -1])]), True)
        assert bytes(raw) == b"\x03\x80" + bytes(7) + b"\x03\x7f" + b"\xff" * 7


def test_value_size_closed_form_matches_the_encoded_length():
    # TestValueSizeOfSignedInt / TestValueSizeOfUnsignedInt: len(encodeSignedInt(v, false)) == valueSizeOfSignedInt(v)
    for v in [64, 8192, 1048576, 134217728, 17179869184, 2199023255552, 281474976710656, 36028797018963968, 4611686018427387904]:
        for x in (v - 10, v, v + 10, -v, -v + 10, -v - 10):
            assert orc.encode_rows(Chunk([_col(abi.I64, [x])])).size == orc.value_size_signed(x), x
    for v in [128, 16384, 2097152, 268435456, 34359738368, 4398046511104, 562949953421312, 72057594037927936, 9223372036854775808]:
        for x in (v - 10, v, v + 10):
            assert orc.encode_rows(Chunk([_col(abi.U64, [x])])).size == orc.value_size_unsigned(x), x
    # the closed form itself on the boundaries the test is built around: 2 bytes (flag + 1) up to 63 / 127, then +1 per 7 bits
    assert [orc.value_size_signed(x) for x in (0, 63, -64, 64, -65, 8191, 8192)] == [2, 2, 2, 3, 3, 3, 4]
    assert [orc.value_size_unsigned(x) for x in (0, 127, 128, 16383, 16384, (1 << 64) - 1)] == [2, 2, 3, 3, 4, 11]


def test_go_stdlib_varint_bytes():
    # encoding/binary: PutVarint zig-zag (0 -> 00, -1 -> 01, 1 -> 02, -2 -> 03, 63 -> 7e, -64 -> 7f, 64 -> 80 01, -65 -> 81 01),
    # PutUvarint(300) = ac 02; MinInt64 -> ff*9 01
    enc = lambda v, tp=abi.I64: bytes(orc.encode_rows(Chunk([_col(tp, [v])])))
    assert [enc(v) for v in (0, -1, 1, -2, 63, -64, 64, -65)] == [b"\x08\x00", b"\x08\x01", b"\x08\x02", b"\x08\x03", b"\x08\x7e", b"\x08\x7f",
                                                                 b"\x08\x80\x01", b"\x08\x81\x01"]
    assert enc(300, abi.U64) == b"\x09\xac\x02" and enc(-(1 << 63)) == b"\x08" + b"\xff" * 9 + b"\x01"
    assert enc((1 << 64) - 1, abi.U64) == b"\x09" + b"\xff" * 9 + b"\x01"


def test_float_codec_round_trip_and_order():
    # TestFloatCodec: round trip, and bytes.Compare of the encodings orders like the floats (:399-441)
    raw = orc.encode_rows(Chunk([_col(abi.F64, F64_TABLE)]))
    st, got, _ = orc.decode_rows(raw, [abi.F64], 100)
    assert st == 0 and raw.size == 9 * len(F64_TABLE)
    assert [struct.pack("<d", x) for x in got.columns[0].data.tolist()] == [struct.pack("<d", x) for x in F64_TABLE]
    enc = lambda f: bytes(orc.encode_rows(Chunk([_col(abi.F64, [f])])))[1:]
    big, f32max, tiny = 1.7976931348623157e308, 3.4028234663852886e38, 5e-324
    for a, b, ret in [(1, -1, 1), (1, 0, 1), (0, -1, 1), (0, 0, 0), (big, 1, 1), (f32max, big, -1), (big, 0, 1), (big, tiny, 1), (-math.inf, 0, -1),
                      (math.inf, 0, 1), (-math.inf, math.inf, -1)]:
        ea, eb = enc(float(a)), enc(float(b))
        assert (ea > eb) - (ea < eb) == ret, (a, b)


def test_decode_one_to_chunk_row_of_the_reference_test():
    # TestDecodeOneToChunk's datumsForTest, fixed-width datums only (strings/blobs are var-len: not on this path):
    # nil, tiny 1, short 1, int24 1, long 1, long -1, longlong 1, uint64 1, float32 1, double 1, year 1 — three rows
    types = [abi.I64] * 7 + [abi.U64, abi.F32, abi.F64, abi.I64]
    vals = [None, 1, 1, 1, 1, -1, 1, 1, 1.0, 1.0, 1]
    cols = []
    for t, v in zip(types, vals):
        c = _col(t, [0 if v is None else v] * 3)
        if v is None:
            c = Column(t, c.data, np.zeros(3, bool))
        cols.append(c)
    raw = orc.encode_rows(Chunk(cols))
    assert bytes(raw[:raw.size // 3]) == b"\x00" + b"\x08\x02" * 4 + b"\x08\x01" + b"\x08\x02" + b"\x09\x01" + (b"\x05\xbf\xf0" + bytes(6)) * 2 + b"\x08\x02"
    st, got, used = orc.decode_rows(raw, types, 32)
    assert st == 0 and got.NumRows() == 3 and used == raw.size
    for r in got.rows():
        assert list(r) == vals
    assert got.columns[8].data.dtype == np.float32  # appendFloatToChunk: TypeFloat -> float32 (codec.go:701-707)


def test_read_rows_data_stops_at_the_chunk_capacity_and_keeps_the_remainder():
    # select_result.go:139-155: decode until chk.IsFull(), keep the remaining bytes for the next call
    chk = Chunk([_col(abi.I64, list(range(-5, 5))), _col(abi.F64, [x / 4 for x in range(10)])])
    raw = orc.encode_rows(chk)
    st, a, used = orc.decode_rows(raw, [abi.I64, abi.F64], 4)
    st2, b, used2 = orc.decode_rows(raw[used:], [abi.I64, abi.F64], 100)
    assert (st, st2) == (0, 0) and a.NumRows() == 4 and b.NumRows() == 6 and used + used2 == raw.size
    assert a.rows() + b.rows() == chk.rows()


def test_decode_errors_of_the_reference():
    ok = orc.encode_rows(Chunk([_col(abi.I64, [1000]), _col(abi.I64, [7])]))
    t2 = [abi.I64, abi.I64]
    assert orc.decode_rows(ok[:-2], t2, 9)[0] == 1                                   # row ends after its first datum: "invalid encoded key"
    assert orc.decode_rows(ok[:2], t2, 9)[0] == 2                                    # varint cut in the middle: insufficient bytes
    assert orc.decode_rows(np.frombuffer(b"\x03\x80\x00", np.uint8), [abi.I64], 9)[0] == 2
    assert orc.decode_rows(np.frombuffer(b"\x08" + b"\xff" * 10 + b"\x01", np.uint8), [abi.I64], 9)[0] == 3   # > 10 bytes: overflow
    assert orc.decode_rows(np.frombuffer(b"\x08" + b"\xff" * 9 + b"\x02", np.uint8), [abi.I64], 9)[0] == 3    # 10th byte > 1: overflow
    assert orc.decode_rows(np.frombuffer(b"\x07\x00", np.uint8), [abi.I64], 9)[0] == 4                        # duration flag: invalid here
    assert orc.decode_rows(np.frombuffer(b"\x02\x02ab", np.uint8), [abi.I64], 9)[0] == 5                      # compact bytes: var-len
    st, chk, used = orc.decode_rows(np.concatenate([ok, ok[:1]]), t2, 9)                                       # complete row, then garbage
    assert st == 2 and chk.NumRows() == 1 and used == ok.size


def test_the_bench_tool_numpy_encoder_writes_the_same_bytes():
    # tools/bench_decode.py generates its input without the oracle; its vectorised EncodeValue must be the reference format
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_decode", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_decode.py"))
    bd = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bd)
    rng = np.random.default_rng(0)
    m = 20_000
    a = rng.integers(-(1 << 62), 1 << 62, m)
    a[:10] = [0, -1, 1, 63, -64, 64, -65, (1 << 63) - 1, -(1 << 63), 300]
    b = rng.integers(0, 2500, m)
    c = rng.random(m) * 1e5 - 5e4
    d = rng.integers(0, 11, m) / 100.0
    want = orc.encode_rows(Chunk([Column(abi.I64, a), Column(abi.I64, b), Column(abi.F64, c), Column(abi.F64, d)]))
    got = bd.encode_value_rows([a, b, c, d])
    assert got.size == want.size and (got == want).all()


def test_storage_boundary_codecs_against_the_transcribed_vectors(orc):
    # tests/golden/codec_cases.json (bytes_test.go:33-78, tablecodec_test.go:55-75, chunk/codec_test.go:29-71)
    import json
    import os

    import numpy as np

    from tinysql_amd import _abi as abi
    from tinysql_amd.chunk import Chunk, Column, StrColumn

    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "codec_cases.json")))
    for c in g["bytes_codec"]:
        enc = bytes([1] + c["enc"])  # the bytesFlag datum around EncodeBytes
        assert orc.encode_rows(Chunk([StrColumn([bytes(c["dec"])])]), comparable=True).tobytes() == enc
        st, rows = orc.decode_rows_chunks(enc, [0, len(enc)], [abi.BYTES])
        assert st == 0 and rows.rows() == [(bytes(c["dec"]),)]
    for e in g["bytes_codec_errors"]:
        enc = bytes([1] + e)
        assert orc.decode_rows_chunks(enc, [0, len(enc)], [abi.BYTES])[0] != 0
    k = g["cut_index_key"]
    t = Chunk([Column(abi.I64, np.array([k["values"][0]])), StrColumn([k["values"][1].encode()]), Column(abi.F64, np.array([k["values"][2]]))])
    keys, offs = orc.encode_index_keys(t, k["table_id"], k["index_id"], np.array([k["handle"]]), np.array([1], np.uint8))
    st, rows = orc.decode_index_kv(keys.tobytes(), offs, None, None, k["cols_len"], [abi.I64, abi.BYTES, abi.F64, abi.I64], 1)
    assert st == 0 and rows.rows() == [(k["values"][0], k["values"][1].encode(), k["values"][2], k["handle"])]
    w = g["chunk_codec"]
    n = w["rows"]
    chk = Chunk([Column(abi.I64, np.zeros(n, np.int64), np.zeros(n, bool)), Column(abi.I64, np.arange(n)), StrColumn([b"%d.12345" % i for i in range(n)]),
                 StrColumn([b"%d.12345" % i for i in range(n)])])
    buf = orc.WireChunk.from_chunk(chk).encode()
    assert len(buf) == w["wire_bytes"] == sum(w["column_bytes"])
    assert list(buf[:8]) == w["first_column_header"] and list(buf[w["column_bytes"][0]:w["column_bytes"][0] + 8]) == w["second_column_header"]
    back = orc.WireChunk([8, 8, -1, -1])
    assert back.decode_to_chunk(buf) == len(buf) and np.frombuffer(back.column(1)[3], np.int64).tolist() == list(range(n))
