"""The oracle's restatement of the stored-row format (oracle/rowcodec.cpp) pinned on the reference's own tests:
util/rowcodec/rowcodec_test.go TestDecodeRowWithHandle (:49-163), TestTypesNewRowCodec (:165-328: small ids, a large column
id, a value of 65536 bytes), TestNilAndDefault (:330-438), TestVarintCompatibility (:440-503: DecodeToBytes output is
byte-identical to tablecodec.EncodeValue — checked against the codec restatement pinned on codec_test.go) and TestCodecUtil
(:505-556, ColumnIsNull).  The reference has no byte-level vectors for this format, so the layout itself is pinned by rows
assembled by hand from row.toBytes (util/rowcodec/row.go:80-99) and encodeInt / EncodeFloat.  CPU only."""
import struct

import numpy as np
import pytest

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column

NP = {abi.I64: np.int64, abi.U64: np.uint64, abi.F64: np.float64, abi.F32: np.float32}


def _col(tp, vals, notnull=None):
    return Column(tp, np.array(vals, dtype=NP[tp]), None if notnull is None else np.array(notnull, dtype=bool))


def _row(b, o, r):
    return b[o[r]:o[r + 1]]


def test_layout_small_row_by_hand():
    # ids 1 (int 1), 22 (uint 1), 3 (double 2), 12 (year 1999), NULL ids 11, 2, 100, 116 (float32 6) — the fixed-width part of
    # TestTypesNewRowCodec.  toBytes: ver, flag, numNotNull, numNull, ids (not-null sorted, then null sorted), end offsets, data
    chk = Chunk([_col(abi.I64, [1]), _col(abi.U64, [1]), _col(abi.F64, [2.0]), _col(abi.I64, [1999]), _col(abi.I64, [0], [False]),
                 _col(abi.I64, [0], [False]), _col(abi.I64, [0], [False]), _col(abi.F32, [6.0])])
    b, o = orc.rowcodec_encode(chk, [1, 22, 3, 12, 11, 2, 100, 116])
    f2 = struct.pack(">Q", struct.unpack(">Q", struct.pack(">d", 2.0))[0] | (1 << 63))  # EncodeFloat: sign bit set for f >= 0
    f6 = struct.pack(">Q", struct.unpack(">Q", struct.pack(">d", 6.0))[0] | (1 << 63))
    want = (bytes([128, 0, 5, 0, 3, 0]) + bytes([1, 3, 12, 22, 116]) + bytes([2, 11, 100]) +
            struct.pack("<5H", 1, 9, 11, 12, 20) + bytes([1]) + f2 + struct.pack("<h", 1999) + bytes([1]) + f6)
    assert bytes(b) == want and o.tolist() == [0, len(want)]


def test_layout_large_ids_by_hand():
    # id 300 -> large: 4-byte ids and 4-byte offsets (row.go:62-68)
    chk = Chunk([_col(abi.I64, [-2]), _col(abi.I64, [70000]), _col(abi.I64, [0], [False])])
    b, o = orc.rowcodec_encode(chk, [300, 7, 9])
    want = (bytes([128, 1, 2, 0, 1, 0]) + struct.pack("<3I", 7, 300, 9) + struct.pack("<2I", 4, 5) + struct.pack("<i", 70000) + bytes([0xfe]))
    assert bytes(b) == want


def test_encode_int_widths():
    # encodeInt / encodeUint (common.go:84-101, 180-197): the narrowest of 1, 2, 4, 8 bytes that holds the value
    for v, w in [(0, 1), (127, 1), (-128, 1), (128, 2), (-129, 2), (32767, 2), (32768, 4), (-32769, 4), ((1 << 31) - 1, 4), (1 << 31, 8), (-(1 << 31) - 1, 8),
                 (-(1 << 63), 8), ((1 << 63) - 1, 8)]:
        b, _ = orc.rowcodec_encode(Chunk([_col(abi.I64, [v])]), [1])
        assert len(b) == 6 + 1 + 2 + w, v
    for v, w in [(0, 1), (255, 1), (256, 2), (65535, 2), (65536, 4), ((1 << 32) - 1, 4), (1 << 32, 8), ((1 << 64) - 1, 8)]:
        b, _ = orc.rowcodec_encode(Chunk([_col(abi.U64, [v])]), [1])
        assert len(b) == 6 + 1 + 2 + w, v


def _roundtrip(chk, ids, specs, handles=None, pad=None):
    b, o = orc.rowcodec_encode(chk, ids, *(pad or ()))
    st, got = orc.rowcodec_decode(b, o, handles, specs)
    assert st == 0
    return b, o, got


def test_decode_row_with_handle():
    # rowcodec_test.go:49-163: the handle column takes its value from the key, not from the row; with the unsigned flag the
    # chunk still gets the int64 bits (AppendInt64) while DecodeToBytes emits a uint datum
    chk = Chunk([_col(abi.I64, [1])])
    for htype, flag in ((abi.I64, 3), (abi.U64, 4)):
        specs = [(-1, htype, abi.RC_HANDLE), (10, abi.I64)]
        b, o, got = _roundtrip(chk, [10], specs, [10000])
        assert got.rows() == [(10000, 1)]
        old = orc.rowcodec_to_old_bytes(b, 10000, specs)
        key = 10000 ^ (1 << 63) if htype == abi.I64 else 10000
        assert bytes(old) == bytes([flag]) + struct.pack(">Q", key) + b"\x08\x02"
        st, dec, used = orc.decode_rows(old, [htype, abi.I64], 1)  # codec.DecodeOne of every old value gives the datum back
        assert st == 0 and used == old.size and dec.rows() == [(10000, 1)]


@pytest.mark.parametrize("first_id,pad_len", [(1, 3), (300, 3), (1, 65536)])
def test_types_new_row_codec(first_id, pad_len):
    # rowcodec_test.go:165-328: every type next to each other; small ids, one large id, one value of 65536 bytes (-> large row)
    chk = Chunk([_col(abi.I64, [1]), _col(abi.U64, [1]), _col(abi.F64, [2.0]), _col(abi.I64, [1999]), _col(abi.I64, [0], [False]),
                 _col(abi.I64, [0], [False]), _col(abi.I64, [0], [False]), _col(abi.F32, [6.0])])
    ids = [first_id, 22, 3, 12, 11, 2, 100, 116]
    types = [abi.I64, abi.U64, abi.F64, abi.I64, abi.I64, abi.I64, abi.I64, abi.F32]
    specs = list(zip(ids, types))
    b, o, got = _roundtrip(chk, ids, specs, pad=(24, [pad_len]))
    assert got.rows() == [(1, 1, 2.0, 1999, None, None, None, 6.0)]
    assert b[1] == (1 if first_id > 255 or pad_len > 65535 else 0)
    old = orc.rowcodec_to_old_bytes(b, -1, specs)
    st, dec, used = orc.decode_rows(old, types, 1)
    assert st == 0 and used == old.size and dec.rows() == [(1, 1, 2.0, 1999, None, None, None, 6.0)]


def test_nil_and_default():
    # rowcodec_test.go:330-438: column 2 is not in the row and has the default 9 -> the chunk gets 9; without a default: NULL
    chk = Chunk([_col(abi.I64, [1])])
    _, _, got = _roundtrip(chk, [1], [(1, abi.I64), (2, abi.U64, abi.RC_HAS_DEFAULT, 9)])
    assert got.rows() == [(1, 9)]
    _, _, got = _roundtrip(chk, [1], [(1, abi.I64), (2, abi.U64)])
    assert got.rows() == [(1, None)]
    # a column that IS in the row as NULL stays NULL even when a default exists (decoder.go:181-184)
    chk = Chunk([_col(abi.I64, [1]), _col(abi.U64, [0], [False])])
    _, _, got = _roundtrip(chk, [1, 2], [(1, abi.I64), (2, abi.U64, abi.RC_HAS_DEFAULT, 9)])
    assert got.rows() == [(1, None)]


def test_varint_compatibility():
    # rowcodec_test.go:440-503: DecodeToBytes(new row) == tablecodec.EncodeValue(datum) byte for byte (varint / varuint forms)
    rng = np.random.default_rng(5)
    iv = np.concatenate([[1, -1, 0, 127, -128, 1 << 40, -(1 << 62)], rng.integers(-(1 << 62), 1 << 62, 50)])
    uv = np.concatenate([np.array([1, 0, 255, 65536, (1 << 64) - 1], dtype=np.uint64), rng.integers(0, 1 << 63, 50).astype(np.uint64)])
    n = min(len(iv), len(uv))
    chk = Chunk([_col(abi.I64, iv[:n]), _col(abi.U64, uv[:n]), _col(abi.F64, rng.standard_normal(n))])
    specs = [(1, abi.I64), (2, abi.U64), (3, abi.F64)]
    b, o = orc.rowcodec_encode(chk, [1, 2, 3])
    old = np.concatenate([orc.rowcodec_to_old_bytes(_row(b, o, r), 1, specs) for r in range(n)])
    assert bytes(old) == bytes(orc.encode_rows(chk))


def test_column_is_null():
    # rowcodec_test.go:505-556: ids 1, 2, 3 = 1, 2, 3 and id 4 = NULL
    chk = Chunk([_col(abi.I64, [1]), _col(abi.I64, [2]), _col(abi.I64, [3]), _col(abi.I64, [0], [False])])
    b, _ = orc.rowcodec_encode(chk, [1, 2, 3, 4])
    assert b[0] == 128  # IsNewFormat
    assert orc.rowcodec_column_is_null(b, 4) == 1 and orc.rowcodec_column_is_null(b, 1) == 0
    assert orc.rowcodec_column_is_null(b, 5) == 1 and orc.rowcodec_column_is_null(b, 5, has_default=True) == 0


def test_both_decoders_agree_on_random_scans():
    # the two paths the reference tests against the same expected datums: ChunkDecoder.DecodeToChunk directly, and
    # BytesDecoder.DecodeToBytes -> RowsData -> readRowsData / DecodeOne (the coprocessor path, pinned on codec_test.go)
    rng = np.random.default_rng(11)
    n = 2000
    edge = np.array([0, 1, -1, 127, -128, 128, -129, 32767, -32768, 32768, (1 << 31) - 1, -(1 << 31), 1 << 31, (1 << 63) - 1, -(1 << 63)])
    chk = Chunk([
        Column(abi.I64, np.where(rng.random(n) < 0.5, rng.choice(edge, n), rng.integers(-(1 << 62), 1 << 62, n)), rng.random(n) > 0.2),
        Column(abi.U64, (rng.integers(0, 1 << 62, n) >> rng.integers(0, 62, n)).astype(np.uint64), rng.random(n) > 0.2),
        Column(abi.F64, rng.standard_normal(n) * 1e6, rng.random(n) > 0.2),
        Column(abi.F32, rng.standard_normal(n).astype(np.float32), rng.random(n) > 0.2),
    ])
    ids = [7, 2, 200, 31]
    specs = [(200, abi.F64), (-1, abi.I64, abi.RC_HANDLE), (7, abi.I64), (31, abi.F32), (2, abi.U64), (99, abi.I64), (98, abi.I64, abi.RC_HAS_DEFAULT, 5)]
    handles = rng.integers(-(1 << 62), 1 << 62, n)
    b, o = orc.rowcodec_encode(chk, ids)
    st, got = orc.rowcodec_decode(b, o, handles, specs)
    assert st == 0 and got.NumRows() == n
    want = Chunk([chk.columns[2], Column(abi.I64, handles), chk.columns[0], chk.columns[3], chk.columns[1],
                  Column(abi.I64, np.zeros(n, np.int64), np.zeros(n, bool)), Column(abi.I64, np.full(n, 5))])
    assert got.rows() == want.rows()
    # through the old datum bytes (no default bytes there: column 98 is NULL on that path, decoder.go:289-301)
    sp2 = specs[:6]
    old = np.concatenate([orc.rowcodec_to_old_bytes(_row(b, o, r), int(handles[r]), sp2) for r in range(n)])
    st, dec, used = orc.decode_rows(old, [s[1] for s in sp2], n)
    assert st == 0 and used == old.size and dec.rows() == Chunk(want.columns[:6]).rows()


def test_errors():
    chk = Chunk([_col(abi.I64, [5, 6, 7]), _col(abi.F64, [1.5, 2.5, 3.5])])
    b, o = orc.rowcodec_encode(chk, [1, 2])
    specs = [(1, abi.I64), (2, abi.F64)]
    bad = b.copy()
    bad[o[1]] = 127  # row.go:54-56: not the new format
    st, got = orc.rowcodec_decode(bad, o, None, specs)
    assert st == 1 and got.rows() == [(5, 1.5)]
    # a real of fewer than 8 bytes: DecodeFloat -> "insufficient bytes to decode value"; an int of 8+ bytes reads its first 8
    st, got = orc.rowcodec_decode(b, o, None, [(2, abi.I64), (1, abi.F64)])
    assert st == 3 and got.NumRows() == 0
    # a row cut in the middle of its offsets array: the reference panics, the restatement reports status 2
    cut = o.copy()
    cut[3] = o[2] + 9
    st, got = orc.rowcodec_decode(b, cut, None, specs)
    assert st == 2 and got.NumRows() == 2


def test_the_bench_tool_numpy_encoder_writes_the_same_bytes():
    # tools/bench_rowcodec.py (and the full-size GPU test) generate stored rows without the oracle; its vectorised Encoder.Encode
    # must be the reference format byte for byte
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_rowcodec", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_rowcodec.py"))
    br = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(br)
    rng = np.random.default_rng(0)
    m = 20_000
    a = rng.integers(-(1 << 62), 1 << 62, m) >> rng.integers(0, 62, m)
    a[:12] = [0, -1, 1, 127, -128, 128, -129, 32767, -32768, 32768, (1 << 63) - 1, -(1 << 63)]
    b = rng.integers(0, 2500, m)
    c = rng.random(m) * 1e5 - 5e4
    u = (rng.integers(0, 1 << 63, m).astype(np.uint64) >> rng.integers(0, 63, m).astype(np.uint64)) * np.uint64(2)
    want_b, want_o = orc.rowcodec_encode(Chunk([Column(abi.I64, a), Column(abi.I64, b), Column(abi.F64, c), Column(abi.U64, u)]), [9, 3, 200, 4])
    got_b, got_o = br.encode_rows_v2([a, b, c, u], [9, 3, 200, 4])
    assert (got_o == want_o).all() and got_b.size == want_b.size and (got_b == want_b).all()


def test_string_columns_round_trip_and_layout():
    # rowcodec_test.go:165-328 (TestTypesNewRowCodec) carries a varchar next to the numbers: EncodeValueDatum writes the bytes as
    # they are (encoder.go:180-181), decodeColToChunk hands them to chk.AppendBytes (decoder.go:226-228); a NULL string is a NULL id
    from tinysql_amd.chunk import StrColumn
    chk = Chunk([Column(abi.I64, np.array([7, -300, 0]), np.array([True, True, False])), StrColumn([b"abc", None, b""]),
                 StrColumn([b"\x00\xff", b"x" * 300, b"tail"])])
    b, o = orc.rowcodec_encode(chk, [1, 2, 9])
    # row 0 by hand (row.toBytes, row.go:80-99): 3 not-null ids 1, 2, 9; end offsets 1, 4, 6; data = 07 | "abc" | 00 ff
    assert bytes(b[o[0]:o[1]]) == bytes([128, 0, 3, 0, 0, 0, 1, 2, 9, 1, 0, 4, 0, 6, 0, 7]) + b"abc" + b"\x00\xff"
    # row 1: the string under id 2 is NULL -> ids 1, 9 then the null id 2; -300 takes two bytes
    assert bytes(b[o[1]:o[1] + 13]) == bytes([128, 0, 2, 0, 1, 0, 1, 9, 2, 2, 0, 46, 1])
    specs = [(9, abi.BYTES), (1, abi.I64), (2, abi.BYTES), (5, abi.BYTES)]
    st, got = orc.rowcodec_decode_chunk(b, o, None, specs)
    assert st == 0
    assert got.rows() == [(b"\x00\xff", 7, b"abc", None), (b"x" * 300, -300, None, None), (b"tail", None, b"", None)]
    # an empty string is not NULL: it is a not-null id whose value has no bytes
    assert got.columns[2].values()[2] == b"" and not got.columns[2].IsNull(2)
    # the fixed-width decoder and the chunk decoder agree where both apply; a damaged row stops both at the same row
    cut = b.copy()
    cut[o[2]] = 127
    st, got = orc.rowcodec_decode_chunk(cut, o, None, specs)
    assert st == 1 and got.NumRows() == 2
