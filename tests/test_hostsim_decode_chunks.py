"""CPU: the code k_decc_count / k_decc_emit run (tinysql_amd/csrc/tsq_decode_dp.h: tsq_decc_value / tsq_decc_uvarint / tsq_decc_store, and
the 12-byte fetch out of aligned words) compiled with g++ through tests/hostsim and walked chunk by chunk like the kernels do —
against the oracle's restatement of readRowsData + DecodeOne with compact-bytes datums (codec.go:623-690, bytes.go:150-160), at every
alignment of the response pointer, including every error the reference reports and its position in the stream."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column, StrColumn

HERE = os.path.dirname(os.path.abspath(__file__))
GUARD = 64
STATUS = {0: "ok", 1: "row cut", 2: "insufficient", 3: "overflow", 4: "bad flag", 5: "bytesFlag", 6: "kind mismatch", 7: "bad marker", 8: "bad padding"}


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "hostsim", "hostsim.so"))
    P = C.c_void_p
    lib.sim_rows_decode_chunks.restype = C.c_uint64
    lib.sim_rows_decode_chunks.argtypes = [P, C.c_int64, C.c_int64, C.c_int64, P, C.c_int64, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    return lib


def run_sim(sim, data, chunk_offs, types, phase=0, cap_rows=None):
    raw = np.frombuffer(bytes(data), dtype=np.uint8)
    buf = np.full(GUARD + 8 + raw.size + GUARD, 0xA5, np.uint8)  # garbage around the response
    buf[GUARD + phase:GUARD + phase + raw.size] = raw
    offs = np.ascontiguousarray(chunk_offs, dtype=np.int64)
    cap = (raw.size // max(len(types), 1) + 1) if cap_rows is None else cap_rows
    bits = [np.zeros(cap + 1, np.uint64) for _ in types]
    lens = [np.zeros(cap + 1, np.int64) for _ in types]
    nns = [np.full(cap + 1, 7, np.uint8) for _ in types]
    pb = (C.c_void_p * len(types))(*[x.ctypes.data for x in bits])
    pl = (C.c_void_p * len(types))(*[x.ctypes.data for x in lens])
    pn = (C.c_void_p * len(types))(*[x.ctypes.data for x in nns])
    rows, bad = C.c_int64(0), C.c_int64(0)
    tp = (C.c_int32 * len(types))(*types)
    err = sim.sim_rows_decode_chunks(buf.ctypes.data_as(C.c_void_p), GUARD, phase, raw.size, offs.ctypes.data_as(C.c_void_p), len(offs) - 1, len(types), tp, pb, pl, pn,
                                     cap, C.byref(rows), C.byref(bad))
    assert bad.value == 0  # no aligned word without a byte of the response was read
    code = 0 if err == (1 << 64) - 1 else err & 15
    n = rows.value
    if n > cap:
        return code, n, None
    cols = []
    for c, t in enumerate(types):
        nn = nns[c][:n].astype(bool)
        assert (nns[c][n:] == 7).all()  # nothing past the rows handed over
        if t == abi.BYTES:
            def cell(r):  # K13f's copy: a grouped (memcomparable) cell skips one marker byte after every 8 data bytes
                pos, ln = int(bits[c][r]), int(lens[c][r])
                if pos >> 62 & 1:
                    pos &= (1 << 62) - 1
                    return bytes(raw[pos + i + i // 8] for i in range(ln))
                return bytes(raw[pos:pos + ln])
            cols.append(StrColumn([cell(r) if nn[r] else None for r in range(n)]))
        elif t == abi.F32:
            cols.append(Column(t, bits[c][:n].astype(np.uint32).view(np.float32).copy(), nn))
        elif t == abi.F64:
            cols.append(Column(t, bits[c][:n].view(np.float64).copy(), nn))
        else:
            cols.append(Column(t, bits[c][:n].view(np.int64).copy() if t == abi.I64 else bits[c][:n].copy(), nn))
    return code, n, Chunk(cols)


def table(rng, n, long_strings=False):
    iv = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64) >> rng.integers(0, 64, n)
    uv = (rng.integers(0, (1 << 64) - 1, n, dtype=np.uint64) >> rng.integers(0, 64, n).astype(np.uint64)).astype(np.uint64)
    words = [None if rng.random() < 0.15 else bytes(rng.integers(0, 256, int(rng.integers(0, 200 if long_strings else 20)), dtype=np.uint8)) for _ in range(n)]
    notes = [None if rng.random() < 0.1 else (b"" if rng.random() < 0.2 else b"n%d" % i) for i in range(n)]
    return Chunk([Column(abi.I64, iv, rng.random(n) >= 0.2), StrColumn(words), Column(abi.F64, rng.standard_normal(n), rng.random(n) >= 0.2), Column(abi.U64, uv, rng.random(n) >= 0.2),
                  StrColumn(notes), Column(abi.F32, rng.standard_normal(n).astype(np.float32), rng.random(n) >= 0.2)])


def response(t, rows_per_chunk=64):
    """fillUpData4SelectResponse by the oracle: the encoded rows + the chunk boundaries"""
    n = t.NumRows()
    parts = [bytes(orc.encode_rows(t.slice(lo, min(lo + rows_per_chunk, n)))) for lo in range(0, n, rows_per_chunk)]
    return b"".join(parts), np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)


@pytest.mark.parametrize("n,per", [(1, 64), (63, 64), (64, 64), (65, 64), (1000, 64), (1000, 7), (300, 1)])
@pytest.mark.parametrize("phase", [0, 1, 3, 7])
def test_walk_equals_the_oracle(sim, n, per, phase):
    rng = np.random.default_rng(n * 8 + phase + per)
    t = table(rng, n, long_strings=(n == 300))
    data, offs = response(t, per)
    st, want = orc.decode_rows_chunks(data, offs, t.types())
    code, rows, got = run_sim(sim, data, offs, t.types(), phase)
    assert st == 0 and code == 0 and rows == n == want.NumRows()
    assert got.rows() == want.rows() == t.rows()


def test_empty_chunks_and_an_output_that_is_too_small(sim):
    rng = np.random.default_rng(2)
    t = table(rng, 100)
    data, offs = response(t, 10)
    offs2 = np.concatenate([[0, 0], offs[1:5], offs[4:5], offs[5:]])  # empty chunks between the real ones
    code, rows, got = run_sim(sim, data, offs2, t.types())
    assert code == 0 and got.rows() == t.rows()
    code, rows, got = run_sim(sim, data, offs, t.types(), cap_rows=99)
    assert code == 0 and rows == 100 and got is None  # nothing written, the rows needed are reported


def damage(data, offs, at, new):
    b = bytearray(data)
    b[at:at + len(new)] = new
    return bytes(b), offs


@pytest.mark.parametrize("case", ["bad_flag", "bytes_flag", "cut_int", "cut_varint", "long_varint", "cut_string", "negative_length", "row_cut", "kind", "kind_reverse", "offsets"])
def test_first_error_in_stream_order(sim, case):
    rng = np.random.default_rng(11)
    types = [abi.I64, abi.BYTES, abi.F64]
    t = Chunk([Column(abi.I64, rng.integers(-5, 5, 200), None), StrColumn([b"s%03d" % i for i in range(200)]), Column(abi.F64, rng.random(200), None)])
    data, offs = response(t, 20)
    row_bytes = 2 + 6 + 9  # varint int (2), compact bytes (1 + 1 + 4), float (9)
    at = int(offs[3]) + 5 * row_bytes  # row 65 = chunk 3, row 5 inside it
    want_rows = 65
    if case == "bad_flag":
        data, offs = damage(data, offs, at, b"\x07")
    elif case == "bytes_flag":
        data, offs = damage(data, offs, at + 2, b"\x01")
    elif case == "cut_int":  # a comparable int datum needs 8 bytes: put one at the very end of a chunk
        data = data[:int(offs[4]) - 9] + b"\x03\x00\x00" + data[int(offs[4]):]
        offs = offs.copy()
        offs[4:] -= 6
        want_rows = 79
    elif case == "cut_varint":  # continuation bits up to the end of the chunk
        data = data[:int(offs[4]) - 9] + b"\x08\x80\x80" + data[int(offs[4]):]
        offs = offs.copy()
        offs[4:] -= 6
        want_rows = 79
    elif case == "long_varint":
        b = bytearray(data)
        b[at:at + 2] = b"\x08\x80"
        b[at + 2:at + 2] = b"\x80" * 10
        data = bytes(b)
        offs = offs.copy()
        offs[4:] += 10
    elif case == "cut_string":  # the length says 100 bytes, the chunk ends first
        data, offs = damage(data, offs, at + 3, b"\xc8\x01"[:1])
    elif case == "negative_length":
        data, offs = damage(data, offs, at + 3, b"\x01")
    elif case == "row_cut":  # the chunk ends after the second value of a row
        data = data[:int(offs[4]) - 9] + data[int(offs[4]):]
        offs = offs.copy()
        offs[4:] -= 9
        want_rows = 79
    elif case == "kind":  # an int datum where the string column's value belongs
        data, offs = damage(data, offs, at + 2, b"\x08\x02\x08\x02\x08\x02"[:6])
    elif case == "kind_reverse":
        types = [abi.I64, abi.I64, abi.F64]
        want_rows = 0
    else:  # chunk boundaries that run backwards: nothing of that chunk can be read
        offs = offs.copy()
        offs[4] = offs[3] - 1
        want_rows = 60
    st, want = orc.decode_rows_chunks(data, offs, types)
    for phase in (0, 5):
        code, rows, got = run_sim(sim, data, offs, types, phase)
        assert st != 0 and code == st, (STATUS[st], STATUS[code])
        assert rows == want.NumRows() == want_rows and got.rows() == want.rows()


# ---- memcomparable (bytesFlag) datums: the EncodeKey form of strings, what index keys hold (round 2)
BYTES_CODEC = [  # util/codec/bytes_test.go:33-47 (the ascending form)
    ([], [0, 0, 0, 0, 0, 0, 0, 0, 247]), ([0], [0, 0, 0, 0, 0, 0, 0, 0, 248]), ([1, 2, 3], [1, 2, 3, 0, 0, 0, 0, 0, 250]), ([1, 2, 3, 0], [1, 2, 3, 0, 0, 0, 0, 0, 251]),
    ([1, 2, 3, 4, 5, 6, 7], [1, 2, 3, 4, 5, 6, 7, 0, 254]), ([0] * 8, [0] * 8 + [255] + [0] * 8 + [247]), ([1, 2, 3, 4, 5, 6, 7, 8], [1, 2, 3, 4, 5, 6, 7, 8, 255, 0, 0, 0, 0, 0, 0, 0, 0, 247]),
    ([1, 2, 3, 4, 5, 6, 7, 8, 9], [1, 2, 3, 4, 5, 6, 7, 8, 255, 9, 0, 0, 0, 0, 0, 0, 0, 248])]
BYTES_CODEC_ERR = [  # bytes_test.go:68-78: DecodeBytes must fail
    [1, 2, 3, 4], [0, 0, 0, 0, 0, 0, 0, 247], [0, 0, 0, 0, 0, 0, 0, 0, 246], [0, 0, 0, 0, 0, 0, 0, 1, 247], [1, 2, 3, 4, 5, 6, 7, 8, 0], [1, 2, 3, 4, 5, 6, 7, 8, 255, 1],
    [1, 2, 3, 4, 5, 6, 7, 8, 255, 1, 2, 3, 4, 5, 6, 7, 8], [1, 2, 3, 4, 5, 6, 7, 8, 255, 1, 2, 3, 4, 5, 6, 7, 8, 255], [1, 2, 3, 4, 5, 6, 7, 8, 255, 1, 2, 3, 4, 5, 6, 7, 8, 0]]


def test_reference_bytes_codec_vectors(sim):
    for dec, enc in BYTES_CODEC:
        assert orc.encode_rows(Chunk([StrColumn([bytes(dec)])]), comparable=True).tolist() == [1] + enc  # EncodeBytes behind the bytesFlag
        data = bytes([1] + enc)
        st, want = orc.decode_rows_chunks(data, [0, len(data)], [abi.BYTES])
        code, rows, got = run_sim(sim, data, [0, len(data)], [abi.BYTES], phase=3)
        assert st == code == 0 and want.rows() == got.rows() == [(bytes(dec),)]
    for enc in BYTES_CODEC_ERR:
        data = bytes([1] + enc)
        st, want = orc.decode_rows_chunks(data, [0, len(data)], [abi.BYTES])
        code, rows, got = run_sim(sim, data, [0, len(data)], [abi.BYTES])
        assert st in (2, 7, 8) and code == st and rows == 0, (enc, STATUS[st], STATUS[code])


@pytest.mark.parametrize("n,per", [(1, 64), (65, 64), (1000, 7), (300, 1)])
@pytest.mark.parametrize("phase", [0, 5])
def test_comparable_responses_walk_like_the_oracle(sim, n, per, phase):
    # every value in its EncodeKey form: ints flag 3, uints flag 4, reals flag 5, strings flag 1 + groups (codec.go:74-109)
    rng = np.random.default_rng(n + per + phase)
    t = table(rng, n, long_strings=(n == 300))
    parts = [bytes(orc.encode_rows(t.slice(lo, min(lo + per, n)), comparable=True)) for lo in range(0, n, per)]
    data, offs = b"".join(parts), np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    st, want = orc.decode_rows_chunks(data, offs, t.types())
    code, rows, got = run_sim(sim, data, offs, t.types(), phase)
    assert st == 0 and code == 0 and rows == n
    assert got.rows() == want.rows() == t.rows()


# ---------------------------------------------------------------- round 5: tsq_rows_decode with a var-len column — the host walk that finds the row boundaries
def _walk(sim, data, n_cols, cap_rows, per):
    raw = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    offs = np.zeros(len(data) + 4, np.int64)
    n_offs, end, dmg = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    sim.sim_dec_walk_rows.restype = C.c_int64
    sim.sim_dec_walk_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    rows = sim.sim_dec_walk_rows(raw.ctypes.data, len(data), n_cols, cap_rows, per, offs.ctypes.data, offs.size, C.byref(n_offs), C.byref(end), C.byref(dmg))
    return rows, offs[:n_offs.value].copy(), end.value, bool(dmg.value)


@pytest.mark.parametrize("n,per", [(1, 64), (64, 64), (65, 64), (1000, 64), (333, 7)])
def test_row_walk_finds_the_boundaries_the_encoder_made(sim, n, per):
    rng = np.random.default_rng(900 + n + per)
    t = table(rng, n, long_strings=(n == 333))
    data, offs = response(t, per)  # the oracle's encoder, `per` rows per chunk: ITS boundaries are the truth
    rows, got, end, dmg = _walk(sim, data, len(t.types()), 1 << 40, per)
    assert rows == n and not dmg and end == len(data) and (got == offs).all()
    # cap_rows: the walk stops after that many rows, the consumed prefix ends there (select_result.go:153 keeps the remainder)
    cap = max(1, n // 3)
    rows, got, end, dmg = _walk(sim, data, len(t.types()), cap, per)
    want_end = len(bytes(orc.encode_rows(t.slice(0, cap))))
    assert rows == cap and not dmg and end == want_end and got[-1] == want_end and got[0] == 0
    # ... and the pieces decode (through the kernels' own walk, on the CPU) to the first `cap` rows
    code, nrows, dec = run_sim(sim, data[:end], got, t.types())
    assert code == 0 and nrows == cap and dec.rows() == t.slice(0, cap).rows()


def test_row_walk_hands_a_damaged_remainder_to_the_kernels(sim):
    rng = np.random.default_rng(77)
    t = table(rng, 200)
    data, offs = response(t, 64)
    bad = bytearray(data)
    at = int(offs[2]) + 0  # the first flag byte of row 128
    bad[at] = 0x7E         # not a flag DecodeOne knows (codec.go:683)
    rows, got, end, dmg = _walk(sim, bytes(bad), len(t.types()), 1 << 40, 64)
    assert dmg and rows == 128 and end == len(bad) and got[-1] == len(bad) and got[-2] == int(offs[2])
    code, nrows, dec = run_sim(sim, bytes(bad), got, t.types())
    assert STATUS[code] == "bad flag" and nrows == 128 and dec.rows() == t.slice(0, 128).rows()
    # a stream that ends inside a row: the rows before it, then "row cut" / "insufficient" from the kernels' walk
    cut = data[:int(offs[1]) + 5]
    rows, got, end, dmg = _walk(sim, cut, len(t.types()), 1 << 40, 64)
    assert dmg and rows >= 64 and got[-1] == len(cut)
    # an empty stream
    assert _walk(sim, b"", 3, 10, 64)[:1] == (0,)
