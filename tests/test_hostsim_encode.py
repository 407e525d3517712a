"""CPU: the code k_enc_size / k_enc_scan / k_enc_emit run (tinysql_amd/csrc/tsq_encode_dp.h: value lengths, datum bytes, the plan
that copies a tile's LDS image to an arbitrarily aligned place in the output) compiled with g++ through tests/hostsim and walked
workgroup by workgroup, tile by tile like the kernels do — against the oracle's restatement of codec.EncodeValue / EncodeKey
(pinned on codec_test.go) byte for byte, at every alignment of the output and of every tile boundary inside it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column, make_cols

HERE = os.path.dirname(os.path.abspath(__file__))
GUARD = 64


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "hostsim", "hostsim.so"))
    P = C.c_void_p
    lib.sim_rows_encode.restype = C.c_int64
    lib.sim_rows_encode.argtypes = [C.POINTER(abi.Col), C.c_int32, C.c_uint32, C.c_int64, C.c_int32, P, C.c_int64, C.c_uint32, P]
    return lib


def run_sim(sim, chunk, comparable_mask=0, n_wg=1024, phase=0, cap=None):
    keep = []
    cols = make_cols(chunk.columns, keep)
    n = chunk.NumRows()
    room = n * len(chunk.columns) * 11 + 16 if cap is None else cap
    buf = np.full(room + 2 * GUARD, 0xEE, np.uint8)
    offs = np.full(n + 2, -7, np.int64)
    total = sim.sim_rows_encode(cols, len(chunk.columns), comparable_mask, n, n_wg, buf[GUARD:].ctypes.data_as(C.c_void_p), room, phase,
                                offs.ctypes.data_as(C.c_void_p))
    assert total >= 0, total
    if total <= room:
        assert (buf[:GUARD] == 0xEE).all() and (buf[GUARD + total:] == 0xEE).all()  # nothing outside [0, total) is touched
        assert offs[n + 1] == -7
    else:
        assert (buf == 0xEE).all() and (offs == -7).all()  # too small: the size is known before a byte is written
    return total, buf[GUARD:GUARD + min(total, room)].copy(), offs[:n + 1].copy()


def rand_chunk(rng, n, null_p=0.2):
    iv = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64) >> rng.integers(0, 64, n)
    iv[:min(n, 8)] = np.array([0, -1, 1, 63, -64, 64, (1 << 63) - 1, -(1 << 63)])[:min(n, 8)]
    uv = (rng.integers(0, (1 << 64) - 1, n, dtype=np.uint64) >> rng.integers(0, 64, n).astype(np.uint64)).astype(np.uint64)
    uv[:min(n, 3)] = np.array([0, 127, (1 << 64) - 1], dtype=np.uint64)[:min(n, 3)]
    fv = np.ldexp(rng.random(n) - 0.5, rng.integers(-60, 60, n))
    fv[:min(n, 6)] = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan])[:min(n, 6)]
    f32 = (rng.random(n) * 100 - 50).astype(np.float32)
    nn = (lambda: rng.random(n) >= null_p) if null_p else (lambda: None)
    return Chunk([Column(abi.I64, iv, nn()), Column(abi.U64, uv, nn()), Column(abi.F64, fv, nn()), Column(abi.F32, f32, nn())])


@pytest.mark.parametrize("comparable", [False, True])
@pytest.mark.parametrize("n", [1, 7, 255, 256, 257, 3000])
def test_bytes_equal_the_oracle(sim, n, comparable):
    rng = np.random.default_rng(n)
    chk = rand_chunk(rng, n)
    want = orc.encode_rows(chk, comparable)
    total, got, offs = run_sim(sim, chk, 0b1111 if comparable else 0)
    assert total == want.size and (got == want).all()
    assert offs[0] == 0 and offs[n] == total and (np.diff(offs) > 0).all()
    for r in sorted(set([0, n // 2, n - 1])):  # the boundaries really delimit row r
        st, row, used = orc.decode_rows(got[offs[r]:offs[r + 1]], chk.types(), 1)
        one = Chunk([Column(c.tp, c.data[r:r + 1], None if c.notnull is None else c.notnull[r:r + 1]) for c in chk.columns])
        assert st == 0 and used == offs[r + 1] - offs[r] and bytes(orc.encode_rows(row, False)) == bytes(orc.encode_rows(one, False))


@pytest.mark.parametrize("phase", range(16))
def test_every_alignment_of_the_output(sim, phase):
    rng = np.random.default_rng(40 + phase)
    chk = rand_chunk(rng, 1500)
    want = orc.encode_rows(chk, False)
    for n_wg in (1, 3, 1024):  # tile boundaries fall on every byte phase anyway; different ownership changes which workgroup carries the base
        total, got, _ = run_sim(sim, chk, 0, n_wg=n_wg, phase=phase)
        assert total == want.size and (got == want).all()


def test_tiny_tiles_and_null_only_rows(sim):
    # rows of 1 byte per column (all NULL): tiles of 256..1024 bytes; single column, single row: tiles below one vector
    n = 700
    chk = Chunk([Column(abi.I64, np.zeros(n, np.int64), np.zeros(n, bool)) for _ in range(3)])
    for phase in (0, 5, 15):
        total, got, offs = run_sim(sim, chk, 0, phase=phase)
        assert total == 3 * n and not got.any() and (np.diff(offs) == 3).all()
    one = Chunk([Column(abi.I64, np.array([5]))])
    for phase in range(16):
        total, got, offs = run_sim(sim, one, 0, phase=phase)
        assert bytes(got) == b"\x08\x0a" and offs.tolist() == [0, 2]


def test_mixed_forms_round_trip(sim):
    # the handle column in the EncodeKey form next to varint columns (rowcodec BytesDecoder.DecodeToBytes): DecodeOne reads both
    rng = np.random.default_rng(8)
    n = 2000
    chk = rand_chunk(rng, n, null_p=0.0)
    total, got, offs = run_sim(sim, chk, 0b0001)
    st, dec, used = orc.decode_rows(got, chk.types(), n)
    assert st == 0 and used == total
    assert bytes(orc.encode_rows(dec, False)) == bytes(orc.encode_rows(chk, False))
    assert (got[offs[:-1]] == 3).all() and ((offs[1:] - offs[:-1]) >= 9 + 2 + 9 + 9).all()  # intFlag + 8 bytes first in every row


def test_output_too_small_writes_nothing(sim):
    rng = np.random.default_rng(2)
    chk = rand_chunk(rng, 500)
    want = orc.encode_rows(chk, False)
    total, got, _ = run_sim(sim, chk, 0, cap=want.size - 1)
    assert total == want.size  # the size is reported; run_sim saw that no byte of the buffer was touched


def string_chunk(rng, n, long_every=0):
    from tinysql_amd.chunk import StrColumn
    words = [None if rng.random() < 0.15 else bytes(rng.integers(0, 256, int(rng.integers(0, 30)), dtype=np.uint8)) for _ in range(n)]
    notes = [(b"L" * 70_000 if long_every and i % long_every == 3 else (None if rng.random() < 0.1 else (b"" if rng.random() < 0.2 else b"note-%d" % i))) for i in range(n)]
    iv = rng.integers(-(1 << 40), 1 << 40, n)
    return Chunk([StrColumn(words), Column(abi.I64, iv, rng.random(n) >= 0.2), StrColumn(notes), Column(abi.F64, rng.standard_normal(n), None)])


@pytest.mark.parametrize("n,long_every", [(1, 0), (255, 0), (256, 0), (257, 0), (3000, 0), (700, 50)])
@pytest.mark.parametrize("phase", [0, 5, 15])
def test_string_columns_as_compact_bytes(sim, n, long_every, phase):
    # a var-len cell = compactBytesFlag + varint(len) + the bytes (codec.go:101-109, bytes.go:141-148); a tile whose rows do not fit
    # the LDS image (70 000-byte cells) is written directly — same bytes either way
    rng = np.random.default_rng(n + phase)
    chk = string_chunk(rng, n, long_every)
    want = orc.encode_rows(chk)
    cap = len(want) + 64
    total, got, offs = run_sim(sim, chk, 0, n_wg=7, phase=phase, cap=cap)
    assert total == len(want) and bytes(got) == bytes(want)
    # the row boundaries cut the byte string into the rows the oracle encodes one by one
    for r in (0, n // 2, n - 1):
        assert bytes(got[offs[r]:offs[r + 1]]) == bytes(orc.encode_rows(chk.slice(r, r + 1)))


@pytest.mark.parametrize("n,long_every", [(1, 0), (257, 0), (3000, 0), (700, 50)])
@pytest.mark.parametrize("phase", [0, 9])
def test_string_columns_in_the_memcomparable_form(sim, n, long_every, phase):
    # EncodeKey of a var-len cell = bytesFlag + groups of 8 bytes, each followed by 0xFF - its pad count; a cell whose length is a
    # multiple of 8 (the empty one too) ends with an all-pad group and 0xF7 (codec.go:86-91, bytes.go:35-67)
    rng = np.random.default_rng(100 + n + phase)
    chk = string_chunk(rng, n, long_every)
    want = orc.encode_rows(chk, comparable=True)
    total, got, offs = run_sim(sim, chk, 0xF, n_wg=5, phase=phase, cap=len(want) + 64)
    assert total == len(want) and bytes(got) == bytes(want)
    for r in (0, n // 2, n - 1):
        assert bytes(got[offs[r]:offs[r + 1]]) == bytes(orc.encode_rows(chk.slice(r, r + 1), comparable=True))


def test_memcomparable_known_answers(sim):
    # the examples of bytes.go:38-42
    from tinysql_amd.chunk import StrColumn
    cells = [b"", bytes([1, 2, 3]), bytes([1, 2, 3, 0]), bytes(range(1, 9))]
    want = [bytes([0] * 8 + [247]), bytes([1, 2, 3, 0, 0, 0, 0, 0, 250]), bytes([1, 2, 3, 0, 0, 0, 0, 0, 251]),
            bytes(list(range(1, 9)) + [255] + [0] * 8 + [247])]
    chk = Chunk([StrColumn(cells)])
    total, got, offs = run_sim(sim, chk, 1, cap=256)
    for r, w in enumerate(want):
        assert bytes(got[offs[r]:offs[r + 1]]) == b"\x01" + w
