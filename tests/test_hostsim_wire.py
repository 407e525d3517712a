"""CPU walk-through of the chunk wire kernels (tsq_wire_dp.h through tests/hostsim) against the oracle's restatement of chunk.Codec /
chunk.Decoder (oracle/chunk_wire.cpp), and the oracle against the reference's own TestCodec (util/chunk/codec_test.go:29-71)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column, StrColumn

HERE = os.path.dirname(os.path.abspath(__file__))
WM_COPY, WM_HDR, WM_BITS, WM_OFFS = 0, 1, 2, 3


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "hostsim", "hostsim.so"))
    lib.sim_wire_walk.restype = None
    lib.sim_wire_walk.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]
    lib.sim_wire_move.restype = None
    lib.sim_wire_move.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    return lib


def _test_codec_chunk(n=10):
    # codec_test.go:38-46: column 0 all NULL, column 1 = i, columns 2 and 3 = "<i>.12345"
    return Chunk([Column(abi.I64, np.zeros(n, np.int64), np.zeros(n, bool)), Column(abi.I64, np.arange(n)), StrColumn([b"%d.12345" % i for i in range(n)]),
                  StrColumn([b"%d.12345" % i for i in range(n)])])


def test_oracle_round_trip_of_the_reference_test_codec():
    chk = _test_codec_chunk()
    buf = orc.WireChunk.from_chunk(chk).encode()
    # 8 + 2 + 80 | 8 + 80 (no bitmap: nullCount 0) | 8 + 11 * 8 + 70 | the same
    assert len(buf) == 90 + 88 + 166 + 166
    assert buf[:8] == bytes([10, 0, 0, 0, 10, 0, 0, 0]) and buf[8:10] == b"\0\0" and buf[90:98] == bytes([10, 0, 0, 0, 0, 0, 0, 0])
    new = orc.WireChunk([8, 8, -1, -1])
    assert new.decode_to_chunk(buf) == len(buf)  # len(remained) == 0
    for c in range(4):
        assert new.column(c)[0] == 10
    assert new.column(0)[1] == b"\0\0" and new.column(1)[1] == b"\xff\xff"  # IsNull(0) everywhere, setAllNotNull for the others
    assert np.frombuffer(new.column(1)[3], np.int64).tolist() == list(range(10))
    offs, data = new.column(2)[2], new.column(2)[3]
    assert [data[offs[i]:offs[i + 1]] for i in range(10)] == [b"%d.12345" % i for i in range(10)]
    assert new.column(3)[2:] == new.column(2)[2:]
    # a buffer cut anywhere is out of range
    for cut in (0, 7, 9, 89, 100, 200, len(buf) - 1):
        assert orc.WireChunk([8, 8, -1, -1]).decode_to_chunk(buf[:cut]) == -1


def _random_chunk(rng, n, null_p=0.2):
    def nn():
        return None if null_p == 0 else rng.random(n) >= null_p
    strs = [None if (null_p and rng.random() < null_p) else bytes(rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8)) for _ in range(n)]
    return Chunk([Column(abi.I64, rng.integers(-1 << 62, 1 << 62, n), nn()), Column(abi.F32, rng.standard_normal(n).astype(np.float32), nn()), StrColumn(strs),
                  Column(abi.F64, rng.standard_normal(n)), Column(abi.U64, rng.integers(0, 1 << 63, n).astype(np.uint64), nn())])


ELEM = [8, 4, -1, 8, 8]


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 64, 1000])
def test_header_walk_finds_the_pieces_where_the_oracle_decoder_finds_them(sim, n):
    rng = np.random.default_rng(n)
    chk = _random_chunk(rng, n)
    buf = orc.WireChunk.from_chunk(chk).encode()
    raw = np.frombuffer(buf + b"\0" * 8, np.uint8)
    elem = (C.c_int32 * 5)(*ELEM)
    for first, mx in [(0, 1 << 40), (8, 16), (n // 8 * 8, 5)]:
        out = np.zeros(20, np.uint64)
        sim.sim_wire_walk(raw.ctypes.data_as(C.c_void_p), len(buf), elem, 5, first, mx, out.ctypes.data_as(C.c_void_p))
        new = orc.WireChunk(ELEM)
        assert new.decode_to_chunk(buf) == len(buf)
        f = min(first, n)
        take = min(mx, n - f)
        for c in range(5):
            ln, bm, offs, data = new.column(c)
            assert int(out[4 * c]) & 0xffffffff == ln == n
            nulls = int(out[4 * c]) >> 32
            assert nulls == n - sum(bin(b).count("1") for b in bm) if nulls else bm == b"\xff" * ((n + 7) // 8)
            assert int(out[4 * c + 1]) == len(data)
            if offs is not None:
                assert (int(out[4 * c + 2]), int(out[4 * c + 3])) == (offs[f], offs[f + take])
    # cut buffers: the walk reports the column in which the buffer ends
    for cut in sorted({0, 5, 8, len(buf) // 2, len(buf) - 1}):
        if cut >= len(buf):
            continue
        out = np.zeros(20, np.uint64)
        sim.sim_wire_walk(raw.ctypes.data_as(C.c_void_p), cut, elem, 5, 0, 1 << 40, out.ctypes.data_as(C.c_void_p))
        assert orc.WireChunk(ELEM).decode_to_chunk(buf[:cut]) == -1 and int(out[16]) == (1 << 64) - 1


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 31, 100, 16384, 16385 + 16, 70001])
def test_copy_lanes_equal_memcpy_at_every_alignment(sim, n):
    rng = np.random.default_rng(n)
    src = rng.integers(0, 256, n + 64, dtype=np.uint8)
    for sa in (0, 1, 7, 8, 15):
        for da in (0, 3, 8, 13):
            dst = np.full(n + 64, 0xAB, np.uint8)
            base = dst.ctypes.data
            off = (-base) % 16 + da  # destination at alignment da
            sim.sim_wire_move(WM_COPY, src.ctypes.data + sa, base + off, n, 0)
            assert dst[off:off + n].tobytes() == src[sa:sa + n].tobytes()
            assert (dst[:off] == 0xAB).all() and (dst[off + n:] == 0xAB).all()  # nothing outside the piece is written


@pytest.mark.parametrize("dst_rows", [0, 1, 5, 8, 13, 64, 1001])
@pytest.mark.parametrize("take", [1, 7, 8, 9, 64, 1000, 40000])
def test_bitmap_append_equals_the_decoder(sim, dst_rows, take):
    # Decoder.decodeColumn (codec.go:325-343) on a destination that already holds dst_rows rows; source with and without a wire bitmap
    rng = np.random.default_rng(dst_rows * 100003 + take)
    for all_notnull in (False, True):
        src_nn = np.ones(take, bool) if all_notnull else rng.random(take) >= 0.3
        if not all_notnull:
            src_nn[0] = False
        dst_nn = rng.random(dst_rows) >= 0.3
        # the oracle: a destination chunk holding dst_rows rows, a decoder over a wire chunk of `take` rows (take rounded by the caller)
        dchunk = orc.WireChunk([8])
        if dst_rows:
            seed = orc.WireChunk.from_chunk(Chunk([Column(abi.I64, np.arange(dst_rows), dst_nn)]))
            inter0 = orc.WireChunk([8])
            inter0.decoder_reset(seed.encode())
            assert inter0.decoder_decode(dchunk, dst_rows + 8) == dst_rows
        inter = orc.WireChunk([8])
        inter.decoder_reset(orc.WireChunk.from_chunk(Chunk([Column(abi.I64, np.arange(take), None if all_notnull else src_nn)])).encode())
        assert inter.decoder_decode(dchunk, take) == take
        want = dchunk.column(0)[1]
        # the lanes: destination bitmap with its dst_rows bits, source bitmap bytes (or none)
        dst = np.full((dst_rows + take + 7) // 8 + 8, 0, np.uint8)
        dst[:(dst_rows + 7) // 8] = np.packbits(dst_nn, bitorder="little")
        guard = dst.copy()
        srcb = np.concatenate([np.packbits(src_nn, bitorder="little"), np.zeros(8, np.uint8)])
        sim.sim_wire_move(WM_BITS, None if all_notnull else srcb.ctypes.data, dst.ctypes.data, take, dst_rows)
        nb = (dst_rows + take + 7) // 8
        assert dst[:nb].tobytes() == want
        assert dst[nb:].tobytes() == guard[nb:].tobytes() and dst[:dst_rows // 8].tobytes() == guard[:dst_rows // 8].tobytes()


@pytest.mark.parametrize("n", [1, 2, 255, 2048, 2049, 10000])
def test_offset_rebase_and_header_lanes(sim, n):
    rng = np.random.default_rng(n)
    offs = np.cumsum(rng.integers(0, 9, n)).astype(np.int64)
    raw = np.concatenate([np.zeros(3, np.uint8), offs.view(np.uint8), np.zeros(8, np.uint8)])  # the offsets at byte position 3
    dst = np.full(n + 2, -7, np.int64)
    sim.sim_wire_move(WM_OFFS, raw.ctypes.data + 3, dst.ctypes.data + 8, n, 1234 - int(offs[0]))
    assert dst[0] == -7 and dst[n + 1] == -7 and (dst[1:n + 1] == offs + (1234 - offs[0])).all()
    hdr = np.full(24, 0xEE, np.uint8)
    sim.sim_wire_move(WM_HDR, None, hdr.ctypes.data + 5, 8, n | (77 << 32))
    assert np.frombuffer(hdr[5:13].tobytes(), np.uint32).tolist() == [n, 77]
    assert (hdr[:5] == 0xEE).all() and (hdr[13:] == 0xEE).all()


def test_decoder_windows_and_reuse_in_the_oracle():
    # Decoder.Decode takes multiples of 8 rows until the intermediate chunk is dry; ReuseIntermChk hands over the rest with offsets from 0
    rng = np.random.default_rng(5)
    chk = _random_chunk(rng, 100)
    buf = orc.WireChunk.from_chunk(chk).encode()
    inter = orc.WireChunk(ELEM)
    assert inter.decoder_reset(buf) == len(buf) and inter.decoder_remained() == 100
    dst = orc.WireChunk(ELEM)
    assert inter.decoder_decode(dst, 30) == 32 and inter.decoder_remained() == 68
    assert inter.decoder_decode(dst, 3) == 8 and dst.column(0)[0] == 40
    rest = orc.WireChunk(ELEM)
    inter.decoder_reuse(rest)
    assert inter.decoder_remained() == 0 and rest.column(2)[0] == 60 and rest.column(2)[2][0] == 0
    whole = orc.WireChunk(ELEM)
    whole.decode_to_chunk(buf)
    for c in range(5):
        ln, bm, offs, data = whole.column(c)
        a, b = dst.column(c), rest.column(c)
        assert a[3] + b[3] == data
        if offs is not None:
            assert a[2] == offs[:41] and b[2] == [o - offs[40] for o in offs[40:]]
