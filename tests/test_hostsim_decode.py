"""CPU: the scalar core of tsq_rows_decode (tinysql_amd/csrc/tsq_decode_dp.h: the backward pass that yields a sub-block's exit
map / counts, and the word-based value decode) compiled with g++ through tests/hostsim and driven over whole byte streams
like the kernels do — against a brute-force walk of every entry offset and against the oracle's sequential decoder.  This
is the part of the GPU algorithm where an off-by-one would silently mis-parse, checked here without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "hostsim", "hostsim.so"))
    P = C.c_void_p
    lib.sim_dec_maps.restype = None
    lib.sim_dec_maps.argtypes = [P, C.c_int64, P, P]
    lib.sim_dec_value.restype = C.c_int32
    lib.sim_dec_value.argtypes = [P, C.c_int64, C.c_int64, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    return lib


def _len_at(b, p):
    """length rule of the format (flag byte + payload), as the reference's decoder consumes bytes"""
    f = b[p] if p < len(b) else 0
    if f in (3, 4, 5):
        return 9
    if f in (8, 9):
        k = 1
        while k < 10 and p + k < len(b) and b[p + k] & 0x80:
            k += 1
        return k + 1
    return 1


def _maps(sim, raw):
    nsb = (raw.size + 31) // 32
    maps = np.zeros(nsb, np.uint64)
    cnts = np.zeros(nsb * 3, np.uint32)
    sim.sim_dec_maps(raw.ctypes.data_as(C.c_void_p), raw.size, maps.ctypes.data_as(C.c_void_p), cnts.ctypes.data_as(C.c_void_p))
    return maps, cnts.reshape(nsb, 3)


def _streams():
    rng = np.random.default_rng(12)
    n = 3000
    wide = (rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64) >> rng.integers(0, 63, n)).astype(np.int64)
    chk = Chunk([Column(abi.I64, wide, rng.random(n) > 0.2), Column(abi.F64, np.ldexp(rng.random(n) - 0.5, rng.integers(-40, 40, n))),
                 Column(abi.U64, (rng.integers(0, (1 << 64) - 1, n, dtype=np.uint64) >> rng.integers(0, 64, n).astype(np.uint64)).astype(np.uint64))])
    yield "EncodeValue", orc.encode_rows(chk), chk
    yield "EncodeKey", orc.encode_rows(chk, True), chk
    alphabet = np.array([0x00, 0x03, 0x05, 0x08, 0x09, 0x80, 0xFF, 0x01], dtype=np.uint8)   # payloads that look like flags
    u = alphabet[rng.integers(0, 8, (n, 8))].copy().view(np.uint64).reshape(n)
    tricky = Chunk([Column(abi.U64, u), Column(abi.I64, u.view(np.int64), rng.random(n) > 0.3)])
    yield "flag-like payloads", orc.encode_rows(tricky, True), tricky
    small = Chunk([Column(abi.I64, rng.integers(-60, 60, 5000), rng.random(5000) > 0.5)])      # 1-2 byte values: up to 32 per sub-block
    yield "tiny values", orc.encode_rows(small), small


def test_subblock_maps_equal_a_brute_force_walk_of_every_entry_offset(sim):
    for name, raw, _ in _streams():
        b = raw.tolist()
        maps, cnts = _maps(sim, raw)
        for sb in range(len(maps)):
            lo = sb * 32
            lim = min(32, len(b) - lo)
            for e in range(11):
                pos, c = e, 0
                while pos < lim:
                    pos += _len_at(b, lo + pos)
                    c += 1
                want_exit = pos - 32 if pos >= 32 else 0
                got_exit = (int(maps[sb]) >> (4 * e)) & 15
                got_cnt = (int(cnts[sb][e >> 2]) >> (8 * (e & 3))) & 255
                assert (got_exit, got_cnt) == (want_exit, c), (name, sb, e)


def test_composed_maps_and_value_decode_equal_the_sequential_decoder(sim):
    for name, raw, chk in _streams():
        b = raw.tolist()
        maps, cnts = _maps(sim, raw)
        types = chk.types()
        st, want, _ = orc.decode_rows(raw, types, chk.NumRows())
        assert st == 0
        # follow entry offset 0 through the sub-blocks (what K13a/K13b/K13c compose), decoding every value on the path
        state, vals = 0, []
        for sb in range(len(maps)):
            lo = sb * 32
            lim = min(32, len(b) - lo)
            pos, c = state, 0
            while pos < lim:
                ln = _len_at(b, lo + pos)
                bits, isnull, real = C.c_uint64(0), C.c_uint8(0), C.c_uint8(0)
                err = sim.sim_dec_value(raw.ctypes.data_as(C.c_void_p), raw.size, lo + pos, ln, C.byref(bits), C.byref(isnull), C.byref(real))
                assert err == 0, (name, lo + pos)
                vals.append(None if isnull.value else bits.value)
                pos += ln
                c += 1
            assert c == (int(cnts[sb][state >> 2]) >> (8 * (state & 3))) & 255
            state = (int(maps[sb]) >> (4 * state)) & 15
        ncols = len(types)
        assert len(vals) == chk.NumRows() * ncols
        for col in range(ncols):
            wc = want.columns[col]
            nn = np.ones(len(wc), bool) if wc.notnull is None else wc.notnull
            got = vals[col::ncols]
            assert [v is not None for v in got] == nn.tolist()
            assert [v for v in got if v is not None] == wc.data.view(np.uint64)[nn].tolist()


def test_value_decode_errors(sim):
    def dec(raw, ln):
        a = np.frombuffer(raw, np.uint8)
        bits, isnull, real = C.c_uint64(0), C.c_uint8(0), C.c_uint8(0)
        return sim.sim_dec_value(a.ctypes.data_as(C.c_void_p), a.size, 0, ln, C.byref(bits), C.byref(isnull), C.byref(real)), bits.value
    assert dec(b"\x08" + b"\xff" * 9 + b"\x01", 11) == (0, (1 << 64) - 1 >> 1 ^ ((1 << 64) - 1))          # MinInt64: ff*9 01
    assert dec(b"\x08" + b"\xff" * 9 + b"\x02", 11)[0] == 3                                                  # 10th byte > 1
    assert dec(b"\x08" + b"\xff" * 10 + b"\x01", 11)[0] == 3                                                 # continuation in the 10th byte
    assert dec(b"\x07\x00", 1)[0] == 4 and dec(b"\x02\x02ab", 1)[0] == 5
    assert dec(b"\x03\x80\x00", 9)[0] == 2                                                                   # cut by the end of the buffer


@pytest.mark.parametrize("tp", [abi.I64, abi.U64, abi.F64, abi.F32])
@pytest.mark.parametrize("desc", [False, True])
def test_sort_key_images_order_like_the_reference_comparators(sim, tp, desc):
    # tsq_sort_image.h: image(a) < image(b) <=> cmp(a, b) < 0 and equal images <=> cmp == 0, for the oracle's restatement of
    # util/chunk/compare.go — checked on all pairs of a value set with the extremes, +-0, +-inf and denormals
    sim.sim_sort_images.restype = None
    sim.sim_sort_images.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    rng = np.random.default_rng(tp)
    if tp == abi.I64:
        v = np.concatenate([np.array([-(1 << 63), (1 << 63) - 1, 0, -1, 1], dtype=np.int64), rng.integers(-(1 << 63), (1 << 63) - 1, 60, dtype=np.int64), rng.integers(-5, 5, 20)])
    elif tp == abi.U64:
        v = np.concatenate([np.array([0, 1, (1 << 63) - 1, 1 << 63, (1 << 64) - 1], dtype=np.uint64), rng.integers(0, (1 << 64) - 1, 60, dtype=np.uint64)])
    else:
        big = 3.4028234663852886e38 if tp == abi.F32 else 1.7976931348623157e308
        v = np.concatenate([[0.0, -0.0, np.inf, -np.inf, 5e-324, -5e-324, 1.0, -1.0, big, -big], np.ldexp(rng.random(60) - 0.5, rng.integers(-100, 100, 60)),
                            rng.integers(-3, 3, 20) / 2.0])
        v = v.astype(np.float32) if tp == abi.F32 else v.astype(np.float64)
    v = np.ascontiguousarray(v)
    img = np.zeros(len(v), np.uint64)
    sim.sim_sort_images(v.ctypes.data_as(C.c_void_p), tp, 1 if desc else 0, len(v), img.ctypes.data_as(C.c_void_p))
    chk = Chunk([Column(tp, v)])
    for i in range(len(v)):
        for j in range(len(v)):
            c = orc.row_compare(chk, [0], [desc], i, j)
            got = -1 if img[i] < img[j] else (1 if img[i] > img[j] else 0)
            assert got == c, (v[i], v[j], desc)


def _string_key_cases():
    rng = np.random.default_rng(77)
    alphabet = [b"", b"\x00", b"\x00\x00", b"a", b"a\x00", b"ab", b"abc", b"abcdefgh", b"abcdefgh\x00", b"abcdefghi", b"abcdefgi", b"\xff", b"\xff\xff" * 5,
                b"abcdefghijklmnop", b"abcdefghijklmnopq", b"abcdefghijklmnoq", b"b", b"B", b"\x80", b"\x7f"]
    vals = list(alphabet) + [bytes(rng.integers(0, 256, int(rng.integers(0, 21)), dtype=np.uint8)) for _ in range(120)]
    vals += [bytes(rng.integers(97, 100, int(rng.integers(0, 5)), dtype=np.uint8)) for _ in range(160)]  # many ties and shared prefixes
    order = rng.permutation(len(vals))
    return [vals[i] for i in order]


@pytest.mark.parametrize("desc", [False, True])
@pytest.mark.parametrize("with_nulls", [False, True])
def test_string_sort_key_walk_orders_like_the_reference_comparator(sim, desc, with_nulls):
    # tsq_sort_image_str + the sub-key sequence of tsq_sort_finish (length, then chunks last to first, then the NULL pass), each a
    # stable sort, against the oracle's stable sort with cmpString (compare.go:71-77): the permutations are the same one
    from tinysql_amd.chunk import StrColumn
    sim.sim_sort_str_key.restype = None
    sim.sim_sort_str_key.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
    vals = _string_key_cases()
    if with_nulls:
        vals = [None if i % 7 == 3 else v for i, v in enumerate(vals)]
    col = StrColumn(vals)
    n = len(vals)
    perm = np.arange(n, dtype=np.int64)
    bm = col.bitmap()
    sim.sim_sort_str_key(col.data.ctypes.data_as(C.c_void_p), col.offsets.ctypes.data_as(C.c_void_p), None if bm is None else bm.ctypes.data_as(C.c_void_p),
                         1 if desc else 0, n, perm.ctypes.data_as(C.c_void_p))
    want = orc.sort_perm(Chunk([col]), [0], [desc])
    assert perm.tolist() == want.tolist()
    # and the oracle's order is Go's string order: bytes compare, NULL before everything (ascending)
    ref = sorted(range(n), key=lambda i: (vals[i] is not None, vals[i] or b""), reverse=False)
    if not desc:
        assert want.tolist() == ref
    else:
        keys = [(vals[i] is not None, vals[i] or b"") for i in want]
        assert all(keys[i] >= keys[i + 1] for i in range(n - 1))
