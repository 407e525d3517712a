"""CPU: the KEY RECORD of a row (tinysql_amd/csrc/tsq_keyrec_dp.h — what k_kr_hist / k_kr_scatter / k_kr_probe / k_kd_assign partition, index and
compare) compiled with g++ through tests/hostsim and checked against the reference's notion of key equality:
  join keys    codec.EqualChunkRow (util/codec/codec.go:363-382): same flag, same bytes, cell by cell; a NULL cell drops the row
               (hash_table.go:161-163, join.go:344); an UNSIGNED cell >= 2^63 never equals a negative BIGINT (codec.go:219-224)
  group keys   the encoded group key (codec.go:700-760): NULL is a key of its own (NilFlag), '' is another
Two rows must have equal records exactly when their key cells are equal in that sense, whatever the columns' mix of integers and strings."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column, StrColumn, make_cols

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "hostsim", "hostsim.so"))
    P = C.c_void_p
    lib.sim_kr_records.restype = None
    lib.sim_kr_records.argtypes = [C.POINTER(abi.Col), C.c_int32, P, C.c_int32, C.c_int32, P, C.c_int64, P, P, P, C.c_int32]
    lib.sim_kr_parse.restype = None
    lib.sim_kr_parse.argtypes = [P, C.c_int32, P, P, P, P, P]
    return lib


def records(sim, chk, key_cols, keep_nulls, selected=None, generic=False):
    keep = []
    cols = make_cols(chk.columns, keep)
    n = chk.NumRows()
    kc = np.array(key_cols, np.int32)
    rec = np.zeros((n, 4), np.uint64)
    st = np.zeros(n, np.uint8)
    h = np.zeros(n, np.uint64)
    sel = None if selected is None else np.ascontiguousarray(selected, np.uint8)
    sim.sim_kr_records(cols, len(chk.columns), kc.ctypes.data, len(key_cols), 1 if keep_nulls else 0, None if sel is None else sel.ctypes.data, n,
                       rec.ctypes.data, st.ctypes.data, h.ctypes.data, 1 if generic else 0)
    return rec, st, h


def cell_class(tp, v):
    """(flag, payload) of a key cell as the reference compares it: integers by (sign class, 64-bit word), strings by their bytes"""
    if v is None:
        return (0, b"")
    if tp == abi.BYTES:
        return (2, bytes(v))
    v = int(v)
    if tp == abi.U64 and v >= 1 << 63:
        return (9, v)
    return (8, v & ((1 << 64) - 1))


def test_known_answer_bytes(sim):
    # (bigint 5, 'ab'): [08][05 00 00 00 00 00 00 00][02][02]['a' 'b'] and zero padding
    chk = Chunk([Column(abi.I64, np.array([5, -1])), StrColumn([b"ab", b""])])
    rec, st, _ = records(sim, chk, [0, 1], True)
    assert st.tolist() == [0, 0]
    b = rec.view(np.uint8).reshape(2, 32)
    assert bytes(b[0]) == bytes([8, 5, 0, 0, 0, 0, 0, 0, 0, 2, 2, 97, 98]) + bytes(19)
    assert bytes(b[1]) == bytes([8] + [255] * 8 + [2, 0]) + bytes(21)


@pytest.mark.parametrize("shape", [[abi.BYTES], [abi.I64, abi.BYTES], [abi.I64, abi.U64, abi.BYTES], [abi.BYTES, abi.U64, abi.BYTES], [abi.I64], [abi.I64, abi.U64],
                                   [abi.I64, abi.U64, abi.I64], [abi.BYTES, abi.I64]])
@pytest.mark.parametrize("keep_nulls", [False, True])
def test_equal_records_iff_equal_keys(sim, shape, keep_nulls):
    rng = np.random.default_rng(len(shape) * 10 + keep_nulls)
    n = 4000
    cols, pyvals = [], []
    for tp in shape:
        if tp == abi.BYTES:
            pool = [b"", b"a", b"a\x00", b"ab", b"b", bytes(range(1, 9)), b"xyz" * 3]
            vals = [None if rng.random() < 0.1 else pool[int(i)] for i in rng.integers(0, len(pool), n)]
            cols.append(StrColumn(vals))
        else:
            pool = np.array([0, 1, 2, (1 << 62), (1 << 63) - 1], dtype=np.uint64)
            if tp == abi.I64:
                data = pool[rng.integers(0, len(pool), n)].astype(np.int64) * rng.choice([1, -1], n)
            else:
                data = np.where(rng.random(n) < 0.3, np.uint64(1 << 63) + pool[rng.integers(0, 3, n)], pool[rng.integers(0, len(pool), n)]).astype(np.uint64)
            nn = rng.random(n) > 0.1
            cols.append(Column(tp, data, nn))
            vals = [None if not nn[i] else int(data[i]) for i in range(n)]
        pyvals.append(vals)
    chk = Chunk(cols)
    rec, st, h = records(sim, chk, list(range(len(shape))), keep_nulls)
    # the builders with compile-time cell positions (keys of 8-byte cells, optionally ending with a string) agree with the general one
    rec_g, st_g, h_g = records(sim, chk, list(range(len(shape))), keep_nulls, generic=True)
    assert (rec == rec_g).all() and (st == st_g).all() and (h == h_g).all()
    keys = [tuple(cell_class(shape[c], pyvals[c][r]) for c in range(len(shape))) for r in range(n)]
    by_rec = {}
    for r in range(n):
        has_null = any(k[0] == 0 for k in keys[r])
        if has_null and not keep_nulls:
            assert st[r] == 1  # a join drops the row
            continue
        assert st[r] == 0
        by_rec.setdefault(rec[r].tobytes(), set()).add(keys[r])
    assert all(len(v) == 1 for v in by_rec.values())          # equal records -> equal keys
    assert len(by_rec) == len(set().union(*by_rec.values()))  # equal keys -> equal records
    # the mix is a function of the record
    seen = {}
    for r in range(n):
        if st[r] == 0:
            assert seen.setdefault(rec[r].tobytes(), int(h[r])) == int(h[r])


def test_signedness_rule_and_selection(sim):
    # BIGINT 3 = BIGINT UNSIGNED 3; BIGINT -1 != BIGINT UNSIGNED 2^64 - 1 (codec.go:219-224); selected == 0: the row has no key
    a = Chunk([Column(abi.I64, np.array([3, -1, 7], np.int64))])
    b = Chunk([Column(abi.U64, np.array([3, (1 << 64) - 1, 7], np.uint64))])
    ra, sa, _ = records(sim, a, [0], False)
    rb, sb, _ = records(sim, b, [0], False, selected=np.array([1, 1, 0]))
    assert (ra[0] == rb[0]).all() and not (ra[1] == rb[1]).all()
    assert sa.tolist() == [0, 0, 0] and sb.tolist() == [0, 0, 1]


def test_cells_that_do_not_fit(sim):
    # one string key: up to 30 bytes; (bigint, string): up to 21; four integers: 36 bytes never fit
    s = Chunk([StrColumn([b"x" * 30, b"x" * 31, b"y" * 300])])
    assert records(sim, s, [0], True)[1].tolist() == [0, 2, 2]
    m = Chunk([Column(abi.I64, np.arange(3)), StrColumn([b"x" * 21, b"x" * 22, None])])
    assert records(sim, m, [0, 1], True)[1].tolist() == [0, 2, 0]
    q = Chunk([Column(abi.I64, np.arange(2)) for _ in range(4)])
    assert records(sim, q, [0, 1, 2, 3], True)[1].tolist() == [2, 2]


def test_the_numpy_restatement_used_by_the_gpu_regression_test(sim):
    # tests/test_agg_keydict_gpu.py searches keys whose 18-bit index tag is all ones with a numpy restatement of the record + mix: same tags here
    from .test_agg_keydict_gpu import _record_tags
    keys = [b"t%d" % i for i in range(20000)] + [b"", b"a" * 30]
    chk = Chunk([StrColumn(keys)])
    _, st, h = records(sim, chk, [0], True)
    assert (st == 0).all()
    assert (((h >> np.uint64(14)) & np.uint64(0x3ffff)) == _record_tags(keys)).all()


def test_null_cells_keep_the_room_of_a_value(sim):
    # GROUP BY (bigint, string): a NULL bigint is nine zero bytes, a NULL string two — the cells after them start where they always do
    chk = Chunk([Column(abi.I64, np.array([7, 7]), np.array([False, True])), StrColumn([b"ab", None])])
    rec, st, _ = records(sim, chk, [0, 1], True)
    b = rec.view(np.uint8).reshape(2, 32)
    assert st.tolist() == [0, 0]
    assert bytes(b[0]) == bytes(9) + bytes([2, 2, 97, 98]) + bytes(19)
    assert bytes(b[1]) == bytes([8, 7, 0, 0, 0, 0, 0, 0, 0]) + bytes(23)


def test_the_mix_spreads_sequential_keys(sim):
    # partition = top bits, LDS slot = low 14 bits, tag = bits 14..31 of the record's mix: keys that differ in a few low bits (order numbers,
    # "k123" strings) must spread over all three
    n = 200_000
    for chk, keys in ((Chunk([Column(abi.I64, np.arange(n)), StrColumn([b"s%d" % (i % 977) for i in range(n)])]), [0, 1]),
                      (Chunk([StrColumn([b"key-%07d" % i for i in range(n)])]), [0]),
                      (Chunk([Column(abi.I64, np.arange(n) * 4096), Column(abi.I64, np.arange(n) % 3)]), [0, 1])):
        _, st, h = records(sim, chk, keys, True)
        assert (st == 0).all() and len(np.unique(h)) == n
        top = np.bincount((h >> np.uint64(53)).astype(np.int64), minlength=2048)    # 2048 partitions: ~98 keys each
        low = np.bincount((h & np.uint64(0x3fff)).astype(np.int64), minlength=16384)  # ~12 per slot
        assert top.min() > 50 and top.max() < 160, (top.min(), top.max())
        assert low.max() < 40
        tags = (h >> np.uint64(14)) & np.uint64(0x3ffff)
        assert len(np.unique(tags)) > 0.6 * min(n, 1 << 18) * (1 - np.exp(-n / (1 << 18))) / (n / (1 << 18)) * (n / (1 << 18)) * 0.9


@pytest.mark.parametrize("shape", [[abi.BYTES], [abi.I64, abi.BYTES], [abi.BYTES, abi.U64, abi.BYTES], [abi.I64, abi.U64, abi.I64], [abi.BYTES, abi.I64]])
def test_records_read_back_to_their_cells(sim, shape):
    # the aggregate's output key columns come back from the dictionary records (k_kd_decode -> kr_parse_cell): every cell of a GROUP BY key,
    # NULL cells included, is what went in
    rng = np.random.default_rng(len(shape) + 40)
    n = 600
    cols, pyvals = [], []
    for tp in shape:
        if tp == abi.BYTES:
            vals = [None if rng.random() < 0.15 else bytes(rng.integers(0, 256, int(rng.integers(0, 6)), dtype=np.uint8)) for _ in range(n)]
            cols.append(StrColumn(vals))
        else:
            data = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64).astype(np.int64 if tp == abi.I64 else np.uint64)
            nn = rng.random(n) > 0.15
            cols.append(Column(tp, data, nn))
            vals = [None if not nn[i] else int(data[i]) for i in range(n)]
        pyvals.append(vals)
    rec, st, _ = records(sim, Chunk(cols), list(range(len(shape))), True)
    assert (st == 0).all()
    is_str = np.array([1 if t == abi.BYTES else 0 for t in shape], np.int32)
    k = len(shape)
    for r in range(n):
        flag, word, off, ln = np.zeros(k, np.uint32), np.zeros(k, np.uint64), np.zeros(k, np.uint32), np.zeros(k, np.uint32)
        raw = np.ascontiguousarray(rec[r]).view(np.uint8)
        sim.sim_kr_parse(raw.ctypes.data, k, is_str.ctypes.data, flag.ctypes.data, word.ctypes.data, off.ctypes.data, ln.ctypes.data)
        for c in range(k):
            v = pyvals[c][r]
            if v is None:
                assert flag[c] == 0
            elif shape[c] == abi.BYTES:
                assert flag[c] == 2 and bytes(raw[off[c]:off[c] + ln[c]]) == v
            else:
                assert flag[c] in (8, 9) and int(word[c]) == v & ((1 << 64) - 1)
