"""CPU: the code k_rowcodec_decode runs per tile and per row (tinysql_amd/csrc/tsq_rowcodec_dp.h: tile plan, header parse, column
id search, value decode) compiled with g++ through tests/hostsim and walked tile by tile like the kernel does — staged copy at
every 16-byte phase of the `values` address, 256-row tiles, ballot-shaped bitmap bytes — against the oracle's restatement of
ChunkDecoder.DecodeToChunk.  The index arithmetic of the staging is where an off-by-one would hide; it is checked here without
a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as orc
from tinysql_amd import _abi as abi
from tinysql_amd.chunk import Chunk, Column, unpack_bitmap

HERE = os.path.dirname(os.path.abspath(__file__))
NP = {abi.I64: np.int64, abi.U64: np.uint64, abi.F64: np.float64, abi.F32: np.float32}


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "hostsim", "hostsim.so"))
    P = C.c_void_p
    lib.sim_rowcodec_decode.restype = C.c_uint64
    lib.sim_rowcodec_decode.argtypes = [P, C.c_int64, C.c_uint64, P, P, C.c_int64, C.POINTER(abi.RowcodecCol), C.c_int32, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.c_uint32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    return lib


def run_sim(sim, values, offsets, handles, specs, base_addr=0, lds_bytes=48 * 1024, fast_layout=1, stats=None):
    values = np.ascontiguousarray(values, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    types = [sp[1] for sp in specs]
    bufs = [np.full(max(n, 1), 0x55, dtype=NP[t]) if NP[t] in (np.int64, np.uint64) else np.zeros(max(n, 1), dtype=NP[t]) for t in types]
    bms = [np.full((n + 7) // 8 + 8, 0xEE, dtype=np.uint8) for _ in types]
    pd = (C.c_void_p * len(types))(*[b.ctypes.data for b in bufs])
    pb = (C.c_void_p * len(types))(*[b.ctypes.data for b in bms])
    h = np.ascontiguousarray(handles, dtype=np.int64) if handles is not None else None
    staged, fastw = C.c_int64(0), C.c_int64(0)
    err = sim.sim_rowcodec_decode(values.ctypes.data_as(C.c_void_p), values.size, base_addr, offsets.ctypes.data_as(C.c_void_p),
                                  h.ctypes.data_as(C.c_void_p) if h is not None else None, n, orc.rowcodec_cols(specs), len(specs), pd, pb, lds_bytes,
                                  fast_layout, C.byref(staged), C.byref(fastw))
    if stats is not None:
        stats.append(fastw.value)
    rows = n if err == (1 << 64) - 1 else err >> 4
    code = 0 if err == (1 << 64) - 1 else err & 15
    for bm in bms:  # nothing is written past the bitmap's last byte
        assert (bm[(n + 7) // 8:] == 0xEE).all()
    chk = Chunk([Column(t, b[:rows].copy(), unpack_bitmap(bm, rows)) for t, b, bm in zip(types, bufs, bms)])
    return code, chk, staged.value


def random_scan(rng, n, null_frac=0.2):
    edge = np.array([0, 1, -1, 127, -128, 128, -129, 32767, -32768, 32768, (1 << 31) - 1, -(1 << 31), 1 << 31, (1 << 63) - 1, -(1 << 63)])
    return Chunk([
        Column(abi.I64, np.where(rng.random(n) < 0.5, rng.choice(edge, n), rng.integers(-(1 << 62), 1 << 62, n)), rng.random(n) >= null_frac),
        Column(abi.U64, (rng.integers(0, 1 << 62, n) >> rng.integers(0, 62, n)).astype(np.uint64), rng.random(n) >= null_frac),
        Column(abi.F64, rng.standard_normal(n) * 1e6, rng.random(n) >= null_frac),
        Column(abi.F32, rng.standard_normal(n).astype(np.float32), rng.random(n) >= null_frac),
    ])


SPECS = [(200, abi.F64), (-1, abi.I64, abi.RC_HANDLE), (7, abi.I64), (31, abi.F32), (2, abi.U64), (99, abi.I64), (98, abi.I64, abi.RC_HAS_DEFAULT, 5)]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 1000, 5000])
def test_tiles_and_bitmap_tail(sim, n):
    rng = np.random.default_rng(n)
    chk = random_scan(rng, n)
    handles = rng.integers(-(1 << 62), 1 << 62, n)
    b, o = orc.rowcodec_encode(chk, [7, 2, 200, 31])
    st, want = orc.rowcodec_decode(b, o, handles, SPECS)
    code, got, staged = run_sim(sim, b, o, handles, SPECS)
    assert st == 0 and code == 0 and got.rows() == want.rows()
    assert staged == (n + 255) // 256


@pytest.mark.parametrize("phase", range(16))
def test_every_alignment_phase_of_the_values_pointer(sim, phase):
    # the staged copy starts at the 16-byte boundary below the tile's first byte: every phase of (address + tile_lo) & 15
    rng = np.random.default_rng(100 + phase)
    chk = random_scan(rng, 700)
    b, o = orc.rowcodec_encode(chk, [7, 2, 200, 31])
    st, want = orc.rowcodec_decode(b, o, None, SPECS[:1] + SPECS[2:])
    code, got, _ = run_sim(sim, b, o, None, SPECS[:1] + SPECS[2:], base_addr=0x7f0000001000 + phase)
    assert st == 0 and code == 0 and got.rows() == want.rows()


def test_large_ids_and_wide_rows_fall_back_to_global_reads(sim):
    # ids above 255 -> 4-byte ids and offsets; a 70000-byte string next to the requested columns -> large offsets, and a tile
    # that no longer fits the LDS budget is parsed straight from `values` (staged == 0 for it)
    rng = np.random.default_rng(3)
    n = 600
    chk = random_scan(rng, n)
    pad = np.where(np.arange(n) % 97 == 5, 70000, rng.integers(0, 40, n))
    b, o = orc.rowcodec_encode(chk, [7, 300, 200, 31], 24, pad)
    specs = [(300, abi.U64), (7, abi.I64), (200, abi.F64), (31, abi.F32), (24, abi.I64)]
    st, want = orc.rowcodec_decode(b, o, None, specs[:4])
    code, got, staged = run_sim(sim, b, o, None, specs[:4])
    assert st == 0 and code == 0 and got.rows() == want.rows() and staged == 0
    code, got, staged = run_sim(sim, b, o, None, specs[:4], lds_bytes=1 << 20)  # the same rows through the staged path
    assert code == 0 and got.rows() == want.rows() and staged == 3


def test_empty_rows_and_schema_without_columns(sim):
    # a row with no columns at all (6-byte header) and columns that are all absent / NULL
    chk = Chunk([Column(abi.I64, np.zeros(300, np.int64), np.zeros(300, bool))])
    b, o = orc.rowcodec_encode(chk, [4])
    specs = [(4, abi.I64), (5, abi.F64), (6, abi.U64, abi.RC_HAS_DEFAULT, 77)]
    st, want = orc.rowcodec_decode(b, o, None, specs)
    code, got, _ = run_sim(sim, b, o, None, specs)
    assert st == 0 and code == 0 and got.rows() == want.rows() == [(None, None, 77)] * 300


@pytest.mark.parametrize("case", ["version", "short_float", "cut_header", "cut_value", "odd_int", "offsets_backwards", "empty_value"])
def test_first_error_in_scan_order(sim, case):
    rng = np.random.default_rng(9)
    n = 900
    chk = random_scan(rng, n, null_frac=0.0)
    b, o = orc.rowcodec_encode(chk, [7, 2, 200, 31])
    specs = [(7, abi.I64), (2, abi.U64), (200, abi.F64), (31, abi.F32)]
    b, o = b.copy(), o.copy()
    at = 517
    if case == "version":
        b[o[at]] = 1
        b[o[at + 100]] = 1  # a later one does not matter
        want_code = 1
    elif case == "short_float":
        specs = [(7, abi.F64)] + specs[1:]  # an int column read as a real: rows whose int is narrower than 8 bytes fail
        want_code = 3
        at = None
    elif case == "cut_header":
        rows = [b[o[r]:o[r + 1]] for r in range(n)]
        rows[at] = rows[at][:9]
        b = np.concatenate(rows)
        o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        want_code = 2
    elif case == "cut_value":
        rows = [b[o[r]:o[r + 1]] for r in range(n)]
        rows[at] = rows[at][:-3]  # the last value (the double, id 200) runs past the end of the row
        b = np.concatenate(rows)
        o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        want_code = 2
    elif case == "odd_int":
        # a 3-byte int value: hand-made row, id 7 -> 3 bytes (LittleEndian.Uint64 on 3 bytes panics in the reference)
        rows = [b[o[r]:o[r + 1]] for r in range(n)]
        rows[at] = np.array([128, 0, 1, 0, 0, 0, 7, 3, 0, 1, 2, 3], dtype=np.uint8)
        b = np.concatenate(rows)
        o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        specs = specs[:1]
        want_code = 2
    elif case == "offsets_backwards":
        o[at + 1] = o[at] - 1
        want_code = 2
    else:
        rows = [b[o[r]:o[r + 1]] for r in range(n)]
        rows[at] = rows[at][:0]
        b = np.concatenate(rows)
        o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        want_code = 2
    if case == "offsets_backwards":
        # the oracle walks rows one by one and would read a negative length: compare with the rows before the damaged one
        st, want = orc.rowcodec_decode(b, o[:at + 1], None, specs)
        st = 2
    else:
        st, want = orc.rowcodec_decode(b, o, None, specs)
    code, got, _ = run_sim(sim, b, o, None, specs)
    assert st == want_code and code == want_code
    if at is not None:
        assert want.NumRows() == at
    assert got.rows() == want.rows()


# ---- waves whose rows share one layout (rc_rows_lds: the column search runs once per wave on the shared signature)
def _uniform_scan(rng, n, n_int=3, null_cols=()):
    """no NULLs except whole columns listed in null_cols (NULL in every row: the layout stays the same)"""
    cols = []
    for j in range(n_int):
        v = np.where(rng.random(n) < 0.5, rng.choice(np.array([0, 1, -1, 127, -128, 128, 32767, -32769, (1 << 31) - 1, 1 << 31, -(1 << 63)]), n),
                     rng.integers(-(1 << 62), 1 << 62, n))
        cols.append(Column(abi.I64, v, np.zeros(n, bool) if j in null_cols else None))
    cols.append(Column(abi.U64, (rng.integers(0, 1 << 62, n) >> rng.integers(0, 62, n)).astype(np.uint64)))
    cols.append(Column(abi.F64, rng.standard_normal(n) * 1e6))
    cols.append(Column(abi.F32, rng.standard_normal(n).astype(np.float32)))
    return Chunk(cols)


def _both_paths(sim, b, o, handles, specs, want_fast=None, **kw):
    st, want = orc.rowcodec_decode(b, o, handles, specs)
    stats = []
    code1, got1, _ = run_sim(sim, b, o, handles, specs, fast_layout=1, stats=stats, **kw)
    code0, got0, _ = run_sim(sim, b, o, handles, specs, fast_layout=0, **kw)
    assert code1 == code0 == st and got1.rows() == want.rows() and got0.rows() == want.rows()
    if want_fast is not None:
        assert stats[0] == want_fast
    return stats[0]


@pytest.mark.parametrize("n", [1, 64, 65, 300, 2000])
def test_shared_layout_waves_take_the_fast_path_and_agree(sim, n):
    rng = np.random.default_rng(500 + n)
    chk = _uniform_scan(rng, n, null_cols=(1,))
    ids = [9, 3, 17, 4, 200, 31]
    b, o = orc.rowcodec_encode(chk, ids)
    handles = rng.integers(-(1 << 62), 1 << 62, n)
    specs = [(200, abi.F64), (-1, abi.I64, abi.RC_HANDLE), (9, abi.I64), (3, abi.I64), (31, abi.F32), (4, abi.U64), (17, abi.I64), (99, abi.I64),
             (98, abi.I64, abi.RC_HAS_DEFAULT, 5), (1 << 40, abi.I64), (-7, abi.I64)]
    _both_paths(sim, b, o, handles, specs, want_fast=(n + 63) // 64)


def test_eight_ids_are_a_signature_nine_are_not(sim):
    rng = np.random.default_rng(8)
    n = 500
    for k, fast in ((8, (n + 63) // 64), (9, 0)):
        chk = Chunk([Column(abi.I64, rng.integers(-300, 300, n)) for _ in range(k)])
        ids = list(range(10, 10 + k))
        b, o = orc.rowcodec_encode(chk, ids)
        _both_paths(sim, b, o, None, [(i, abi.I64) for i in reversed(ids)] + [(5, abi.I64)], want_fast=fast)


def test_mixed_layouts_vote_wave_by_wave(sim):
    # rows 0..639 share a layout except row 200 (one NULL) and rows 320..383 (a different table shape); large ids never qualify
    rng = np.random.default_rng(77)
    n = 640
    nn = np.ones(n, bool)
    nn[200] = False
    chk = Chunk([Column(abi.I64, rng.integers(-9999, 9999, n), nn), Column(abi.F64, rng.random(n)), Column(abi.I64, rng.integers(0, 9, n))])
    b1, o1 = orc.rowcodec_encode(chk, [1, 2, 3])
    other = Chunk([Column(abi.I64, rng.integers(-9999, 9999, 64)), Column(abi.F64, rng.random(64))])
    b2, o2 = orc.rowcodec_encode(other, [1, 2])
    rows = [b1[o1[r]:o1[r + 1]] for r in range(n)]
    rows[320:384] = [b2[o2[r]:o2[r + 1]] for r in range(64)]
    b = np.concatenate(rows)
    o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    fast = _both_paths(sim, b, o, None, [(3, abi.I64), (1, abi.I64), (2, abi.F64)])
    assert fast == 9  # 10 waves, the one holding row 200 is mixed; rows 320..383 are uniform among themselves
    big, ob = orc.rowcodec_encode(chk, [1, 2, 300])
    assert _both_paths(sim, big, ob, None, [(300, abi.I64), (1, abi.I64), (2, abi.F64)]) == 0


@pytest.mark.parametrize("case", ["cut_value", "odd_int", "short_float", "version_first_lane", "version_other_lane"])
def test_errors_inside_shared_layout_waves(sim, case):
    rng = np.random.default_rng(31)
    n = 700
    chk = _uniform_scan(rng, n)
    ids = [9, 3, 17, 4, 200, 31]
    b, o = orc.rowcodec_encode(chk, ids)
    specs = [(9, abi.I64), (3, abi.I64), (17, abi.I64), (4, abi.U64), (200, abi.F64), (31, abi.F32)]
    rows = [b[o[r]:o[r + 1]].copy() for r in range(n)]
    at = 453
    if case == "cut_value":
        rows[at] = rows[at][:-5]          # header and ids intact (the wave still votes "same layout"), the last value runs past the row
    elif case == "odd_int":
        r = rows[at]
        nnc = int(r[2])
        offs_at = 6 + nnc
        ends = r[offs_at:offs_at + 2 * nnc].view("<u2").astype(np.int64)
        w0 = int(ends[0])                 # first value (id 3) gets 3 bytes: shift the later offsets, keep the layout
        ends = (ends + (3 - w0)).astype("<u2")
        rows[at] = np.concatenate([r[:offs_at], ends.view(np.uint8), np.array([1, 2, 3], np.uint8), r[offs_at + 2 * nnc + w0:]])
    elif case == "short_float":
        specs = [(9, abi.F64)] + specs[1:]
    elif case == "version_first_lane":
        at = 448                          # lane 0 of its wave: the wave has no valid signature and takes the general path
        rows[at][0] = 7
    else:
        rows[at][0] = 7
    b = np.concatenate(rows)
    o = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    st, want = orc.rowcodec_decode(b, o, None, specs)
    assert st != 0 and (case == "short_float" or want.NumRows() == at)
    _both_paths(sim, b, o, None, specs)


def test_string_columns_become_references_into_the_row(sim):
    # a TSQ_BYTES column: the per-row code leaves (start inside the row) << 32 | length — what k_rowcodec_var_len / _copy turn into
    # offsets and bytes — checked here against the cells of the oracle's DecodeToChunk restatement, on both wave paths and for rows
    # parsed from global memory
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(77)
    n = 3000
    words = [None if rng.random() < 0.15 else bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)) for _ in range(n)]
    wide = [None if rng.random() < 0.1 else b"w" * int(rng.integers(0, 3) * 120) for _ in range(n)]
    chk = Chunk([Column(abi.I64, rng.integers(-1000, 1000, n), rng.random(n) >= 0.2), StrColumn(words), StrColumn(wide)])
    b, o = orc.rowcodec_encode(chk, [3, 1, 8])
    specs = [(1, abi.BYTES), (3, abi.I64), (8, abi.BYTES), (4, abi.BYTES)]
    st, want = orc.rowcodec_decode_chunk(b, o, None, specs)
    assert st == 0
    for lds, fast in ((48 * 1024, 1), (48 * 1024, 0), (8 * 1024, 1)):  # 8 KB: the wide tiles fall back to global reads
        isim = [(sp[0], abi.I64 if sp[1] == abi.BYTES else sp[1]) + tuple(sp[2:]) for sp in specs]  # buffers: 8 bytes per row
        isim_specs = [(sp[0], sp[1]) for sp in specs]
        values = np.ascontiguousarray(b, dtype=np.uint8)
        offsets = np.ascontiguousarray(o, dtype=np.int64)
        bufs = [np.zeros(n, np.uint64) for _ in specs]
        bms = [np.zeros((n + 7) // 8 + 8, np.uint8) for _ in specs]
        pd = (C.c_void_p * len(specs))(*[x.ctypes.data for x in bufs])
        pb = (C.c_void_p * len(specs))(*[x.ctypes.data for x in bms])
        staged, fastw = C.c_int64(0), C.c_int64(0)
        err = sim.sim_rowcodec_decode(values.ctypes.data_as(C.c_void_p), values.size, 5, offsets.ctypes.data_as(C.c_void_p), None, n, orc.rowcodec_cols(isim_specs),
                                      len(specs), pd, pb, lds, fast, C.byref(staged), C.byref(fastw))
        assert err == (1 << 64) - 1 and isim
        for c, sp in enumerate(specs):
            nn = unpack_bitmap(bms[c], n)
            col = want.columns[c]
            if sp[1] != abi.BYTES:
                assert [None if not nn[r] else int(bufs[c][r].astype(np.int64)) for r in range(n)] == col.values()
                continue
            for r in range(n):
                ref = int(bufs[c][r])
                if col.IsNull(r):
                    assert not nn[r] and ref == 0  # a NULL cell has no bytes
                else:
                    start, ln = ref >> 32, ref & 0xffffffff
                    assert nn[r] and bytes(values[o[r] + start:o[r] + start + ln]) == col.values()[r]


def test_default_string_reference_is_handed_out_for_an_absent_column(sim):
    # a TSQ_BYTES column with TSQ_RC_HAS_DEFAULT: the host puts the cell reference of the default string — (1 << 63) | where << 32 |
    # length, into the pool of default strings — into def_bits, and the per-row code hands it out wherever the row lacks the column
    # (defDatum, decoder.go:186-194); a column that is present (or NULL in the row) is untouched by the default
    from tinysql_amd.chunk import StrColumn
    rng = np.random.default_rng(5)
    n = 500
    words = [None if rng.random() < 0.3 else b"w%d" % i for i in range(n)]
    chk = Chunk([Column(abi.I64, rng.integers(0, 9, n)), StrColumn(words)])
    b, o = orc.rowcodec_encode(chk, [1, 2])
    default = b"the default"
    specs = [(2, abi.BYTES, abi.RC_HAS_DEFAULT, default), (7, abi.BYTES, abi.RC_HAS_DEFAULT, default), (1, abi.I64)]
    st, want = orc.rowcodec_decode_chunk(b, o, None, specs)
    assert st == 0 and want.columns[0].values() == words and want.columns[1].values() == [default] * n
    ref = (1 << 63) | (4 << 32) | len(default)  # the pool holds 4 bytes of another column's default first
    isim = [(2, abi.BYTES, abi.RC_HAS_DEFAULT, ref), (7, abi.BYTES, abi.RC_HAS_DEFAULT, ref), (1, abi.I64)]
    values, offsets = np.ascontiguousarray(b, dtype=np.uint8), np.ascontiguousarray(o, dtype=np.int64)
    for fast in (1, 0):
        bufs = [np.zeros(n, np.uint64) for _ in isim]
        bms = [np.zeros((n + 7) // 8 + 8, np.uint8) for _ in isim]
        pd = (C.c_void_p * 3)(*[x.ctypes.data for x in bufs])
        pb = (C.c_void_p * 3)(*[x.ctypes.data for x in bms])
        staged, fastw = C.c_int64(0), C.c_int64(0)
        err = sim.sim_rowcodec_decode(values.ctypes.data_as(C.c_void_p), values.size, 3, offsets.ctypes.data_as(C.c_void_p), None, n, orc.rowcodec_cols(isim), 3, pd, pb,
                                      48 * 1024, fast, C.byref(staged), C.byref(fastw))
        assert err == (1 << 64) - 1
        assert (bufs[1] == np.uint64(ref)).all() and unpack_bitmap(bms[1], n).all()  # column 7 is in no row
        nn = unpack_bitmap(bms[0], n)
        for r in range(n):  # column 2 is in every row (NULL or not): never the default
            if words[r] is None:
                assert not nn[r]
            else:
                v = int(bufs[0][r])
                assert nn[r] and v >> 63 == 0 and bytes(values[o[r] + (v >> 32):o[r] + (v >> 32) + (v & 0xffffffff)]) == words[r]
