"""GPU parity of the chunk wire format (tsq_chunk_encode / tsq_chunk_decode / tsq_chunk_decode_peek; SURVEY.md §8 a/A "wire Codec")
against the oracle's restatement of chunk.Codec and chunk.Decoder (util/chunk/codec.go:28-143, 233-353): byte-exact wire buffers,
byte-exact Column state (nullBitmap, offsets, data) after every Decoder step; the reference's TestCodec through the mirror."""
import ctypes as C

import numpy as np
import pytest

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd import chunk_codec as CC
from tinysql_amd.chunk import Chunk, Column, StrColumn

from . import gpu_helpers as G

pytestmark = pytest.mark.gpu

TYPES = [abi.I64, abi.F32, abi.BYTES, abi.F64, abi.U64, abi.BYTES]
ELEM = [8, 4, -1, 8, 8, -1]


def _chunk(rng, n, null_p=0.2, cell=9):
    def nn(p=null_p):
        return None if p == 0 else rng.random(n) >= p
    def strs(p):
        return [None if (p and rng.random() < p) else bytes(rng.integers(0, 256, int(rng.integers(0, cell)), dtype=np.uint8)) for _ in range(n)]
    return Chunk([Column(abi.I64, rng.integers(-1 << 62, 1 << 62, n), nn()), Column(abi.F32, rng.standard_normal(n).astype(np.float32), nn()), StrColumn(strs(null_p)),
                  Column(abi.F64, rng.standard_normal(n)), Column(abi.U64, rng.integers(0, 1 << 63, n).astype(np.uint64), nn(0.5)), StrColumn(strs(0))])


def _same_state(wire_chunk, orc_chunk):
    """Column by column what the reference would hold: length, nullBitmap bytes, offsets, data bytes."""
    for c, col in enumerate(wire_chunk.columns):
        ln, bm, offs, data = orc_chunk.column(c)
        assert col.length == ln
        assert col.nullBitmap[:(ln + 7) // 8].tobytes() == bm, c
        if offs is not None:
            assert col.offsets[:ln + 1].tolist() == offs, c
        assert col.data[:len(data)].tobytes() == data, c


def test_reference_test_codec_through_the_mirror(ctx, orc):
    # util/chunk/codec_test.go:29-71
    numRows = 10
    colTypes = [abi.I64, abi.I64, abi.BYTES, abi.BYTES]
    oldChk = Chunk([Column(abi.I64, np.zeros(numRows, np.int64), np.zeros(numRows, bool)), Column(abi.I64, np.arange(numRows)),
                    StrColumn([b"%d.12345" % i for i in range(numRows)]), StrColumn([b"%d.12345" % i for i in range(numRows)])])
    codec = CC.Codec(ctx, colTypes)
    buffer = codec.Encode(oldChk)
    assert buffer == orc.WireChunk.from_chunk(oldChk).encode()
    newChk = CC.WireChunk(colTypes, numRows)
    remained = codec.DecodeToChunk(buffer, newChk)
    assert len(remained) == 0 and newChk.NumCols() == 4 and newChk.NumRows() == numRows
    rows = newChk.to_chunk().rows()
    for i in range(numRows):
        assert rows[i] == (None, i, b"%d.12345" % i, b"%d.12345" % i)


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 63, 64, 65, 1000, 1024, 50_001])
@pytest.mark.parametrize("null_p", [0.0, 0.2])
def test_encode_is_byte_exact_and_decode_restores_the_columns(ctx, orc, n, null_p):
    rng = np.random.default_rng(n * 3 + int(null_p * 10))
    chk = _chunk(rng, n, null_p)
    want = orc.WireChunk.from_chunk(chk).encode()
    codec = CC.Codec(ctx, TYPES)
    buf = codec.Encode(chk)
    assert buf == want
    # DecodeToChunk + a column behind the chunk stays with the caller (codec.go:93)
    got, remained = codec.Decode(buf + b"tail")
    assert remained == b"tail"
    inter, dst = orc.WireChunk(ELEM), orc.WireChunk(ELEM)
    inter.decoder_reset(want)
    assert inter.decoder_decode(dst, n) == n
    _same_state(got, dst)
    assert got.to_chunk().rows() == chk.rows()


@pytest.mark.parametrize("n,required", [(100, 32), (1000, 100), (1024, 1024), (77, 8), (4099, 1000)])
def test_decoder_steps_equal_the_reference_decoder(ctx, orc, n, required):
    # select_result-style loop: Reset(data); while !IsFinished: Decode(chk) appends a multiple of 8 rows; the destination is
    # emptied only every other time, so rows are also appended behind rows (bit offsets != 0 when the tail was not a multiple of 8)
    rng = np.random.default_rng(n + required)
    chk = _chunk(rng, n)
    data = orc.WireChunk.from_chunk(chk).encode()
    dec = CC.Decoder(ctx, CC.WireChunk(TYPES, required), TYPES)
    dec.Reset(data)
    inter = orc.WireChunk(ELEM)
    inter.decoder_reset(data)
    assert dec.RemainedRows() == inter.decoder_remained() == n
    dst, odst = CC.WireChunk(TYPES, required), orc.WireChunk(ELEM)
    step = 0
    while not dec.IsFinished():
        if step % 2 == 0:
            dst.Reset()
            odst = orc.WireChunk(ELEM)
            dst.requiredRows = required
        else:
            dst.requiredRows = dst.NumRows() + required  # more room: append behind the rows of the previous step
        before = dst.NumRows()
        dec.Decode(dst)
        inter.decoder_decode(odst, dst.requiredRows - before)
        assert dec.RemainedRows() == inter.decoder_remained()
        _same_state(dst, odst)
        step += 1
    assert step >= 1


def test_append_behind_rows_that_are_not_a_multiple_of_eight(ctx, orc):
    # the last window of one response (5 rows) is followed by the first window of the next one: destination bit offset 5
    rng = np.random.default_rng(2)
    a, b = _chunk(rng, 13), _chunk(rng, 300)
    da, db = orc.WireChunk.from_chunk(a).encode(), orc.WireChunk.from_chunk(b).encode()
    dst, odst = CC.WireChunk(TYPES, 1024), orc.WireChunk(ELEM)
    for data in (da, db):
        dec = CC.Decoder(ctx, CC.WireChunk(TYPES), TYPES)
        dec.Reset(data)
        inter = orc.WireChunk(ELEM)
        inter.decoder_reset(data)
        dec.Decode(dst)
        inter.decoder_decode(odst, 1024 - odst.column(0)[0])
        _same_state(dst, odst)
    assert dst.NumRows() == 313 and dst.to_chunk().rows() == a.rows() + b.rows()


def test_reuse_interm_chk(ctx, orc):
    rng = np.random.default_rng(3)
    chk = _chunk(rng, 500)
    data = orc.WireChunk.from_chunk(chk).encode()
    dec = CC.Decoder(ctx, CC.WireChunk(TYPES), TYPES)
    dec.Reset(data)
    inter = orc.WireChunk(ELEM)
    inter.decoder_reset(data)
    first, ofirst = CC.WireChunk(TYPES, 96), orc.WireChunk(ELEM)
    dec.Decode(first)
    inter.decoder_decode(ofirst, 96)
    rest, orest = CC.WireChunk(TYPES), orc.WireChunk(ELEM)
    dec.ReuseIntermChk(rest)
    inter.decoder_reuse(orest)
    assert dec.IsFinished() and rest.NumRows() == 404
    for c, col in enumerate(rest.columns):  # (the reference's reused bitmap keeps the wire bytes: compare the valid bits)
        ln, bm, offs, data_c = orest.column(c)
        assert col.length == ln
        nb = (ln + 7) // 8
        mask = np.unpackbits(np.frombuffer(bm[:nb], np.uint8), bitorder="little")[:ln]
        assert (np.unpackbits(col.nullBitmap[:nb], bitorder="little")[:ln] == mask).all()
        if offs is not None:
            assert col.offsets[:ln + 1].tolist() == offs[:ln + 1]
            assert col.data[:offs[ln]].tobytes() == data_c[:offs[ln]]
        else:
            assert col.data[:ln * ELEM[c]].tobytes() == data_c[:ln * ELEM[c]]
    assert first.to_chunk().rows() + rest.to_chunk().rows() == chk.rows()


def test_device_resident_encode_and_decode(ctx, orc):
    # device columns -> device wire buffer -> device columns (no host copy of the data), appended behind 11 rows
    rng = np.random.default_rng(4)
    n = 20_000
    chk = Chunk(_chunk(rng, n).columns[:3])
    types, elem = TYPES[:3], ELEM[:3]
    want = orc.WireChunk.from_chunk(chk).encode()
    lib = ctx.lib
    d0, d1 = G.DevCol(ctx, abi.I64, n, True), G.DevCol(ctx, abi.F32, n, True)
    ctx.h2d(d0.data, np.ascontiguousarray(chk.columns[0].data)); ctx.h2d(d0.bitmap, chk.columns[0].bitmap())
    ctx.h2d(d1.data, np.ascontiguousarray(chk.columns[1].data)); ctx.h2d(d1.bitmap, chk.columns[1].bitmap())
    d2 = G.DevStrCol(ctx, chk.columns[2])
    cols = G.dev_cols([d0, d1, d2])
    need = C.c_int64(0)
    _lib.check(lib.tsq_chunk_encode(ctx.h, cols, 3, n, None, 0, abi.COL_DEVICE, C.byref(need)), ctx.h)
    assert need.value == len(want)
    dbuf = ctx.alloc(need.value + 64)
    _lib.check(lib.tsq_chunk_encode(ctx.h, cols, 3, n, C.c_void_p(dbuf + 3), need.value, abi.COL_DEVICE, C.byref(need)), ctx.h)  # an odd byte position
    back = np.zeros(need.value, np.uint8)
    ctx.d2h(back, dbuf + 3)
    assert back.tobytes() == want
    # decode rows [8, 8 + 5000) behind 11 rows that are already there
    head = Chunk([c.slice(0, 11) for c in chk.columns])
    hbuf = orc.WireChunk.from_chunk(head).encode()
    o0, o1 = G.DevCol(ctx, abi.I64, 6000, True), G.DevCol(ctx, abi.F32, 6000, True)
    o2 = G.DevStrCol(ctx, nrows=6000, nbytes=len(chk.columns[2].data) + 64)
    out = G.dev_cols([o0, o1, o2])
    for c in range(3):
        out[c].length = 0
    tp = (C.c_int32 * 3)(*types)
    hraw = np.frombuffer(hbuf + b"\0" * 8, np.uint8)
    nrows, used = C.c_int64(0), C.c_int64(0)
    _lib.check(lib.tsq_chunk_decode(ctx.h, hraw.ctypes.data_as(C.c_void_p), len(hbuf), 0, tp, 3, 0, 1 << 40, out, C.byref(nrows), C.byref(used)), ctx.h)
    assert nrows.value == 11 and used.value == len(hbuf) and out[0].length == 11
    total, take, used2 = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    nb = (C.c_int64 * 3)()
    _lib.check(lib.tsq_chunk_decode_peek(ctx.h, C.c_void_p(dbuf + 3), need.value, abi.COL_DEVICE, tp, 3, 8, 5000, C.byref(total), C.byref(take), nb, C.byref(used2)), ctx.h)
    offs = chk.columns[2].offsets
    assert (total.value, take.value, used2.value, list(nb)) == (n, 5000, len(want), [0, 0, int(offs[5008] - offs[8])])
    _lib.check(lib.tsq_chunk_decode(ctx.h, C.c_void_p(dbuf + 3), need.value, abi.COL_DEVICE, tp, 3, 8, 5000, out, C.byref(nrows), C.byref(used)), ctx.h)
    assert nrows.value == 5000 and out[2].length == 5011
    o0.n = o1.n = 5011
    wantc = Chunk([c.slice(0, 11) for c in chk.columns]).rows() + Chunk([c.slice(8, 5008) for c in chk.columns]).rows()
    got = Chunk([o0.to_host(), o1.to_host(), o2.to_host(5011, int(offs[11] + offs[5008] - offs[8]))])
    assert got.rows() == wantc
    for d in (d0, d1, d2, o0, o1, o2):
        d.free()
    ctx.free(dbuf)


def test_damaged_buffers_and_bad_arguments(ctx, orc):
    rng = np.random.default_rng(6)
    chk = _chunk(rng, 40)
    buf = orc.WireChunk.from_chunk(chk).encode()
    codec = CC.Codec(ctx, TYPES)
    for cut in (0, 5, 8, 30, len(buf) // 2, len(buf) - 1):
        assert orc.WireChunk(ELEM).decode_to_chunk(buf[:cut]) == -1
        with pytest.raises(_lib.TsqError) as e:
            codec.Decode(buf[:cut])
        assert e.value.status == abi.ERR_INVALID and "ends inside column" in str(e.value)
    # offsets that run backwards / beyond the data
    pos = 8 + 5 + 320 + 8 + 5 + 160 + 8 + 5  # header + bitmap + data of columns 0, 1; header + bitmap of column 2 -> its offsets
    bad = bytearray(buf)
    bad[pos + 8 * 40:pos + 8 * 41] = (1 << 40).to_bytes(8, "little")  # offsets[length]: data bytes beyond the buffer
    with pytest.raises(_lib.TsqError):
        codec.Decode(bytes(bad))
    # columns of different lengths
    other = orc.WireChunk.from_chunk(Chunk(_chunk(rng, 41).columns[1:])).encode()
    first_col = orc.WireChunk.from_chunk(Chunk(chk.columns[:1])).encode()
    with pytest.raises(_lib.TsqError) as e:
        codec.Decode(first_col + other)
    assert "different lengths" in str(e.value)
    # first_row must be a multiple of 8
    raw = np.frombuffer(buf + b"\0" * 8, np.uint8)
    tp = (C.c_int32 * 6)(*TYPES)
    tot = C.c_int64(0)
    st = ctx.lib.tsq_chunk_decode_peek(ctx.h, raw.ctypes.data_as(C.c_void_p), len(buf), 0, tp, 6, 3, 8, C.byref(tot), None, None, None)
    assert st == abi.ERR_INVALID
    # a too small out buffer: the bytes needed are reported
    from tinysql_amd.chunk import make_cols
    keep = []
    cols = make_cols(chk.columns, keep)
    need = C.c_int64(0)
    out = np.zeros(16, np.uint8)
    assert ctx.lib.tsq_chunk_encode(ctx.h, cols, 6, 40, out.ctypes.data_as(C.c_void_p), 16, 0, C.byref(need)) == abi.ERR_INVALID and need.value == len(buf)


def test_full_size_round_trip_on_device(ctx):
    # 2e7 rows x (bigint with NULLs, double, bigint without NULLs): generated on the device, encoded, decoded, compared on the device
    # through the wire buffer's own structure: the decoded columns re-encode to the same bytes (encode o decode = identity)
    import zlib
    n = 20_000_000
    lib = ctx.lib
    cols = [G.DevCol(ctx, abi.I64, n, True), G.DevCol(ctx, abi.F64, n, False), G.DevCol(ctx, abi.I64, n, False)]
    ctx.gen_column(G.gen_spec(2, table=3, col=0, m=1 << 40, null_pct=10), n, cols[0].data, cols[0].bitmap)
    ctx.gen_column(G.gen_spec(3, table=3, col=1), n, cols[1].data)
    ctx.gen_column(G.gen_spec(0, start=-5), n, cols[2].data)
    arr = G.dev_cols(cols)
    need = C.c_int64(0)
    _lib.check(lib.tsq_chunk_encode(ctx.h, arr, 3, n, None, 0, abi.COL_DEVICE, C.byref(need)), ctx.h)
    assert need.value == 3 * 8 + (n + 7) // 8 + 3 * 8 * n
    w1, w2 = ctx.alloc(need.value + 64), ctx.alloc(need.value + 64)
    _lib.check(lib.tsq_chunk_encode(ctx.h, arr, 3, n, C.c_void_p(w1), need.value, abi.COL_DEVICE, C.byref(need)), ctx.h)
    outs = [G.DevCol(ctx, abi.I64, n, True), G.DevCol(ctx, abi.F64, n, True), G.DevCol(ctx, abi.I64, n, True)]
    oarr = G.dev_cols(outs)
    for c in range(3):
        oarr[c].length = 0
    tp = (C.c_int32 * 3)(abi.I64, abi.F64, abi.I64)
    nrows, used = C.c_int64(0), C.c_int64(0)
    _lib.check(lib.tsq_chunk_decode(ctx.h, C.c_void_p(w1), need.value, abi.COL_DEVICE, tp, 3, 0, n, oarr, C.byref(nrows), C.byref(used)), ctx.h)
    assert nrows.value == n and used.value == need.value
    need2 = C.c_int64(0)
    _lib.check(lib.tsq_chunk_encode(ctx.h, oarr, 3, n, C.c_void_p(w2), need.value, abi.COL_DEVICE, C.byref(need2)), ctx.h)
    assert need2.value == need.value
    a, b = np.zeros(need.value, np.uint8), np.zeros(need.value, np.uint8)
    ctx.d2h(a, w1)
    ctx.d2h(b, w2)
    assert zlib.crc32(a) == zlib.crc32(b) and (a == b).all()
    # and the decoded first column is the generated one
    h0, g0 = outs[0].to_host(), cols[0].to_host()
    assert (h0.data == g0.data).all() and (h0.notnull == g0.notnull).all()
    for c in cols + outs:
        c.free()
    ctx.free(w1)
    ctx.free(w2)


def test_bits_beyond_the_last_row_do_not_travel(ctx, orc):
    # a device column whose bitmap was preset to all ones (only NULLs cleared, like tsq_rows_decode leaves it) has set bits beyond its
    # last row; a Go Column never has (column.go:113-125), so the encoder clears them: the wire bytes are those of the clean column
    rng = np.random.default_rng(7)
    n = 1003
    nn = rng.random(n) >= 0.3
    clean = Chunk([Column(abi.I64, rng.integers(0, 99, n), nn)])
    want = orc.WireChunk.from_chunk(clean).encode()
    d = G.DevCol(ctx, abi.I64, n, True)
    ctx.h2d(d.data, np.ascontiguousarray(clean.columns[0].data))
    bm = clean.columns[0].bitmap().copy()
    bm[n // 8] |= 0xFF ^ ((1 << (n % 8)) - 1)  # garbage in the unused bits of the last byte
    ctx.h2d(d.bitmap, bm)
    need = C.c_int64(0)
    out = np.zeros(len(want) + 8, np.uint8)
    _lib.check(ctx.lib.tsq_chunk_encode(ctx.h, G.dev_cols([d]), 1, n, out.ctypes.data_as(C.c_void_p), len(want), 0, C.byref(need)), ctx.h)
    assert need.value == len(want) and out[:len(want)].tobytes() == want
    d.free()


def test_offsets_that_decrease_in_the_middle_are_refused_before_anything_is_written(ctx, orc):
    # offsets[0] == 0 and a plausible offsets[rows] are not enough — an offset in between that runs backwards (or far
    # beyond the data) would put negative lengths / out-of-buffer cells into the decoded column.  The window's offsets are checked on
    # the device (k_wire_check_offs) before a byte of the destination changes.
    rng = np.random.default_rng(16)
    chk = _chunk(rng, 40)
    buf = orc.WireChunk.from_chunk(chk).encode()
    codec = CC.Codec(ctx, TYPES)
    pos = 8 + 5 + 320 + 8 + 5 + 160 + 8 + 5  # the offsets of column 2 (see test_damaged_buffers_and_bad_arguments)
    for row, val in ((17, 1 << 40), (17, -5), (3, 10**6), (39, 0)):
        bad = bytearray(buf)
        end = int.from_bytes(buf[pos + 8 * 40:pos + 8 * 41], "little")
        if val == 0 and end == 0:
            continue
        bad[pos + 8 * row:pos + 8 * row + 8] = int(val).to_bytes(8, "little", signed=True)
        with pytest.raises(_lib.TsqError) as e:
            codec.Decode(bytes(bad))
        assert e.value.status == abi.ERR_INVALID and "damaged" in str(e.value), (row, val, str(e.value))
    assert codec.Decode(buf)[0].to_chunk().rows() == chk.rows()  # the undamaged chunk still decodes


def test_encode_of_a_var_len_view_whose_offsets_start_anywhere(ctx, orc):
    # DeviceColumn.view / tsq_colset_slice hand out var-len columns whose offsets do not start at 0.  The wire chunk carries
    # offsets from 0 and only the view's bytes — byte-identical to encoding the sliced column (Codec.Encode of chk.Slice, codec.go:42-76)
    rng = np.random.default_rng(18)
    n = 5000
    chk = Chunk(_chunk(rng, n).columns[:3])
    lo, hi = 1000, 4200  # a multiple of 8: the bitmap of the view starts on a byte
    want = orc.WireChunk.from_chunk(Chunk([c.slice(lo, hi) for c in chk.columns])).encode()
    lib = ctx.lib
    # host columns: pointers into the middle of the arrays
    from tinysql_amd.chunk import make_cols
    keep = []
    cols = make_cols(chk.columns, keep)
    for c in range(3):
        es = ELEM[c] if ELEM[c] > 0 else 0
        if es:
            cols[c].data = cols[c].data + lo * es
        else:
            cols[c].offsets = cols[c].offsets + lo * 8
        if cols[c].null_bitmap:
            cols[c].null_bitmap = cols[c].null_bitmap + lo // 8
        cols[c].length = hi - lo
    need = C.c_int64(0)
    _lib.check(lib.tsq_chunk_encode(ctx.h, cols, 3, hi - lo, None, 0, 0, C.byref(need)), ctx.h)
    assert need.value == len(want)
    out = np.zeros(need.value, np.uint8)
    _lib.check(lib.tsq_chunk_encode(ctx.h, cols, 3, hi - lo, out.ctypes.data_as(C.c_void_p), need.value, 0, C.byref(need)), ctx.h)
    assert out.tobytes() == want
    # device columns: the same view of device-resident columns
    d0, d1 = G.DevCol(ctx, abi.I64, n, True), G.DevCol(ctx, abi.F32, n, True)
    ctx.h2d(d0.data, np.ascontiguousarray(chk.columns[0].data)); ctx.h2d(d0.bitmap, chk.columns[0].bitmap())
    ctx.h2d(d1.data, np.ascontiguousarray(chk.columns[1].data)); ctx.h2d(d1.bitmap, chk.columns[1].bitmap())
    d2 = G.DevStrCol(ctx, chk.columns[2])
    dcols = G.dev_cols([d0, d1, d2])
    for c in range(3):
        if ELEM[c] > 0:
            dcols[c].data = dcols[c].data + lo * ELEM[c]
        else:
            dcols[c].offsets = dcols[c].offsets + lo * 8
        if dcols[c].null_bitmap:
            dcols[c].null_bitmap = dcols[c].null_bitmap + lo // 8
        dcols[c].length = hi - lo
    dbuf = ctx.alloc(len(want) + 64)
    _lib.check(lib.tsq_chunk_encode(ctx.h, dcols, 3, hi - lo, C.c_void_p(dbuf + 5), len(want), abi.COL_DEVICE, C.byref(need)), ctx.h)
    back = np.zeros(len(want), np.uint8)
    ctx.d2h(back, dbuf + 5)
    assert back.tobytes() == want
    for d in (d0, d1, d2):
        d.free()
    ctx.free(dbuf)
