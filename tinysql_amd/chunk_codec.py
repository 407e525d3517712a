"""chunk.Codec and chunk.Decoder over the C-ABI (util/chunk/codec.go:28-143, 233-353; SURVEY.md §8 a/A "wire Codec").

The wire format is a chunk's columns one after the other — u32 length | u32 nullCount | [bitmap] | [offsets] | data — i.e. the
column-major answer a coprocessor can give instead of datum rows; decoding it into a device chunk is a copy at HBM rate
(tsq_chunk_decode).  Same names, argument meaning and results as the reference:

    Codec(colTypes).Encode(chk) -> bytes                        codec.go:42-48
    Codec(colTypes).Decode(buffer) -> (chk, remained)           codec.go:78-86
    Codec(colTypes).DecodeToChunk(buffer, chk) -> remained      codec.go:88-93
    Decoder(chk, colTypes): Reset(data) / Decode(chk) / IsFinished() / RemainedRows() / ReuseIntermChk(chk)    codec.go:246-308

`colTypes` are the ABI's column types (abi.I64 ... abi.BYTES; getFixedLen, codec.go:169-181, is their element size).  A chunk here
is a WireChunk: per column the raw Column state of the reference (length, nullBitmap bytes, offsets, data bytes) in numpy arrays.
"""
import ctypes as C

import numpy as np

from . import _abi as abi
from . import _lib
from .chunk import Chunk, Column, StrColumn, elem_size, unpack_bitmap, np_dtype


class WireColumn:
    """util/chunk/column.go:28-34 as raw buffers (capacity grows like append)."""

    def __init__(self, tp):
        self.tp = tp
        self.length = 0
        self.nullBitmap = np.zeros(16, np.uint8)
        self.offsets = np.zeros(2, np.int64) if tp == abi.BYTES else None  # newVarLenColumn: offsets = [0]
        self.data = np.zeros(64, np.uint8)

    def data_bytes(self):
        return int(self.offsets[self.length]) if self.tp == abi.BYTES else self.length * elem_size(self.tp)

    def reserve(self, rows, data_bytes):
        def grow(a, n):
            if a.size >= n:
                return a
            b = np.zeros(max(n, a.size * 2), a.dtype)
            b[:a.size] = a
            return b
        self.nullBitmap = grow(self.nullBitmap, (rows + 7) // 8 + 8)
        self.data = grow(self.data, data_bytes + 16)
        if self.offsets is not None:
            self.offsets = grow(self.offsets, rows + 2)

    def to_column(self):
        n = self.length
        nn = unpack_bitmap(self.nullBitmap, n)
        if self.tp == abi.BYTES:
            raw = self.data.tobytes()
            return StrColumn([raw[int(self.offsets[i]):int(self.offsets[i + 1])] if nn[i] else None for i in range(n)])
        return Column(self.tp, self.data[:n * elem_size(self.tp)].view(np_dtype(self.tp)).copy(), None if nn.all() else nn)


class WireChunk:
    def __init__(self, colTypes, requiredRows=1024):
        self.columns = [WireColumn(tp) for tp in colTypes]
        self.requiredRows = requiredRows

    def NumRows(self):
        return self.columns[0].length if self.columns else 0

    def NumCols(self):
        return len(self.columns)

    def RequiredRows(self):
        return self.requiredRows

    def Reset(self):
        for c in self.columns:
            c.length = 0
            if c.offsets is not None:
                c.offsets[0] = 0

    def SwapColumns(self, other):
        self.columns, other.columns = other.columns, self.columns

    def to_chunk(self):
        return Chunk([c.to_column() for c in self.columns])

    def _out_cols(self, keep):
        arr = (abi.Col * len(self.columns))()
        for i, c in enumerate(self.columns):
            arr[i].data = c.data.ctypes.data_as(C.c_void_p)
            arr[i].null_bitmap = c.nullBitmap.ctypes.data_as(C.c_void_p)
            if c.offsets is not None:
                arr[i].offsets = c.offsets.ctypes.data_as(C.c_void_p)
            arr[i].length = c.length
            arr[i].elem_size = -1 if c.tp == abi.BYTES else elem_size(c.tp)
            arr[i].type = c.tp
            arr[i].flags = 0
        keep.append(arr)
        return arr


def _types_arr(colTypes):
    return (C.c_int32 * len(colTypes))(*colTypes)


class Codec:
    def __init__(self, ctx, colTypes):
        self.ctx = ctx
        self.colTypes = list(colTypes)

    def Encode(self, chk):
        """chk: a tinysql_amd.chunk.Chunk (host columns) -> the wire bytes (codec.go:42-76)."""
        from .chunk import make_cols
        lib = self.ctx.lib
        keep = []
        cols = make_cols(chk.columns, keep)
        n = chk.NumRows()
        need = C.c_int64(0)
        _lib.check(lib.tsq_chunk_encode(self.ctx.h, cols, len(chk.columns), n, None, 0, 0, C.byref(need)), self.ctx.h)
        out = np.zeros(need.value + 8, np.uint8)
        _lib.check(lib.tsq_chunk_encode(self.ctx.h, cols, len(chk.columns), n, out.ctypes.data_as(C.c_void_p), need.value, 0, C.byref(need)), self.ctx.h)
        return out[:need.value].tobytes()

    def DecodeToChunk(self, buffer, chk):
        """fills the (emptied) WireChunk from the buffer's first len(colTypes) columns; returns the remained bytes (codec.go:88-93)."""
        chk.Reset()
        used = _decode_window(self.ctx, buffer, self.colTypes, 0, 1 << 40, chk)[1]
        return buffer[used:]

    def Decode(self, buffer):
        chk = WireChunk(self.colTypes)
        return chk, self.DecodeToChunk(buffer, chk)


class _WireBuf:
    """the wire bytes as the C-ABI takes them: host memory, or (dev=True) uploaded to HBM once — what a Decoder does with a response
    it decodes window by window (a host buffer would be staged again for every window)"""

    def __init__(self, ctx, buffer, dev=False):
        self.ctx, self.n = ctx, len(buffer)
        self.raw = np.frombuffer(bytes(buffer) + b"\0" * 8, np.uint8)
        self.dptr, self.flags = None, 0
        if dev and self.n:
            self.dptr = ctx.alloc(self.n + 64)
            ctx.h2d(self.dptr, self.raw[:self.n])
            self.flags = abi.COL_DEVICE

    def ptr(self):
        return C.c_void_p(self.dptr) if self.dptr else self.raw.ctypes.data_as(C.c_void_p)

    def free(self):
        if self.dptr:
            self.ctx.free(self.dptr)
            self.dptr = None


def _peek(ctx, wb, colTypes, first, max_rows):
    tp = _types_arr(colTypes)
    total, take, used = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    nbytes = (C.c_int64 * len(colTypes))()
    _lib.check(ctx.lib.tsq_chunk_decode_peek(ctx.h, wb.ptr(), wb.n, wb.flags, tp, len(colTypes), first, max_rows, C.byref(total),
                                             C.byref(take), nbytes, C.byref(used)), ctx.h)
    return total.value, take.value, list(nbytes), used.value


def _decode_window(ctx, buffer, colTypes, first, max_rows, chk):
    """appends rows [first, first + max_rows) of the wire chunk in `buffer` (bytes, or a _WireBuf) to chk; returns (rows appended,
    bytes of the wire chunk)."""
    wb = buffer if isinstance(buffer, _WireBuf) else _WireBuf(ctx, buffer)
    total, take, nbytes, used = _peek(ctx, wb, colTypes, first, max_rows)
    for c, nb in zip(chk.columns, nbytes):
        c.reserve(c.length + take, c.data_bytes() + (nb if c.tp == abi.BYTES else take * elem_size(c.tp)))
    keep = []
    out = chk._out_cols(keep)
    n, u = C.c_int64(0), C.c_int64(0)
    _lib.check(ctx.lib.tsq_chunk_decode(ctx.h, wb.ptr(), wb.n, wb.flags, _types_arr(colTypes), len(colTypes), first, max_rows, out,
                                        C.byref(n), C.byref(u)), ctx.h)
    for i, c in enumerate(chk.columns):
        c.length = out[i].length
    return n.value, u.value


class Decoder:
    """codec.go:233-353.  The intermediate chunk is the wire buffer itself (Reset keeps it; decoding a window of it is the copy the
    reference does from intermChk), `intermChk` is only used by ReuseIntermChk."""

    def __init__(self, ctx, chk, colTypes):
        self.ctx = ctx
        self.intermChk = chk
        self.colTypes = list(colTypes)
        self.remainedRows = 0
        self._data = None
        self._next = 0

    def Reset(self, data):
        """the response's bytes go to HBM once; every Decode / ReuseIntermChk then copies a window of them (codec.go:272-275)"""
        self.Close()
        self._data = _WireBuf(self.ctx, data, dev=True)
        self.remainedRows = _peek(self.ctx, self._data, self.colTypes, 0, 0)[0]
        self._next = 0

    def Close(self):
        if self._data is not None:
            self._data.free()
            self._data = None

    def __del__(self):
        try:
            self.Close()
        except Exception:
            pass

    def Decode(self, chk):
        requiredRows = chk.RequiredRows() - chk.NumRows()
        requiredRows = (requiredRows + 7) >> 3 << 3  # a multiple of 8 (codec.go:259)
        requiredRows = min(requiredRows, self.remainedRows)
        n, _ = _decode_window(self.ctx, self._data, self.colTypes, self._next, requiredRows, chk)
        assert n == requiredRows
        self._next += requiredRows
        self.remainedRows -= requiredRows

    def IsFinished(self):
        return self.remainedRows == 0

    def RemainedRows(self):
        return self.remainedRows

    def ReuseIntermChk(self, chk):
        """chk takes the remaining rows without a second copy (codec.go:291-308): they are decoded straight into the emptied
        intermediate chunk, which is swapped with chk."""
        self.intermChk.Reset()
        _decode_window(self.ctx, self._data, self.colTypes, self._next, self.remainedRows, self.intermChk)
        chk.SwapColumns(self.intermChk)
        self._next += self.remainedRows
        self.remainedRows = 0
