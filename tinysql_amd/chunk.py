"""numpy-backed mirror of util/chunk (the Go caller's side of the column ABI).

`Column` mirrors util/chunk/column.go:28-34 (fixed-width data + null bitmap with bit==1 meaning
NOT NULL, LSB first); `Chunk` mirrors util/chunk/chunk.go:31-46 (columns + optional sel).
Only the harness uses this module: tests, bench and the Python `Executor` mirrors.
"""
import ctypes as C

import numpy as np

from . import _abi as abi

_NP = {abi.I64: np.int64, abi.U64: np.uint64, abi.F32: np.float32, abi.F64: np.float64}


def np_dtype(tp):
    return _NP[tp]


def elem_size(tp):
    return 4 if tp == abi.F32 else 8


def pack_bitmap(notnull):
    """bool/uint8 array (1 = NOT NULL) -> column.go bitmap bytes (LSB first)."""
    return np.packbits(np.asarray(notnull, dtype=np.uint8), bitorder="little")


def unpack_bitmap(bitmap, n):
    return np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")[:n].astype(bool)


class Column:
    """One column of a chunk.  `notnull` is None when the column has no NULLs."""

    def __init__(self, tp, data, notnull=None):
        self.tp = tp
        self.data = np.ascontiguousarray(data, dtype=_NP[tp])
        if notnull is not None:
            notnull = np.asarray(notnull, dtype=bool)
            if notnull.all():
                notnull = None
        self.notnull = notnull
        if self.notnull is not None:
            # NULL slots hold zero bytes (column.go:150-158 AppendNull)
            self.data = self.data.copy()
            self.data[~self.notnull] = 0
        self._bitmap = None

    def __len__(self):
        return len(self.data)

    def IsNull(self, i):
        return self.notnull is not None and not self.notnull[i]

    def bitmap(self):
        if self.notnull is None:
            return None
        if self._bitmap is None:
            bm = pack_bitmap(self.notnull)
            self._bitmap = np.concatenate([bm, np.zeros(8, np.uint8)])  # slack for word reads
        return self._bitmap

    def as_col(self, keep):
        """tsq_col view (host pointers).  `keep` collects references that must outlive the call."""
        c = abi.Col()
        c.data = self.data.ctypes.data_as(C.c_void_p)
        bm = self.bitmap()
        c.null_bitmap = bm.ctypes.data_as(C.c_void_p) if bm is not None else None
        c.offsets = None
        c.length = len(self.data)
        c.elem_size = elem_size(self.tp)
        c.type = self.tp
        c.flags = 0
        keep.append(self.data)
        keep.append(bm)
        return c

    def slice(self, lo, hi):
        return Column(self.tp, self.data[lo:hi], None if self.notnull is None else self.notnull[lo:hi])

    def values(self):
        """python list with None for NULL (tests)."""
        out = self.data.tolist()
        if self.notnull is not None:
            for i in np.nonzero(~self.notnull)[0]:
                out[i] = None
        return out


class StrColumn:
    """A var-len column (util/chunk/column.go:28-34: offsets[n + 1] + concatenated data bytes, binary strings).
    `values`: list of bytes / str (utf-8 encoded) / None (NULL)."""

    tp = abi.BYTES

    def __init__(self, values):
        vals = [None if v is None else (v.encode() if isinstance(v, str) else bytes(v)) for v in values]
        self._vals = vals
        self.notnull = None if all(v is not None for v in vals) else np.array([v is not None for v in vals], dtype=bool)
        lens = np.array([0 if v is None else len(v) for v in vals], dtype=np.int64)  # a NULL cell has no bytes (AppendNull)
        self.offsets = np.zeros(len(vals) + 1, dtype=np.int64)
        np.cumsum(lens, out=self.offsets[1:])
        self.data = np.frombuffer(b"".join(v for v in vals if v is not None) + b"\0" * 8, dtype=np.uint8).copy()
        self._bitmap = None

    def __len__(self):
        return len(self._vals)

    def IsNull(self, i):
        return self._vals[i] is None

    def bitmap(self):
        if self.notnull is None:
            return None
        if self._bitmap is None:
            self._bitmap = np.concatenate([pack_bitmap(self.notnull), np.zeros(8, np.uint8)])
        return self._bitmap

    def as_col(self, keep):
        c = abi.Col()
        c.data = self.data.ctypes.data_as(C.c_void_p)
        bm = self.bitmap()
        c.null_bitmap = bm.ctypes.data_as(C.c_void_p) if bm is not None else None
        c.offsets = self.offsets.ctypes.data_as(C.c_void_p)
        c.length = len(self._vals)
        c.elem_size = -1
        c.type = abi.BYTES
        c.flags = 0
        keep += [self.data, bm, self.offsets]
        return c

    def slice(self, lo, hi):
        return StrColumn(self._vals[lo:hi])

    def values(self):
        return list(self._vals)


def make_cols(columns, keep):
    arr = (abi.Col * len(columns))()
    for i, c in enumerate(columns):
        arr[i] = c.as_col(keep)
    return arr


class Chunk:
    """util/chunk.Chunk: columns + optional selection vector."""

    def __init__(self, columns, sel=None):
        self.columns = list(columns)
        self.sel = None if sel is None else np.ascontiguousarray(sel, dtype=np.int32)

    def NumCols(self):
        return len(self.columns)

    def NumRows(self):  # chunk.go:308-316
        if self.sel is not None:
            return len(self.sel)
        return len(self.columns[0]) if self.columns else 0

    def types(self):
        return [c.tp for c in self.columns]

    def slice(self, lo, hi):
        assert self.sel is None
        return Chunk([c.slice(lo, hi) for c in self.columns])

    def rows(self):
        """list of row tuples honouring sel (tests; Result.Check-style comparisons)."""
        cols = [c.values() for c in self.columns]
        idx = range(len(self.columns[0])) if self.sel is None else self.sel.tolist()
        return [tuple(col[i] for col in cols) for i in idx]


def out_buffers(types, cap, keep, var_bytes=None):
    """allocates output columns for a pull of up to `cap` rows; returns (tsq_col array, [(data, bitmap[, offsets])]).
    var_bytes[i] = data bytes of var-len column i for this pull (tsq_join_peek)."""
    arr = (abi.Col * len(types))()
    bufs = []
    for i, tp in enumerate(types):
        bm = np.zeros((cap + 7) // 8 + 8, dtype=np.uint8)
        if tp == abi.BYTES:
            data = np.zeros((var_bytes[i] if var_bytes is not None else 0) + 8, dtype=np.uint8)
            offs = np.zeros(cap + 1, dtype=np.int64)
            arr[i].offsets = offs.ctypes.data_as(C.c_void_p)
            arr[i].elem_size = -1
            bufs.append((data, bm, offs))
        else:
            data = np.zeros(cap, dtype=_NP[tp])
            arr[i].elem_size = elem_size(tp)
            bufs.append((data, bm))
        arr[i].data = data.ctypes.data_as(C.c_void_p)
        arr[i].null_bitmap = bm.ctypes.data_as(C.c_void_p)
        arr[i].length = cap
        arr[i].type = tp
        arr[i].flags = 0
    keep.append(bufs)
    return arr, bufs


def chunk_from_buffers(types, bufs, n):
    cols = []
    for tp, b in zip(types, bufs):
        if tp == abi.BYTES:
            data, bm, offs = b
            nn = unpack_bitmap(bm, n)
            raw = data.tobytes()
            cols.append(StrColumn([raw[offs[i]:offs[i + 1]] if nn[i] else None for i in range(n)]))
        else:
            cols.append(Column(tp, b[0][:n].copy(), unpack_bitmap(b[1], n)))
    return Chunk(cols)


def concat(chunks, types):
    if not chunks:
        return Chunk([StrColumn([]) if tp == abi.BYTES else Column(tp, np.zeros(0, _NP[tp])) for tp in types])
    cols = []
    for i, tp in enumerate(types):
        if tp == abi.BYTES:
            cols.append(StrColumn([v for c in chunks for v in c.columns[i].values()]))
            continue
        data = np.concatenate([c.columns[i].data for c in chunks])
        if any(c.columns[i].notnull is not None for c in chunks):
            nn = np.concatenate([
                c.columns[i].notnull if c.columns[i].notnull is not None else np.ones(len(c.columns[i]), bool)
                for c in chunks])
        else:
            nn = None
        cols.append(Column(tp, data, nn))
    return Chunk(cols)
