"""Mirror of distsql.selectResult's decode step (distsql/select_result.go:102-155) for the harness: the RowsData byte
string of a coprocessor response -> chunk columns, decoded on the GPU by libtsq (`tsq_rows_decode`, SURVEY.md §8 f rank 2).

`SelectResult.Next` keeps the reference's contract: it fills a chunk with at most `max_rows` rows and keeps the
undecoded remainder of the response for the next call (select_result.go:153).  Errors are the reference's
(`codec.Decoder.DecodeOne`, util/codec/codec.go:623-690) and surface as `_lib.TsqError` with the same message.
"""
import ctypes as C

import numpy as np

from . import _abi as abi
from . import _lib
from .chunk import Chunk, Column, chunk_from_buffers, concat, np_dtype, out_buffers

ROWS_PER_CHUNK = 64  # store/mockstore/mocktikv/cop_handler_dag.go:510


def decode_rows(ctx, data, types, cap_rows):
    """host bytes -> (host Chunk, bytes consumed).  data: bytes / np.uint8 array."""
    raw = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data, dtype=np.uint8)
    keep = []
    out, bufs = out_buffers(types, max(cap_rows, 1), keep)
    tp = (C.c_int32 * len(types))(*types)
    n, used = C.c_int64(0), C.c_int64(0)
    _lib.check(ctx.lib.tsq_rows_decode(ctx.h, raw.ctypes.data_as(C.c_void_p), raw.size, 0, len(types), tp, out, cap_rows, C.byref(n), C.byref(used)), ctx.h)
    return chunk_from_buffers(types, bufs, n.value), used.value


def decode_chunks(ctx, chunks, types, cap_rows=None):
    """the RowsData byte strings of a response's tipb.Chunks -> host Chunk, decoded on the GPU by `tsq_rows_decode_chunks` (one lane
    per chunk; bytes datums become var-len cells).  cap_rows: rows the output buffers hold (default: 64 per chunk, the storage side's
    cut, cop_handler_dag.go:510; the call is repeated with the reported size if a chunk holds more)."""
    parts = [np.frombuffer(c, dtype=np.uint8) if isinstance(c, (bytes, bytearray)) else np.asarray(c, dtype=np.uint8) for c in chunks]
    raw = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    offs = np.zeros(len(parts) + 1, np.int64)
    np.cumsum([p.size for p in parts], out=offs[1:])
    cap = ROWS_PER_CHUNK * max(len(parts), 1) if cap_rows is None else cap_rows
    tp = (C.c_int32 * len(types))(*types)
    buf = raw if raw.size else np.zeros(1, np.uint8)
    while True:
        keep = []
        out, bufs = out_buffers(types, max(cap, 1), keep, var_bytes=[raw.size if t == abi.BYTES else 0 for t in types])
        n = C.c_int64(0)
        st = ctx.lib.tsq_rows_decode_chunks(ctx.h, buf.ctypes.data_as(C.c_void_p), raw.size, offs.ctypes.data_as(C.c_void_p), len(parts), 0, len(types), tp, out, cap,
                                            C.byref(n))
        if st == abi.ERR_INVALID and n.value > cap:
            cap = n.value  # a chunk with more than 64 rows: size the buffers from the count the library reported
            continue
        _lib.check(st, ctx.h)
        return chunk_from_buffers(types, bufs, n.value)


class SelectResult:
    """distsql.selectResult over a list of response chunks (each a RowsData byte string)."""

    def __init__(self, ctx, responses, types):
        self.ctx, self.types = ctx, list(types)
        self.responses = [np.frombuffer(r, dtype=np.uint8) if isinstance(r, (bytes, bytearray)) else np.asarray(r, dtype=np.uint8) for r in responses]
        self.idx = 0

    def Next(self, max_rows=1024):
        """select_result.go:102-128: decode until the chunk is full or the responses are used up; empty chunk = EOS."""
        if abi.BYTES in self.types:
            return self._next_varlen(max_rows)
        got = []
        want = max_rows
        while want > 0 and self.idx < len(self.responses):
            raw = self.responses[self.idx]
            if raw.size == 0:
                self.idx += 1
                continue
            chk, used = decode_rows(self.ctx, raw, self.types, want)
            self.responses[self.idx] = raw[used:]
            got.append(chk)
            want -= chk.NumRows()
        return concat(got, self.types) if got else Chunk([Column(t, np.zeros(0, np_dtype(t))) for t in self.types])


def _next_varlen(self, max_rows):
    """schemas with var-len columns: all chunks are decoded at once (tsq_rows_decode_chunks) the first time, then handed out in
    pieces of max_rows rows — the reference's contract (a chunk is filled up to its capacity, the rest waits) without re-parsing."""
    if getattr(self, "_all", None) is None:
        self._all, self._pos = decode_chunks(self.ctx, self.responses, self.types), 0
        self.idx = len(self.responses)
    lo, hi = self._pos, min(self._all.NumRows(), self._pos + max_rows)
    self._pos = hi
    return self._all.slice(lo, hi)


SelectResult._next_varlen = _next_varlen


def encode_rows(ctx, chunk, comparable_cols=()):
    """The storage side's inverse of decode_rows: the rows of a chunk (var-len columns as compact-bytes datums) -> (RowsData bytes, row offsets[n + 1]), encoded
    on the GPU by libtsq (`tsq_rows_encode`; codec.EncodeValue per value, util/codec/codec.go:74-99,205-209).  Columns listed in
    `comparable_cols` use the EncodeKey form (the handle column of a table scan, util/rowcodec/decoder.go:263-273)."""
    from .chunk import make_cols
    n = chunk.NumRows()
    keep = []
    cols = make_cols(chunk.columns, keep)
    flags = (C.c_uint32 * len(chunk.columns))(*[abi.ENC_COMPARABLE if i in comparable_cols else 0 for i in range(len(chunk.columns))])
    # 11 bytes bound every fixed-width datum; a string cell adds its bytes to its (<= 11-byte) header
    # (the memcomparable form: 9 bytes per 8 + one more group)
    var = sum(int(c.offsets[-1] - c.offsets[0]) for c in chunk.columns if c.tp == abi.BYTES and len(c.offsets))
    cap = n * len(chunk.columns) * 11 + 16 + var + (var // 8 + n * len(chunk.columns))
    out = np.zeros(cap, np.uint8)
    offs = np.zeros(n + 1, np.int64)
    got = C.c_int64(0)
    _lib.check(ctx.lib.tsq_rows_encode(ctx.h, cols, len(chunk.columns), flags, n, out.ctypes.data_as(C.c_void_p), cap, 0, offs.ctypes.data_as(C.c_void_p),
                                       C.byref(got)), ctx.h)
    return out[:got.value].copy(), offs




def response_chunks(raw, offsets):
    """fillUpData4SelectResponse / appendRow (cop_handler_dag.go:414-425, :512-519): the byte string cut into tipb.Chunk.RowsData
    pieces of 64 rows."""
    n = len(offsets) - 1
    return [bytes(raw[offsets[lo]:offsets[min(lo + ROWS_PER_CHUNK, n)]]) for lo in range(0, n, ROWS_PER_CHUNK)]
