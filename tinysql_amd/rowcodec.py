"""Mirror of util/rowcodec's chunk decoder for the harness: the KV values of a table scan (rowcodec v2 rows) -> chunk columns,
decoded on the GPU by libtsq (`tsq_rowcodec_decode`, SURVEY.md §8 f rank 4).

`ColInfo` / `NewChunkDecoder(columns, handleColID, defDatum)` keep the reference's shape (util/rowcodec/decoder.go:45-55,
:146-156).  The reference decodes ONE row per `DecodeToChunk(rowData, handle, chk)` call; a GPU wants the whole scan batch, so
`DecodeToChunk` here takes the rows of a batch (`values` back to back + `offsets`, `handles`) and returns the chunk — row for
row what the loop over the reference's method appends.  Errors are the reference's ("invalid codec version", "insufficient
bytes to decode value"; rows on which the reference would panic give "malformed row") and surface as `_lib.TsqError`; a
var-len column type raises with TSQ_ERR_UNSUPPORTED (that scan keeps the Go decoder).  No CPU fallback.
"""
import ctypes as C
import struct

import numpy as np

from . import _abi as abi
from . import _lib
from .chunk import chunk_from_buffers, out_buffers

# mysql type codes the decoder switches on (parser/mysql/type.go; decoder.go:201-236)
TypeTiny, TypeShort, TypeLong, TypeFloat, TypeDouble, TypeLonglong, TypeInt24, TypeYear = 1, 2, 3, 4, 5, 8, 9, 13
TypeVarchar, TypeBit, TypeBlob, TypeVarString, TypeString = 15, 16, 252, 253, 254
UnsignedFlag = 32  # parser/mysql/const.go
_INT_TYPES = (TypeLonglong, TypeLong, TypeInt24, TypeShort, TypeTiny, TypeYear)
TypeTinyBlob, TypeMediumBlob, TypeLongBlob = 249, 250, 251
_STRING_TYPES = (TypeVarString, TypeVarchar, TypeString, TypeBlob, TypeTinyBlob, TypeMediumBlob, TypeLongBlob)


class ColInfo:
    """rowcodec.ColInfo (decoder.go:45-55)."""

    def __init__(self, ID, Tp, Flag=0, IsPKHandle=False, Flen=0):
        self.ID, self.Tp, self.Flag, self.IsPKHandle, self.Flen = ID, Tp, Flag, IsPKHandle, Flen

    def tsq_type(self):
        if self.Tp in _INT_TYPES:
            return abi.U64 if self.Flag & UnsignedFlag else abi.I64
        if self.Tp == TypeFloat:
            return abi.F32
        if self.Tp == TypeDouble:
            return abi.F64
        if self.Tp in _STRING_TYPES:
            return abi.BYTES  # chk.AppendBytes of the value (decoder.go:226-228)
        if self.Tp == TypeBit and 1 <= self.Flen <= 64:
            return abi.BYTES  # a binary literal of (Flen + 7) / 8 bytes built from the stored uint (decoder.go:229-231)
        return None  # the types TinySQL does not have: the Go decoder


def _def_bits(tp, v):
    if tp == abi.F64:
        return struct.unpack("<Q", struct.pack("<d", float(v)))[0]
    if tp == abi.F32:
        return struct.unpack("<I", struct.pack("<f", float(v)))[0]
    return int(v) & ((1 << 64) - 1)


class ChunkDecoder:
    """rowcodec.ChunkDecoder (decoder.go:140-156) over libtsq."""

    def __init__(self, ctx, columns, handleColID=-1, defDatum=None):
        self.ctx, self.columns, self.handleColID, self.defDatum = ctx, list(columns), handleColID, defDatum
        self.types = [c.tsq_type() for c in self.columns]
        if any(t is None for t in self.types):
            raise _lib.TsqError(abi.ERR_UNSUPPORTED, "unknown type")  # decodeColToChunk's default branch: the Go decoder
        self.cols = (abi.RowcodecCol * len(self.columns))()
        self._keep, self.def_len = [], [0] * len(self.columns)
        for i, c in enumerate(self.columns):
            self.cols[i].col_id, self.cols[i].type, self.cols[i].flags, self.cols[i].def_bits = c.ID, self.types[i], 0, 0
            bit = 0
            if c.Tp == TypeBit:
                bit = abi.RC_BIT | (((c.Flen + 7) >> 3) << 8)
                self.cols[i].flags = bit
            if c.ID == handleColID:  # decoder.go:165
                self.cols[i].flags = abi.RC_HANDLE
            elif defDatum is not None:
                d = defDatum(i)  # a NULL default datum is the same as no default (AppendDatum of a NULL datum appends NULL)
                if d is not None:
                    self.cols[i].flags = abi.RC_HAS_DEFAULT | bit
                    if self.types[i] == abi.BYTES:  # a string default: its bytes (chk.AppendDatum -> AppendBytes)
                        b = d.encode() if isinstance(d, str) else bytes(d)
                        buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b or b"\0")
                        self._keep.append(buf)
                        self.cols[i].def_bytes, self.cols[i].def_len = C.cast(buf, C.c_void_p), len(b)
                        self.def_len[i] = len(b)
                    else:
                        self.cols[i].def_bits = _def_bits(self.types[i], d)

    def DecodeToChunk(self, values, offsets, handles=None):
        """values: bytes / np.uint8 (the rows back to back); offsets: n+1 row boundaries; handles: n int64 or None."""
        raw = np.frombuffer(values, dtype=np.uint8) if isinstance(values, (bytes, bytearray)) else np.ascontiguousarray(values, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offs) - 1
        hd = np.ascontiguousarray(handles, dtype=np.int64) if handles is not None else None
        keep = []
        # a var-len cell is a piece of its row (or the column's default string): the column's data cannot exceed the rows' bytes + n defaults
        out, bufs = out_buffers(self.types, max(n, 1), keep, var_bytes=[raw.size + n * self.def_len[i] if t == abi.BYTES else 0 for i, t in enumerate(self.types)])
        got = C.c_int64(0)
        _lib.check(self.ctx.lib.tsq_rowcodec_decode(self.ctx.h, raw.ctypes.data_as(C.c_void_p), raw.size, offs.ctypes.data_as(C.c_void_p),
                                                    hd.ctypes.data_as(C.c_void_p) if hd is not None else None, n, 0, len(self.columns), self.cols, out,
                                                    C.byref(got)), self.ctx.h)
        return chunk_from_buffers(self.types, bufs, got.value)


def NewChunkDecoder(ctx, columns, handleColID=-1, defDatum=None):
    """rowcodec.NewChunkDecoder(columns, handleColID, defDatum, loc) (decoder.go:146-156); loc is unused without time types."""
    return ChunkDecoder(ctx, columns, handleColID, defDatum)
