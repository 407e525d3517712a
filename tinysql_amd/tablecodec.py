"""Mirror of package tablecodec's record-key functions for the harness (tablecodec/tablecodec.go): the keys of a table scan
<-> handles, computed on the GPU by libtsq (`tsq_rowkeys_decode` / `tsq_rowkeys_encode`, SURVEY.md §8 f rank 4).

    record key = 't' | EncodeInt(tableID) | "_r" | EncodeInt(handle)        RecordRowKeyLen = 19 bytes

The batch forms are what a scan uses: mocktikv's tableScanExec calls DecodeRowKey once per KV pair (store/mockstore/mocktikv/
executor.go:124-196); the handles then feed `rowcodec.ChunkDecoder` for the PK-handle column.  Errors are the reference's
("invalid key", tablecodec.go:237) and surface as `_lib.TsqError`.
"""
import ctypes as C

import numpy as np

from . import _abi as abi
from . import _lib

RecordRowKeyLen = 19  # tablecodec.go:39-44


def EncodeRowKeysWithHandles(ctx, tableID, handles):
    """EncodeRowKeyWithHandle (tablecodec.go:65-70) of every handle: np.uint8 array of 19 * n bytes."""
    h = np.ascontiguousarray(handles, dtype=np.int64)
    out = np.zeros(max(h.size, 1) * RecordRowKeyLen, np.uint8)
    _lib.check(ctx.lib.tsq_rowkeys_encode(ctx.h, tableID, h.ctypes.data_as(C.c_void_p), h.size, 0, out.ctypes.data_as(C.c_void_p)), ctx.h)
    return out[:h.size * RecordRowKeyLen]


def EncodeRowKeyWithHandle(ctx, tableID, handle):
    return bytes(EncodeRowKeysWithHandles(ctx, tableID, [handle]))


def DecodeRowKeys(ctx, keys, key_offsets=None, want_table_ids=False):
    """DecodeRowKey (tablecodec.go:235-242) of every key of `keys` (bytes / np.uint8; 19-byte keys back to back, or cut by
    key_offsets[n + 1]).  Returns handles (and the table ids DecodeRecordKey hands back).  The first key in scan order that is
    not a record key raises TsqError("invalid key") — the scan ends there, as tableScanExec returns the error."""
    raw = np.frombuffer(keys, dtype=np.uint8) if isinstance(keys, (bytes, bytearray)) else np.ascontiguousarray(keys, dtype=np.uint8)
    offs = None if key_offsets is None else np.ascontiguousarray(key_offsets, dtype=np.int64)
    if offs is None and raw.size % RecordRowKeyLen:
        raise _lib.TsqError(abi.ERR_INVALID, "invalid key")  # a trailing partial key: len(key) != RecordRowKeyLen
    n = raw.size // RecordRowKeyLen if offs is None else len(offs) - 1
    handles = np.zeros(max(n, 1), np.int64)
    tids = np.zeros(max(n, 1), np.int64) if want_table_ids else None
    got = C.c_int64(0)
    buf = raw if raw.size else np.zeros(1, np.uint8)
    _lib.check(ctx.lib.tsq_rowkeys_decode(ctx.h, buf.ctypes.data_as(C.c_void_p), raw.size, None if offs is None else offs.ctypes.data_as(C.c_void_p), n, 0,
                                          handles.ctypes.data_as(C.c_void_p), None if tids is None else tids.ctypes.data_as(C.c_void_p), C.byref(got)), ctx.h)
    return (handles[:n], tids[:n]) if want_table_ids else handles[:n]


def DecodeRowKey(ctx, key):
    return int(DecodeRowKeys(ctx, key, key_offsets=[0, len(key)])[0])


def DecodeRecordKey(ctx, key):
    """(tableID, handle) — tablecodec.go:73-77"""
    h, t = DecodeRowKeys(ctx, key, key_offsets=[0, len(key)], want_table_ids=True)
    return int(t[0]), int(h[0])
