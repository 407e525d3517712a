"""The storage side's coprocessor executors over libtsq — the harness mirror of store/mockstore/mocktikv's DAG handler
(cop_handler_dag.go:49-83 handleCopDAGRequest, :149-160 buildDAG), string columns included (SURVEY.md §8 f rank 4).

The reference runs the pushed-down plan row at a time over `[][]byte` datums: tableScanExec (executor.go:48-196: DecodeRowKey +
rowcodec.BytesDecoder.DecodeToBytes per KV pair) -> selectionExec (:322-390: decode the related columns, evalBool) -> hashAggExec
(aggregate.go:30-182: group key = EncodeValue of the group-by datums, aggregation.Aggregation.Update per row) | topNExec (:392-470 +
topn.go: a heap of `limit` rows) | limitExec (:472-507), then fillUpData4SelectResponse cuts the encoded rows into tipb.Chunks of 64
rows (cop_handler_dag.go:414-425, :510-519).  Here every executor is batch at a time and its chunks stay in HBM (gpu_pipeline.py's
device chunks); all compute runs in libtsq:
    tableScanExec  = tsq_rowkeys_decode (record keys -> handles) + tsq_rowcodec_decode (stored rows -> columns)
    indexScanExec  = tsq_indexkeys_decode (index keys [+ values] -> the index columns and the handle; memcomparable strings)
    selectionExec  = tsq_filter_eval + tsq_chunk_compact
    hashAggExec    = tsq_agg_* in the partial layout of the reference: per function its partial results (AVG: count, sum —
                     avg.go:78-81), then the group-by values (aggregate.go:96-113)
    topNExec       = tsq_sort_* with a row limit (radix select + sort of the candidates)
    limitExec      = the first `limit` rows
    response       = tsq_rows_encode of the requested output offsets + the 64-row cut, or (encodeType "chunk") one wire chunk per
                     batch (tsq_chunk_encode), read back by chunk.Decoder
Same names and argument meaning as the reference's executors; differences that follow from running in parallel are the ones
DESIGN.md lists (group order, FIRST_ROW of a column that is not functionally dependent on the group key, the running-sum overflow).
"""
import ctypes as C

import numpy as np

from . import _abi as abi
from . import _lib
from . import rowcodec as RC
from .executor import AggFuncDesc
from .gpu_pipeline import EOS, DeviceChunk, DeviceColumn, GpuExecutor, GpuHashAggExec, GpuSelectionExec, GpuSortExec

ROWS_PER_CHUNK = 64  # cop_handler_dag.go:510


class _DevBytes:
    """a byte / int64 array uploaded to HBM once"""

    def __init__(self, ctx, arr):
        self.ctx, self.n = ctx, arr.nbytes
        self.p = ctx.alloc(max(arr.nbytes, 1) + 64)
        if arr.nbytes:
            ctx.h2d(self.p, np.ascontiguousarray(arr))

    def free(self):
        if self.p:
            self.ctx.free(self.p)
            self.p = None


class tableScanExec(GpuExecutor):
    """mocktikv.tableScanExec (executor.go:48-196) over the KV pairs of its ranges, in scan order: `keys` = the record keys back to
    back (19 bytes each), `values` / `value_offsets` = the stored rows (rowcodec v2).  columns: rowcodec.ColInfo of the scan's
    tipb.ColumnInfos (cop_handler_dag.go:162-208); a PK-handle column takes its value from the key (decoder.go:263-273)."""

    def __init__(self, ctx, columns, keys, values, value_offsets, batch_rows=1 << 22):
        self.columns = list(columns)
        types = [c.tsq_type() for c in self.columns]
        if any(t is None for t in types):
            raise _lib.TsqError(abi.ERR_UNSUPPORTED, "unknown type")  # a type the decoder does not know: the Go executors
        super().__init__(ctx, types)
        self.raw_keys = np.frombuffer(keys, dtype=np.uint8) if isinstance(keys, (bytes, bytearray)) else np.ascontiguousarray(keys, dtype=np.uint8)
        self.raw_vals = np.frombuffer(values, dtype=np.uint8) if isinstance(values, (bytes, bytearray)) else np.ascontiguousarray(values, dtype=np.uint8)
        self.offs = np.ascontiguousarray(value_offsets, dtype=np.int64)
        self.n = len(self.offs) - 1
        if self.raw_keys.size != 19 * self.n:
            raise _lib.TsqError(abi.ERR_INVALID, "invalid key")
        self.batch = (batch_rows + 7) & ~7
        self.rc = (abi.RowcodecCol * len(self.columns))()
        for i, c in enumerate(self.columns):
            self.rc[i].col_id, self.rc[i].type, self.rc[i].def_bits = c.ID, types[i], 0
            self.rc[i].flags = abi.RC_HANDLE if c.IsPKHandle else 0
            if c.Tp == RC.TypeBit:  # the stored uint as a binary literal of (Flen + 7) / 8 bytes (decoder.go:229-231)
                self.rc[i].flags |= abi.RC_BIT | (((c.Flen + 7) >> 3) << 8)
        self.dk = self.dv = self.do = self.dh = None
        self.out, self.pos, self.count = None, 0, 0

    def Open(self):
        ctx = self.ctx
        self.dk, self.dv, self.do = _DevBytes(ctx, self.raw_keys), _DevBytes(ctx, self.raw_vals), _DevBytes(ctx, self.offs)
        self.dh = ctx.alloc(max(self.n, 1) * 8 + 64)
        # a string cell is a piece of its row: a var-len column of a batch cannot hold more bytes than the scanned values
        self.out = self._buffers(min(self.batch, max(self.n, 8)), var_bytes=[self.raw_vals.size] * len(self.types))
        self.pos = self.count = 0
        # DecodeRowKey of every pair (one launch for the whole scan: 27 bytes of traffic per row)
        got = C.c_int64(0)
        if self.n:
            _lib.check(self.lib.tsq_rowkeys_decode(ctx.h, C.c_void_p(self.dk.p), self.raw_keys.size, None, self.n, abi.COL_DEVICE, C.c_void_p(self.dh), None,
                                                   C.byref(got)), ctx.h)

    def Next(self):
        if self.pos >= self.n:
            return EOS
        lo, hi = self.pos, min(self.n, self.pos + self.batch)
        oc = (abi.Col * len(self.out))(*[c.col(hi - lo) for c in self.out])
        got = C.c_int64(0)
        # rows [lo, hi): their boundaries are offsets[lo .. hi] (absolute byte positions into `values`)
        _lib.check(self.lib.tsq_rowcodec_decode(self.ctx.h, C.c_void_p(self.dv.p), self.raw_vals.size, C.c_void_p(self.do.p + 8 * lo),
                                                C.c_void_p(self.dh + 8 * lo), hi - lo, abi.COL_DEVICE, len(self.columns), self.rc, oc, C.byref(got)), self.ctx.h)
        self.pos = hi
        self.count += got.value
        return DeviceChunk(self.out, got.value)

    def Counts(self):
        return [self.count]  # one range: rows handed on (executor.go:76-85)

    def Close(self):
        for b in (self.dk, self.dv, self.do):
            if b:
                b.free()
        if self.dh:
            self.ctx.free(self.dh)
        if self.out:
            for c in self.out:
                c.free()
        self.dk = self.dv = self.do = self.dh = self.out = None


class indexScanExec(GpuExecutor):
    """mocktikv.indexScanExec (executor.go:191-320) over the KV pairs of its ranges, in scan order: every pair through
    tablecodec.DecodeIndexKV (tablecodec.go:376-434).  `keys` / `key_offsets` = the index keys back to back; `values` /
    `value_offsets` = the pairs' values (the handle of a unique index, 8 bytes big endian); `types` = the index columns' types
    (+ the handle column, I64 or U64, when pkStatus != PrimaryKeyNotExists); colsLen = len(IndexScan.Columns) without the handle."""

    PrimaryKeyNotExists, PrimaryKeyIsSigned, PrimaryKeyIsUnsigned = 0, 1, 2  # tablecodec.go:394-403

    def __init__(self, ctx, types, colsLen, pkStatus, keys, key_offsets, values=b"", value_offsets=None, batch_rows=1 << 22):
        super().__init__(ctx, list(types))
        assert len(self.types) == colsLen + (1 if pkStatus else 0)
        self.colsLen, self.pkStatus = colsLen, pkStatus
        self.raw_keys = np.frombuffer(bytes(keys), dtype=np.uint8) if isinstance(keys, (bytes, bytearray)) else np.ascontiguousarray(keys, dtype=np.uint8)
        self.koffs = np.ascontiguousarray(key_offsets, dtype=np.int64)
        self.n = len(self.koffs) - 1
        self.raw_vals = np.frombuffer(bytes(values), dtype=np.uint8) if isinstance(values, (bytes, bytearray)) else np.ascontiguousarray(values, dtype=np.uint8)
        self.voffs = None if value_offsets is None else np.ascontiguousarray(value_offsets, dtype=np.int64)
        self.batch = batch_rows
        self.dk = self.dko = self.dv = self.dvo = None
        self.out, self.pos, self.count = None, 0, 0

    def Open(self):
        ctx = self.ctx
        self.dk, self.dko = _DevBytes(ctx, self.raw_keys), _DevBytes(ctx, self.koffs)
        if self.voffs is not None:
            self.dv, self.dvo = _DevBytes(ctx, self.raw_vals), _DevBytes(ctx, self.voffs)
        self.out = self._buffers(min(self.batch, max(self.n, 8)), var_bytes=[self.raw_keys.size] * len(self.types))  # a string cell is a piece of its key
        self.pos = self.count = 0

    def Next(self):
        if self.pos >= self.n:
            return EOS
        lo, hi = self.pos, min(self.n, self.pos + self.batch)
        oc = (abi.Col * len(self.out))(*[c.col(hi - lo) for c in self.out])
        tp = (C.c_int32 * len(self.types))(*self.types)
        got = C.c_int64(0)
        # pairs [lo, hi): their boundaries are key_offsets[lo .. hi] / value_offsets[lo .. hi] (absolute byte positions)
        _lib.check(self.lib.tsq_indexkeys_decode(self.ctx.h, C.c_void_p(self.dk.p), self.raw_keys.size, C.c_void_p(self.dko.p + 8 * lo), hi - lo,
                                                 C.c_void_p(self.dv.p) if self.dv else None, self.raw_vals.size if self.dv else 0,
                                                 C.c_void_p(self.dvo.p + 8 * lo) if self.dvo else None, abi.COL_DEVICE, self.colsLen, tp, self.pkStatus, oc,
                                                 C.byref(got)), self.ctx.h)
        self.pos = hi
        self.count += got.value
        return DeviceChunk(self.out, got.value)

    def Counts(self):
        return [self.count]

    def Close(self):
        for b in (self.dk, self.dko, self.dv, self.dvo):
            if b:
                b.free()
        if self.out:
            for c in self.out:
                c.free()
        self.dk = self.dko = self.dv = self.dvo = self.out = None


class selectionExec(GpuSelectionExec):
    """mocktikv.selectionExec (executor.go:322-390): rows for which every condition is true (evalBool: NULL and 0 drop the row)."""

    def __init__(self, ctx, src, conditions):
        super().__init__(ctx, src, conditions)


class hashAggExec(GpuHashAggExec):
    """mocktikv.hashAggExec (aggregate.go:30-182).  aggFuncs: [(TSQ_AGG_*, argument column | -1 for the constant of COUNT(*))];
    groupByCols: bare columns (group-by EXPRESSIONS are evaluated by a projection below, as the planner does for the SQL side).
    Output row = partial results of every function in order, then the group-by values."""

    def __init__(self, ctx, src, aggFuncs, groupByCols, est_groups=0):
        in_types = src.Schema()
        descs = []
        for func, arg in aggFuncs:
            at = in_types[arg] if arg >= 0 else abi.I64
            # only AVG has a partial form that differs from its result (count, sum)
            descs.append(AggFuncDesc(func, arg, at, abi.MODE_PARTIAL1 if func == abi.AGG_AVG else abi.MODE_COMPLETE))
        for g in groupByCols:
            descs.append(AggFuncDesc(abi.AGG_FIRSTROW, g, in_types[g]))  # the group-by datum of the row that made the group
        super().__init__(ctx, src, list(groupByCols), descs, est_groups=est_groups)


class topNExec(GpuSortExec):
    """mocktikv.topNExec (executor.go:392-470, topn.go): the `limit` smallest rows under the ORDER BY items, in that order."""

    def __init__(self, ctx, src, orderByCols, desc, limit):
        super().__init__(ctx, src, list(orderByCols), list(desc), offset=0, count=limit)


class limitExec(GpuExecutor):
    """mocktikv.limitExec (executor.go:472-507): the first `limit` rows of its source, in the source's order."""

    def __init__(self, ctx, src, limit):
        super().__init__(ctx, src.Schema(), (src,))
        self.src, self.limit, self.cursor = src, int(limit), 0

    def Open(self):
        super().Open()
        self.cursor = 0

    def Next(self):
        if self.cursor >= self.limit:
            return EOS
        chk = self.src.Next()
        n = min(chk.NumRows(), self.limit - self.cursor)
        self.cursor += n
        return DeviceChunk(chk.columns, n) if n else EOS


def fillUpData4SelectResponse(ctx, chunk, outputOffsets, chunks=None, rowCnt=0):
    """cop_handler_dag.go:414-425 + appendRow :512-519: the requested columns of every row, encoded value by value
    (codec.EncodeValue) on the GPU, appended to `chunks` — RowsData pieces of 64 rows, a new piece whenever the running row count
    is a multiple of 64, whatever batches the rows arrive in.  chunk: a DeviceChunk.  Returns (chunks, rowCnt)."""
    chunks = [] if chunks is None else chunks
    n = chunk.NumRows()
    if n == 0:
        return chunks, rowCnt
    cols = (abi.Col * len(outputOffsets))(*[chunk.columns[o].col(n) for o in outputOffsets])
    cap = n * len(outputOffsets) * 11 + 16
    got = C.c_int64(0)
    if any(chunk.columns[o].tp == abi.BYTES for o in outputOffsets):  # string cells: the size of the response is asked first
        st = ctx.lib.tsq_rows_encode(ctx.h, cols, len(outputOffsets), None, n, None, 0, abi.COL_DEVICE, None, C.byref(got))
        if st not in (abi.OK, abi.ERR_INVALID):
            _lib.check(st, ctx.h)
        cap = got.value + 16
    dout = ctx.alloc(cap + 64)
    doffs = ctx.alloc((n + 1) * 8 + 64)
    try:
        got = C.c_int64(0)
        _lib.check(ctx.lib.tsq_rows_encode(ctx.h, cols, len(outputOffsets), None, n, C.c_void_p(dout), cap, abi.COL_DEVICE, C.c_void_p(doffs), C.byref(got)), ctx.h)
        raw = np.zeros(max(got.value, 1), np.uint8)
        offs = np.zeros(n + 1, np.int64)
        ctx.d2h(raw, dout)
        ctx.d2h(offs, doffs)
    finally:
        ctx.free(dout)
        ctx.free(doffs)
    lo = 0
    while lo < n:
        room = ROWS_PER_CHUNK - rowCnt % ROWS_PER_CHUNK  # rows the open piece still takes (64 when a new one starts)
        hi = min(n, lo + room)
        piece = bytes(raw[offs[lo]:offs[hi]])
        if rowCnt % ROWS_PER_CHUNK == 0:
            chunks.append(piece)
        else:
            chunks[-1] += piece
        rowCnt += hi - lo
        lo = hi
    return chunks, rowCnt


def fillUpChunkResponse(ctx, chunk, outputOffsets, chunks=None):
    """The column-major answer (upstream's tipb.EncodeType_TypeChunk; TinySQL ships the codec, util/chunk/codec.go:28-143, and answers
    with datum rows): every batch of the DAG's output becomes ONE wire chunk — chunk.Codec.Encode of the requested columns on the GPU
    (tsq_chunk_encode).  The SQL side reads it back with chunk.Decoder (tinysql_amd/chunk_codec.py): a copy instead of a parse."""
    chunks = [] if chunks is None else chunks
    n = chunk.NumRows()
    if n == 0:
        return chunks
    cols = (abi.Col * len(outputOffsets))(*[chunk.columns[o].col(n) for o in outputOffsets])
    need = C.c_int64(0)
    _lib.check(ctx.lib.tsq_chunk_encode(ctx.h, cols, len(outputOffsets), n, None, 0, abi.COL_DEVICE, C.byref(need)), ctx.h)
    dout = ctx.alloc(need.value + 64)
    try:
        _lib.check(ctx.lib.tsq_chunk_encode(ctx.h, cols, len(outputOffsets), n, C.c_void_p(dout), need.value, abi.COL_DEVICE, C.byref(need)), ctx.h)
        raw = np.zeros(need.value, np.uint8)
        ctx.d2h(raw, dout)
    finally:
        ctx.free(dout)
    chunks.append(raw.tobytes())
    return chunks


class SelectResponse:
    """tipb.SelectResponse as far as this path fills it: Chunks (RowsData byte strings), OutputCounts, Error."""

    def __init__(self):
        self.Chunks, self.OutputCounts, self.Error = [], None, None


def buildDAG(ctx, executors, pairs):
    """cop_handler_dag.go:149-160: chain the executors, each taking the previous one as its source.  executors: a list of
    ("TableScan", [ColInfo...]) | ("Selection", [conditions]) | ("Aggregation", aggFuncs, groupByCols) |
    ("TopN", orderByCols, desc, limit) | ("Limit", n); pairs = (keys, values, value_offsets) of the scanned ranges."""
    src = None
    for e in executors:
        tp = e[0]
        if tp == "TableScan":
            cur = tableScanExec(ctx, e[1], *pairs)
        elif tp == "Selection":
            cur = selectionExec(ctx, src, e[1])
        elif tp == "Aggregation":
            cur = hashAggExec(ctx, src, e[1], e[2])
        elif tp == "TopN":
            cur = topNExec(ctx, src, e[1], e[2], e[3])
        elif tp == "Limit":
            cur = limitExec(ctx, src, e[1])
        else:
            raise _lib.TsqError(abi.ERR_UNSUPPORTED, "this exec type %s doesn't support yet." % tp)  # cop_handler_dag.go:141-144
        src = cur
    return src


def handleCopDAGRequest(ctx, executors, outputOffsets, pairs, encodeType="default"):
    """cop_handler_dag.go:49-83: run the DAG to its end, encode the rows, cut them into chunks.  An executor error becomes
    SelectResponse.Error (toPBError) and no chunks, like the reference.  encodeType "chunk": the response's Chunks are wire chunks
    (fillUpChunkResponse) instead of 64-row pieces of datum rows."""
    resp = SelectResponse()
    resp.EncodeType = encodeType
    e = buildDAG(ctx, executors, pairs)
    scan = e
    while scan.children:
        scan = scan.children[0]
    rows = 0
    try:
        e.Open()
        while True:
            chk = e.Next()
            if chk.NumRows() == 0:
                break
            if encodeType == "chunk":
                resp.Chunks = fillUpChunkResponse(ctx, chk, outputOffsets, resp.Chunks)
            else:
                resp.Chunks, rows = fillUpData4SelectResponse(ctx, chk, outputOffsets, resp.Chunks, rows)
        resp.OutputCounts = scan.Counts() if hasattr(scan, "Counts") else None
    except _lib.TsqError as err:
        resp.Chunks, resp.Error = [], err.message
    finally:
        e.Close()
    return resp
