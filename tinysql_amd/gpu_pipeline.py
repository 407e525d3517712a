"""Device-resident volcano operators: the same Open / Next / Close contract as `executor.py`
(executor/executor.go:146-162), but the chunks handed from child to parent stay in HBM.

SURVEY.md §8(f) rank 1: when parent and child are both GPU operators the D2H -> H2D hop of a 1024-row
host chunk is replaced by a pointer hand-off; `tidb_max_chunk_size` has no upper bound
(sessionctx/variable/varsutil.go:251), so a "chunk" here is millions of rows.  Everything below is host
plumbing around the C-ABI (`include/tsq.h`); all compute runs in libtsq:
  GpuSelectionExec   = tsq_filter_eval (device flags) + tsq_chunk_compact  (executor.go:346-438, column.go:504-552)
  GpuProjectionExec  = tsq_expr_eval per output column                     (projection.go:54-434, evaluator.go:121-133)
  GpuHashJoinExec    = tsq_join_* with TSQ_COL_DEVICE columns              (join.go:31-146)
  GpuHashAggExec     = tsq_agg_* with TSQ_COL_DEVICE columns               (aggregate.go:134-588)
  GpuSortExec        = tsq_sort_* (ORDER BY / TopN) with TSQ_COL_DEVICE columns (sort.go:27-318)
A chunk returned by Next is valid until the next call of Next on the same operator (the Go operators
recycle their chunks the same way, join.go:62-78).
"""
import ctypes as C

import numpy as np

from . import _abi as abi
from . import _lib
from .chunk import Chunk, Column, StrColumn, np_dtype
from .expression import Column as Column_
from .expression import ETReal, ETString, CompiledExpr, Unsupported


def _es(tp):
    return 4 if tp == abi.F32 else 8


class DeviceColumn:
    """one column in HBM: data + optional null bitmap (bit 1 = NOT NULL, util/chunk/column.go:89-92); a var-len (abi.BYTES)
    column also has offsets[cap + 1] and its data buffer holds `cap_bytes` bytes (column.go:28-34)."""

    def __init__(self, ctx, tp, cap_rows, with_bitmap=True, data=None, bitmap=None, offsets=None, cap_bytes=0):
        self.ctx, self.tp, self.cap = ctx, tp, cap_rows
        self.var = tp == abi.BYTES
        self.cap_bytes = cap_bytes
        self.owned = data is None
        if self.owned:
            self.data = ctx.alloc((max(cap_bytes, 1) if self.var else max(cap_rows, 1) * _es(tp)) + 64)
            self.bitmap = ctx.alloc((cap_rows + 7) // 8 + 64) if with_bitmap else None
            self.offsets = ctx.alloc((cap_rows + 1) * 8 + 64) if self.var else None
        else:
            self.data, self.bitmap, self.offsets = data, bitmap, offsets

    def col(self, nrows):
        c = abi.Col()
        c.data, c.null_bitmap, c.length = self.data, self.bitmap, nrows
        c.offsets = self.offsets if self.var else None
        c.elem_size, c.type, c.flags = (-1 if self.var else _es(self.tp)), self.tp, abi.COL_DEVICE
        return c

    def view(self, lo):
        """rows [lo, ...) — lo must be a multiple of 8 so that the bitmap stays byte aligned.  A var-len view keeps the data
        base: the offsets index it (the rows' offsets need not start at 0, tsq_colset_slice)."""
        assert lo % 8 == 0
        if self.var:
            return DeviceColumn(self.ctx, self.tp, self.cap - lo, data=self.data, offsets=self.offsets + 8 * lo, cap_bytes=self.cap_bytes,
                                bitmap=None if self.bitmap is None else self.bitmap + lo // 8)
        return DeviceColumn(self.ctx, self.tp, self.cap - lo, data=self.data + lo * _es(self.tp),
                            bitmap=None if self.bitmap is None else self.bitmap + lo // 8)

    def nbytes(self, nrows):
        """data bytes spanned by the first nrows cells of a var-len column (one 16-byte read-back)"""
        if not self.var or nrows == 0:
            return 0
        o = np.zeros(1, np.int64)
        e = np.zeros(1, np.int64)
        self.ctx.d2h(o, self.offsets)
        self.ctx.d2h(e, self.offsets + 8 * nrows)
        return int(e[0] - o[0])

    def ensure_bytes(self, nbytes):
        """grow the data buffer of an owned var-len column (contents are not kept: called before a pull fills it)"""
        if self.var and nbytes > self.cap_bytes:
            assert self.owned
            self.ctx.free(self.data)
            self.cap_bytes = int(nbytes * 1.25) + 64
            self.data = self.ctx.alloc(self.cap_bytes + 64)

    def to_host_pinned(self, nrows):
        """fixed-width column -> (data array, bitmap bytes or None) in PINNED host memory (ctx.host_array): one DMA each, no
        pageable bounce, no unpacking — the form a host operator consumes (util/chunk/column.go:28-34 keeps the bitmap packed too)"""
        assert not self.var
        data = self.ctx.host_array(max(nrows, 1), np_dtype(self.tp))
        self.ctx.d2h(data, self.data)
        bm = None
        if self.bitmap is not None:
            bm = self.ctx.host_array((nrows + 7) // 8 + 1, np.uint8)
            self.ctx.d2h(bm, self.bitmap)
        return data, bm

    def to_host(self, nrows):
        nn = None
        if self.bitmap is not None:
            bm = np.zeros((nrows + 7) // 8 + 1, np.uint8)
            self.ctx.d2h(bm, self.bitmap)
            nn = np.unpackbits(bm, bitorder="little")[:nrows].astype(bool)
        if self.var:
            offs = np.zeros(nrows + 1, np.int64)
            self.ctx.d2h(offs, self.offsets)
            lo, hi = int(offs[0]), int(offs[nrows])
            raw = np.zeros(max(hi - lo, 1), np.uint8)
            if hi > lo:
                self.ctx.d2h(raw, self.data + lo)
            return StrColumn([bytes(raw[offs[r] - lo:offs[r + 1] - lo]) if (nn is None or nn[r]) else None for r in range(nrows)])
        arr = np.zeros(max(nrows, 1), dtype=np_dtype(self.tp))
        self.ctx.d2h(arr, self.data)
        return Column(self.tp, arr[:nrows].copy(), nn)

    def free(self):
        if self.owned:
            self.ctx.free(self.data)
            if self.bitmap is not None:
                self.ctx.free(self.bitmap)
            if self.offsets is not None:
                self.ctx.free(self.offsets)
        self.data = self.bitmap = self.offsets = None


class DeviceChunk:
    def __init__(self, columns, nrows, sel=None):
        # sel: device pointer to one byte per row, 0 = the row was filtered out (Chunk.sel of the reference is a selection VECTOR,
        # util/chunk/chunk.go:31-46; a flag per row is its device form).  Only operators that say so accept a chunk with sel.
        self.columns, self.nrows, self.sel = list(columns), nrows, sel

    def NumRows(self):
        return self.nrows

    def types(self):
        return [c.tp for c in self.columns]

    def cols(self):
        return (abi.Col * len(self.columns))(*[c.col(self.nrows) for c in self.columns])

    def to_host(self):
        return Chunk([c.to_host(self.nrows) for c in self.columns])

    def free(self):
        for c in self.columns:
            c.free()

    @staticmethod
    def from_host(ctx, chunk):
        """a host chunk copied to HBM; a column without a NOT-NULL array (a NOT NULL column) gets no null bitmap"""
        cols = []
        for c in chunk.columns:
            has_nulls = c.notnull is not None
            if c.tp == abi.BYTES:
                nb = int(c.offsets[-1]) if len(c.offsets) else 0
                d = DeviceColumn(ctx, c.tp, len(c), with_bitmap=has_nulls, cap_bytes=nb)
                if nb:
                    ctx.h2d(d.data, np.ascontiguousarray(c.data[:nb]))
                ctx.h2d(d.offsets, np.ascontiguousarray(c.offsets))
            else:
                d = DeviceColumn(ctx, c.tp, len(c), with_bitmap=has_nulls)
                ctx.h2d(d.data, np.ascontiguousarray(c.data))
            if len(c) and has_nulls:
                ctx.h2d(d.bitmap, np.packbits(c.notnull, bitorder="little"))
            cols.append(d)
        return DeviceChunk(cols, chunk.NumRows())


class GpuExecutor:
    def __init__(self, ctx, types, children=()):
        self.ctx, self.lib, self.types, self.children = ctx, ctx.lib, list(types), list(children)

    def Schema(self):
        return self.types

    def Open(self):
        for c in self.children:
            c.Open()

    def Close(self):
        for c in self.children:
            c.Close()

    def _buffers(self, cap_rows, var_bytes=None):
        """output columns for up to cap_rows rows; var_bytes[i]: data bytes of var-len column i (grown later by ensure_bytes)"""
        return [DeviceColumn(self.ctx, t, cap_rows, cap_bytes=(var_bytes[i] if var_bytes else 0) if t == abi.BYTES else 0) for i, t in enumerate(self.types)]

    def _size_varlen(self, peek, handle, out, cap_rows):
        """before a pull: ask how many data bytes the var-len output columns of the next pull need and grow their buffers"""
        if abi.BYTES not in self.types:
            return
        vb, nr = (C.c_int64 * len(self.types))(), C.c_int64(0)
        _lib.check(peek(handle, cap_rows, C.byref(nr), vb, len(self.types)), handle)
        for c, b in zip(out, vb):
            c.ensure_bytes(b)


EOS = DeviceChunk([], 0)


def drain_device(exe):
    """Open / Next* / Close; returns host Chunks (the root of a plan copies its result out like writeChunks does)."""
    exe.Open()
    out = []
    try:
        while True:
            chk = exe.Next()
            if chk.NumRows() == 0:
                break
            out.append(chk.to_host())
    finally:
        exe.Close()
    return out


class DeviceTableScan(GpuExecutor):
    """a table that already lives in HBM (what a GPU-side TableReader / cop-response decoder would produce), handed out in
    batches of `batch_rows` rows as pointer views."""

    def __init__(self, ctx, table, batch_rows=1 << 24):
        super().__init__(ctx, table.types())
        self.table, self.batch, self.pos = table, (batch_rows + 7) & ~7, 0

    def Open(self):
        self.pos = 0

    def Next(self):
        n = self.table.nrows
        if self.pos >= n:
            return EOS
        hi = min(n, self.pos + self.batch)
        out = DeviceChunk([c.view(self.pos) for c in self.table.columns], hi - self.pos)
        self.pos = hi
        return out


class GpuSelectionExec(GpuExecutor):
    def __init__(self, ctx, child, filters, jit=None, compact=True):
        """compact=False: the chunk is handed on UNCOMPACTED with its selection flags (DeviceChunk.sel) — for a parent that takes them
        (GpuHashJoinExec's probe side: tsq_join_probe_push(selected)): the filtered rows are never copied."""
        super().__init__(ctx, child.Schema(), (child,))
        self.child, self.filters, self.jit, self.compact = child, list(filters), jit, compact
        self.expr, self.out, self.flags, self.cap = None, None, None, 0

    def Open(self):
        super().Open()
        self.expr = CompiledExpr(self.ctx, self.filters, jit=self.jit)

    def Next(self):
        while True:
            chk = self.child.Next()
            n = chk.NumRows()
            if n == 0:
                return EOS
            if n > self.cap:
                self._release()
                self.cap = n
                self.out = self._buffers(n) if self.compact else []
                self.flags = self.ctx.alloc(n + 64)
            w = C.c_int64(0)
            _lib.check(self.lib.tsq_filter_eval(self.expr.h, chk.cols(), len(chk.columns), n, None, self.flags, None, C.byref(w)), self.expr.h)
            self.expr.warnings += w.value
            if not self.compact:
                return DeviceChunk(chk.columns, n, sel=self.flags)
            for src, dst in zip(chk.columns, self.out):  # a selection never grows a var-len column
                dst.ensure_bytes(src.nbytes(n))
            oc = (abi.Col * len(self.out))(*[c.col(n) for c in self.out])
            m = C.c_int64(0)
            _lib.check(self.lib.tsq_chunk_compact(self.ctx.h, chk.cols(), len(chk.columns), n, self.flags, oc, C.byref(m)), self.ctx.h)
            if m.value:
                return DeviceChunk(self.out, m.value)

    def _release(self):
        if self.out:
            for c in self.out:
                c.free()
        if self.flags:
            self.ctx.free(self.flags)
        self.out, self.flags, self.cap = None, None, 0

    def Close(self):
        self._release()
        if self.expr:
            self.expr.close()
            self.expr = None
        super().Close()


class GpuProjectionExec(GpuExecutor):
    def __init__(self, ctx, child, exprs, jit=None):
        # a string-valued expression has no fixed-width output column here: the planner must learn that when it BUILDS the plan
        # (Unsupported = keep the Go ProjectionExec), not from a failing Next
        for e in exprs:
            if e.eval_type == ETString:
                raise Unsupported("GpuProjectionExec: a string-valued projection expression keeps the Go operator (VecEvalString exists on chunk level only)")
        types = [abi.F64 if e.eval_type == ETReal else (abi.U64 if e.unsigned else abi.I64) for e in exprs]
        super().__init__(ctx, types, (child,))
        self.child, self.exprs, self.jit = child, list(exprs), jit
        self.compiled, self.out, self.cap = [], None, 0

    def Open(self):
        super().Open()
        # a bare 8-byte column needs no kernel: the child's column is handed on (the reference copies it, projection.go:54-62
        # -> Column.CopyConstruct; a device pointer that stays valid until the child's next Next is equivalent here)
        self.passthru = [e.index if isinstance(e, Column_) and e.tp != abi.F32 else None for e in self.exprs]
        self.compiled = [None if p is not None else CompiledExpr(self.ctx, [e], jit=self.jit) for p, e in zip(self.passthru, self.exprs)]

    def Next(self):
        chk = self.child.Next()
        n = chk.NumRows()
        if n == 0:
            return EOS
        if n > self.cap:
            self._release()
            self.cap = n
            self.out = [None if p is not None else DeviceColumn(self.ctx, t, n) for p, t in zip(self.passthru, self.types)]
        cols = []
        for p, ce, dst in zip(self.passthru, self.compiled, self.out):
            if p is not None:
                cols.append(chk.columns[p])
                continue
            oc = dst.col(n)
            oc.type = abi.F64 if dst.tp == abi.F64 else abi.I64
            w = C.c_int64(0)
            _lib.check(self.lib.tsq_expr_eval(ce.h, chk.cols(), len(chk.columns), n, None, C.byref(oc), C.byref(w)), ce.h)
            ce.warnings += w.value
            cols.append(dst)
        return DeviceChunk(cols, n)

    def _release(self):
        if self.out:
            for c in self.out:
                if c is not None:
                    c.free()
        self.out, self.cap = None, 0

    def Close(self):
        self._release()
        for ce in self.compiled:
            if ce is not None:
                ce.close()
        self.compiled = []
        super().Close()


class GpuHashJoinExec(GpuExecutor):
    def __init__(self, ctx, left, right, left_keys, right_keys, join_type=abi.JOIN_INNER, inner_child_idx=1, pull_rows=1 << 24, used=None):
        """used: the output columns (indices into left's + right's columns) the parent reads — the planner's column pruning
        (tsq_join_set_used_columns); the others are not materialised and come out as columns without data."""
        super().__init__(ctx, left.Schema() + right.Schema(), (left, right))
        self.used = None if used is None else sorted(set(used))
        self.build_is_right = inner_child_idx == 1
        self.build = right if self.build_is_right else left
        self.probe = left if self.build_is_right else right
        bkeys = right_keys if self.build_is_right else left_keys
        pkeys = left_keys if self.build_is_right else right_keys
        cfg = abi.JoinCfg()
        cfg.join_type, cfg.build_is_right, cfg.n_keys = join_type, 1 if self.build_is_right else 0, len(bkeys)
        for i, (b, p) in enumerate(zip(bkeys, pkeys)):
            cfg.build_key_idx[i], cfg.probe_key_idx[i] = b, p
        cfg.n_build_cols, cfg.n_probe_cols = len(self.build.Schema()), len(self.probe.Schema())
        for i, t in enumerate(self.build.Schema()):
            cfg.build_types[i] = t
        for i, t in enumerate(self.probe.Schema()):
            cfg.probe_types[i] = t
        cfg.max_chunk_size = 1024
        self.cfg, self.h, self.prepared = cfg, None, False
        self.pull_rows = (pull_rows + 7) & ~7
        self.out = None

    def Open(self):
        super().Open()
        h = C.c_void_p()
        _lib.check(self.lib.tsq_join_create(self.ctx.h, C.byref(self.cfg), C.byref(h)), self.ctx.h)
        self.h, self.prepared = h, False
        if self.used is not None:
            flags = (C.c_uint8 * len(self.types))(*[1 if i in self.used else 0 for i in range(len(self.types))])
            _lib.check(self.lib.tsq_join_set_used_columns(h, flags, len(self.types)), h)
        # fixed-width outputs are BORROWED from the operator's result batch (TSQ_COL_BORROW: pointers valid until the next pull) — the
        # device form of Chunk.SwapColumns; a var-len output column keeps the copying pull
        self.borrow = abi.BYTES not in self.types
        self.out = None if self.borrow else self._buffers(self.pull_rows)

    def Next(self):
        if not self.prepared:
            while True:
                chk = self.build.Next()
                if chk.NumRows() == 0:
                    break
                if chk.sel:  # tsq_join_build_push has no `selected`: flags on the build side would be ignored silently
                    raise ValueError("GpuHashJoinExec: selection flags on the build side: use GpuSelectionExec(compact=True)")
                _lib.check(self.lib.tsq_join_build_push(self.h, chk.cols(), len(chk.columns), chk.NumRows()), self.h)
            _lib.check(self.lib.tsq_join_build_finish(self.h), self.h)
            self.prepared = True
        n, eos = C.c_int64(0), C.c_int32(0)
        while True:
            if self.borrow:
                oc = (abi.Col * len(self.types))()
                for i, t in enumerate(self.types):
                    oc[i].type, oc[i].elem_size, oc[i].flags = t, _es(t), abi.COL_DEVICE | abi.COL_BORROW
                _lib.check(self.lib.tsq_join_pull(self.h, oc, len(self.types), self.pull_rows, C.byref(n), C.byref(eos)), self.h)
                if n.value > 0:
                    # a column the parent does not use has no data: it aliases a materialised column's buffer so that a consumer which
                    # passes whole chunks on (it never reads the cells) still hands out valid pointers
                    alias = next(oc[i].data for i in range(len(self.types)) if oc[i].data)
                    cols = [DeviceColumn(self.ctx, t, n.value, data=oc[i].data or alias, bitmap=oc[i].null_bitmap if oc[i].data else None) for i, t in enumerate(self.types)]
                    return DeviceChunk(cols, n.value)
            else:
                self._size_varlen(self.lib.tsq_join_peek, self.h, self.out, self.pull_rows)
                oc = (abi.Col * len(self.out))(*[c.col(self.pull_rows) for c in self.out])
                _lib.check(self.lib.tsq_join_pull(self.h, oc, len(self.out), self.pull_rows, C.byref(n), C.byref(eos)), self.h)
                if n.value > 0:
                    return DeviceChunk(self.out, n.value)
            if eos.value:
                return EOS
            chk = self.probe.Next()
            if chk.NumRows() == 0:
                _lib.check(self.lib.tsq_join_probe_finish(self.h), self.h)
                continue
            # a probe-side chunk may carry selection flags (GpuSelectionExec(compact=False)): the join takes them as `selected`.
            # Under an OUTER join a row with selected == 0 is emitted NULL-padded (onMissMatch, executor/join.go:344-345), not dropped:
            # flags are a WHERE filter only below an inner join — anything else must compact first
            if chk.sel and self.cfg.join_type != abi.JOIN_INNER:
                raise ValueError("GpuHashJoinExec: selection flags on the probe side of an outer join: use GpuSelectionExec(compact=True)")
            _lib.check(self.lib.tsq_join_probe_push(self.h, chk.cols(), len(chk.columns), chk.NumRows(), C.c_void_p(chk.sel) if chk.sel else None), self.h)

    def Close(self):
        if self.h:
            st = abi.Stats()
            if self.lib.tsq_join_stats(self.h, C.byref(st)) == abi.OK:
                self.last_stats = st  # which route the probe batches took (tools/q3.py reports it)
            self.lib.tsq_join_cancel(self.h)
            self.lib.tsq_join_destroy(self.h)
            self.h = None
        if self.out:
            for c in self.out:
                c.free()
            self.out = None
        super().Close()


class GpuHashAggExec(GpuExecutor):
    """HashAggExec on device-resident chunks (tsq_agg_*); stream=True: StreamAggExec — the child delivers its rows ordered by the group
    keys (a GpuSortExec, an ordered scan) and the groups come out in that order (tsq_agg_set_stream)."""

    def __init__(self, ctx, child, group_by_cols, agg_funcs, est_groups=0, pull_rows=1 << 22, stream=False):
        self.stream = stream
        types = []
        for f in agg_funcs:
            types += f.out_types()
        super().__init__(ctx, types, (child,))
        self.child = child
        cfg = abi.AggCfg()
        in_types = child.Schema()
        cfg.n_group_keys = len(group_by_cols)
        for i, c in enumerate(group_by_cols):
            cfg.group_key_col[i], cfg.group_key_type[i] = c, in_types[c]
        cfg.n_aggs = len(agg_funcs)
        for i, f in enumerate(agg_funcs):
            cfg.aggs[i].func, cfg.aggs[i].mode = f.func, f.mode
            cfg.aggs[i].arg_col, cfg.aggs[i].arg_col2, cfg.aggs[i].arg_type = f.arg_col, f.arg_col2, f.arg_type
        cfg.n_input_cols = len(in_types)
        for i, t in enumerate(in_types):
            cfg.input_types[i] = t
        cfg.est_groups, cfg.max_chunk_size = est_groups, 1024
        self.cfg, self.h, self.prepared, self.out = cfg, None, False, None
        self.pull_rows = (pull_rows + 7) & ~7

    def Open(self):
        super().Open()
        h = C.c_void_p()
        _lib.check(self.lib.tsq_agg_create(self.ctx.h, C.byref(self.cfg), C.byref(h)), self.ctx.h)
        if self.stream:
            _lib.check(self.lib.tsq_agg_set_stream(h, 1), h)
        self.h, self.prepared = h, False
        self.out = self._buffers(self.pull_rows)

    def Next(self):
        if not self.prepared:
            self.pushed = False
            while True:
                chk = self.child.Next()
                if chk.NumRows() == 0:
                    break
                _lib.check(self.lib.tsq_agg_push(self.h, chk.cols(), len(chk.columns), chk.NumRows()), self.h)
                self.pushed = True
            _lib.check(self.lib.tsq_agg_finish(self.h), self.h)
            self.prepared = True
        if not self.pushed:
            # no input chunk: GROUP BY yields no rows (aggregate_test.go:58-59).  The single default row of a key-less
            # aggregate over empty input (aggregate.go:572-574) is served by the host-chunk HashAggExec.
            return EOS
        n, eos = C.c_int64(0), C.c_int32(0)
        self._size_varlen(self.lib.tsq_agg_peek, self.h, self.out, self.pull_rows)
        oc = (abi.Col * len(self.out))(*[c.col(self.pull_rows) for c in self.out])
        _lib.check(self.lib.tsq_agg_pull(self.h, oc, len(self.out), self.pull_rows, C.byref(n), C.byref(eos)), self.h)
        if n.value == 0:
            return EOS
        return DeviceChunk(self.out, n.value)

    def Close(self):
        if self.h:
            self.lib.tsq_agg_cancel(self.h)
            self.lib.tsq_agg_destroy(self.h)
            self.h = None
        if self.out:
            for c in self.out:
                c.free()
            self.out = None
        super().Close()


class GpuSortExec(GpuExecutor):
    """SortExec / TopNExec on device-resident chunks: tsq_sort_* with TSQ_COL_DEVICE columns (executor/sort.go:27-318)."""

    def __init__(self, ctx, child, by_cols, by_desc, offset=0, count=-1, pull_rows=1 << 22):
        super().__init__(ctx, child.Schema(), (child,))
        self.child = child
        cfg = abi.SortCfg()
        cfg.n_cols = len(self.types)
        for i, t in enumerate(self.types):
            cfg.col_types[i] = t
        cfg.n_keys = len(by_cols)
        for i, (c, d) in enumerate(zip(by_cols, by_desc)):
            cfg.key_col[i], cfg.key_desc[i] = c, 1 if d else 0
        cfg.limit_offset, cfg.limit_count, cfg.max_chunk_size = offset, count, 1024
        self.cfg, self.h, self.fetched, self.out = cfg, None, False, None
        self.pull_rows = (pull_rows + 7) & ~7

    def Open(self):
        super().Open()
        h = C.c_void_p()
        _lib.check(self.lib.tsq_sort_create(self.ctx.h, C.byref(self.cfg), C.byref(h)), self.ctx.h)
        self.h, self.fetched = h, False
        self.out = self._buffers(self.pull_rows)

    def Next(self):
        if not self.fetched:
            while True:
                chk = self.child.Next()
                if chk.NumRows() == 0:
                    break
                _lib.check(self.lib.tsq_sort_push(self.h, chk.cols(), len(chk.columns), chk.NumRows()), self.h)
            _lib.check(self.lib.tsq_sort_finish(self.h), self.h)
            self.fetched = True
        n, eos = C.c_int64(0), C.c_int32(0)
        self._size_varlen(self.lib.tsq_sort_peek, self.h, self.out, self.pull_rows)
        oc = (abi.Col * len(self.out))(*[c.col(self.pull_rows) for c in self.out])
        _lib.check(self.lib.tsq_sort_pull(self.h, oc, len(self.out), self.pull_rows, C.byref(n), C.byref(eos)), self.h)
        return DeviceChunk(self.out, n.value) if n.value else EOS

    def Close(self):
        if self.h:
            self.lib.tsq_sort_cancel(self.h)
            self.lib.tsq_sort_destroy(self.h)
            self.h = None
        if self.out:
            for c in self.out:
                c.free()
            self.out = None
        super().Close()
