"""Mirror of package `executor`'s volcano interface for the harness (tests, bench).

`Executor` keeps the reference contract (executor/executor.go:146-162): Open(); Next(req) fills
`req` with at most max_chunk_size rows, an EMPTY chunk means end of stream and Next stays
idempotent after it; Close().  HashJoinExec / HashAggExec / SelectionExec / ProjectionExec here
are what the Go shim's GPUHashJoinExec / GPUHashAggExec do: they drain their
children chunk by chunk into libtsq and hand result chunks back.  All compute is in libtsq.
"""
import ctypes as C

import numpy as np

from . import _abi as abi
from . import _lib
from .chunk import Chunk, Column, StrColumn, chunk_from_buffers, make_cols, np_dtype, out_buffers
from .expression import CompiledExpr


class Executor:
    def __init__(self, ctx, types, children=(), max_chunk_size=1024):
        self.ctx = ctx
        self.types = list(types)  # Schema(): output column types
        self.children = list(children)
        self.max_chunk_size = max_chunk_size

    def Schema(self):
        return self.types

    def Open(self):  # baseExecutor.Open (executor.go:71-79): open children
        for c in self.children:
            c.Open()

    def Next(self):
        """returns a Chunk; NumRows()==0 means EOS (server/conn.go:955-957)."""
        raise NotImplementedError

    def Close(self):
        for c in self.children:
            c.Close()

    def empty(self):
        return Chunk([StrColumn([]) if tp == abi.BYTES else Column(tp, np.zeros(0, np_dtype(tp))) for tp in self.types])


def drain(exe):
    """runs Open/Next*/Close like recordSet + writeChunks (adapter.go:93-119, conn.go:931-975)."""
    exe.Open()
    out = []
    try:
        while True:
            chk = exe.Next()
            if chk.NumRows() == 0:
                break
            out.append(chk)
    finally:
        exe.Close()
    return out


class MockDataSource(Executor):
    """executor/benchmark_test.go:50-144 mockDataSource: hands out pre-built chunks."""

    def __init__(self, ctx, chunk, max_chunk_size=1024):
        super().__init__(ctx, chunk.types(), (), max_chunk_size)
        self.chunk = chunk
        self.pos = 0

    def Open(self):
        self.pos = 0

    def Next(self):
        n = self.chunk.NumRows()
        if self.pos >= n:
            return self.empty()
        hi = min(n, self.pos + self.max_chunk_size)
        out = self.chunk.slice(self.pos, hi)
        self.pos = hi
        return out


class HashJoinExec(Executor):
    """GPU HashJoinExec (replaces executor/join.go:31-146).

    left/right: child executors; inner_child_idx: which child is hashed (build side),
    exactly like PhysicalHashJoin.InnerChildIdx (planner/core/physical_plans.go:219-224).
    """

    def __init__(self, ctx, left, right, left_keys, right_keys, join_type=abi.JOIN_INNER, inner_child_idx=1,
                 other_conditions=(), outer_filter=(), max_chunk_size=1024, probe_batch_rows=0):
        super().__init__(ctx, left.Schema() + right.Schema(), (left, right), max_chunk_size)
        self.lib = ctx.lib
        self.build_is_right = inner_child_idx == 1
        self.build = right if self.build_is_right else left
        self.probe = left if self.build_is_right else right
        bkeys = right_keys if self.build_is_right else left_keys
        pkeys = left_keys if self.build_is_right else right_keys
        cfg = abi.JoinCfg()
        cfg.join_type = join_type
        cfg.build_is_right = 1 if self.build_is_right else 0
        cfg.n_keys = len(bkeys)
        for i, (b, p) in enumerate(zip(bkeys, pkeys)):
            cfg.build_key_idx[i] = b
            cfg.probe_key_idx[i] = p
        cfg.n_build_cols = len(self.build.Schema())
        cfg.n_probe_cols = len(self.probe.Schema())
        for i, t in enumerate(self.build.Schema()):
            cfg.build_types[i] = t
        for i, t in enumerate(self.probe.Schema()):
            cfg.probe_types[i] = t
        cfg.max_chunk_size = max_chunk_size
        cfg.concurrency = 5  # tidb_hash_join_concurrency default (tidb_vars.go:249); unused on the GPU
        cfg.probe_batch_rows = probe_batch_rows
        self._keep = []
        if other_conditions:
            from .expression import compile_list
            arr = compile_list(list(other_conditions))
            self._keep.append(arr)
            cfg.other_conds = arr
            cfg.n_other_conds = len(other_conditions)
        if outer_filter:
            from .expression import compile_list
            arr = compile_list(list(outer_filter))
            self._keep.append(arr)
            cfg.outer_filters = arr
            cfg.n_outer_filters = len(outer_filter)
        self.cfg = cfg
        self.h = None
        self.prepared = False
        self.probe_eos = False

    def Open(self):
        super().Open()
        h = C.c_void_p()
        _lib.check(self.lib.tsq_join_create(self.ctx.h, C.byref(self.cfg), C.byref(h)), self.ctx.h)
        self.h = h
        self.prepared = False
        self.probe_eos = False
        if getattr(self, "ordered", False):
            _lib.check(self.lib.tsq_join_set_ordered(h, 1), h)

    def _build(self):  # fetchAndBuildHashTable (join.go:148-158)
        while True:
            chk = self.build.Next()
            if chk.NumRows() == 0:
                break
            keep = []
            cols = make_cols(chk.columns, keep)
            _lib.check(self.lib.tsq_join_build_push(self.h, cols, len(chk.columns), chk.NumRows()), self.h)
        _lib.check(self.lib.tsq_join_build_finish(self.h), self.h)

    def Next(self):  # HashJoinExec.Next (join.go:125-146)
        if not self.prepared:
            self._build()
            self.prepared = True
        keep = []
        out, bufs = out_buffers(self.types, self.max_chunk_size, keep)
        n = C.c_int64(0)
        eos = C.c_int32(0)
        while True:
            _lib.check(self.lib.tsq_join_pull(self.h, out, len(self.types), self.max_chunk_size, C.byref(n), C.byref(eos)), self.h)
            if n.value > 0:
                return chunk_from_buffers(self.types, bufs, n.value)
            if eos.value:
                return self.empty()
            # needs more probe input: feed one outer chunk (fetchOuterSideChunks, join.go:194-221)
            chk = self.probe.Next()
            if chk.NumRows() == 0:
                _lib.check(self.lib.tsq_join_probe_finish(self.h), self.h)
                continue
            k2 = []
            cols = make_cols(chk.columns, k2)
            _lib.check(self.lib.tsq_join_probe_push(self.h, cols, len(chk.columns), chk.NumRows(), None), self.h)

    def stats(self):
        st = abi.Stats()
        _lib.check(self.lib.tsq_join_stats(self.h, C.byref(st)), self.h)
        return st

    def Close(self):  # join.go:81-107: stop promptly, free everything
        if self.h:
            self.lib.tsq_join_cancel(self.h)
            self.lib.tsq_join_destroy(self.h)
            self.h = None
        super().Close()


class MergeJoinExec(HashJoinExec):
    """GPU MergeJoinExec (replaces executor/merge_join.go:31-373).  The reference walks two SORTED children with two cursors;
    its result is the outer rows in order, each joined with its inner group in order (NULL-key inner rows skipped, unmatched
    outer rows padded for outer joins).  That is the hash join with ordered output (tsq_join_set_ordered): the outer child is
    the probe side, the inner child the build side — and the inputs need not even be sorted for the order to hold."""

    def __init__(self, ctx, left, right, left_keys, right_keys, join_type=abi.JOIN_INNER, inner_child_idx=1, other_conditions=(),
                 outer_filter=(), max_chunk_size=1024):
        super().__init__(ctx, left, right, left_keys, right_keys, join_type, inner_child_idx, other_conditions, outer_filter, max_chunk_size)
        self.ordered = True


class AggFuncDesc:
    """expression/aggregation.AggFuncDesc (descriptor.go): name, mode, argument column(s)."""

    def __init__(self, func, arg_col, arg_type=abi.I64, mode=abi.MODE_COMPLETE, arg_col2=-1):
        self.func, self.arg_col, self.arg_type, self.mode, self.arg_col2 = func, arg_col, arg_type, mode, arg_col2

    def out_types(self):
        real = self.arg_type in (abi.F32, abi.F64)
        partial = self.mode in (abi.MODE_PARTIAL1, abi.MODE_PARTIAL2)
        if self.func == abi.AGG_COUNT:
            return [abi.I64]
        if self.func == abi.AGG_SUM:
            return [abi.F64 if real else abi.I64]
        if self.func == abi.AGG_AVG:
            s = abi.F64 if real else abi.I64
            return [abi.I64, s] if partial else [s]
        return [self.arg_type]


class HashAggExec(Executor):
    """GPU HashAggExec (replaces executor/aggregate.go:134-155,482-588)."""

    def __init__(self, ctx, child, group_by_cols, agg_funcs, max_chunk_size=1024, est_groups=0):
        types = []
        for f in agg_funcs:
            types += f.out_types()
        super().__init__(ctx, types, (child,), max_chunk_size)
        self.lib = ctx.lib
        self.child = child
        cfg = abi.AggCfg()
        cfg.n_group_keys = len(group_by_cols)
        in_types = child.Schema()
        for i, c in enumerate(group_by_cols):
            cfg.group_key_col[i] = c
            cfg.group_key_type[i] = in_types[c]
        cfg.n_aggs = len(agg_funcs)
        for i, f in enumerate(agg_funcs):
            cfg.aggs[i].func, cfg.aggs[i].mode = f.func, f.mode
            cfg.aggs[i].arg_col, cfg.aggs[i].arg_col2, cfg.aggs[i].arg_type = f.arg_col, f.arg_col2, f.arg_type
        cfg.n_input_cols = len(in_types)
        for i, t in enumerate(in_types):
            cfg.input_types[i] = t
        cfg.est_groups = est_groups
        cfg.max_chunk_size = max_chunk_size
        self.cfg = cfg
        self.h = None
        self.prepared = False

    def Open(self):
        super().Open()
        h = C.c_void_p()
        _lib.check(self.lib.tsq_agg_create(self.ctx.h, C.byref(self.cfg), C.byref(h)), self.ctx.h)
        self.h = h
        self.prepared = False

    def Next(self):  # parallelExec (aggregate.go:559-588)
        if not self.prepared:
            while True:  # fetchChildData (aggregate.go:487-522)
                chk = self.child.Next()
                if chk.NumRows() == 0:
                    break
                keep = []
                cols = make_cols(chk.columns, keep)
                _lib.check(self.lib.tsq_agg_push(self.h, cols, len(chk.columns), chk.NumRows()), self.h)
            _lib.check(self.lib.tsq_agg_finish(self.h), self.h)
            self.prepared = True
        keep = []
        var_bytes = None
        if abi.BYTES in self.types:  # string outputs (firstRow4String / maxMin4String): size their data arrays for this pull
            pn, pb = C.c_int64(0), (C.c_int64 * len(self.types))()
            _lib.check(self.lib.tsq_agg_peek(self.h, self.max_chunk_size, C.byref(pn), pb, len(self.types)), self.h)
            var_bytes = list(pb)
        out, bufs = out_buffers(self.types, self.max_chunk_size, keep, var_bytes)
        n = C.c_int64(0)
        eos = C.c_int32(0)
        _lib.check(self.lib.tsq_agg_pull(self.h, out, len(self.types), self.max_chunk_size, C.byref(n), C.byref(eos)), self.h)
        if n.value == 0:
            return self.empty()
        return chunk_from_buffers(self.types, bufs, n.value)

    def Close(self):
        if self.h:
            self.lib.tsq_agg_cancel(self.h)
            self.lib.tsq_agg_destroy(self.h)
            self.h = None
        super().Close()


class StreamAggExec(HashAggExec):
    """GPU StreamAggExec (BASELINE.json north star; the reference has only the plan name "StreamAgg", planner/core/cbo_test.go:200-212):
    the child delivers rows ordered by the group-by columns, groups come out in that order (tsq_agg_set_stream, csrc/tsq_streamagg.h).
    `child_is_ordered=False` puts a SortExec on the group-by columns below the operator — the planner's order enforcement."""

    def __init__(self, ctx, child, group_by_cols, agg_funcs, max_chunk_size=1024, child_is_ordered=True):
        if not child_is_ordered and group_by_cols:
            child = SortExec(ctx, child, list(group_by_cols), [False] * len(group_by_cols), max_chunk_size)
        super().__init__(ctx, child, group_by_cols, agg_funcs, max_chunk_size)

    def Open(self):
        super().Open()
        _lib.check(self.lib.tsq_agg_set_stream(self.h, 1), self.h)


class SelectionExec(Executor):
    """SelectionExec.Next vectorized branch (executor/executor.go:393-409, STUB in the reference):
    VectorizedFilter over the child chunk, keep the selected rows."""

    def __init__(self, ctx, child, filters, max_chunk_size=1024):
        super().__init__(ctx, child.Schema(), (child,), max_chunk_size)
        self.child = child
        self.filters = list(filters)
        self.expr = None

    def Open(self):
        super().Open()
        self.expr = CompiledExpr(self.ctx, self.filters)

    def Next(self):
        while True:
            chk = self.child.Next()
            if chk.NumRows() == 0:
                return self.empty()
            selected = self.expr.VectorizedFilter(chk)
            if selected.any():
                idx = np.nonzero(selected)[0]
                cols = [Column(c.tp, c.data[idx], None if c.notnull is None else c.notnull[idx]) for c in chk.columns]
                return Chunk(cols)

    def Close(self):
        if self.expr:
            self.expr.close()
            self.expr = None
        super().Close()


class ProjectionExec(Executor):
    """ProjectionExec (executor/projection.go) via EvaluatorSuite.Run (expression/evaluator.go:121-133)."""

    def __init__(self, ctx, child, exprs, max_chunk_size=1024):
        from .expression import ETReal
        types = [abi.F64 if e.eval_type == ETReal else (abi.U64 if e.unsigned else abi.I64) for e in exprs]
        super().__init__(ctx, types, (child,), max_chunk_size)
        self.child = child
        self.exprs = list(exprs)
        self.compiled = []

    def Open(self):
        super().Open()
        self.compiled = [CompiledExpr(self.ctx, [e]) for e in self.exprs]

    def Next(self):
        chk = self.child.Next()
        if chk.NumRows() == 0:
            return self.empty()
        return Chunk([ce.VecEval(chk) for ce in self.compiled])

    def Close(self):
        for ce in self.compiled:
            ce.close()
        self.compiled = []
        super().Close()


class SortExec(Executor):
    """GPU SortExec (replaces executor/sort.go:27-144): ORDER BY over bare columns, each ascending or descending."""

    def __init__(self, ctx, child, by_cols, by_desc, max_chunk_size=1024, offset=0, count=-1):
        super().__init__(ctx, child.Schema(), (child,), max_chunk_size)
        self.lib = ctx.lib
        cfg = abi.SortCfg()
        cfg.n_cols = len(self.types)
        for i, t in enumerate(self.types):
            cfg.col_types[i] = t
        cfg.n_keys = len(by_cols)
        for i, (c, d) in enumerate(zip(by_cols, by_desc)):
            cfg.key_col[i], cfg.key_desc[i] = c, 1 if d else 0
        cfg.limit_offset, cfg.limit_count, cfg.max_chunk_size = offset, count, max_chunk_size
        self.cfg, self.h, self.fetched = cfg, None, False

    def Open(self):
        super().Open()
        h = C.c_void_p()
        _lib.check(self.lib.tsq_sort_create(self.ctx.h, C.byref(self.cfg), C.byref(h)), self.ctx.h)
        self.h, self.fetched = h, False

    def Next(self):
        if not self.fetched:  # fetchRowChunks (sort.go:80-97), then the sort
            while True:
                chk = self.children[0].Next()
                if chk.NumRows() == 0:
                    break
                keep = []
                cols = make_cols(chk.columns, keep)
                _lib.check(self.lib.tsq_sort_push(self.h, cols, len(chk.columns), chk.NumRows()), self.h)
            _lib.check(self.lib.tsq_sort_finish(self.h), self.h)
            self.fetched = True
        keep = []
        var_bytes = None
        if abi.BYTES in self.types:  # how many data bytes the var-len columns of the next chunk need
            vb, nr = (C.c_int64 * len(self.types))(), C.c_int64(0)
            _lib.check(self.lib.tsq_sort_peek(self.h, self.max_chunk_size, C.byref(nr), vb, len(self.types)), self.h)
            var_bytes = list(vb)
        out, bufs = out_buffers(self.types, self.max_chunk_size, keep, var_bytes=var_bytes)
        n, eos = C.c_int64(0), C.c_int32(0)
        _lib.check(self.lib.tsq_sort_pull(self.h, out, len(self.types), self.max_chunk_size, C.byref(n), C.byref(eos)), self.h)
        return chunk_from_buffers(self.types, bufs, n.value) if n.value else self.empty()

    def Close(self):
        if self.h:
            self.lib.tsq_sort_cancel(self.h)
            self.lib.tsq_sort_destroy(self.h)
            self.h = None
        super().Close()


class TopNExec(SortExec):
    """GPU TopNExec (replaces executor/sort.go:146-318): rows [offset, offset + count) of the ORDER BY order."""

    def __init__(self, ctx, child, by_cols, by_desc, offset, count, max_chunk_size=1024):
        super().__init__(ctx, child, by_cols, by_desc, max_chunk_size, offset=offset, count=count)
