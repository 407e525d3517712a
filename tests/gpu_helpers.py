"""Helpers for the `-m gpu` tests: drive libtsq through the C-ABI exactly like the cgo shim would."""
import ctypes as C

import numpy as np

from tinysql_amd import _abi as abi
from tinysql_amd import _lib
from tinysql_amd.chunk import Chunk, Column, chunk_from_buffers, concat, make_cols, np_dtype, out_buffers


def push_chunked(fn, handle, chunk, chunk_rows, selected=None, lib_handle_for_err=None):
    """one C call per <=chunk_rows rows (tidb_max_chunk_size sized pushes)."""
    n = chunk.NumRows()
    for lo in range(0, n, chunk_rows):
        hi = min(n, lo + chunk_rows)
        part = chunk.slice(lo, hi)
        keep = []
        cols = make_cols(part.columns, keep)
        if selected is None:
            _lib.check(fn(handle, cols, len(part.columns), hi - lo), handle)
        else:
            s = np.ascontiguousarray(selected[lo:hi], dtype=np.uint8)
            _lib.check(fn(handle, cols, len(part.columns), hi - lo, s.ctypes.data_as(C.c_void_p)), handle)


def run_join(ctx, cfg, build, probe, chunk_rows=1024, selected=None, pull_rows=1024, count_only=False, checksum=False, radix=None,
             stats_out=None, ordered=None, packing=None):
    """build_push* -> build_finish -> (probe_push, pull*)* -> probe_finish -> pull*; returns Chunk (or count[,sum,xor])."""
    lib = ctx.lib
    h = C.c_void_p()
    _lib.check(lib.tsq_join_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        if radix is not None:
            _lib.check(lib.tsq_join_set_radix(h, radix), h)
        if ordered is not None:
            _lib.check(lib.tsq_join_set_ordered(h, 1 if ordered else 0), h)
        if packing is not None:
            _lib.check(lib.tsq_join_set_key_packing(h, packing), h)
        push_chunked(lib.tsq_join_build_push, h, build, chunk_rows)
        _lib.check(lib.tsq_join_build_finish(h), h)
        if count_only:
            _lib.check(lib.tsq_join_set_count_only(h, 1), h)
            if checksum:
                _lib.check(lib.tsq_join_set_checksum(h, 1), h)
        probe_is_left = cfg.build_is_right != 0
        ptypes = [cfg.probe_types[i] for i in range(cfg.n_probe_cols)]
        btypes = [cfg.build_types[i] for i in range(cfg.n_build_cols)]
        out_types = ptypes + btypes if probe_is_left else btypes + ptypes
        got = []

        has_var = abi.BYTES in out_types

        def pull_all():
            while True:
                keep = []
                var_bytes = None
                if has_var:  # size the data arrays of the var-len columns for this pull (tsq_join_peek)
                    pn, pb = C.c_int64(0), (C.c_int64 * len(out_types))()
                    _lib.check(lib.tsq_join_peek(h, pull_rows, C.byref(pn), pb, len(out_types)), h)
                    var_bytes = list(pb)
                out, bufs = out_buffers(out_types, pull_rows, keep, var_bytes)
                n, eos = C.c_int64(0), C.c_int32(0)
                _lib.check(lib.tsq_join_pull(h, out, len(out_types), pull_rows, C.byref(n), C.byref(eos)), h)
                if n.value == 0:
                    return bool(eos.value)
                got.append(chunk_from_buffers(out_types, bufs, n.value))

        n = probe.NumRows()
        for lo in range(0, n, chunk_rows):
            hi = min(n, lo + chunk_rows)
            part = probe.slice(lo, hi)
            keep = []
            cols = make_cols(part.columns, keep)
            sel = None
            if selected is not None:
                s = np.ascontiguousarray(selected[lo:hi], dtype=np.uint8)
                keep.append(s)
                sel = s.ctypes.data_as(C.c_void_p)
            _lib.check(lib.tsq_join_probe_push(h, cols, len(part.columns), hi - lo, sel), h)
            if not count_only:
                pull_all()
        _lib.check(lib.tsq_join_probe_finish(h), h)
        if count_only:
            c = C.c_int64(0)
            _lib.check(lib.tsq_join_count(h, C.byref(c)), h)
            if stats_out is not None:
                st = abi.Stats()
                _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
                stats_out.append(st)
            if checksum:
                s, x = C.c_uint64(0), C.c_uint64(0)
                _lib.check(lib.tsq_join_checksum(h, C.byref(s), C.byref(x)), h)
                return c.value, s.value, x.value
            return c.value
        assert pull_all() is True
        if stats_out is not None:
            st = abi.Stats()
            _lib.check(lib.tsq_join_stats(h, C.byref(st)), h)
            stats_out.append(st)
        return concat(got, out_types)
    finally:
        lib.tsq_join_destroy(h)


def run_agg(ctx, cfg, chunk, out_types, chunk_rows=1024, pull_rows=1024, fast=None, stats_out=None, stream=False):
    lib = ctx.lib
    h = C.c_void_p()
    _lib.check(lib.tsq_agg_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        if fast is not None:
            _lib.check(lib.tsq_agg_set_fast(h, fast), h)
        if stream:  # StreamAggExec: the rows of `chunk` are ordered by the group keys
            _lib.check(lib.tsq_agg_set_stream(h, 1), h)
        push_chunked(lib.tsq_agg_push, h, chunk, chunk_rows)
        _lib.check(lib.tsq_agg_finish(h), h)
        if stats_out is not None:
            st = abi.Stats()
            _lib.check(lib.tsq_agg_stats(h, C.byref(st)), h)
            stats_out.append(st)
        got = []
        has_var = abi.BYTES in out_types
        while True:
            keep = []
            var_bytes = None
            if has_var:  # size the data arrays of the var-len columns for this pull (tsq_agg_peek)
                pn, pb = C.c_int64(0), (C.c_int64 * len(out_types))()
                _lib.check(lib.tsq_agg_peek(h, pull_rows, C.byref(pn), pb, len(out_types)), h)
                var_bytes = list(pb)
            out, bufs = out_buffers(out_types, pull_rows, keep, var_bytes)
            n, eos = C.c_int64(0), C.c_int32(0)
            _lib.check(lib.tsq_agg_pull(h, out, len(out_types), pull_rows, C.byref(n), C.byref(eos)), h)
            if n.value == 0:
                assert eos.value == 1
                break
            got.append(chunk_from_buffers(out_types, bufs, n.value))
        return concat(got, out_types)
    finally:
        lib.tsq_agg_destroy(h)


def run_sort(ctx, chunk, key_cols, key_desc, chunk_rows=1024, pull_rows=1024, offset=0, count=-1, stats_out=None):
    """push* -> finish -> pull* through the C-ABI; returns the ordered Chunk."""
    lib = ctx.lib
    types = chunk.types()
    cfg = abi.SortCfg()
    cfg.n_cols = len(types)
    for i, t in enumerate(types):
        cfg.col_types[i] = t
    cfg.n_keys = len(key_cols)
    for i, (c, d) in enumerate(zip(key_cols, key_desc)):
        cfg.key_col[i], cfg.key_desc[i] = c, 1 if d else 0
    cfg.limit_offset, cfg.limit_count, cfg.max_chunk_size = offset, count, 1024
    h = C.c_void_p()
    _lib.check(lib.tsq_sort_create(ctx.h, C.byref(cfg), C.byref(h)), ctx.h)
    try:
        push_chunked(lib.tsq_sort_push, h, chunk, chunk_rows)
        _lib.check(lib.tsq_sort_finish(h), h)
        if stats_out is not None:
            rows, p, sk, ms = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_double(0)
            _lib.check(lib.tsq_sort_stats(h, C.byref(rows), C.byref(p), C.byref(sk), C.byref(ms)), h)
            stats_out.append({"rows": rows.value, "passes": p.value, "passes_skipped": sk.value, "sort_kernel_ms": ms.value})
        got = []
        while True:
            keep = []
            var_bytes = None
            if abi.BYTES in types:
                vb, nr = (C.c_int64 * len(types))(), C.c_int64(0)
                _lib.check(lib.tsq_sort_peek(h, pull_rows, C.byref(nr), vb, len(types)), h)
                var_bytes = list(vb)
            out, bufs = out_buffers(types, pull_rows, keep, var_bytes=var_bytes)
            n, eos = C.c_int64(0), C.c_int32(0)
            _lib.check(lib.tsq_sort_pull(h, out, len(types), pull_rows, C.byref(n), C.byref(eos)), h)
            if var_bytes is not None:
                assert n.value == nr.value
            if n.value == 0:
                assert eos.value == 1
                break
            got.append(chunk_from_buffers(types, bufs, n.value))
        return concat(got, types)
    finally:
        lib.tsq_sort_destroy(h)


class DevCol:
    """a device-resident column allocated through the C-ABI (tsq_dev_alloc)."""

    def __init__(self, ctx, tp, nrows, with_nulls=False):
        self.ctx, self.tp, self.n = ctx, tp, nrows
        es = 4 if tp == abi.F32 else 8
        self.data = ctx.alloc(max(nrows, 1) * es + 64)
        self.bitmap = ctx.alloc((nrows + 7) // 8 + 64) if with_nulls else None

    def col(self):
        c = abi.Col()
        c.data = self.data
        c.null_bitmap = self.bitmap
        c.length = self.n
        c.elem_size = 4 if self.tp == abi.F32 else 8
        c.type = self.tp
        c.flags = abi.COL_DEVICE
        return c

    def to_host(self):
        arr = np.zeros(max(self.n, 1), dtype=np_dtype(self.tp))
        self.ctx.d2h(arr, self.data)
        nn = None
        if self.bitmap:
            bm = np.zeros((self.n + 7) // 8 + 1, np.uint8)
            self.ctx.d2h(bm, self.bitmap)
            nn = np.unpackbits(bm, bitorder="little")[:self.n].astype(bool)
        return Column(self.tp, arr[:self.n], nn)

    def free(self):
        self.ctx.free(self.data)
        if self.bitmap:
            self.ctx.free(self.bitmap)
        self.data = self.bitmap = None


class DevStrCol:
    """a device-resident var-len column (offsets[n + 1] + data [+ bitmap]) holding the cells of a StrColumn, or an empty
    output column with room for `nrows` cells and `nbytes` data bytes."""

    tp = abi.BYTES

    def __init__(self, ctx, column=None, nrows=0, nbytes=0):
        self.ctx = ctx
        if column is not None:
            self.n = len(column)
            self.data = ctx.alloc(len(column.data) + 64)
            self.offsets = ctx.alloc((self.n + 1) * 8 + 64)
            ctx.h2d(self.data, column.data)
            ctx.h2d(self.offsets, column.offsets)
            self.bitmap = None
            if column.notnull is not None:
                self.bitmap = ctx.alloc((self.n + 7) // 8 + 64)
                ctx.h2d(self.bitmap, column.bitmap())
        else:
            self.n = nrows
            self.data = ctx.alloc(nbytes + 64)
            self.offsets = ctx.alloc((nrows + 1) * 8 + 64)
            self.bitmap = ctx.alloc((nrows + 7) // 8 + 64)

    def col(self):
        c = abi.Col()
        c.data, c.null_bitmap, c.offsets = self.data, self.bitmap, self.offsets
        c.length, c.elem_size, c.type, c.flags = self.n, -1, abi.BYTES, abi.COL_DEVICE
        return c

    def to_host(self, n, nbytes):
        from tinysql_amd.chunk import StrColumn, unpack_bitmap

        offs = np.zeros(n + 1, np.int64)
        self.ctx.d2h(offs, self.offsets)
        data = np.zeros(nbytes + 8, np.uint8)
        self.ctx.d2h(data, self.data)
        bm = np.zeros((n + 7) // 8 + 1, np.uint8)
        self.ctx.d2h(bm, self.bitmap)
        nn = unpack_bitmap(bm, n)
        raw = data.tobytes()
        return StrColumn([raw[offs[i]:offs[i + 1]] if nn[i] else None for i in range(n)])

    def free(self):
        for p in (self.data, self.offsets, self.bitmap):
            if p:
                self.ctx.free(p)


def dev_cols(cols):
    arr = (abi.Col * len(cols))()
    for i, c in enumerate(cols):
        arr[i] = c.col()
    return arr


def gen_spec(kind, table=0, col=0, seed=42, start=0, a=0, b=0, m=0, null_pct=0):
    s = abi.GenSpec()
    s.kind, s.table, s.col, s.seed, s.start, s.a, s.b, s.m, s.null_pct = kind, table, col, seed, start, a, b, m, null_pct
    return s


def to_device(ctx, column):
    d = DevCol(ctx, column.tp, len(column), with_nulls=column.notnull is not None)
    if len(column):
        ctx.h2d(d.data, column.data)
        if column.notnull is not None:
            ctx.h2d(d.bitmap, column.bitmap())
    return d


# numpy restatement of splitmix64 / rowhash for size-independent property checks at full size
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def np_splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def np_rowhash(cols):
    """cols: list of uint64 arrays (no NULLs); returns per-row hash (same as tsq_rowhash_step chain)."""
    with np.errstate(over="ignore"):
        h = np.full(len(cols[0]), 0x243F6A8885A308D3, dtype=np.uint64)
        for c, v in enumerate(cols):
            h = np_splitmix64(h ^ (v + np.uint64(((c + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)))
        return h


def np_gen_r(seed, table, col, i):
    return np_splitmix64(np.uint64(seed) ^ np.uint64((table << 56) & 0xFFFFFFFFFFFFFFFF) ^ np.uint64(col << 48) ^ i)
