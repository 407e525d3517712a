"""Multi-GPU hash-radix redistribute: one process per GPU.

`Comm` is a thin caller of libtsq's tsq_comm_* / tsq_redistribute (RCCL send/recv over xGMI inside the library,
csrc/tsq_comm.hip) — nothing but plain pointers and sizes crosses the C-ABI, a Go host calls the same sequence
(INTEGRATION.md §6).  `DistHashJoinCount` and `dist_hash_agg` are the two distributed plans built from it.

CPU analogue in the reference: HashAggExec's partial->final shuffle (executor/aggregate.go:352-356) and the probe-chunk
dispatch of HashJoinExec (executor/join.go:219).  Equi-join and GROUP BY are partitionable by any function of the key, so the
ONLY data-path collective is this exchange: rank r keeps / receives every row whose key ranks to r (tsq_key_rank), then runs
the single-GPU operator locally.  xGMI is point-to-point (7 links per GPU), so an all-to-all drives all links at once;
ring-style collectives would be bound by one link.

The exchange's bookkeeping (who sends which bytes to whom, where they land, how received var-len offsets are rebased) lives in
csrc/tsq_comm_plan.h and is walked without a GPU by tests/test_dist_cpu.py (world sizes 1-8 in one process, and two processes
over gloo).  Round 2's torch.distributed copy of that bookkeeping is gone: nothing in the product called it.
"""
import ctypes as C

from . import _abi as abi
from . import _lib


# ====================================================================== the C-ABI path (RCCL inside libtsq)
import os
import time


_generation = [0]  # communicators created by this process so far: every rank creates them in the same order


def _launch_start_time():
    """start time of the launcher (the parent of every rank); an id file older than that belongs to an earlier, crashed launch"""
    try:
        import psutil
        return psutil.Process(os.getppid()).create_time()
    except Exception:
        pass
    try:  # without psutil: field 22 of /proc/<ppid>/stat = start time in clock ticks since boot, next to the boot time
        with open("/proc/%d/stat" % os.getppid()) as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/stat") as f:
            btime = next(int(l.split()[1]) for l in f if l.startswith("btime"))
        return btime + ticks / float(os.sysconf("SC_CLK_TCK"))
    except Exception:
        return 0.0


def rendezvous_unique_id(lib, rank, world, timeout_s=180.0):
    """rank 0 creates the RCCL unique id, the other ranks of this launch read it from a file named after the launch (the launcher's
    pid, MASTER_PORT / TORCHELASTIC_RUN_ID) AND the number of the communicator inside the process (a second Comm of the same launch
    cannot read the first one's id).  Rank 0 removes what may lie at that path before it writes; a reader ignores a file older than
    the launcher (a crashed earlier launch with a recycled pid); every reader leaves an acknowledgement that rank 0 waits for
    before it unlinks the id in Comm.close()."""
    gen = _generation[0]
    _generation[0] += 1
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "tsq_rdzv_%d_%s_%s_%d" % (os.getppid(), os.environ.get("MASTER_PORT", "0"),
                                                                                   os.environ.get("TORCHELASTIC_RUN_ID", "0"), gen))
    buf = (C.c_uint8 * abi.COMM_ID_BYTES)()
    if world == 1:
        _lib.check(lib.tsq_comm_unique_id(buf))
        return bytes(buf), None
    if rank == 0:
        for stale in [path] + [path + ".ack%d" % r for r in range(1, world)]:
            try:
                os.remove(stale)
            except OSError:
                pass
        _lib.check(lib.tsq_comm_unique_id(buf))
        with open(path + ".tmp", "wb") as f:
            f.write(bytes(buf))
        os.replace(path + ".tmp", path)
        return bytes(buf), path
    t0, born = time.time(), _launch_start_time()
    while True:
        try:
            if os.path.getmtime(path) + 1.0 >= born:
                with open(path, "rb") as f:
                    b = f.read()
                if len(b) == abi.COMM_ID_BYTES:
                    with open(path + ".ack%d" % rank, "wb") as f:
                        f.write(b"1")
                    return b, None
        except (FileNotFoundError, OSError):
            pass
        if time.time() - t0 > timeout_s:
            raise RuntimeError("rendezvous: no unique id at %s after %.0f s" % (path, timeout_s))
        time.sleep(0.01)


class Comm:
    """tsq_comm handle: one per process / GPU."""

    SUM, MAX, MIN = 0, 1, 2

    def __init__(self, ctx, rank=None, world=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        uid, self._rdzv = rendezvous_unique_id(self.lib, self.rank, self.world)
        idbuf = (C.c_uint8 * abi.COMM_ID_BYTES).from_buffer_copy(uid)
        self.h = C.c_void_p()
        _lib.check(self.lib.tsq_comm_create(ctx.h, self.rank, self.world, idbuf, C.byref(self.h)), ctx.h)

    def close(self):
        if self.h:
            self.barrier()
            self.lib.tsq_comm_destroy(self.h)
            self.h = None
            if self._rdzv:  # rank 0: every reader has acknowledged the id (they all passed the barrier above, after reading it)
                for f in [self._rdzv] + [self._rdzv + ".ack%d" % r for r in range(1, self.world)]:
                    try:
                        os.remove(f)
                    except OSError:
                        pass

    def allreduce_i64(self, values, op=0):
        a = (C.c_int64 * len(values))(*values)
        _lib.check(self.lib.tsq_comm_allreduce_i64(self.h, a, len(values), op), self.h)
        return list(a)

    def allreduce_f64(self, values, op=0):
        a = (C.c_double * len(values))(*values)
        _lib.check(self.lib.tsq_comm_allreduce_f64(self.h, a, len(values), op), self.h)
        return list(a)

    def barrier(self):
        _lib.check(self.lib.tsq_comm_barrier(self.h), self.h)

    def info(self):
        """(rank, ranks, library version) as RCCL itself reports them (ncclCommUserRank / ncclCommCount / ncclGetVersion)"""
        r, n, v = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        _lib.check(self.lib.tsq_comm_info(self.h, C.byref(r), C.byref(n), C.byref(v)), self.h)
        return r.value, n.value, v.value

    def redistribute(self, cols, key_col, key_mode, nrows, slot=0):
        """cols: list of abi.Col (device resident).  Returns (received abi.Col array, n_received); call wait(slot) before the
        consumer of the received columns is queued."""
        arr = (abi.Col * len(cols))(*cols)
        out = (abi.Col * len(cols))()
        n = C.c_int64(0)
        _lib.check(self.lib.tsq_redistribute(self.h, arr, len(cols), key_col, key_mode, nrows, slot, out, C.byref(n)), self.h)
        return out, n.value

    def wait(self, slot=0):
        _lib.check(self.lib.tsq_redistribute_wait(self.h, slot), self.h)

    def allgather(self, cols, nrows, slot=0):
        """every rank receives every rank's rows, in rank order (tsq_redistribute with TSQ_KEYMODE_BROADCAST): the small side of a
        broadcast join.  Returns (received abi.Col array, total rows); wait(slot) before the consumer is queued."""
        return self.redistribute(cols, 0, abi.KEYMODE_BROADCAST, nrows, slot)


def col_slice(col, lo, hi):
    """rows [lo, hi) of a device-resident fixed-width column (lo a multiple of 8 when the column has a null bitmap)"""
    c = abi.Col()
    c.data = (col.data or 0) + lo * col.elem_size
    assert not col.null_bitmap or lo % 8 == 0
    c.null_bitmap, c.offsets = (col.null_bitmap + lo // 8) if col.null_bitmap else None, None
    c.length, c.elem_size, c.type, c.flags = hi - lo, col.elem_size, col.type, col.flags
    return c


def redistribute_pieces(comm, cols, key_col, key_mode, nrows, n_pieces, consume, batched_counts=False):
    """The rows in n_pieces pieces (piece c -> slot c); consume(received cols, n) queues the operator's work on the context's stream.

    Default: piece c + 1 is split and put on the wire BEFORE piece c's consumer is queued, so the exchange of c + 1 overlaps the
    operator kernels of c and the split of c + 1 overlaps the wire time of c — one count all-gather (a host-side wait) per piece.
    batched_counts: every piece is split first, ONE all-gather carries the run sizes of all pieces (tsq_redistribute_prepare /
    _counts / _issue), then the exchanges are queued one behind the other with no host-side wait in between; the splits are not
    hidden behind the wire then (the 8-byte split moves 16 B per row: DESIGN.md §6 has the arithmetic of which one wins where)."""
    n_pieces = max(1, min(int(n_pieces), 8))
    bounds = [min(nrows, ((nrows * c // n_pieces) + 63) & ~63) for c in range(n_pieces)] + [nrows]
    if not batched_counts:
        pending = None
        for c in range(n_pieces):
            lo, hi = bounds[c], bounds[c + 1]
            got = comm.redistribute([col_slice(x, lo, hi) for x in cols], key_col, key_mode, hi - lo, slot=c % 8)
            if pending is not None:
                comm.wait(pending[0])
                consume(pending[1], pending[2])
            pending = (c % 8, got[0], got[1])
        comm.wait(pending[0])
        consume(pending[1], pending[2])
        return
    lib, n_cols = comm.lib, len(cols)
    for c in range(n_pieces):
        lo, hi = bounds[c], bounds[c + 1]
        arr = (abi.Col * n_cols)(*[col_slice(x, lo, hi) for x in cols])
        _lib.check(lib.tsq_redistribute_prepare(comm.h, arr, n_cols, key_col, key_mode, hi - lo, c), comm.h)
    slots = (C.c_int32 * n_pieces)(*range(n_pieces))
    _lib.check(lib.tsq_redistribute_counts(comm.h, slots, n_pieces), comm.h)
    got = []
    for c in range(n_pieces):
        out, n = (abi.Col * n_cols)(), C.c_int64(0)
        _lib.check(lib.tsq_redistribute_issue(comm.h, c, out, n_cols, C.byref(n)), comm.h)
        got.append((out, n.value))
    for c in range(n_pieces):
        comm.wait(c)
        consume(got[c][0], got[c][1])


class DistHashJoinCount:
    """SELECT count(*) FROM probe JOIN build ON k across the ranks of `comm`.

    Two plans, chosen by the build side (the same choice on every rank):
    * SHARED IMAGES (tsq_join_build_finish_shared): every rank pushes its own build rows, the packed direct-address images of
      the whole build side are summed across the ranks once, and every rank probes its OWN probe rows — no probe row ever
      crosses xGMI, the probe phase scales with the number of GPUs.  The reference's shape: N join workers probing one shared,
      read-only hash table (executor/join.go:233-239).
    * HASH-RADIX EXCHANGE (the fallback for build sides the images cannot hold: key ranges beyond 31 bits, ...): both sides are
      redistributed by rank(key) (tsq_redistribute), every rank builds and probes what it owns.
    The counts are summed with one 8-byte all-reduce either way."""

    def __init__(self, comm, cfg, radix_mode=None, packing_mode=None, shared=True):
        self.comm, self.lib, self.ctx = comm, comm.lib, comm.ctx
        self.cfg, self.radix_mode, self.packing_mode, self.want_shared = cfg, radix_mode, packing_mode, shared
        self.h = None
        self.shared = False
        self._create()
        self.probed_local = 0
        self.probe_batches = 0
        self.wire_bytes_probe = 0  # bytes of probe rows this rank put on the wire (0 on the shared-images plan)

    def _create(self):
        self.h = C.c_void_p()
        _lib.check(self.lib.tsq_join_create(self.ctx.h, C.byref(self.cfg), C.byref(self.h)), self.ctx.h)
        if self.radix_mode is not None:
            _lib.check(self.lib.tsq_join_set_radix(self.h, self.radix_mode), self.h)
        if self.packing_mode is not None:
            _lib.check(self.lib.tsq_join_set_key_packing(self.h, self.packing_mode), self.h)

    def build(self, cols, key_col, nrows):
        if self.want_shared:
            arr = (abi.Col * len(cols))(*cols)
            if nrows:
                _lib.check(self.lib.tsq_join_build_push(self.h, arr, len(cols), nrows), self.h)
            ok = C.c_int32(0)
            _lib.check(self.lib.tsq_join_build_finish_shared(self.h, self.comm.h, C.byref(ok)), self.h)
            if ok.value:
                self.shared = True
                return nrows
            self.lib.tsq_join_destroy(self.h)  # not packable (every rank got the same answer): the exchange plan, from scratch
            self._create()
        got, n = self.comm.redistribute(cols, key_col, 0, nrows, slot=0)
        self.comm.wait(0)
        if n:
            _lib.check(self.lib.tsq_join_build_push(self.h, got, len(cols), n), self.h)
        _lib.check(self.lib.tsq_join_build_finish(self.h), self.h)
        _lib.check(self.lib.tsq_join_set_count_only(self.h, 1), self.h)
        return n

    def probe(self, cols, key_col, nrows, n_pieces=4, batched_counts=False):
        if self.shared:  # this rank's own rows against the images of the whole build side
            if nrows:
                arr = (abi.Col * len(cols))(*cols)
                _lib.check(self.lib.tsq_join_probe_push(self.h, arr, len(cols), nrows, None), self.h)
                self.probed_local += nrows
                self.probe_batches += 1
            return

        def consume(got, n):
            if n:
                _lib.check(self.lib.tsq_join_probe_push(self.h, got, len(cols), n, None), self.h)
                self.probed_local += n
                self.probe_batches += 1
        redistribute_pieces(self.comm, cols, key_col, 0, nrows, n_pieces, consume, batched_counts)
        self.wire_bytes_probe += nrows * 8 * len(cols) * (self.comm.world - 1) // self.comm.world

    def count(self):
        c = C.c_int64(0)
        _lib.check(self.lib.tsq_join_count(self.h, C.byref(c)), self.h)
        return self.comm.allreduce_i64([c.value])[0]

    def stats(self):
        st = abi.Stats()
        _lib.check(self.lib.tsq_join_stats(self.h, C.byref(st)), self.h)
        return st

    def close(self):
        if self.h:
            self.lib.tsq_join_destroy(self.h)
            self.h = None


def _pull_buffers(ctx, lib, handle, peek, types, n):
    """device buffers for a pull of n rows with the given column types (var-len columns sized by the operator's peek):
    returns (abi.Col array, [(data, bitmap[, offsets, data bytes])])"""
    cols = (abi.Col * len(types))()
    var_bytes = [0] * len(types)
    if n and abi.BYTES in types:
        vb, nr = (C.c_int64 * len(types))(), C.c_int64(0)
        _lib.check(peek(handle, (n + 7) & ~7, C.byref(nr), vb, len(types)), handle)
        var_bytes = list(vb)
    bufs = []
    for i, tp in enumerate(types):
        q = ctx.alloc((max(n, 1) + 7) // 8 + 8)
        if tp == abi.BYTES:
            p, o = ctx.alloc(var_bytes[i] + 64), ctx.alloc((max(n, 1) + 1) * 8 + 64)
            cols[i].offsets, cols[i].elem_size = o, -1
            bufs.append((p, q, o, var_bytes[i]))
        else:
            es = 4 if tp == abi.F32 else 8
            p = ctx.alloc(max(n, 1) * es)
            cols[i].elem_size = es
            bufs.append((p, q))
        cols[i].data, cols[i].null_bitmap, cols[i].length, cols[i].type, cols[i].flags = p, q, n, tp, abi.COL_DEVICE
    return cols, bufs


def dist_hash_agg(comm, partial_cfg, final_cfg, cols, nrows, partial_types, key_col_partial=0, cap=None, out_types=None):
    """HashAggExec across the ranks of `comm`, the reference's own three stages (executor/aggregate.go:96-133):
      1. every rank pre-aggregates ITS rows (partial workers, Partial1Mode): one row per local group;
      2. the partial rows are redistributed by rank(group key) — shuffleIntermData (aggregate.go:352-356) over xGMI;
      3. every rank merges the partial rows of the groups it owns (final workers, FinalMode) and returns them.
    partial_cfg / final_cfg: abi.AggCfg of the two stages; partial_types: column types of the partial rows (the input schema of
    final_cfg); key_col_partial: the group key's column in the partial rows (a string group key travels like any other: the rank
    of its bytes' hash); out_types: types of the final columns (default: 8-byte columns).  Returns (device column buffers —
    (data, bitmap) or (data, bitmap, offsets, data bytes) for a var-len column —, n_groups_local): the caller owns the buffers
    (ctx.free)."""
    lib, ctx = comm.lib, comm.ctx
    hp = C.c_void_p()
    _lib.check(lib.tsq_agg_create(ctx.h, C.byref(partial_cfg), C.byref(hp)), ctx.h)
    pbufs = []
    try:
        arr = (abi.Col * len(cols))(*cols)
        if nrows:
            _lib.check(lib.tsq_agg_push(hp, arr, len(cols), nrows), hp)
        _lib.check(lib.tsq_agg_finish(hp), hp)
        ng = C.c_int64(0)
        _lib.check(lib.tsq_agg_num_groups(hp, C.byref(ng)), hp)
        n_part = ng.value
        pcols, pbufs = _pull_buffers(ctx, lib, hp, lib.tsq_agg_peek, list(partial_types), n_part)
        if n_part:
            n, eos = C.c_int64(0), C.c_int32(0)
            _lib.check(lib.tsq_agg_pull(hp, pcols, len(partial_types), (n_part + 7) & ~7, C.byref(n), C.byref(eos)), hp)
            assert n.value == n_part
    finally:
        lib.tsq_agg_destroy(hp)
    hf = C.c_void_p()
    _lib.check(lib.tsq_agg_create(ctx.h, C.byref(final_cfg), C.byref(hf)), ctx.h)
    try:
        got, n = comm.redistribute(list(pcols), key_col_partial, 1, n_part, slot=0)
        comm.wait(0)
        if n:
            _lib.check(lib.tsq_agg_push(hf, got, len(partial_types), n), hf)
        _lib.check(lib.tsq_agg_finish(hf), hf)
        ng = C.c_int64(0)
        _lib.check(lib.tsq_agg_num_groups(hf, C.byref(ng)), hf)
        n_out = ng.value
        n_out_cols = final_cfg.n_aggs
        otypes = list(out_types) if out_types is not None else [abi.I64] * n_out_cols
        ocols, out_bufs = _pull_buffers(ctx, lib, hf, lib.tsq_agg_peek, otypes, n_out)
        if n_out:
            m, eos = C.c_int64(0), C.c_int32(0)
            _lib.check(lib.tsq_agg_pull(hf, ocols, n_out_cols, (n_out + 7) & ~7, C.byref(m), C.byref(eos)), hf)
            assert m.value == n_out
        ctx.sync()
    finally:
        lib.tsq_agg_destroy(hf)
        for b in pbufs:
            for p in b[:3]:
                ctx.free(p)
    return out_bufs, n_out


# ====================================================================== distributed operator plans on device-resident executors
from . import gpu_pipeline as G  # noqa: E402


class _Collected(G.GpuExecutor):
    """helper: drains a child into ONE device chunk (owned buffers); the exchange executors hand that chunk to the communicator"""

    def __init__(self, ctx, child):
        super().__init__(ctx, child.Schema(), (child,))
        self.child, self.buf, self.n, self.cap = child, None, 0, 0

    def collect(self):
        """-> (owned DeviceColumns of all the child's rows, row count).  A column gets a null bitmap only if some batch brings one
        (the batches of a NOT NULL column have none); bitmaps are appended at arbitrary bit positions (tsq_bitmap_append): a child's
        batches need not be multiples of 8 rows"""
        parts, total = [], 0
        while True:
            chk = self.child.Next()
            m = chk.NumRows()
            if m == 0:
                break
            # the child's buffers are reused by its next Next(): copy the batch (device to device, on the context's stream)
            cols = []
            for src, t in zip(chk.columns, self.types):
                assert not src.var, "the exchange executors of the distributed plans move fixed-width columns"
                dst = G.DeviceColumn(self.ctx, t, m, with_bitmap=src.bitmap is not None)
                _lib.check(self.lib.tsq_copy_d2d(self.ctx.h, dst.data, src.data, m * G._es(src.tp)), self.ctx.h)
                if src.bitmap is not None:
                    _lib.check(self.lib.tsq_copy_d2d(self.ctx.h, dst.bitmap, src.bitmap, (m + 7) // 8), self.ctx.h)
                cols.append(dst)
            parts.append((cols, m))
            total += m
        if len(parts) == 1:
            return parts[0][0], total
        if not parts:
            return [G.DeviceColumn(self.ctx, t, 1, with_bitmap=False) for t in self.types], 0
        out = []
        for i, t in enumerate(self.types):
            nullable = any(cols[i].bitmap is not None for cols, _ in parts)
            dst = G.DeviceColumn(self.ctx, t, total + 64, with_bitmap=nullable)
            at = 0
            for cols, m in parts:
                src = cols[i]
                _lib.check(self.lib.tsq_copy_d2d(self.ctx.h, dst.data + at * G._es(t), src.data, m * G._es(t)), self.ctx.h)
                if nullable:
                    _lib.check(self.lib.tsq_bitmap_append(self.ctx.h, dst.bitmap, at, src.bitmap, m), self.ctx.h)
                at += m
            out.append(dst)
        self.ctx.sync()
        for cols, _ in parts:
            for c in cols:
                c.free()
        return out, total


class GpuBroadcastExec(_Collected):
    """all-gather of the child's rows: every rank's Next() returns the rows of ALL ranks (one chunk).  The small side of a broadcast
    join — a filtered dimension table, the result of an earlier join — goes to every GPU once; the big side is never moved."""

    def __init__(self, ctx, comm, child, slot=0):
        super().__init__(ctx, child)
        self.comm, self.slot, self.done, self.mine = comm, slot, False, None

    def Open(self):
        super().Open()
        self.done = False

    def Next(self):
        if self.done:
            return G.EOS
        self.done = True
        self.mine, n = self.collect()
        got, total = self.comm.allgather([c.col(n) for c in self.mine], n, slot=self.slot)
        self.comm.wait(self.slot)
        self.wire_bytes = (total - n) * sum(G._es(t) for t in self.types)  # received from the other ranks
        if total == 0:
            return G.EOS
        cols = [G.DeviceColumn(self.ctx, t, total, data=g.data, bitmap=g.null_bitmap) for t, g in zip(self.types, got)]
        return G.DeviceChunk(cols, total)

    def Close(self):
        if self.mine:
            for c in self.mine:
                c.free()
            self.mine = None
        super().Close()


class GpuShuffleExec(_Collected):
    """hash-radix redistribute of the child's rows by rank(key column): shuffleIntermData (executor/aggregate.go:352-356) across GPUs.
    key_mode 1: group-key equality (partial aggregate rows meet their group's owner), 0: join-key equality."""

    def __init__(self, ctx, comm, child, key_col, key_mode=1, slot=1):
        super().__init__(ctx, child)
        self.comm, self.key_col, self.key_mode, self.slot, self.done, self.mine = comm, key_col, key_mode, slot, False, None

    def Open(self):
        super().Open()
        self.done = False

    def Next(self):
        if self.done:
            return G.EOS
        self.done = True
        self.mine, n = self.collect()
        got, total = self.comm.redistribute([c.col(n) for c in self.mine], self.key_col, self.key_mode, n, slot=self.slot)
        self.comm.wait(self.slot)
        self.wire_bytes = n * sum(G._es(t) for t in self.types) * (self.comm.world - 1) // self.comm.world
        if total == 0:
            return G.EOS
        cols = [G.DeviceColumn(self.ctx, t, total, data=g.data, bitmap=g.null_bitmap) for t, g in zip(self.types, got)]
        return G.DeviceChunk(cols, total)

    def Close(self):
        if self.mine:
            for c in self.mine:
                c.free()
            self.mine = None
        super().Close()


def dist_q3_plan(ctx, comm, customer_d, orders_d, lineitem_d, seg, day, batch_rows=1 << 24, jit=None):
    """BASELINE configs[4]: the TPC-H Q3-shaped query over tables ROW-SHARDED across the ranks of `comm` (every rank holds 1 / N of
    customer, orders and lineitem, in any order).  The plan moves only what is small:

        customer' = Broadcast( Selection(customer_r, c_mktsegment = seg) -> c_custkey )              all-gather: 3e6 x 8 B at SF 100
        orders'   = Broadcast( Join(Selection(orders_r, o_orderdate < day), customer') -> 3 columns )  all-gather: 1.5e7 x 24 B
        partial   = HashAgg_partial( Projection( Join(Selection(lineitem_r, l_shipdate > day), orders') ) )   lineitem never moves
        result    = HashAgg_final( Shuffle(partial, by l_orderkey) )                                   partial groups: <= 3.1e7 x 32 B / N per rank

    — the reference's shapes: a hash join whose build side is complete before the probe starts (executor/join.go:233-239: every
    worker sees the whole build side), and HashAggExec's partial -> shuffle -> final (executor/aggregate.go:96-133, 352-356).
    Every rank returns the groups it owns (rank(l_orderkey)); the union over the ranks is the query's result.
    Returns (root executor, [join executors], [exchange executors])."""
    from . import expression as E
    from .executor import AggFuncDesc
    F, Col, K = E.ScalarFunction, E.Column, E.Constant
    I, R = abi.I64, abi.F64
    cust = G.GpuSelectionExec(ctx, G.DeviceTableScan(ctx, customer_d, batch_rows), [F("eq", Col(1, I), K(seg))], jit=jit)
    cust_all = GpuBroadcastExec(ctx, comm, G.GpuProjectionExec(ctx, cust, [Col(0, I)], jit=jit), slot=0)
    ords = G.GpuSelectionExec(ctx, G.DeviceTableScan(ctx, orders_d, batch_rows), [F("lt", Col(2, I), K(day))], jit=jit)
    # orders_r (probe: o_orderkey, o_custkey, o_orderdate, o_shippriority) JOIN customer' (build: c_custkey)
    j1 = G.GpuHashJoinExec(ctx, ords, cust_all, [1], [0], abi.JOIN_INNER, 1)
    ords_all = GpuBroadcastExec(ctx, comm, G.GpuProjectionExec(ctx, j1, [Col(0, I), Col(2, I), Col(3, I)], jit=jit), slot=1)
    line = G.GpuSelectionExec(ctx, G.DeviceTableScan(ctx, lineitem_d, batch_rows), [F("gt", Col(1, I), K(day))], jit=jit)
    # lineitem_r (probe: l_orderkey, l_shipdate, price, disc) JOIN orders' (build: o_orderkey, o_orderdate, o_shippriority)
    j2 = G.GpuHashJoinExec(ctx, line, ords_all, [0], [0], abi.JOIN_INNER, 1)
    proj = G.GpuProjectionExec(ctx, j2, [Col(0, I), Col(5, I), Col(6, I), F("mul", Col(2, R), F("minus", K(1.0), Col(3, R)))], jit=jit)
    P1, FIN = abi.MODE_PARTIAL1, abi.MODE_FINAL
    partial = G.GpuHashAggExec(ctx, proj, [0, 1, 2], [AggFuncDesc(abi.AGG_FIRSTROW, 0, I, P1), AggFuncDesc(abi.AGG_FIRSTROW, 1, I, P1),
                                                       AggFuncDesc(abi.AGG_FIRSTROW, 2, I, P1), AggFuncDesc(abi.AGG_SUM, 3, R, P1)])
    shuffled = GpuShuffleExec(ctx, comm, partial, 0, key_mode=1, slot=2)
    final = G.GpuHashAggExec(ctx, shuffled, [0, 1, 2], [AggFuncDesc(abi.AGG_FIRSTROW, 0, I, FIN), AggFuncDesc(abi.AGG_FIRSTROW, 1, I, FIN),
                                                        AggFuncDesc(abi.AGG_FIRSTROW, 2, I, FIN), AggFuncDesc(abi.AGG_SUM, 3, R, FIN)])
    return final, [j1, j2], [cust_all, ords_all, shuffled]
