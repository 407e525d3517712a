"""Multi-GPU hash-radix redistribute: one process per GPU, torch.distributed (backend "nccl" == RCCL
on ROCm) all-to-all(v) over xGMI.

CPU analogue in the reference: HashAggExec's partial->final shuffle (executor/aggregate.go:352-356)
and the probe-chunk dispatch of HashJoinExec (executor/join.go:219).  Equi-join and GROUP BY are
partitionable by any function of the key, so the ONLY data-path collective is this exchange:
rank r keeps/receives every row whose key ranks to r (tsq_key_rank), then runs the single-GPU
operator locally.  xGMI is point-to-point (7 links per GPU), so an all-to-all drives all links at
once; ring-style collectives would be bound by one link.

The functions below only move torch tensors (device tensors with nccl, CPU tensors with gloo in the
CPU test-suite); the split itself is libtsq's tsq_radix_split on the GPU.
"""
import ctypes as C

from . import _abi as abi
from . import _lib


def exchange_counts(dist, torch, send_counts, device):
    """all-to-all of the per-destination row counts; returns the per-source counts this rank receives."""
    w = dist.get_world_size()
    send = torch.tensor(list(send_counts), dtype=torch.int64, device=device)
    recv = torch.empty(w, dtype=torch.int64, device=device)
    dist.all_to_all_single(recv, send)
    return [int(x) for x in recv.tolist()]


def exchange_runs(dist, torch, tensors, send_counts, recv_counts):
    """all-to-all(v) of contiguous runs.  tensors[i] holds this rank's rows already grouped by
    destination (run p = rows for rank p, send_counts[p] rows); returns the received tensors."""
    out = []
    total = sum(recv_counts)
    for t in tensors:
        r = torch.empty(max(total, 1), dtype=t.dtype, device=t.device)[:total]
        dist.all_to_all_single(r, t[: sum(send_counts)], list(recv_counts), list(send_counts))
        out.append(r)
    return out


def dev_col_from_tensor(t, tp, nrows, bitmap=None):
    """tsq_col view of a device torch tensor (plain pointer + size: no torch types cross the C-ABI)."""
    c = abi.Col()
    c.data = t.data_ptr()
    c.null_bitmap = bitmap.data_ptr() if bitmap is not None else None
    c.offsets = None
    c.length = nrows
    c.elem_size = 4 if tp == abi.F32 else 8
    c.type = tp
    c.flags = abi.COL_DEVICE
    return c


def radix_split(ctx, cols, key_col, key_mode, nrows, n_parts, out_cols):
    """tsq_radix_split wrapper: returns the per-part row counts (host ints)."""
    counts = (C.c_int64 * n_parts)()
    arr_in = (abi.Col * len(cols))(*cols)
    arr_out = (abi.Col * len(out_cols))(*out_cols)
    _lib.check(ctx.lib.tsq_radix_split(ctx.h, arr_in, len(cols), key_col, key_mode, nrows, n_parts, arr_out, counts), ctx.h)
    return list(counts)


def redistribute(ctx, dist, torch, tensors, types, key_col, key_mode, nrows):
    """split by rank(key) on the GPU, exchange counts, all-to-all the runs.
    tensors: device torch tensors (one per column, no NULLs).  Returns (received tensors, n_received)."""
    w = dist.get_world_size()
    outs = [torch.empty_like(t) for t in tensors]
    cols = [dev_col_from_tensor(t, tp, nrows) for t, tp in zip(tensors, types)]
    ocols = [dev_col_from_tensor(t, tp, nrows) for t, tp in zip(outs, types)]
    send_counts = radix_split(ctx, cols, key_col, key_mode, nrows, w, ocols)
    recv_counts = exchange_counts(dist, torch, send_counts, tensors[0].device)
    got = exchange_runs(dist, torch, outs, send_counts, recv_counts)
    return got, sum(recv_counts)


def redistribute_pipelined(ctx, dist, torch, tensors, types, key_col, key_mode, nrows, n_chunks=4, split=None):
    """redistribute() in n_chunks pieces, as a generator of (received tensors, n_received): the rows are split chunk by
    chunk, ONE count exchange covers all chunks, all data exchanges are issued asynchronously back to back, and piece c
    is handed to the caller as soon as its exchange has completed — the single-GPU operator consumes piece c while pieces
    c+1.. are still on the wire (xGMI moves 8 B/row at a fraction of what the probe kernel consumes, so the wire is the
    longer leg; only the last piece's probe is exposed).  The union of the pieces is the same multiset of rows that
    redistribute() returns.  `split(cols_as_tensors, lo, hi, n_parts) -> (out tensors, counts)` replaces the GPU split
    in the CPU (gloo) test-suite."""
    w = dist.get_world_size()
    n_chunks = max(1, int(n_chunks))  # every rank must use the same value: the count exchange carries w * n_chunks words
    bounds = [((nrows * c // n_chunks) + 7) & ~7 for c in range(n_chunks)] + [nrows]
    bounds = [min(b, nrows) for b in bounds]
    pieces, counts = [], []
    for c in range(n_chunks):
        lo, hi = bounds[c], bounds[c + 1]
        if split is not None:
            outs, cnt = split(tensors, lo, hi, w)
        else:
            outs = [torch.empty(max(hi - lo, 1), dtype=t.dtype, device=t.device)[: hi - lo] for t in tensors]
            cols = [dev_col_from_tensor(t[lo:hi], tp, hi - lo) for t, tp in zip(tensors, types)]
            ocols = [dev_col_from_tensor(t, tp, hi - lo) for t, tp in zip(outs, types)]
            cnt = radix_split(ctx, cols, key_col, key_mode, hi - lo, w, ocols) if hi > lo else [0] * w
        pieces.append(outs)
        counts.append(cnt)
    # one exchange for all counts: send[p * n_chunks + c] = rows of chunk c that go to rank p
    dev = tensors[0].device
    send = torch.tensor([counts[c][p] for p in range(w) for c in range(n_chunks)], dtype=torch.int64, device=dev)
    recv = torch.empty(w * n_chunks, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send)
    recv = recv.tolist()
    inflight = []
    for c in range(n_chunks):
        rc = [int(recv[s * n_chunks + c]) for s in range(w)]
        total = sum(rc)
        got, works = [], []
        for t in pieces[c]:
            r = torch.empty(max(total, 1), dtype=t.dtype, device=t.device)[:total]
            works.append(dist.all_to_all_single(r, t[: sum(counts[c])], rc, list(counts[c]), async_op=True))
            got.append(r)
        inflight.append((got, total, works))
    for got, total, works in inflight:
        for wk in works:
            wk.wait()  # nccl: the current stream waits for the exchange; gloo: the host does
        yield got, total
    del pieces  # send buffers stay alive until every exchange has been waited for
