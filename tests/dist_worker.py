"""Worker for tests/test_dist_cpu.py: a world-size-2 (or more) gloo run of the SHIPPED exchange bookkeeping.

tsq_redistribute (csrc/tsq_comm.hip) = split on the GPU + count all-gather + the transfers and offset shifts that
csrc/tsq_comm_plan.h computes from the gathered matrix.  Here two real processes run that plan: the count vectors are
all-gathered over gloo, every process gets ITS plan from the same header (tests/hostsim: sim_comm_plan), and the transfers are
executed with torch.distributed send / recv on CPU byte tensors in the plan's issue order (a fixed-width nullable key, a double,
a var-len column).  The split kernel is replaced by a stable numpy partition by tsq_key_rank, the wire by gloo — nothing else.
Then the local join of what every rank received equals the whole join (the oracle, on rank 0)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding as orc  # noqa: E402
from tests import helpers as H  # noqa: E402
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd.chunk import Chunk, Column, StrColumn  # noqa: E402

SIM = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "hostsim.so"))
SIM.sim_key_rank.restype = C.c_uint32
SIM.sim_key_rank.argtypes = [C.c_uint64, C.c_uint32]
DATA, OFFS, NOTNULL = 0, 1, 2


def key_ranks(keys, notnull, world):
    return np.array([SIM.sim_key_rank(int(k) & ((1 << 64) - 1), world) if nn else 0 for k, nn in zip(keys.tolist(), notnull.tolist())], dtype=np.int64)


def redistribute(rank, world, cols):
    """cols: list of ('fixed', np array of 8-byte cells, notnull bool[] or None) / ('var', list of bytes-or-None).  Column 0 is the key.
    Returns the received columns in the same form."""
    n = len(cols[0][1])
    knn = cols[0][2] if cols[0][2] is not None else np.ones(n, bool)
    dest = key_ranks(cols[0][1].view(np.int64), knn, world)
    order = np.argsort(dest, kind="stable")
    sendc = np.bincount(dest, minlength=world).tolist()
    es = [8 if c[0] == "fixed" else 0 for c in cols]
    var_cols = [i for i, c in enumerate(cols) if c[0] == "var"]
    L = world + 1 + len(var_cols) * world
    # ---- the split: send buffers per (column, kind) as byte arrays
    send = {}
    mask = 0
    for i, c in enumerate(cols):
        if c[0] == "fixed":
            send[(i, DATA)] = np.ascontiguousarray(c[1][order]).view(np.uint8)
            nn = c[2][order] if c[2] is not None else np.ones(n, bool)
            if c[2] is not None:
                mask |= 1 << i
        else:
            vals = [c[1][j] for j in order.tolist()]
            offs = np.zeros(n + 1, dtype=np.int64)
            np.cumsum([0 if v is None else len(v) for v in vals], out=offs[1:])
            send[(i, OFFS)] = offs.view(np.uint8)
            send[(i, DATA)] = np.frombuffer(b"".join(v for v in vals if v is not None), dtype=np.uint8).copy()
            nn = np.array([v is not None for v in vals], dtype=bool)
            if not nn.all():
                mask |= 1 << i
        send[(i, NOTNULL)] = nn.astype(np.uint8)
    vec = np.zeros(L, dtype=np.int64)
    vec[:world] = sendc
    vec[world] = mask
    for v, i in enumerate(var_cols):
        offs = send[(i, OFFS)].view(np.int64)
        row = 0
        for p in range(world):
            vec[world + 1 + v * world + p] = offs[row + sendc[p]] - offs[row]
            row += sendc[p]
    gathered = [torch.zeros(L, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(vec))  # = the ncclAllGather of the count vectors
    M = np.concatenate([g.numpy() for g in gathered]).astype(np.uint64)
    # ---- this rank's plan, from the header tsq_comm.hip executes
    cap = 16 * len(cols) * world + 64
    xf, sh = np.zeros(7 * cap, dtype=np.int64), np.zeros(5 * cap, dtype=np.int64)
    total, pmask, nx, ns = C.c_int64(0), C.c_uint64(0), C.c_int32(0), C.c_int32(0)
    rbytes = np.zeros(len(cols), dtype=np.int64)
    rc = SIM.sim_comm_plan(rank, world, len(cols), (C.c_int32 * len(cols))(*es), M.ctypes.data_as(C.c_void_p), C.byref(total), C.byref(pmask),
                           rbytes.ctypes.data_as(C.c_void_p), xf.ctypes.data_as(C.c_void_p), cap, C.byref(nx), sh.ctypes.data_as(C.c_void_p), cap, C.byref(ns))
    assert rc == 0
    total = total.value
    recv = {}
    for i, c in enumerate(cols):
        recv[(i, DATA)] = np.full(total * 8 if c[0] == "fixed" else int(rbytes[i]), 0xEE, dtype=np.uint8)
        recv[(i, OFFS)] = np.full((total + world + 1) * 8, 0xEE, dtype=np.uint8)
        recv[(i, NOTNULL)] = np.full(total, 0xEE, dtype=np.uint8)
    # ---- the group of sends and receives, in the plan's order (isend / irecv = inside ncclGroupStart / End)
    works, keep = [], []
    for k in range(nx.value):
        col, kind, peer, so, sl, ro, rl = xf[7 * k: 7 * k + 7].tolist()
        if peer == rank:
            assert sl == rl
            recv[(col, kind)][ro:ro + rl] = send[(col, kind)][so:so + sl]
            continue
        if sl:
            t = torch.from_numpy(np.ascontiguousarray(send[(col, kind)][so:so + sl]))
            keep.append(t)
            works.append(dist.isend(t, peer))
        if rl:
            t = torch.from_numpy(recv[(col, kind)][ro:ro + rl])  # a view: the bytes land in place
            works.append(dist.irecv(t, peer))
    for w in works:
        w.wait()
    out = []
    for i, c in enumerate(cols):
        nn = recv[(i, NOTNULL)].astype(bool) if (pmask.value >> i) & 1 else np.ones(total, bool)
        if c[0] == "fixed":
            out.append(("fixed", recv[(i, DATA)].view(np.int64).copy(), nn))
        else:
            tmp = recv[(i, OFFS)].view(np.int64)
            offs = np.full(total + 1, -1, dtype=np.int64)
            offs[0] = 0
            for k in range(ns.value):
                scol, src, dst, rows, delta = sh[5 * k: 5 * k + 5].tolist()
                if scol == i:
                    offs[dst:dst + rows] = tmp[src:src + rows] + delta
            assert (np.diff(offs) >= 0).all() and offs[total] == len(recv[(i, DATA)])
            data = recv[(i, DATA)].tobytes()
            out.append(("var", [data[offs[j]:offs[j + 1]] if nn[j] else None for j in range(total)]))
    return out


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nb, npr = 20000 - 13 * rank, 30000 + 7 * rank  # ragged: the ranks own different row counts
    rng = np.random.default_rng(1000 + rank)

    def table(n, keys_hi, null_keys):
        k = rng.integers(0, keys_hi, n).astype(np.int64)
        knn = rng.random(n) > 0.03 if null_keys else None
        v = rng.random(n).view(np.int64)
        s = [None if rng.random() < 0.1 else bytes(rng.integers(97, 123, int(rng.integers(0, 12)), dtype=np.uint8)) for _ in range(n)]
        return [("fixed", k, knn), ("fixed", v, None), ("var", s)]

    # rank 1's build keys carry NULLs, rank 0's do not: the column must travel as nullable on BOTH
    build, probe = table(nb, 5000, rank == 1), table(npr, 6000, True)
    rb, rp = redistribute(rank, world, build), redistribute(rank, world, probe)
    for got in (rb, rp):  # every received key ranks to this rank (NULL keys to rank 0); nothing lost
        assert (key_ranks(got[0][1], got[0][2], world) == rank).all()
    tot = torch.tensor([len(rb[0][1]), len(rp[0][1])], dtype=torch.int64)
    dist.all_reduce(tot)
    nbs, nps = torch.tensor([nb]), torch.tensor([npr])
    dist.all_reduce(nbs)
    dist.all_reduce(nps)
    assert tot.tolist() == [int(nbs), int(nps)]

    def chunk(t):
        return Chunk([Column(abi.I64, t[0][1], t[0][2]), Column(abi.F64, t[1][1].view(np.float64)), StrColumn(t[2][1])])

    types = [abi.I64, abi.F64, abi.BYTES]
    cfg = H.join_cfg(types, types, [0], [0], abi.JOIN_INNER, 1)
    local = orc.hash_join(cfg, chunk(rb), chunk(rp))
    mine = sorted(H.multiset(local))
    allrows, allb, allp = [None] * world, [None] * world, [None] * world
    dist.all_gather_object(allrows, mine)
    dist.all_gather_object(allb, build)
    dist.all_gather_object(allp, probe)
    if rank == 0:  # reference: the whole (unpartitioned) join
        def cat(parts):
            return [("fixed", np.concatenate([p[0][1] for p in parts]), np.concatenate([p[0][2] if p[0][2] is not None else np.ones(len(p[0][1]), bool) for p in parts])),
                    ("fixed", np.concatenate([p[1][1] for p in parts]), None), ("var", sum((p[2][1] for p in parts), []))]
        whole = orc.hash_join(cfg, chunk(cat(allb)), chunk(cat(allp)))
        got = sorted(sum(allrows, []), key=lambda t: tuple((x is None, str(x)) for x in t))
        assert got == H.multiset(whole), (len(got), whole.NumRows())
        print("DIST_OK rows=%d" % whole.NumRows())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
