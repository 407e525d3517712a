"""Worker for tests/test_dist_cpu.py: world_size-2 gloo run of the N>1 redistribute logic
(tinysql_amd/parallel.py) with CPU stand-ins for the two GPU pieces (numpy split in place of
tsq_radix_split, the oracle join in place of the HIP join)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding as orc  # noqa: E402
from tests import gpu_helpers as G  # noqa: E402
from tests import helpers as H  # noqa: E402
from tinysql_amd import _abi as abi  # noqa: E402
from tinysql_amd import parallel  # noqa: E402
from tinysql_amd.chunk import Chunk, Column  # noqa: E402


def np_mix64(k):
    with np.errstate(over="ignore"):
        k = k ^ (k >> np.uint64(33))
        k = k * np.uint64(0xFF51AFD7ED558CCD)
        k = k ^ (k >> np.uint64(33))
        k = k * np.uint64(0xC4CEB9FE1A85EC53)
        return k ^ (k >> np.uint64(33))


def np_rank(keys_u64, parts):
    return (((np_mix64(keys_u64) & np.uint64(0xFFFF)) * np.uint64(parts)) >> np.uint64(16)).astype(np.int64)


def split(cols, key, parts):
    r = np_rank(cols[key].view(np.uint64), parts)
    order = np.argsort(r, kind="stable")
    counts = np.bincount(r, minlength=parts).tolist()
    return [c[order] for c in cols], counts


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nb, npr = 20000, 30000
    rng = np.random.default_rng(1000 + rank)
    bk = rng.integers(0, 5000, nb).astype(np.int64)
    bv = rng.integers(-99, 99, nb).astype(np.int64)
    pk = rng.integers(0, 6000, npr).astype(np.int64)
    pv = rng.integers(-99, 99, npr).astype(np.int64)

    def redistribute(cols):
        parts, counts = split(cols, 0, world)
        rc = parallel.exchange_counts(dist, torch, counts, "cpu")
        got = parallel.exchange_runs(dist, torch, [torch.from_numpy(np.ascontiguousarray(c)) for c in parts], counts, rc)
        return [g.numpy() for g in got], counts, rc

    (rbk, rbv), sc, rc = redistribute([bk, bv])
    (rpk, rpv), _, _ = redistribute([pk, pv])
    # every received key ranks to this rank; nothing lost
    assert (np_rank(rbk.view(np.uint64), world) == rank).all() and (np_rank(rpk.view(np.uint64), world) == rank).all()
    tot = torch.tensor([len(rbk), len(rpk)], dtype=torch.int64)
    dist.all_reduce(tot)
    assert tot.tolist() == [nb * world, npr * world]

    # the pipelined exchange (bench.py's N>1 step) delivers the same multiset of rows in pieces; ranks own different row
    # counts here (rank 1 holds 7 rows fewer)
    def split_t(tensors, lo, hi, parts):
        cols, counts = split([t[lo:hi].numpy() for t in tensors], 0, parts)
        return [torch.from_numpy(np.ascontiguousarray(c)) for c in cols], counts

    n_mine = npr - 7 * rank
    got_k, got_v, pieces = [], [], 0
    for (k, v), n in parallel.redistribute_pipelined(None, dist, torch, [torch.from_numpy(pk[:n_mine]), torch.from_numpy(pv[:n_mine])],
                                                     [abi.I64, abi.I64], 0, 0, n_mine, 5, split=split_t):
        assert len(k) == len(v) == n
        got_k.append(k.numpy().copy())
        got_v.append(v.numpy().copy())
        pieces += 1
    assert pieces == 5
    gk, gv = np.concatenate(got_k), np.concatenate(got_v)
    assert (np_rank(gk.view(np.uint64), world) == rank).all()
    allp2 = [None] * world
    dist.all_gather_object(allp2, (pk[:npr - 7 * rank], pv[:npr - 7 * rank]))
    wk = np.concatenate([a[0] for a in allp2])
    wv = np.concatenate([a[1] for a in allp2])
    mine = np_rank(wk.view(np.uint64), world) == rank
    assert sorted(zip(gk.tolist(), gv.tolist())) == sorted(zip(wk[mine].tolist(), wv[mine].tolist()))

    cfg = H.join_cfg([abi.I64] * 2, [abi.I64] * 2, [0], [0], abi.JOIN_INNER, 1)
    local = orc.hash_join(cfg, Chunk([Column(abi.I64, rbk), Column(abi.I64, rbv)]), Chunk([Column(abi.I64, rpk), Column(abi.I64, rpv)]))
    s, x = orc.rows_checksum(local)
    agg = torch.tensor([local.NumRows(), s & 0x7FFFFFFFFFFFFFFF, s >> 63], dtype=torch.int64)
    dist.all_reduce(agg)
    xs = [None] * world
    dist.all_gather_object(xs, x)

    # reference: the whole (unpartitioned) join on rank 0
    allb = [None] * world
    allp = [None] * world
    dist.all_gather_object(allb, (bk, bv))
    dist.all_gather_object(allp, (pk, pv))
    if rank == 0:
        B = Chunk([Column(abi.I64, np.concatenate([b[0] for b in allb])), Column(abi.I64, np.concatenate([b[1] for b in allb]))])
        P = Chunk([Column(abi.I64, np.concatenate([p[0] for p in allp])), Column(abi.I64, np.concatenate([p[1] for p in allp]))])
        whole = orc.hash_join(cfg, B, P)
        ws, wx = orc.rows_checksum(whole)
        assert whole.NumRows() == int(agg[0]), (whole.NumRows(), int(agg[0]))
        gx = 0
        for v in xs:
            gx ^= v
        assert gx == wx
        gs = (int(agg[1]) + (int(agg[2]) << 63)) & ((1 << 64) - 1)
        assert gs == ws
        print("DIST_OK rows=%d" % whole.NumRows())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
