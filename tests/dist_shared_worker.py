"""Worker for tests/test_shared_plan_cpu.py: a world-size-2 (or more) gloo run of the SHARED-IMAGES join plan
(tsq_join_build_finish_shared, csrc/tsq_join.hip: da_prepare with a communicator).

Real processes, real collectives: the key range and the usable rows are all-reduced (MIN / SUM over int64, as the product does over
RCCL), every process asks the product's host arithmetic for the plan (tests/hostsim: tsq_da_plan through sim_shared_rank_plan),
assembles the image of ITS build rows, the images are summed with ONE all-reduce (uint8 elements for byte cells, int32 for the words
of bit cells: the arithmetic of ncclSum), every process checks the population of the sum (tsq_da_shared_images_ok) and probes its OWN
probe rows.  The kernels are loops here and gloo is the wire; the decisions are the product's.  The sum of the per-rank counts must
equal the whole-table join (numpy over the union of all ranks' rows, recomputed on every rank from the seeds)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "hostsim.so"))
I64P, U8P = C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
SIM.sim_shared_rank_plan.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32, I64P]
SIM.sim_shared_rank_image.argtypes = [I64P, C.c_int64, I64P, U8P]
SIM.sim_shared_rank_image.restype = C.c_int32
SIM.sim_shared_population.argtypes = [U8P, C.c_int64, C.c_int32]
SIM.sim_shared_population.restype = C.c_int64
SIM.sim_shared_images_ok.argtypes = [C.c_int64, C.c_int64]
SIM.sim_shared_images_ok.restype = C.c_int32
SIM.sim_shared_rank_probe.argtypes = [I64P, C.c_int64, I64P, U8P]
SIM.sim_shared_rank_probe.restype = C.c_int64


def ptr(a, t):
    return a.ctypes.data_as(t)


def tables(case, world):
    """every rank's (build keys, probe keys) of a case, from fixed seeds — any rank can recompute the union for the expected count"""
    out = []
    for r in range(world):
        rng = np.random.default_rng(1000 * case + r)
        if case == 0:    # duplicates inside and across ranks: byte cells
            b = rng.integers(-20_000, 40_000, 30_000 + 777 * r)
            p = rng.integers(-40_000, 80_000, 50_000 + r)
        elif case == 1:  # a unique build side spread over 29 bits: bit cells; rank 1 holds no build rows at all
            allb = (np.arange(200_000, dtype=np.int64) * 2654 + 17) % (1 << 29)
            allb = np.unique(allb)
            b = allb[0::1] if (r == 0 and world == 2) else allb[r::world]
            if world == 2 and r == 1:
                b = allb[:0]
            p = rng.choice(allb, 40_000) + rng.integers(0, 2, 40_000)
        elif case == 2:  # the same, but ONE key lives on two ranks: the bit of its cell carries in the sum -> the population check fails everywhere
            allb = np.unique((np.arange(100_000, dtype=np.int64) * 7919 + 5) % (1 << 29))
            b = allb[r::world]
            if r == world - 1:
                b = np.concatenate([b, allb[:1]])
            p = rng.choice(allb, 20_000)
        else:            # a hot key with 300 rows on each rank: its byte cell wraps in the sum (2 x 300 = 600 > 255 ... mod 256)
            b = np.concatenate([rng.integers(0, 50_000, 20_000), np.full(120, 4242)])
            p = rng.integers(0, 50_000, 30_000)
        out.append((np.ascontiguousarray(b, dtype=np.int64), np.ascontiguousarray(p, dtype=np.int64)))
    return out


def expected(tabs):
    allb = np.concatenate([t[0] for t in tabs])
    keys, mult = np.unique(allb, return_counts=True)
    want = 0
    for _, p in tabs:
        idx = np.searchsorted(keys, p)
        idx[idx >= len(keys)] = 0
        hit = keys[idx] == p if len(keys) else np.zeros(len(p), bool)
        want += int(mult[idx][hit].sum()) if len(keys) else 0
    return want


def shared_join(rank, world, b, p, force):
    """-> (shared, count of this rank or None)"""
    big = np.iinfo(np.int64).max
    rng_t = torch.tensor([int(b.min()) if len(b) else big, -int(b.max()) if len(b) else big], dtype=torch.int64)
    dist.all_reduce(rng_t, op=dist.ReduceOp.MIN)
    usable = torch.tensor([len(b)], dtype=torch.int64)
    dist.all_reduce(usable, op=dist.ReduceOp.SUM)
    if usable.item() == 0:
        return False, None
    plan = np.zeros(8, dtype=np.int64)
    SIM.sim_shared_rank_plan(int(rng_t[0]), -int(rng_t[1]), int(usable.item()), 1 if force else 0, ptr(plan, I64P))
    if not plan[0]:
        return False, None
    img = np.zeros(int(plan[7]), dtype=np.uint8)
    lf = SIM.sim_shared_rank_image(ptr(b, I64P), len(b), ptr(plan, I64P), ptr(img, U8P))
    assert lf >= 0, "a build key outside the all-reduced range"
    flags = torch.tensor([lf], dtype=torch.int64)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX)  # every rank learns of a local overflow
    if flags.item():
        return False, None
    t = torch.from_numpy(img.view(np.int32)) if plan[1] else torch.from_numpy(img)  # bit cells travel as 32-bit words
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    pop = SIM.sim_shared_population(ptr(img, U8P), len(img), int(plan[1]))
    if not SIM.sim_shared_images_ok(pop, int(usable.item())):
        return False, None
    return True, int(SIM.sim_shared_rank_probe(ptr(p, I64P), len(p), ptr(plan, I64P), ptr(img, U8P)))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    verdicts = []
    for case, want_shared in ((0, True), (1, True), (2, False), (3, None)):
        tabs = tables(case, world)
        b, p = tabs[rank]
        shared, got = shared_join(rank, world, b, p, force=True)
        v = torch.tensor([1 if shared else 0], dtype=torch.int64)
        lo, hi = v.clone(), v.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert lo.item() == hi.item(), "case %d: the ranks disagree about the plan" % case
        if want_shared is not None:
            assert shared == want_shared, (case, shared)
        if shared:
            total = torch.tensor([got], dtype=torch.int64)
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
            want = expected(tabs)
            assert total.item() == want, (case, total.item(), want)
        elif case == 3:
            pass  # 2 x 120 rows of one key fit a byte cell (240): shared; more ranks wrap it: not shared — either is correct, and agreed
        verdicts.append((case, shared))
    dist.barrier()
    if rank == 0:
        print("SHARED_OK", verdicts)
    dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
