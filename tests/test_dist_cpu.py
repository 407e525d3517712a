"""CPU: the N>1 path's bookkeeping — the part of tsq_redistribute (csrc/tsq_comm.hip) that no single-GPU box can exercise.

csrc/tsq_comm_plan.h turns the gathered count matrix into the transfers and offset shifts one rank performs; tsq_comm.hip executes
them with RCCL.  (1) tests/hostsim/comm_sim.cpp walks that header for world sizes 2, 4 and 8 inside one process (memcpy as the
wire): ragged row counts, empty ranks, nullable-on-one-rank columns, var-len columns, a hot key.  (2) Two real processes run the
same plan over gloo (tests/dist_worker.py) and the local joins of what they received add up to the whole join."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim():
    lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "hostsim.so"))
    lib.sim_comm_exchange.restype = C.c_int32
    lib.sim_comm_exchange.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_uint64, C.POINTER(C.c_int64), C.c_uint64, C.c_int32, C.c_char_p, C.c_int32]
    return lib


def _run(sim, world, kinds, nullable_mask, rows, seed, skew=0):
    err = C.create_string_buffer(256)
    rc = sim.sim_comm_exchange(world, len(kinds), (C.c_int32 * len(kinds))(*kinds), nullable_mask, (C.c_int64 * world)(*rows), seed, skew, err, 256)
    assert rc == 0, (rc, err.value.decode())


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("kinds", [[8], [8, 8], [8, 0], [8, 4, 0, 8, 0]], ids=["key", "key+v", "key+str", "5cols"])
def test_exchange_plan_delivers_every_row_to_its_owner(sim, world, kinds):
    n = len(kinds)
    rows = [1000 + 37 * r for r in range(world)]
    _run(sim, world, kinds, 0, rows, 1)                                   # no NULLs anywhere: no NOT-NULL bytes travel
    _run(sim, world, kinds, (1 << (n * world)) - 1, rows, 2)              # every column nullable on every rank
    _run(sim, world, kinds, 1 << (n * (world - 1)), rows, 3)              # only the LAST rank's key column holds NULLs: nullable for all
    _run(sim, world, kinds, 0b10 if n > 1 else 0, rows, 4, skew=1)        # a hot key: one rank receives most of the rows


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_plan_ragged_and_empty_ranks(sim, world):
    kinds = [8, 0, 8]
    _run(sim, world, kinds, 0b111, [0] * world, 5)                        # nobody has rows
    _run(sim, world, kinds, 0b111, [5000] + [0] * (world - 1), 6)         # one rank holds everything
    _run(sim, world, kinds, 0b111 << 3, [0] + [300] * (world - 1), 7)     # rank 0 sends nothing but receives
    _run(sim, world, kinds, 0, [1] * world, 8)                            # one row each: most runs are empty


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_broadcast_mode_is_an_all_gather_of_the_columns(sim, world):
    # key_mode 2 of tsq_redistribute (the small sides of a broadcast join: tinysql_amd/parallel.py, dist_q3): every rank receives every
    # rank's rows in rank order; fixed-width, nullable and var-len columns, ragged and empty ranks
    kinds = [8, 0, 4, 8]
    _run(sim, world, kinds, 0, [700 + 13 * r for r in range(world)], 11, skew=2)
    _run(sim, world, kinds, (1 << (len(kinds) * world)) - 1, [0 if r % 2 else 900 for r in range(world)], 12, skew=2)
    _run(sim, world, kinds, 0b0110, [1] * world, 13, skew=2)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_count_exchange_of_several_pieces_at_once(sim, world):
    # tsq_redistribute_counts: pieces with different vector lengths (a piece with var-len columns carries their byte counts too)
    sim.sim_comm_counts.restype = C.c_int32
    sim.sim_comm_counts.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_uint64]
    for ls in ([world + 1], [world + 1] * 4, [world + 1, 3 * world + 1, world + 1, 2 * world + 1, world + 1, world + 1, world + 1, 5 * world + 1]):
        assert sim.sim_comm_counts(world, len(ls), (C.c_int32 * len(ls))(*ls), 99 + world) == 0


def test_world_size_2_two_processes_run_the_shipped_plan_over_gloo():
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "DIST_OK" in p.stdout
