"""CPU: the SHARED-IMAGES join plan (tsq_join_build_finish_shared) walked without a GPU for world sizes 1, 2, 4 and 8.

The plan: every rank assembles the packed direct-address images of ITS build rows over the key range of the WHOLE build side, the
images are summed across the ranks once (one all-reduce per build side), every rank probes its OWN probe rows — nothing crosses
xGMI in the probe phase.  tests/hostsim/shared_sim.cpp runs the host arithmetic of the product (tsq_da_plan, tsq_da_mix,
tsq_da_shared_images_ok, the wire-byte formulas: csrc/tsq_dapack.h, csrc/tsq_comm_plan.h) with loops in place of the kernels and
the arithmetic of ncclSum on uint8 / uint32 in place of the wire, and compares the plan's count with a hash map over the whole
build side.  The real path (HIP + RCCL) is tests/dist_gpu_worker.py."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim():
    lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "hostsim.so"))
    lib.sim_shared_join.restype = C.c_int32
    lib.sim_shared_join.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_uint64, C.POINTER(C.c_int64), C.c_char_p, C.c_int32]
    lib.sim_shared_projection.restype = C.c_int32
    lib.sim_shared_projection.argtypes = [C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
    return lib


def run(sim, world, brows, prows, kind, key_lo, span, stride=1, hot=0, force=1, seed=1):
    out = (C.c_int64 * 8)()
    err = C.create_string_buffer(256)
    rc = sim.sim_shared_join(world, (C.c_int64 * world)(*brows), (C.c_int64 * world)(*prows), kind, key_lo, span, stride, hot, force, seed, out, err, 256)
    assert rc == 0, (rc, err.value.decode())
    return dict(shared=out[0], got=out[1], want=out[2], bits=out[3], image_bytes=out[4], b=out[5], build_wire=out[6], probe_wire=out[7])


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_byte_cells_duplicates_inside_and_across_ranks(sim, world):
    r = run(sim, world, [20_000 + 311 * i for i in range(world)], [50_000 + 7 * i for i in range(world)], 0, -30_000, 70_000, seed=10 + world)
    assert r["shared"] == 1 and r["bits"] == 0 and r["got"] == r["want"] > 0 and r["probe_wire"] == 0
    assert r["build_wire"] == (0 if world == 1 else 2 * r["image_bytes"] * (world - 1) // world)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_bit_cells_unique_build_side_over_a_wide_range(sim, world):
    # 30-bit range: 2^21 lattice points 512 apart; one bit per cell, 128 MiB of images
    n = 40_000
    r = run(sim, world, [n] * world, [60_000] * world, 1, 5, 0, stride=(1 << 29) // (n * world), seed=20 + world)
    assert r["shared"] == 1 and r["bits"] == 1 and r["b"] >= 29 and r["got"] == r["want"] > 0


@pytest.mark.parametrize("world", [2, 4, 8])
def test_a_key_on_two_ranks_of_a_wide_range_is_noticed_by_every_rank(sim, world):
    n = 30_000
    r = run(sim, world, [n] * world, [1000] * world, 2, 5, 0, stride=(1 << 29) // (n * world), seed=30 + world)
    assert r["shared"] == 0  # the carry of the summed words costs population: the exchange plan takes the join


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_a_cell_beyond_255_only_in_the_sum(sim, world):
    hot = 300 // world + 1  # every rank alone stays below 256 (world > 1): only the summed byte wraps
    r = run(sim, world, [5_000] * world, [20_000] * world, 0, 0, 40_000, hot=hot, seed=40 + world)
    assert r["shared"] == 0
    ok = run(sim, world, [5_000] * world, [20_000] * world, 0, 0, 40_000, hot=200 // world, seed=40 + world)
    assert ok["shared"] == 1 and ok["got"] == ok["want"]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ragged_and_empty_ranks(sim, world):
    r = run(sim, world, [0] + [9_000] * (world - 1), [4_000] * world, 0, 100, 30_000, seed=50)
    assert r["shared"] == 1 and r["got"] == r["want"] > 0
    r = run(sim, world, [25_000] + [0] * (world - 1), [0] * (world - 1) + [30_000], 0, -5, 30_000, seed=51)
    assert r["shared"] == 1 and r["got"] == r["want"] > 0
    r = run(sim, world, [0] * world, [100] * world, 0, 0, 1000, seed=52)
    assert r["shared"] == 0 and r["want"] == 0  # no usable build row anywhere: nothing to share (and nothing joins)


def test_sparse_ranges_keep_the_exchange_plan_unless_forced(sim):
    r = run(sim, 4, [1000] * 4, [1000] * 4, 0, 0, 1 << 27, force=0, seed=60)
    assert r["shared"] == 0
    r = run(sim, 4, [1000] * 4, [1000] * 4, 0, 0, 1 << 27, force=1, seed=60)
    assert r["shared"] == 1 and r["got"] == r["want"]


@pytest.mark.parametrize("world,bits,b,img", [(1, 0, 27, 1 << 27), (2, 0, 28, 1 << 28), (4, 1, 29, 1 << 26), (8, 1, 30, 1 << 27)])
def test_projection_of_the_bench_shape(sim, world, bits, b, img):
    """bench.py's weak scaling: 1e8 unique build keys + 1e8 probe rows per rank (DESIGN.md §6's table)"""
    out = (C.c_int64 * 8)()
    assert sim.sim_shared_projection(world, 100_000_000, 100_000_000, 1, out) == 0
    assert out[0] == 1 and out[1] == bits and out[2] == b and out[3] == img
    assert out[5] == 0 and out[6] == (0 if world == 1 else 100_000_000 * 8 * (world - 1) // world)
    assert out[4] == (0 if world == 1 else 2 * img * (world - 1) // world)
    # a build side WITH duplicate keys beyond 28 bits has no images: the exchange plan
    assert sim.sim_shared_projection(world, 100_000_000, 100_000_000, 0, out) == 0 and out[0] == (1 if b <= 28 else 0)


@pytest.mark.parametrize("world", [2, 3])
def test_two_real_processes_run_the_shared_plan_over_gloo(world):
    """tests/dist_shared_worker.py: the range / usable-rows / flag / image all-reduces are real collectives (gloo), the plan and the
    verdicts are the product's host arithmetic — byte cells with duplicates across ranks, bit cells with a rank that holds no build
    rows, a key on two ranks that must make EVERY rank drop the plan, a hot key whose byte cell approaches the wrap."""
    import subprocess
    import sys
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29531 + world), os.path.join(ROOT, "tests", "dist_shared_worker.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "SHARED_OK" in p.stdout
