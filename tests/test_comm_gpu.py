"""GPU: the multi-GPU exchange behind the C-ABI (csrc/tsq_comm.hip): tsq_radix_split + RCCL send/recv + the HIP operators,
against the oracle's whole-table join and aggregate.  World size 1 runs on any box (RCCL initialises, the rank exchanges
with itself); world size 2 runs when two GPUs are visible (the round-end driver's multi-GPU box), one process per GPU."""
import os
import subprocess
import sys

import pytest

from tinysql_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world):
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29600 + world), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    procs = []
    for r in range(world):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py")], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d/%d OK" % (r, world)) in o, "rank %d failed:\n%s" % (r, o[-4000:])


def test_exchange_join_and_aggregate_world_size_1():
    _run(1)


def test_exchange_join_and_aggregate_world_size_2():
    if _lib.load().tsq_device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU over RCCL)")
    _run(2)


def _run_dist_q3(world, sf="0.05"):
    """tools/q3.py --dist: the distributed Q3-shaped plan (broadcast joins + partial -> shuffle -> final aggregate) on row-sharded
    tables; rank 0's JSON line says whether every rank's groups equal the numpy restatement of the whole query"""
    import json
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29700 + world), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    procs = []
    for r in range(world):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "q3.py"), sf, "--dist"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s\n%s" % (r, o[-2000:], e[-4000:])
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["verified_against_numpy"] is True and line["n_gpus"] == world and line["groups"] > 1000
    return line


def test_distributed_q3_plan_world_size_1():
    line = _run_dist_q3(1)
    assert line["wire_bytes_this_rank"] == {"broadcast_customer": 0, "broadcast_orders": 0, "shuffle_partial_groups": 0}


def test_distributed_q3_plan_world_size_2():
    if _lib.load().tsq_device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU over RCCL)")
    line = _run_dist_q3(2)
    assert line["wire_bytes_this_rank"]["broadcast_orders"] > 0
