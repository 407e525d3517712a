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
