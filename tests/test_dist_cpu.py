"""CPU: the N>1 path (hash-radix redistribute + local operator + tiny all-reduce) with world_size 2
over gloo.  The GPU split and join are replaced by CPU stand-ins; the exchange code under test is
tinysql_amd/parallel.py itself."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_2_partitioned_join_equals_whole_join():
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tests", "dist_worker.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "DIST_OK" in p.stdout
