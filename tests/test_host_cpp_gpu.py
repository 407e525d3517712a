"""The C++ host executors (tinysql_amd/host/tsq_host.hpp — the cgo shim's role, written in C++ because the image has no
Go toolchain) replaying the reference's own SQL-level cases on the GPU: tsq_host_test.cpp cites join_test.go /
aggregate_test.go / aggfuncs tests case by case and exits non-zero on any mismatch."""
import os
import subprocess

import pytest

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tinysql_amd", "host")


@pytest.mark.gpu
def test_reference_cases_through_the_cpp_host_executors():
    exe = os.path.join(HOST, "tsq_host_test")
    # always through make: a binary built against an older include/tsq.h (struct layouts) must not be run
    subprocess.run(["make", "-C", HOST], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed, 0 failed" in r.stdout and "PASS join_test.go:134-146" in r.stdout


def test_cpp_host_layer_builds_against_the_c_abi_without_hip_headers():
    # the host layer is plain C++17 over include/tsq.h: it must compile with g++ alone (no hipcc, no torch)
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", os.path.join(HOST, "tsq_host_test.cpp")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    src = open(os.path.join(HOST, "tsq_host.hpp")).read()
    assert "oracle" not in src.replace("touches the oracle", "") and "hip/hip_runtime" not in src
