#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2d2}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 tinysql_amd/host/tsq_host_test > $O/host_test.txt 2>&1; echo "rc=$?"; grep -c PASS $O/host_test.txt; grep -A3 FAIL $O/host_test.txt | head -30; tail -2 $O/host_test.txt
