#!/bin/bash
# C++ host strings, multi-key tag collisions, NULLs through the exchange (world size 1 on this box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2p}
mkdir -p $O
timeout 900 python -m pytest tests/test_host_cpp_gpu.py tests/test_agg_gpu.py tests/test_agg_string_gpu.py -x -q > $O/tests_a.txt 2>&1
echo "host/agg tests rc=$?" >> $O/rc.txt
TSQ_TEST_TIMING=1 timeout 900 python -m pytest tests/test_comm_gpu.py -x -q > $O/tests_comm.txt 2>&1
echo "comm tests rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -30 $O/tests_a.txt | cut -c1-600; tail -30 $O/tests_comm.txt | cut -c1-400
