#!/bin/bash
# string aggregate tests, then the whole GPU suite (the 2-rank RCCL test is left to the final run: its first init costs minutes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2o}
mkdir -p $O
timeout 600 python -m pytest tests/test_agg_string_gpu.py tests/test_agg_gpu.py -x -q > $O/tests_str.txt 2>&1
echo "string tests rc=$?" >> $O/rc.txt
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_comm_gpu.py > $O/tests_all.txt 2>&1
echo "all tests rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -30 $O/tests_str.txt | cut -c1-400; tail -15 $O/tests_all.txt | cut -c1-300
