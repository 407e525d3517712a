#!/bin/bash
# round 5, call t: the whole GPU suite (timed), smoke
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
( time timeout 1100 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > $O/r05_pytest_gpu.txt 2>&1
cat $O/r05_pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
