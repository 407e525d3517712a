#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2x}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sort_gpu.py tests/test_join_ordered_gpu.py -q --durations=5 > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
