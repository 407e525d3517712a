#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2e2}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_storage_coprocessor_gpu.py -q -x --durations=4 > $O/pytest.txt 2>&1
tail -40 $O/pytest.txt
