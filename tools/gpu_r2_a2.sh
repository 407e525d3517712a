#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2a2}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_decode_chunks_gpu.py tests/test_decode_gpu.py -q --durations=5 > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
