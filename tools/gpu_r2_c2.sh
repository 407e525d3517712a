#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2c2}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_storage_rows_encode_gpu.py tests/test_decode_chunks_gpu.py -q --durations=3 > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
