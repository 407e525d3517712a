#!/bin/bash
# round 5, call s: outer joins on key records
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_join_keyrec_gpu.py tests/test_join_varlen_gpu.py tests/test_join_gpu.py -m gpu -x -q 2>&1 | tail -25
