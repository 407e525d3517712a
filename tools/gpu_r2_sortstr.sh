#!/bin/bash
# round 2: string ORDER BY items — the sort suite and the sort microbench
out=gpurun_out/${1:-r2s}
mkdir -p $out
timeout 600 python -m pytest tests/test_sort_gpu.py -x -q -m gpu > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
