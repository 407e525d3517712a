#!/bin/bash
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_agg_packed_gpu.py tests/test_agg_keydict_gpu.py tests/test_agg_fast_gpu.py tests/test_agg_gpu.py tests/test_agg_string_gpu.py tests/test_pipeline_gpu.py tests/test_host_cpp_gpu.py -m gpu -x -q 2>&1 | tail -3
for shape in "10000000 100000" "10000000 5000000"; do
  timeout 200 python3 tools/bench_agg_string.py $shape 2>&1 | tail -1
done
