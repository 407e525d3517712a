#!/bin/bash
# round 5, call q: GROUP BY string keys: dictionary route vs the several-column upsert, three shapes
mkdir -p gpurun_out
timeout 300 python3 -m pytest tests/test_agg_keydict_gpu.py -m gpu -x -q 2>&1 | tail -3
for shape in "10000000 100000" "100000000 1000000" "100000000 10000000"; do
  timeout 200 python3 tools/bench_agg_string.py $shape 2>&1 | tail -1
  timeout 200 python3 tools/bench_agg_string.py $shape --no-dict 2>&1 | tail -1
done
