#!/bin/bash
# round 5: 64-byte slots in the dictionary aggregate's scatter pass (A/B: knob KEYREC=2 = separate arrays)
mkdir -p gpurun_out
timeout 300 python3 -m pytest tests/test_agg_keydict_gpu.py -m gpu -x -q 2>&1 | tail -3
for shape in "10000000 100000" "10000000 5000000" "100000000 10000000"; do
  timeout 200 python3 tools/bench_agg_string.py $shape 2>&1 | tail -1
  timeout 200 python3 tools/bench_agg_string.py $shape --arrays 2>&1 | tail -1
done
