#!/bin/bash
# round 5, call w: dense emit with one device atomic per 4096-cell chunk
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_agg_packed_gpu.py tests/test_agg_keydict_gpu.py tests/test_agg_fast_gpu.py tests/test_agg_gpu.py -m gpu -x -q 2>&1 | tail -4
for shape in "10000000 100000" "10000000 5000000" "100000000 10000000"; do
  timeout 200 python3 tools/bench_agg_string.py $shape 2>&1 | tail -1
done
timeout 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras c3_agg_1e9_1e6,c3_zipf_s1,agg_two_keys_1000x100 --extras-file r05_w_x.json 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['sides'])"
