#!/bin/bash
# round 5: fixed-position record builders + the 5-multiply mix: parity and timings
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_join_keyrec_gpu.py tests/test_agg_keydict_gpu.py tests/test_join_varlen_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras two_key_bigint_string_count,agg_string_keys_1e7_1e5,agg_string_keys_1e7_5e6 --extras-file r05_y_x.json 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['sides'])"
timeout 200 python3 tools/bench_agg_string.py 100000000 10000000 2>&1 | tail -1
