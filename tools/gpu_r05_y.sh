#!/bin/bash
# round 5, call y: non-temporal stores in the key-record scatter
mkdir -p gpurun_out
timeout 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras two_key_bigint_string_count,agg_string_keys_1e7_1e5,agg_string_keys_1e7_5e6 --extras-file r05_y_x.json 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['sides'])"
timeout 600 python3 -m pytest tests/test_join_keyrec_gpu.py tests/test_agg_keydict_gpu.py -m gpu -x -q 2>&1 | tail -3
