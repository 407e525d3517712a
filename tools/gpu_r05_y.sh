#!/bin/bash
mkdir -p gpurun_out
timeout 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras two_key_bigint_string_count,agg_string_keys_1e7_1e5,agg_string_keys_1e7_5e6 --extras-file r05_y_x.json 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['sides'])"
