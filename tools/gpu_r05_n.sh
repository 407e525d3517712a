#!/bin/bash
# round 5, call n: materialising key-record join with per-partition output bases
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_join_keyrec_gpu.py tests/test_join_varlen_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras two_key_bigint_string_count --extras-file r05_n_extras.json 2>gpurun_out/r05_n.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps(d.get('sides'), indent=0))"
python3 - <<'P'
import json
d = json.load(open('gpurun_out/r05_n_extras.json'))
print(json.dumps(d['extras'].get('two_key_bigint_string_count'), indent=1)[:1500])
P
