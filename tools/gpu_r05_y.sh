#!/bin/bash
# round 5, call y: non-temporal stores of the emit kernel's output cells
mkdir -p gpurun_out
timeout 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras materialising,materialising_nullable_left_outer --extras-file r05_y_x.json 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['sides'])"
python3 - <<'P'
import json
d = json.load(open('gpurun_out/r05_y_x.json'))
ex = d.get('extras', d)
for k in ('materialising', 'materialising_nullable_left_outer'):
    v = ex[k]; print(k, {q: v[q] for q in ('ms', 'repeated_probe_pass_ms', 'packed_prepare_ms', 'verified')})
P
timeout 600 python3 -m pytest tests/test_join_packed_gpu.py -m gpu -x -q -k "travelling or golden" 2>&1 | tail -3
