#!/bin/bash
# round 5, call u: cursor atomics performed in the XCD's L2 (A/B on the headline), packed-route parity
mkdir -p gpurun_out
for v in 0 1 0 1; do
  timeout 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --only-extras rho_0.5 --knob XCD_ATOMICS=$v --extras-file r05_u_x.json 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('XCD_ATOMICS=$v', 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'second', r['second_kernel']['kernel_ms'], 'verified', d['verified'], d['sides'])"
done
timeout 600 python3 -m pytest tests/test_join_packed_gpu.py tests/test_packed_auto_gpu.py tests/test_large_vs_oracle_gpu.py -m gpu -x -q 2>&1 | tail -4
