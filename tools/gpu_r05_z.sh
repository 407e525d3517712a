#!/bin/bash
# round 5, call z: the N > 1 code path on one rank (both plans), after the last join changes
mkdir -p gpurun_out
for plan in auto exchange; do
  timeout 300 python3 bench.py --force-dist --dist-plan $plan --no-cpu-baseline --steps 5 --warmup 2 --only-extras none --extras-file r05_z_x.json 2>gpurun_out/r05_z.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$plan', d['value'], d['ms_per_step'], d['verified'], d['config'].get('parallelism'), d.get('dist_plan'))" || tail -5 gpurun_out/r05_z.err
done
