#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2m}
mkdir -p $O
timeout 900 python -m pytest tests/test_join_radix_gpu.py -x -q > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
TSQ_LDS_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof.json 2> $O/prof.err
cat $O/rc.txt; tail -3 $O/tests.txt
grep lds-prof $O/prof.err | head -2
cat $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['kernel_ms'], d['roofline']['partition']['kernel_ms'], d['roofline']['probe_phase']['frac'])
for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6'): print(k, {q:d[k].get(q) for q in ('ms','ms_per_probe_pass','frac','verified')})
"
