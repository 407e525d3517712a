#!/bin/bash
# quick A/B: parity of the radix tests + phase profile + bench (1e8 and 1e7 build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2f}
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_join_radix_gpu.py -x -q > $O/tests_join.txt 2>&1
echo "tests_join rc=$?" >> $O/rc.txt
TSQ_LDS_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_lds.json 2> $O/prof_lds.err
TSQ_LDS_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --build-rows 10000000 > $O/prof_c2.json 2> $O/prof_c2.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_lds.json 2> $O/bench_lds.err
echo "bench_lds rc=$?" >> $O/rc.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --build-rows 10000000 > $O/bench_c2.json 2> $O/bench_c2.err
echo "bench_c2 rc=$?" >> $O/rc.txt
cat $O/rc.txt
tail -3 $O/tests_join.txt
grep "lds-prof" $O/*.err | awk 'NR%3==1'
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f'))
r=d.get('roofline',{})
print(d['value'], d['ms_per_step'], d['verified'], d.get('probe_strategy'), r.get('kernel_ms'), r.get('partition',{}).get('kernel_ms'), r.get('probe_phase',{}).get('frac'), d['build_kernel_ms'], d['table_bytes'])
" 2>&1 | tail -1; done
