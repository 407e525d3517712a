#!/bin/bash
# round 2, GPU call I: the whole GPU suite (with the row-encode tests un-gated), the exchange tests, bench N=1 with extras, bench --force-dist
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2i}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_comm_gpu.py -x -q > $O/tests_comm.txt 2>&1
echo "tests_comm rc=$?" >> $O/rc.txt
TSQ_RUN_UNVERIFIED=1 timeout 1500 python -m pytest tests -m gpu -q > $O/tests_all.txt 2>&1
echo "tests_all rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/rc.txt
timeout 300 python bench.py --steps 5 --warmup 2 --force-dist --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err
echo "bench_dist rc=$?" >> $O/rc.txt
cat $O/rc.txt
tail -15 $O/tests_comm.txt
tail -12 $O/tests_all.txt
cat $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','verified','probe_strategy')})
print(d['roofline'])
for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6','cpu_baseline'): print(k, d.get(k))
"
tail -3 $O/bench.err
cat $O/bench_dist1.json | cut -c1-600; tail -5 $O/bench_dist1.err
