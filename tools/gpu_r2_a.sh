#!/bin/bash
# round 2, GPU call A: parity of the sliced table + LDS probe, then the bench with both probe routes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_join_radix_gpu.py tests/test_join_gpu.py tests/test_build_partitioned_gpu.py tests/test_join_ordered_gpu.py -x -q > $O/tests_join.txt 2>&1
echo "tests_join rc=$?" >> $O/rc.txt
for v in lds l2; do
  TSQ_RADIX_KERNEL=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  echo "bench_$v rc=$?" >> $O/rc.txt
done
for lf in 0.5 0.85; do
  TSQ_TABLE_LF=$lf timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_lds_lf$lf.json 2> $O/bench_lds_lf$lf.err
  echo "bench_lf$lf rc=$?" >> $O/rc.txt
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --build-rows 10000000 > $O/bench_c2.json 2> $O/bench_c2.err
echo "bench_c2 rc=$?" >> $O/rc.txt
R=$(pwd)
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err
cd $R
python tools/summarize_prof.py $O/prof_summary.txt --stats $O/prof/p_kernel_stats.csv 2>&1
timeout 600 python -m pytest tests/test_agg_fast_gpu.py tests/test_split_gpu.py tests/test_agg_gpu.py -x -q > $O/tests_agg.txt 2>&1
echo "tests_agg rc=$?" >> $O/rc.txt
cat $O/rc.txt
tail -5 $O/tests_join.txt
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f'))
r=d.get('roofline',{})
print(d['value'], d['ms_per_step'], d['verified'], d.get('probe_strategy'), r.get('kernel_ms'), r.get('partition',{}).get('kernel_ms'), r.get('probe_phase',{}).get('frac'), d['build_kernel_ms'], d['table_bytes'])
" 2>&1 | tail -1; done
head -30 $O/prof_summary.txt
