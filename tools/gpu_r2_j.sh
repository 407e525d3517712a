#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2j}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_join_radix_gpu.py tests/test_join_gpu.py tests/test_pipeline_gpu.py tests/test_build_partitioned_gpu.py -x -q > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/rc.txt
R=$(pwd)
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_bench.json 2> $R/$O/prof_bench.err
cd $R
python tools/summarize_prof.py $O/prof_summary.txt --stats $O/prof/p_kernel_stats.csv 2>&1
cat $O/rc.txt
tail -15 $O/tests.txt
cat $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','verified','probe_strategy')})
for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6'): print(k, d.get(k))
"
tail -3 $O/bench.err
head -30 $O/prof_summary.txt
