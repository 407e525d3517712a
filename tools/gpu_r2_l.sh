#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2m}
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/$n -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/$n.json 2> $R/$O/$n.err
done
cd $R
python tools/summarize_prof.py $O/pmc_summary.txt --pmc $O/FETCH_SIZE/p_counter_collection.csv --pmc $O/TCC_HIT_sum_TCC_MISS_sum/p_counter_collection.csv
grep -E "k_lds_probe" $O/pmc_summary.txt
cat $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','verified')}, d['roofline']['kernel_ms'], d['roofline']['partition']['kernel_ms'])
for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6'): print(k, {q:d[k].get(q) for q in ('ms','ms_per_probe_pass','frac','verified')})
"
