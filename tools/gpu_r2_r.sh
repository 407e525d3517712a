#!/bin/bash
# partition kernels with all tile loads issued before the hashing: parity of everything that partitions, then timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2r}
mkdir -p $O
timeout 900 python -m pytest tests/test_join_radix_gpu.py tests/test_build_partitioned_gpu.py tests/test_agg_fast_gpu.py tests/test_large_vs_oracle_gpu.py tests/test_split_gpu.py -x -q > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/rc.txt
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/summarize_prof.py $O/kernels.txt --stats $O/prof/p_kernel_stats.csv 2>&1 | tail -2
cat $O/rc.txt; tail -4 $O/tests.txt | cut -c1-300; head -24 $O/kernels.txt | cut -c1-150
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','verified','build_ms','build_kernel_ms')}, d['roofline'].get('kernel_ms'), d['roofline']['frac'], d['roofline']['probe_phase'])
for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6'):
    if k in d: print(' ', k, {q:d[k].get(q) for q in ('ms','ms_per_probe_pass','frac','verified','first_pass_ms')})
PY
