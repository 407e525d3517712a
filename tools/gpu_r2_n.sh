#!/bin/bash
# build-image (counting) + pipelined LDS aggregate: correctness, then timings with and without the new kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2n}
mkdir -p $O
timeout 900 python -m pytest tests/test_build_partitioned_gpu.py tests/test_join_radix_gpu.py tests/test_agg_gpu.py tests/test_agg_fast_gpu.py tests/test_large_vs_oracle_gpu.py -x -q > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/rc.txt
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
TSQ_BUILD_IMAGES=cas timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_cas.json 2> $O/bench_cas.err
python tools/summarize_prof.py $O/kernels.txt --stats $(ls $O/prof/*kernel_stats.csv $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1) 2>&1
cat $O/rc.txt; tail -3 $O/tests.txt
head -30 $O/kernels.txt
python - <<PY
import json
for f in ("bench.json","bench_cas.json"):
    try:
        d=json.loads(open("$O/"+f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, {k:d.get(k) for k in ('value','ms_per_step','verified')}, d['roofline'].get('kernel_ms'), d.get('build'))
    for k in ('c2_1e8x1e7','materialising','c3_agg_1e9_1e6'):
        if k in d: print(' ', k, d[k])
PY
