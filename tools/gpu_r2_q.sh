#!/bin/bash
# LDS aggregate with per-word descriptors + hashed tags: parity, then the C3 timing under rocprofv3 --kernel-trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2q}
mkdir -p $O
timeout 900 python -m pytest tests/test_agg_fast_gpu.py tests/test_agg_gpu.py tests/test_large_vs_oracle_gpu.py tests/test_pipeline_gpu.py tests/test_agg_string_gpu.py -x -q > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/rc.txt
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python tools/bench_configs.py --skip-join > $O/c3.json 2> $O/c3.err
python tools/summarize_prof.py $O/kernels.txt --stats $O/prof/p_kernel_stats.csv 2>&1 | tail -2
cat $O/rc.txt; tail -5 $O/tests.txt | cut -c1-300; cat $O/c3.json; head -14 $O/kernels.txt | cut -c1-150
