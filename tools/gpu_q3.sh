#!/bin/bash
# Q3-shaped device pipeline: tests, wall clock, kernel-time breakdown.  Run from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_agg_gpu.py tests/test_agg_fast_gpu.py tests/test_join_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/q3.py 10 2>&1 | tail -2
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_q3 -o q3 --output-format csv -- python $R/tools/q3.py 10 > $O/prof_q3.txt 2>&1
python $R/tools/summarize_prof.py $O/q3_prof_summary.txt --stats $O/prof_q3/q3_kernel_stats.csv; head -30 $O/q3_prof_summary.txt | cut -c1-140; tail -2 $O/prof_q3.txt
