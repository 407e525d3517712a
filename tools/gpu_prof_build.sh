#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_build -o b --output-format csv -- python $R/tools/bench_build.py 100000000 > $O/prof_build.txt 2>&1
python $R/tools/summarize_prof.py $O/build_prof_summary.txt --stats $O/prof_build/b_kernel_stats.csv; head -14 $O/build_prof_summary.txt | cut -c1-150
