#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c --output-format csv -- python $R/tools/bench_configs.py > $O/prof_c3.txt 2>&1
grep "C3" $O/prof_c3.txt | cut -c1-300
python $R/tools/summarize_prof.py $O/c3_prof_summary.txt --stats $O/prof_c3/c_kernel_stats.csv; head -16 $O/c3_prof_summary.txt | cut -c1-150
