#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_emit -o e --output-format csv -- python $R/tools/bench_emit.py > $O/prof_emit.txt 2>&1
tail -2 $O/prof_emit.txt | cut -c1-400
python $R/tools/summarize_prof.py $O/emit_prof_summary.txt --stats $O/prof_emit/e_kernel_stats.csv; head -12 $O/emit_prof_summary.txt | cut -c1-150
