#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python tools/bench_decode.py > $O/decode.json 2> $O/decode.err; cat $O/decode.json; tail -3 $O/decode.err
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_dec -o d --output-format csv -- python $R/tools/bench_decode.py > $O/prof_dec.txt 2>&1
python $R/tools/summarize_prof.py $O/decode_prof_summary.txt --stats $O/prof_dec/d_kernel_stats.csv; head -10 $O/decode_prof_summary.txt | cut -c1-150
