#!/bin/bash
# rocprofv3 kernel stats of the chunk wire codec bench and the index key bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2wp}
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/tools/bench_wire.py 5e7 > $R/$O/wire.json 2> $R/$O/prof.err
cd $R
python tools/summarize_prof.py $O/rocprof_wire.txt --stats $O/prof/p_kernel_stats.csv --note "python tools/bench_wire.py 5e7 under rocprofv3 --kernel-trace --stats (6 encodes + 6 decodes of a 2.0 GB wire chunk)"
head -12 $O/rocprof_wire.txt | cut -c1-160
cat $O/wire.json | cut -c1-300
