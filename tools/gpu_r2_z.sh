#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2z}
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/bench_configs.py --skip-join > $O/c3.json 2> $O/c3.err; cat $O/c3.json | cut -c1-400; tail -2 $O/c3.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o p --output-format csv -- python $OLDPWD/tools/bench_configs.py --skip-join > /dev/null 2> $OLDPWD/$O/prof.err
cd $OLDPWD
python tools/summarize_prof.py $O/rocprof_c3.txt --stats $O/prof/p_kernel_stats.csv --note "tools/bench_configs.py --skip-join (C3: 1e9 rows / 1e6 groups) under rocprofv3 --kernel-trace --stats"
head -10 $O/rocprof_c3.txt | cut -c1-150
