#!/bin/bash
# round 5, run I: kernel times of the key-record route (1e7 x 1e7 on (bigint, 16-byte varstring))
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_i -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras two_key_bigint_string_count --extras-file r05_i_x.json > $O/r05_i.line 2> $O/r05_i.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_i -name "*kernel_stats.csv" | head -1) > $O/r05_i_rocprof.txt 2>&1
rm -rf $O/prof_i
grep -E "k_kr_|kernel  " $O/r05_i_rocprof.txt
