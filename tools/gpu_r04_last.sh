#!/bin/bash
# round 4, last run: the default bench line and its rocprofv3 kernel stats at the final commit
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
timeout 100 python bench.py > $O/r04z_bench.json 2> $O/r04z_bench.err
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_z -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 > $O/r04z_bench_prof.json 2> $O/r04z_bench_prof.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_z -name "*kernel_stats.csv" | head -1) > $O/r04z_bench_rocprof.txt 2>&1
rm -rf $O/prof_z
echo done
