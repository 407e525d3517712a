#!/bin/bash
# round 5, final evidence: the driver's bench command (line kept), then rocprofv3 kernel stats + PMC (FETCH_SIZE / WRITE_SIZE, separate passes) of the bench command
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench.json 2> $O/r05_bench.err ) 2> $O/r05_bench.time
cp $O/bench_extras.json $O/r05_bench_extras.json
tail -3 $O/r05_bench.time; wc -c $O/r05_bench.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_g -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --extras-file r05_g_prof_extras.json > $O/r05_bench_prof.json 2> $O/r05_bench_prof.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_g -name "*kernel_stats.csv" | head -1) > $O/r05_bench_rocprof.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_g_$c -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --extras-file r05_g_pmc_extras.json > $O/r05_pmc_$c.json 2> $O/r05_pmc_$c.err
done
python $GRAFT_REPO_ROOT/tools/summarise_prof.py pmc-by-grid $(find $O/pmc_g_FETCH_SIZE $O/pmc_g_WRITE_SIZE -name "*counter_collection.csv") > $O/r05_bench_pmc.txt 2>&1
rm -rf $O/prof_g $O/pmc_g_* 2>/dev/null
head -12 $O/r05_bench_rocprof.txt
grep -c . $O/r05_bench_pmc.txt
