#!/bin/bash
# round 4, run H: packed aggregate (dense state / narrow cells / overflow store) re-checked; Q3 under the profiler at the final code
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_agg_packed_gpu.py tests/test_agg_gpu.py -m gpu -q 2>&1 | tail -25
timeout 900 python -m pytest tests/test_join_packed_gpu.py tests/test_pipeline_gpu.py tests/test_packed_auto_gpu.py -m gpu -q 2>&1 | tail -25
} > $O/r04h_pytest.txt 2>&1
C3=c3_agg_1e9_1e6,c3_agg_1e9_1e6_double,c3_zipf_s1,c3_sparse_keys,agg_two_keys_1000x100,agg_two_keys_50x20
timeout 600 python bench.py --no-cpu-baseline --steps 5 --only-extras $C3 > $O/r04h_bench_c3.json 2> $O/r04h_bench_c3.err
timeout 300 python tools/q3.py 100 --device-gen 2>&1 | tail -1 > $O/r04h_q3.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3h -o c3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --only-extras $C3 > $O/r04h_c3_prof.json 2> $O/r04h_c3_prof.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_c3h -name "*kernel_stats.csv" | head -1) > $O/r04h_c3_rocprof.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_c3h_$c -o c3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --only-extras $C3 > $O/r04h_c3_pmc_$c.json 2> $O/r04h_c3_pmc_$c.err
done
python $GRAFT_REPO_ROOT/tools/summarise_prof.py pmc-by-grid $(find $O/pmc_c3h_FETCH_SIZE $O/pmc_c3h_WRITE_SIZE -name "*counter_collection.csv") > $O/r04h_c3_pmc.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_q3h -o q3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/q3.py 100 --device-gen > $O/r04h_q3_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_q3h -name "*kernel_stats.csv" | head -1) > $O/r04h_q3_rocprof.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_q3h_$c -o q3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/q3.py 100 --device-gen > $O/r04h_q3_pmc_$c.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/summarise_prof.py pmc-by-grid $(find $O/pmc_q3h_FETCH_SIZE $O/pmc_q3h_WRITE_SIZE -name "*counter_collection.csv") > $O/r04h_q3_pmc.txt 2>&1
rm -rf $O/prof_c3h $O/prof_q3h $O/pmc_c3h_* $O/pmc_q3h_* 2>/dev/null
ls -la $O | grep r04h
echo done
