#!/bin/bash
# round 4, GPU call E: partition2 fix, bit-cell pairs route, wide composite aggregate, unit-loop sort kernel, Q3 plans, bench kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
: > $O/r04e_partition_check.txt
tools/partition_check | while read n hot b var; do
  timeout 60 tools/partition_check $n $hot $b $var >> $O/r04e_partition_check.txt 2>&1 || echo "   -> exit $? for n=$n hot=$hot b=$b var=$var" >> $O/r04e_partition_check.txt
done
echo "partition_check: $(grep -c ' ok$' $O/r04e_partition_check.txt) ok, $(grep -c -E 'MISMATCH|exit' $O/r04e_partition_check.txt) bad"
(timeout 1700 python -m pytest tests/test_join_packed_gpu.py tests/test_agg_packed_gpu.py tests/test_agg_gpu.py tests/test_pipeline_gpu.py tests/test_comm_gpu.py tests/test_packed_auto_gpu.py \
   tests/test_large_vs_oracle_gpu.py tests/test_agg_fast_gpu.py -m gpu -q --maxfail=40 > $O/r04e_pytest.txt 2>&1; echo "rc=$?" >> $O/r04e_pytest.txt)
tail -6 $O/r04e_pytest.txt
: > $O/r04e_q3.txt
for args in "10" "100 --device-gen" "100 --device-gen --classic" "100 --device-gen --dist"; do
  echo "== q3.py $args" >> $O/r04e_q3.txt
  timeout 600 python tools/q3.py $args 2>&1 | tail -1 >> $O/r04e_q3.txt
done
cut -c1-300 $O/r04e_q3.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_q3e -o q3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/q3.py 100 --device-gen > $O/r04e_q3_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_q3e -name "*kernel_stats.csv" | head -1) > $O/r04e_q3_rocprof.txt 2>&1
head -30 $O/r04e_q3_rocprof.txt | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_benche -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 > $O/r04e_bench_prof.json 2> $O/r04e_bench_prof.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_benche -name "*kernel_stats.csv" | head -1) > $O/r04e_bench_rocprof.txt 2>&1
rm -rf $O/prof_q3e/*/*.db $O/prof_benche/*/*.db 2>/dev/null
echo done
