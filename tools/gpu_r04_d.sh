#!/bin/bash
# round 4, GPU call D: partition differential (one process per case), new Q3 plan + distributed plan, kernel stats of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
: > $O/r04d_partition_check.txt
tools/partition_check | while read n hot b var; do
  timeout 60 tools/partition_check $n $hot $b $var >> $O/r04d_partition_check.txt 2>&1 || echo "   -> exit $? for n=$n hot=$hot b=$b var=$var" >> $O/r04d_partition_check.txt
done
grep -E "MISMATCH|FAULT|exit|fault" $O/r04d_partition_check.txt | head -40
(timeout 1500 python -m pytest tests/test_join_packed_gpu.py tests/test_pipeline_gpu.py tests/test_comm_gpu.py tests/test_join_gpu.py tests/test_packed_auto_gpu.py \
   -m gpu -q --maxfail=40 -k "not (two_wg_per_cu and u32) and not ragged and not golden" > $O/r04d_pytest.txt 2>&1; echo "rc=$?" >> $O/r04d_pytest.txt)
tail -4 $O/r04d_pytest.txt
for args in "10" "10 --classic" "100 --device-gen" "100 --device-gen --classic"; do
  echo "== q3.py $args" >> $O/r04d_q3.txt
  timeout 600 python tools/q3.py $args 2>&1 | tail -1 >> $O/r04d_q3.txt
done
timeout 300 python tools/q3.py 100 --device-gen --dist 2>&1 | tail -1 >> $O/r04d_q3.txt
cut -c1-400 $O/r04d_q3.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_q3b -o q3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/q3.py 100 --device-gen > $O/r04d_q3_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_q3b -name "*kernel_stats.csv" | head -1) > $O/r04d_q3_rocprof.txt 2>&1
head -24 $O/r04d_q3_rocprof.txt | cut -c1-150
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 > $O/r04d_bench_prof.json 2> $O/r04d_bench_prof.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) > $O/r04d_bench_rocprof.txt 2>&1
rm -rf $O/prof_q3b/*/*.db $O/prof_bench/*/*.db 2>/dev/null
echo done
