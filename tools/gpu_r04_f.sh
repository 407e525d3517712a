#!/bin/bash
# round 4, GPU call F: wide several-column COUNT(*) join, sel fast path, sort kernel occupancy, distributed Q3; the default bench line;
# kernel stats + FETCH_SIZE / WRITE_SIZE passes of the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
(timeout 1500 python -m pytest tests/test_join_packed_gpu.py tests/test_agg_gpu.py tests/test_comm_gpu.py tests/test_pipeline_gpu.py tests/test_join_radix_gpu.py tests/test_packed_auto_gpu.py \
   -m gpu -q --maxfail=40 > $O/r04f_pytest.txt 2>&1; echo "rc=$?" >> $O/r04f_pytest.txt)
tail -6 $O/r04f_pytest.txt
: > $O/r04f_q3.txt
for args in "10" "100 --device-gen" "100 --device-gen --dist"; do
  echo "== q3.py $args" >> $O/r04f_q3.txt
  timeout 600 python tools/q3.py $args 2>/dev/null | tail -1 >> $O/r04f_q3.txt
done
cut -c1-330 $O/r04f_q3.txt
/usr/bin/time -v -o $O/r04f_bench.time timeout 900 python bench.py > $O/r04f_bench.json 2> $O/r04f_bench.err
grep -E "Elapsed" $O/r04f_bench.time
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_benchf -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 > $O/r04f_bench_prof.json 2> $O/r04f_bench_prof.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_benchf -name "*kernel_stats.csv" | head -1) > $O/r04f_bench_rocprof.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_benchf_$c -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 > $O/r04f_bench_pmc_$c.json 2> $O/r04f_bench_pmc_$c.err
done
python $GRAFT_REPO_ROOT/tools/summarise_prof.py pmc-by-grid $(find $O/pmc_benchf_FETCH_SIZE $O/pmc_benchf_WRITE_SIZE -name "*counter_collection.csv") > $O/r04f_bench_pmc.txt 2>&1
rm -rf $O/prof_benchf/*/*.db $O/pmc_benchf_*/*/*.db $O/pmc_benchf_*/*/*counter_collection.csv 2>/dev/null
ls -la $O | tail -5
echo done
