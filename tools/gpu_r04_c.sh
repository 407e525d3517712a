#!/bin/bash
# round 4, GPU call C: partition differential, the new build-side preparation + lazy table under the packed tests, bench, Q3 + bench profiles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 300 tools/partition_check > $O/r04c_partition_check.txt 2>&1; echo "rc=$?" >> $O/r04c_partition_check.txt
grep -E "MISMATCH|FAULT|ALL OK|FAILED|rc=" $O/r04c_partition_check.txt | head -40
(timeout 1500 python -m pytest tests/test_join_packed_gpu.py tests/test_packed_auto_gpu.py tests/test_join_gpu.py tests/test_join_radix_gpu.py tests/test_large_vs_oracle_gpu.py \
   tests/test_pipeline_gpu.py tests/test_host_cpp_gpu.py tests/test_join_varlen_gpu.py tests/test_join_ordered_gpu.py -m gpu -q --maxfail=40 -k "not (two_wg_per_cu and u32)" > $O/r04c_pytest.txt 2>&1; echo "rc=$?" >> $O/r04c_pytest.txt)
tail -4 $O/r04c_pytest.txt
timeout 900 python bench.py > $O/r04c_bench.json 2> $O/r04c_bench.err
tail -c 600 $O/r04c_bench.err
timeout 300 python tools/q3.py 10 > $O/r04c_q3_sf10.txt 2>&1
timeout 600 python tools/q3.py 100 --device-gen > $O/r04c_q3_sf100.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_q3 -o q3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/q3.py 100 --device-gen > $O/r04c_q3_prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_q3 -name "*kernel_stats.csv" | head -1) > $O/r04c_q3_rocprof.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_q3_$c -o q3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/q3.py 100 --device-gen > $O/r04c_q3_pmc_$c.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/summarise_prof.py pmc $(find $O/pmc_q3_FETCH_SIZE $O/pmc_q3_WRITE_SIZE -name "*counter_collection.csv") > $O/r04c_q3_pmc.txt 2>&1
head -30 $O/r04c_q3_rocprof.txt | cut -c1-160
rm -rf $O/prof_q3/*/*.db $O/pmc_q3_*/*/*.db 2>/dev/null
du -sh $O | tail -1
echo done
