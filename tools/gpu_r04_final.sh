#!/bin/bash
# round 4, final run: the default bench line, the whole GPU suite, smoke, rocprofv3 kernel stats of the bench command,
# an A/B of the packed aggregate's one-word count+sum variant (DAAGG_SIG=2), PMC passes of the C3 lines
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
SECONDS=0
timeout 900 python bench.py > $O/r04_bench.json 2> $O/r04_bench.err
echo "default bench.py wall seconds: $SECONDS" > $O/r04_bench.time
OPS="tests/test_agg_gpu.py tests/test_agg_packed_gpu.py tests/test_agg_string_gpu.py tests/test_join_gpu.py tests/test_join_packed_gpu.py tests/test_join_radix_gpu.py tests/test_packed_auto_gpu.py tests/test_pipeline_gpu.py tests/test_comm_gpu.py"
DESEL=""; for f in $OPS; do DESEL="$DESEL --deselect $f"; done
{
echo "== python -m pytest tests -m gpu -q  (two processes: the operators | everything else)"
timeout 1200 python -m pytest $OPS -m gpu -q 2>&1 | tail -12
timeout 1200 python -m pytest tests -m gpu -q $DESEL 2>&1 | tail -12
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
} > $O/r04_pytest_gpu.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_final -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 > $O/r04_bench_prof.json 2> $O/r04_bench_prof.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_final -name "*kernel_stats.csv" | head -1) > $O/r04_bench_rocprof.txt 2>&1
cd $GRAFT_REPO_ROOT
{
echo "== TSQ_TEST_KNOBS=DAAGG_SIG=2 python -m pytest tests/test_agg_packed_gpu.py tests/test_agg_gpu.py -m gpu -q"
TSQ_TEST_KNOBS=DAAGG_SIG=2 timeout 600 python -m pytest tests/test_agg_packed_gpu.py tests/test_agg_gpu.py -m gpu -q 2>&1 | tail -8
} > $O/r04_sig3_pytest.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 3 --only-extras c3_agg_1e9_1e6,c3_zipf_s1 --knob DAAGG_SIG=2 > $O/r04_sig3_bench.json 2> $O/r04_sig3_bench.err
cd /tmp
C3=c3_agg_1e9_1e6,c3_agg_1e9_1e6_double,c3_zipf_s1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_final_$c -o c3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --only-extras $C3 > $O/r04_c3_pmc_$c.json 2> $O/r04_c3_pmc_$c.err
done
python $GRAFT_REPO_ROOT/tools/summarise_prof.py pmc-by-grid $(find $O/pmc_final_FETCH_SIZE $O/pmc_final_WRITE_SIZE -name "*counter_collection.csv") > $O/r04_c3_pmc.txt 2>&1
rm -rf $O/prof_final $O/pmc_final_* 2>/dev/null
ls -la $O | grep "r04_"
echo done
