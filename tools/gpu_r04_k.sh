#!/bin/bash
# round 4, run K: the sort pass of the travelling build columns back at run E's shape; the C++ host mirror's new case; the bench line and
# its rocprofv3 kernel stats at the final code
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
{
echo "== python -m pytest tests/test_join_packed_gpu.py tests/test_packed_auto_gpu.py tests/test_host_cpp_gpu.py tests/test_join_gpu.py -m gpu -q"
timeout 900 python -m pytest tests/test_join_packed_gpu.py tests/test_packed_auto_gpu.py tests/test_host_cpp_gpu.py tests/test_join_gpu.py -m gpu -q 2>&1 | tail -12
} > $O/r04k_pytest.txt 2>&1
SECONDS=0
timeout 600 python bench.py > $O/r04k_bench.json 2> $O/r04k_bench.err
echo "default bench.py wall seconds: $SECONDS" > $O/r04k_bench.time
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_k -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 > $O/r04k_bench_prof.json 2> $O/r04k_bench_prof.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_k -name "*kernel_stats.csv" | head -1) > $O/r04k_bench_rocprof.txt 2>&1
rm -rf $O/prof_k 2>/dev/null
ls -la $O | grep r04k
echo done
