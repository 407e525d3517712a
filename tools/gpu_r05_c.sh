#!/bin/bash
# round 5, run C: kernel times of the Zipf / uniform / sparse C3 variants (one rocprofv3 pass each), the stream aggregate + decode tests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_streamagg_gpu.py tests/test_decode_gpu.py -m gpu -q -x > $O/r05_c_tests.txt 2>&1
tail -25 $O/r05_c_tests.txt
cd /tmp
for v in c3_zipf_s1 c3_agg_1e9_1e6; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c_$v -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras $v --extras-file r05_c_$v.json > $O/r05_c_$v.line 2> $O/r05_c_$v.err
  python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_c_$v -name "*kernel_stats.csv" | head -1) > $O/r05_c_${v}_rocprof.txt 2>&1
  rm -rf $O/prof_c_$v
  head -30 $O/r05_c_${v}_rocprof.txt
done
