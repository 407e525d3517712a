#!/bin/bash
# round 5, run D: hot keys of the packed aggregate (sampled, absorbed in the partition kernel) — tests + C3 variants with / without
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_agg_packed_gpu.py tests/test_agg_fast_gpu.py tests/test_agg_gpu.py -m gpu -q > $O/r05_d_tests.txt 2>&1
tail -25 $O/r05_d_tests.txt
for knob in "" "--knob DAAGG_HOT=0"; do
  timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $knob --only-extras c3_agg_1e9_1e6,c3_agg_1e9_1e6_double,c3_zipf_s1,c3_sparse_keys --extras-file r05_d_x.json 2> $O/r05_d.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$knob', json.dumps(d['sides']))"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_d -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras c3_zipf_s1 --extras-file r05_d_zipf.json > $O/r05_d_zipf.line 2> $O/r05_d_zipf.err
python $GRAFT_REPO_ROOT/tools/summarise_prof.py stats $(find $O/prof_d -name "*kernel_stats.csv" | head -1) > $O/r05_d_zipf_rocprof.txt 2>&1
rm -rf $O/prof_d
head -12 $O/r05_d_zipf_rocprof.txt
