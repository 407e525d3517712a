#!/bin/bash
# round 5, last call: the driver's bench command once more on the final code (the rocprof / PMC summaries stay those of tools/gpu_r05_final.sh)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
( time timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench.json 2> $O/r05_bench.err ) 2> $O/r05_bench.time
cp $O/bench_extras.json $O/r05_bench_extras.json
tail -3 $O/r05_bench.time; wc -c $O/r05_bench.json
