#!/bin/bash
# round 5, run F: both distributed plans emulated on one GPU (W = 2, 4, 8), the whole GPU suite with durations, the bench line
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dist_emulate.py > $O/r05_dist_emulated.json 2> $O/r05_dist_emulated.err
tail -3 $O/r05_dist_emulated.err; head -c 600 $O/r05_dist_emulated.json; echo
SECONDS=0
timeout 1100 python -m pytest tests -m gpu -q --durations=40 > $O/r05_pytest_f.txt 2>&1
echo "suite wall seconds: $SECONDS" >> $O/r05_pytest_f.txt
tail -50 $O/r05_pytest_f.txt
SECONDS=0
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_f.json 2> $O/r05_bench_f.err
echo "bench wall seconds: $SECONDS"; wc -c $O/r05_bench_f.json; tail -2 $O/r05_bench_f.err
