#!/bin/bash
# full GPU round: parity tests, headline bench (+rocprof), Q3-shaped pipeline (+rocprof), build and kernel side benches.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu4.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu4.txt
tail -6 $O/pytest_gpu4.txt
timeout 600 python bench.py > $O/bench_v3.json 2> $O/bench_v3.err; echo "bench exit $?"; cut -c1-400 $O/bench_v3.json; tail -2 $O/bench_v3.err
timeout 300 python tools/q3.py 10 > $O/q3.json 2> $O/q3.err; cat $O/q3.json
timeout 300 python tools/bench_build.py > $O/build.json 2>&1; cat $O/build.json | cut -c1-260
timeout 600 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err; cat $O/configs.json | cut -c1-400
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_v3 -o v3 --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_v3_bench.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_q3 -o q3 --output-format csv -- python $R/tools/q3.py 10 > $O/prof_q3.txt 2>&1
python $R/tools/summarize_prof.py $O/bench_v3_prof_summary.txt --stats $O/prof_v3/v3_kernel_stats.csv
python $R/tools/summarize_prof.py $O/q3_prof_summary.txt --stats $O/prof_q3/q3_kernel_stats.csv
head -8 $O/bench_v3_prof_summary.txt | cut -c1-150
