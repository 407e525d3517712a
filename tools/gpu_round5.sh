#!/bin/bash
# full GPU round: parity tests, smoke, headline bench (+rocprof), side benches (Q3 pipeline, build, C2/C3, decode, sort).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu5.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu5.txt
tail -4 $O/pytest_gpu5.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_v4.json 2> $O/bench_v4.err; echo "bench exit $?"; cut -c1-300 $O/bench_v4.json; tail -2 $O/bench_v4.err
timeout 300 python tools/q3.py 10 > $O/q3.json 2> $O/q3.err; cut -c1-400 $O/q3.json
timeout 500 python tools/q3.py 100 --device-gen > $O/q3_sf100.json 2>> $O/q3.err; cut -c1-400 $O/q3_sf100.json
timeout 300 python tools/bench_build.py > $O/build.json 2>&1
timeout 600 python tools/bench_configs.py > $O/configs.json 2> $O/configs.err
timeout 600 python tools/bench_decode.py > $O/decode.json 2> $O/decode.err; cut -c1-300 $O/decode.json
timeout 600 python tools/bench_sort.py > $O/sort.json 2> $O/sort.err; cut -c1-200 $O/sort.json
timeout 600 python tools/bench_emit.py > $O/emit.json 2> $O/emit.err; cut -c1-300 $O/emit.json
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_v4 -o v4 --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_v4_bench.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_q3 -o q3 --output-format csv -- python $R/tools/q3.py 10 > $O/prof_q3.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_dec -o d --output-format csv -- python $R/tools/bench_decode.py > $O/prof_dec.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_sort -o s --output-format csv -- python $R/tools/bench_sort.py > $O/prof_sort.txt 2>&1
python $R/tools/summarize_prof.py $O/bench_v4_prof_summary.txt --stats $O/prof_v4/v4_kernel_stats.csv
python $R/tools/summarize_prof.py $O/q3_prof_summary.txt --stats $O/prof_q3/q3_kernel_stats.csv
python $R/tools/summarize_prof.py $O/decode_prof_summary.txt --stats $O/prof_dec/d_kernel_stats.csv
python $R/tools/summarize_prof.py $O/sort_prof_summary.txt --stats $O/prof_sort/s_kernel_stats.csv
head -7 $O/bench_v4_prof_summary.txt | cut -c1-150; head -9 $O/sort_prof_summary.txt | cut -c1-150
