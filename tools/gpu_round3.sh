#!/bin/bash
# parity tests + bench (radix auto and off) + rocprofv3 of the radix bench.  Run from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu2.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu2.txt
tail -15 $O/pytest_gpu2.txt
timeout 600 python bench.py > $O/bench_radix.json 2> $O/bench_radix.err; echo "bench exit $?"; cat $O/bench_radix.json | cut -c1-1500; tail -3 $O/bench_radix.err
timeout 600 python bench.py --radix off --no-cpu-baseline > $O/bench_direct.json 2>> $O/bench_radix.err; cat $O/bench_direct.json | cut -c1-600
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_v2 -o v2 --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_v2_bench.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_v2 -o v2 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_v2.txt 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_v2 -o v2 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write_v2.txt 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $O/pmc_tcc_v2 -o v2 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_tcc_v2.txt 2>&1
cat $O/prof_v2/v2_kernel_stats.csv | head -8
