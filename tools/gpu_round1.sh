#!/bin/bash
# GPU call: parity tests, radix sweep, rocprof of the current bench.  Run from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
timeout 600 ./tools/radix_ubench > $O/radix_ubench.txt 2>&1; echo "ubench exit $?" >> $O/radix_ubench.txt
tail -5 $O/radix_ubench.txt
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_v1 -o v1 --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_v1_bench.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_v1 -o v1 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch_v1.txt 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_v1 -o v1 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write_v1.txt 2>&1
ls -R $O | head -50
