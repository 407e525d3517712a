#!/bin/bash
# first hardware run of the record-key kernels + the coprocessor chain; clean rocprofv3 stats of the bench workload alone
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2u}
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 600 python -m pytest tests/test_storage_coprocessor_gpu.py -q -x > $O/pytest_cop.txt 2>&1
tail -15 $O/pytest_cop.txt
timeout 200 python tools/bench_tablecodec.py 1e8 > $O/tablecodec.json 2> $O/tablecodec.err; cat $O/tablecodec.json; tail -3 $O/tablecodec.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/$O/prof.json 2> $R/$O/prof.err
cd $R
python tools/summarize_prof.py $O/rocprof_noextras.txt --stats $O/prof/p_kernel_stats.csv --note "python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras under rocprofv3 --kernel-trace --stats: the bench workload alone (build once + 25 probe passes of 1e8 rows)"
head -12 $O/rocprof_noextras.txt | cut -c1-150
