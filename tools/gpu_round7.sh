#!/bin/bash
# rowcodec slice: full-size parity test, 25M-row bench with the LDS tile sweep, rocprofv3 kernel trace of the bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 60 python -m pytest tests/test_storage_rowcodec_gpu.py -q -x > $O/rowcodec_tests2.txt 2>&1; echo "pytest exit $?" >> $O/rowcodec_tests2.txt
tail -3 $O/rowcodec_tests2.txt
TSQ_RC_SWEEP=1 timeout 90 python tools/bench_rowcodec.py 25e6 > $O/rowcodec_bench2.json 2> $O/rowcodec_bench2.err; echo "bench exit $?"; cut -c1-900 $O/rowcodec_bench2.json; tail -2 $O/rowcodec_bench2.err
export TMPDIR=/tmp
cd /tmp
timeout 90 rocprofv3 --kernel-trace --stats -d $O/prof_rc -o rc --output-format csv -- python $R/tools/bench_rowcodec.py 25e6 > $O/prof_rc.txt 2>&1
python $R/tools/summarize_prof.py $O/rowcodec_prof_summary.txt --stats $O/prof_rc/rc_kernel_stats.csv
head -8 $O/rowcodec_prof_summary.txt | cut -c1-160
