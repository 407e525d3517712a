#!/bin/bash
# round 4, run G: the dense partial state / narrow argument cells / overflow store of the packed aggregate
cd /root/repo
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_agg_packed_gpu.py -m gpu -q -x -k "dense or narrow" 2>&1 | tail -15
echo "rc=$?"
timeout 1500 python -m pytest tests/test_agg_packed_gpu.py tests/test_agg_gpu.py tests/test_pipeline_gpu.py -m gpu -q 2>&1 | tail -15
timeout 300 python -m pytest tests/test_join_packed_gpu.py -m gpu -q -k "mixed_signedness" 2>&1 | tail -5
} > gpurun_out/r04g_pytest.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --steps 5 --only-extras c3_agg_1e9_1e6,c3_agg_1e9_1e6_double,c3_zipf_s1,c3_sparse_keys,agg_two_keys_1000x100 > gpurun_out/r04g_bench_c3.json 2> gpurun_out/r04g_bench_c3.err
timeout 300 python tools/q3.py 100 --device-gen > gpurun_out/r04g_q3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o c3 --output-format csv -- python /root/repo/bench.py --no-cpu-baseline --steps 2 --only-extras c3_agg_1e9_1e6,c3_zipf_s1 > /root/repo/gpurun_out/r04g_prof.json 2> /root/repo/gpurun_out/r04g_prof.err
python /root/repo/tools/summarise_prof.py stats $(find /tmp/prof_g -name "*kernel_stats.csv" | head -1) > /root/repo/gpurun_out/r04g_c3_rocprof.txt 2>&1
ls -la /root/repo/gpurun_out | tail -12
echo done
