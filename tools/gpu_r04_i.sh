#!/bin/bash
# round 4, run I: A/B of the packed aggregate's partition kernel (two 512-thread workgroups per CU vs one of 1024), pinned-buffer pool
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_agg_packed_gpu.py tests/test_agg_gpu.py -m gpu -q 2>&1 | tail -15 > $O/r04i_pytest.txt
C3=c3_agg_1e9_1e6,c3_agg_1e9_1e6_double,c3_zipf_s1,pcie_inclusive_1e7
timeout 400 python bench.py --no-cpu-baseline --steps 3 --only-extras $C3 > $O/r04i_bench_part2.json 2> $O/r04i_bench_part2.err
timeout 400 python bench.py --no-cpu-baseline --steps 3 --only-extras $C3 --knob DAAGG_PART2=0 > $O/r04i_bench_part1.json 2> $O/r04i_bench_part1.err
ls -la $O | grep r04i
echo done
