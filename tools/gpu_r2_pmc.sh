#!/bin/bash
# PMC passes for roofline.traffic of the bench's two kernels (separate passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2), + hit rates
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2pmc}
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/$n -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/$n.json 2> $R/$O/$n.err
done
cd $R
python tools/summarize_prof.py $O/pmc_summary.txt --pmc $O/FETCH_SIZE/p_counter_collection.csv --pmc $O/WRITE_SIZE/p_counter_collection.csv --pmc $O/TCC_HIT_sum_TCC_MISS_sum/p_counter_collection.csv --pmc $O/TCC_EA0_RDREQ_sum_TCC_EA0_WRREQ_sum/p_counter_collection.csv
grep -E "k_lds_probe|k_radix_partition<1024, 16" $O/pmc_summary.txt
TSQ_TEST_TIMING=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=29999 timeout 600 python tests/dist_gpu_worker.py 2>&1 | tail -12
