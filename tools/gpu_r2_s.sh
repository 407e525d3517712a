#!/bin/bash
# SQ counters of the C3 aggregate kernels (two passes of <= 8 SQ counters)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/${1:-r2s}
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $R/$O/p1 -o p --output-format csv -- python $R/tools/bench_configs.py --skip-join --agg-rows 3e8 > $R/$O/p1.json 2> $R/$O/p1.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD -d $R/$O/p2 -o p --output-format csv -- python $R/tools/bench_configs.py --skip-join --agg-rows 3e8 > $R/$O/p2.json 2> $R/$O/p2.err
cd $R
python tools/summarize_prof.py $O/pmc.txt --pmc $O/p1/p_counter_collection.csv --pmc $O/p2/p_counter_collection.csv 2>&1 | tail -3
grep -E "k_agg_lds<1|k_radix_partition" $O/pmc.txt | cut -c1-60,90-140
tail -2 $O/p2.err | cut -c1-300
