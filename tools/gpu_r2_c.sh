#!/bin/bash
# round 2, GPU call C: where does k_lds_probe_count spend its cycles?  s_memtime phases + SQ counters
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c
export TMPDIR=/tmp
R=$(pwd)
TSQ_LDS_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_lds.json 2> $O/prof_lds.err
TSQ_LDS_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --build-rows 10000000 > $O/prof_c2.json 2> $O/prof_c2.err
TSQ_LDS_PROF=1 TSQ_TABLE_LF=0.5 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_lf5.json 2> $O/prof_lf5.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d $R/$O/pmc1 -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc1.json 2> $R/$O/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $R/$O/pmc2 -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc2.json 2> $R/$O/pmc2.err
cd $R
python tools/summarize_prof.py $O/pmc_summary.txt --pmc $O/pmc1/p_counter_collection.csv --pmc $O/pmc2/p_counter_collection.csv 2>&1
grep "lds-prof" $O/*.err
grep -E "k_lds_probe|k_radix_partition<1024, 16" $O/pmc_summary.txt
tail -3 $O/pmc1.err
