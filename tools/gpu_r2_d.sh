#!/bin/bash
# round 2, GPU call D: re-queueing probe rounds: parity, phase profile, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r2e
O=gpurun_out/r2e
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_join_radix_gpu.py tests/test_build_partitioned_gpu.py -x -q > $O/tests_join.txt 2>&1
echo "tests_join rc=$?" >> $O/rc.txt
TSQ_LDS_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_lds.json 2> $O/prof_lds.err
TSQ_LDS_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --build-rows 10000000 > $O/prof_c2.json 2> $O/prof_c2.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_lds.json 2> $O/bench_lds.err
echo "bench_lds rc=$?" >> $O/rc.txt
for lf in 0.5 0.65; do
  TSQ_TABLE_LF=$lf timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_lds_lf$lf.json 2> $O/bench_lds_lf$lf.err
  echo "bench_lf$lf rc=$?" >> $O/rc.txt
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --build-rows 10000000 > $O/bench_c2.json 2> $O/bench_c2.err
echo "bench_c2 rc=$?" >> $O/rc.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU -d $R/$O/pmc1 -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc1.json 2> $R/$O/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES -d $R/$O/pmc2 -o p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/pmc2.json 2> $R/$O/pmc2.err
cd $R
python tools/summarize_prof.py $O/pmc_summary.txt --pmc $O/pmc1/p_counter_collection.csv --pmc $O/pmc2/p_counter_collection.csv 2>&1
cat $O/rc.txt
tail -4 $O/tests_join.txt
grep "lds-prof" $O/*.err | head -4
for f in $O/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f'))
r=d.get('roofline',{})
print(d['value'], d['ms_per_step'], d['verified'], d.get('probe_strategy'), r.get('kernel_ms'), r.get('partition',{}).get('kernel_ms'), r.get('probe_phase',{}).get('frac'), d['build_kernel_ms'], d['table_bytes'])
" 2>&1 | tail -1; done
grep -E "k_lds_probe" $O/pmc_summary.txt
