#!/bin/bash
# PMC passes for k_rowcodec_decode_pipe (tools/bench_rowcodec.py 1e7): HBM traffic (FETCH_SIZE / WRITE_SIZE, one per pass) and the SQ
# instruction / wait mix.  Counters only — no trace domains in the same run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 40 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_rc_F -o f --output-format csv -- python $R/tools/bench_rowcodec.py 1e7 > $O/pmc_rc_F.txt 2>&1
timeout 40 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_rc_W -o w --output-format csv -- python $R/tools/bench_rowcodec.py 1e7 > $O/pmc_rc_W.txt 2>&1
timeout 40 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc_rc_S -o s --output-format csv -- python $R/tools/bench_rowcodec.py 1e7 > $O/pmc_rc_S.txt 2>&1
python $R/tools/summarize_prof.py $O/rowcodec_pmc_summary.txt --pmc $O/pmc_rc_F/f_counter_collection.csv --pmc $O/pmc_rc_W/w_counter_collection.csv --pmc $O/pmc_rc_S/s_counter_collection.csv \
  --note "k_rowcodec_decode_pipe<6>, 1e7 rows x 5 columns (439.5 MB of stored rows, 80 MB of offsets, 400 MB of values out); FETCH_SIZE / WRITE_SIZE in KiB per dispatch"
grep -E "rowcodec" $O/rowcodec_pmc_summary.txt | cut -c1-170
tail -3 $O/pmc_rc_S.txt | cut -c1-300
