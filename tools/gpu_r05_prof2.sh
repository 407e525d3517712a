#!/bin/bash
# round 5: kernel stats of the Q3 pipeline and of the key-record / dictionary kernels (string join + string aggregate), SQ counters of the latter
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_q3 -o q --output-format csv -- python3 $R/tools/q3.py 100 --device-gen > $O/r05_q3.line 2> $O/r05_q3.err
( echo "# TPC-H Q3-shaped pipeline at SF 100 on one MI355X (tools/q3.py 100 --device-gen), round 5: rocprofv3 --kernel-trace --stats; table generation + 4 runs of the plan (per run: calls / 4)"; echo "# result line:"; tail -1 $O/r05_q3.line | cut -c1-900; python3 $R/tools/summarise_prof.py stats $(find $O/prof_q3 -name "*kernel_stats.csv" | head -1) ) > $O/r05_q3_rocprof.txt 2>&1
rm -rf $O/prof_q3
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_k -o k --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras two_key_bigint_string_count,agg_string_keys_1e7_1e5,agg_string_keys_1e7_5e6 --extras-file r05_k_x.json > $O/r05_k.line 2> $O/r05_k.err
( echo "# key-record join (1e7 x 1e7 (bigint, varstring) keys: COUNT(*) and materialised) and dictionary aggregate (1e7 rows, 1e5 / 4.3e6 string groups): rocprofv3 --kernel-trace --stats of"; echo "# bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras two_key_bigint_string_count,agg_string_keys_1e7_1e5,agg_string_keys_1e7_5e6"; python3 $R/tools/summarise_prof.py stats $(find $O/prof_k -name "*kernel_stats.csv" | head -1) ) > $O/r05_keyrec_rocprof.txt 2>&1
rm -rf $O/prof_k
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_k_$tag -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras two_key_bigint_string_count,agg_string_keys_1e7_1e5 --extras-file r05_k_x.json > $O/r05_k.line 2> $O/r05_k.err
  python3 - "$O/pmc_k_$tag" <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "k_kr_" in n or "k_kd_" in n:
            k = (n.split("(")[0][:24], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
for k, (c, v) in sorted(acc.items()):
    print("%-26s %-22s launches %3d  avg per launch %.4e" % (k[0], k[1], c, v / c))
P
  rm -rf $O/pmc_k_$tag
done > $O/r05_keyrec_sq.txt 2>&1
head -14 $O/r05_q3_rocprof.txt | cut -c1-160; head -16 $O/r05_keyrec_rocprof.txt | cut -c1-160; cat $O/r05_keyrec_sq.txt | head -60
