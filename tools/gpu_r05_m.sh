#!/bin/bash
# round 5, run M: SQ instruction counters of the stream-aggregate kernels (two passes)
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_m_$tag -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras stream_agg_1e8_ordered --extras-file r05_m_x.json > $O/r05_m.line 2> $O/r05_m.err
  python3 - "$O/pmc_m_$tag" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "k_sa_" in n:
            k = (n[:28], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
for k, (c, v) in sorted(acc.items()):
    print("%-30s %-20s calls %d avg %.3e" % (k[0], k[1], c, v / c))
P
  rm -rf $O/pmc_m_$tag
done
