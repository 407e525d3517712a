#!/bin/bash
# round 5, run L: every dispatch of the stream-aggregate kernels (duration, grid) of the side measurement
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace -d $O/prof_l -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --only-extras stream_agg_1e8_ordered --extras-file r05_l_x.json > $O/r05_l.line 2> $O/r05_l.err
python3 - <<'P'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_l/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    n = r["Kernel_Name"]
    if "k_sa_" in n or "k_agg_finalize" in n or "fillBuffer" in n and False:
        print(n[:40], "grid", r.get("Grid_Size") or r.get("Grid_Size_X"), "wg", r.get("Workgroup_Size") or r.get("Workgroup_Size_X"), "us %.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
P
rm -rf $O/prof_l
