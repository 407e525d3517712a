#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into one small text file under profiles/.
usage: summarize_prof.py OUT.txt --stats DIR/xxx_kernel_stats.csv [--pmc DIR/xxx_counter_collection.csv ...] [--note TEXT]
PMC values are averaged per (kernel, counter) over the dispatches of the run; FETCH_SIZE/WRITE_SIZE are KiB
(rocprofv3); on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x (MI355X_MICROARCH.md §HBM)."""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--stats", action="append", default=[])
ap.add_argument("--pmc", action="append", default=[])
ap.add_argument("--note", action="append", default=[])
a = ap.parse_args()
with open(a.out, "w") as f:
    for n in a.note:
        f.write("# %s\n" % n)
    for s in a.stats:
        f.write("\n== rocprofv3 --kernel-trace --stats : %s\n" % s)
        f.write("%-90s %6s %14s %12s %7s\n" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in csv.DictReader(open(s)):
            f.write("%-90s %6s %14s %12.0f %7s\n" % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"]))
    for p in a.pmc:
        f.write("\n== rocprofv3 --pmc : %s\n" % p)
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(p)):
            agg.setdefault((r["Kernel_Name"][:90], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        f.write("%-90s %-22s %6s %16s\n" % ("kernel", "counter", "calls", "avg_value"))
        for (k, c), v in agg.items():
            f.write("%-90s %-22s %6d %16.1f\n" % (k, c, len(v), sum(v) / len(v)))
