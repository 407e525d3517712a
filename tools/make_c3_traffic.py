#!/usr/bin/env python3
"""profiles/rNN_c3_pmc.txt: the aggregate's HBM traffic per 2.5e8-row batch, from the committed counter summary (profiles/traffic_rNN.json):
   python tools/make_c3_traffic.py [r05g]
(r05g = round 5's run G, taken before the string-key aggregate existed: its child runs the same kernels on 1e7-row batches, and the
per-(kernel, grid) averages of the final run mix those with the 2.5e8-row batches of C3)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05g"
tk = json.load(open(os.path.join(ROOT, "profiles", "traffic_%s.json" % rnd)))["kernels_KiB_per_launch"]


def row(name):
    for k, v in tk.items():
        if k.startswith(name):
            return k, v
    raise KeyError(name)


rows = [("C3 BIGINT (SUM, COUNT(*)), 2.5e8 rows per batch, algorithmic 16 B/row = 4.00 GB",
         ["void k_daagg_partition<512, 8, 1, 2, false>", "void k_agg_da<3, 4096, 1>"]),
        ("C3 DOUBLE (8-byte argument cells travel), algorithmic 16 B/row = 4.00 GB",
         ["void k_daagg_partition<1024, 8, 1, 8, true>", "void k_agg_da<2, 4096, 2>"]),
        ("C3 sparse 64-bit keys (radix + LDS hash tables), algorithmic 16 B/row = 4.00 GB",
         ["void k_radix_partition<1024, 8, 4, 1, false, true>", "void k_agg_lds<1, 3>"])]
out = ["# HBM traffic of the aggregate's kernels per launch (one 2.5e8-row batch), from profiles/%s_bench_pmc.txt = rocprofv3 --pmc FETCH_SIZE and" % rnd,
       "# --pmc WRITE_SIZE passes of `bench.py --no-cpu-baseline --steps 5`; bytes = FETCH_SIZE KiB x 2 (gfx950: 64 B units counted as 32 for wide",
       "# reads, MI355X_MICROARCH.md) + WRITE_SIZE KiB.  (k_agg_da<3,4096,1>'s average mixes uniform and Zipf batches.)", ""]
for title, names in rows:
    out.append(title)
    tot = 0.0
    for n in names:
        k, v = row(n)
        b = (v["FETCH_SIZE"] * 2 + v["WRITE_SIZE"]) * 1024 / 1e9
        tot += b
        out.append("  %-100s launches %3d  read %.2f GB  written %.2f GB" % (k[:100], v["launches"], v["FETCH_SIZE"] * 2 * 1024 / 1e9, v["WRITE_SIZE"] * 1024 / 1e9))
    out.append("  total %.2f GB per batch = %.2fx the algorithmic bytes (the partition store is written once and read once)" % (tot, tot / 4.0))
    out.append("")
open(os.path.join(ROOT, "profiles", "%s_c3_pmc.txt" % rnd.rstrip("g")), "w").write("\n".join(out))
print("\n".join(out))
