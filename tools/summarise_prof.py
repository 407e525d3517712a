#!/usr/bin/env python3
"""rocprofv3 CSV output -> the text summaries committed under profiles/.
   summarise_prof.py stats <kernel_stats.csv>                 per-kernel calls / total / average / share
   summarise_prof.py pmc <counter_collection.csv> [...]       per-kernel, per-counter average over the launches
   summarise_prof.py pmc-by-grid <counter_collection.csv> ..  the same, one line per (kernel, grid size): shapes are not mixed"""
import collections
import csv
import sys


def stats(path):
    rows = list(csv.DictReader(open(path)))
    print("%-110s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "share"))
    for r in rows:
        print("%-110s %6d %12.3f %12.1f %6.2f%%" % (r["Name"][:110], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def pmc(paths, by_grid=False):
    """by_grid: one line per (kernel, grid size) — a kernel that serves several shapes of a run (the 1e8-row joins and the 1e7-row PCIe
    case of bench.py) gets one average per shape instead of a mixed one (profile hygiene)"""
    for path in paths:
        acc = collections.OrderedDict()
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"] + ((" [grid %s]" % r.get("Grid_Size", "?")) if by_grid else "")
            key = (name, r["Counter_Name"])
            d = acc.setdefault(key, {})
            d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])  # one row per (dispatch, counter instance)
        print("== rocprofv3 --pmc : %s" % path)
        print("%-110s %-24s %6s %16s" % ("kernel", "counter", "calls", "avg_value"))
        for (k, c), d in acc.items():
            kk = k if len(k) <= 110 else k[:110 - len(k[k.rindex(" [grid"):])] + k[k.rindex(" [grid"):]  # (a long name keeps its "[grid N]")
            print("%-110s %-24s %6d %16.1f" % (kk, c, len(d), sum(d.values()) / len(d)))
        print()


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "pmc-by-grid":
        pmc(sys.argv[2:], by_grid=True)
    else:
        pmc(sys.argv[2:])
