#!/usr/bin/env python3
"""rocprofv3 CSV output -> the text summaries committed under profiles/.
   summarise_prof.py stats <kernel_stats.csv>                 per-kernel calls / total / average / share
   summarise_prof.py pmc <counter_collection.csv> [...]       per-kernel, per-counter average over the launches"""
import collections
import csv
import sys


def stats(path):
    rows = list(csv.DictReader(open(path)))
    print("%-110s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "share"))
    for r in rows:
        print("%-110s %6d %12.3f %12.1f %6.2f%%" % (r["Name"][:110], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def pmc(paths):
    for path in paths:
        acc = collections.OrderedDict()
        for r in csv.DictReader(open(path)):
            key = (r["Kernel_Name"], r["Counter_Name"])
            d = acc.setdefault(key, {})
            d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])  # one row per (dispatch, counter instance)
        print("== rocprofv3 --pmc : %s" % path)
        print("%-110s %-24s %6s %16s" % ("kernel", "counter", "calls", "avg_value"))
        for (k, c), d in acc.items():
            print("%-110s %-24s %6d %16.1f" % (k[:110], c, len(d), sum(d.values()) / len(d)))
        print()


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
