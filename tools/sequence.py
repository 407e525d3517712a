#!/usr/bin/env python3
"""The order of dispatches and memory copies on the device, out of rocprofv3 --kernel-trace --memory-copy-trace CSVs: one line per event
(start in µs since the first listed one, duration, kernel name or copy direction + bytes).  usage: sequence.py DIR [--last N]
Used to find the small copies / fills a pipeline issues per batch (Q3: 165 copies and 113 fills of a few bytes per run, ~9 µs each)."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 400
    ev = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:70] + "  grid " + r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?")))))
    ev.sort()
    ev = ev[-last:]
    t0 = ev[0][0]
    for s, e, n in ev:
        print("%10.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))


if __name__ == "__main__":
    main()
