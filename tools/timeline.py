#!/usr/bin/env python3
"""Timeline of the named kernels out of a rocprofv3 --kernel-trace CSV: start / end (µs since the first listed dispatch), duration, and how much of
each dispatch ran beside the dispatch listed before it.  usage: timeline.py kernel_trace.csv name_substring [name_substring ...] [--last N]"""
import csv
import sys


def main():
    args = sys.argv[1:]
    last = 24
    if "--last" in args:
        i = args.index("--last")
        last = int(args[i + 1])
        del args[i:i + 2]
    path, names = args[0], args[1:]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r.get("Kernel_Name", "")
            if any(n in k for n in names):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k.split("(")[0][:60], r.get("Queue_Id", "")))
    rows.sort()
    rows = rows[-last:]
    if not rows:
        print("no such dispatches")
        return
    t0 = rows[0][0]
    prev_end = None
    print(f"{'kernel':60s} {'queue':>6s} {'start_us':>10s} {'end_us':>10s} {'dur_us':>9s} {'beside_previous_us':>19s}")
    for s, e, k, q in rows:
        ov = max(0, min(e, prev_end) - s) if prev_end is not None else 0
        print(f"{k:60s} {q:>6s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {ov / 1e3:19.1f}")
        prev_end = e if prev_end is None else max(prev_end, e)
    print(f"span: {(max(r[1] for r in rows) - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
