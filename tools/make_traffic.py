#!/usr/bin/env python3
"""profiles/traffic_rNN.json from the per-(kernel, grid) PMC summary of a bench.py run (tools/summarise_prof.py pmc-by-grid over the
FETCH_SIZE and the WRITE_SIZE pass):  make_traffic.py <pmc-by-grid.txt> <out.json> [source note]
bench.py reads the file for `roofline.traffic` (the two kernels of the headline step) and for the `traffic_KiB_per_launch` of its side
measurements.  FETCH_SIZE / WRITE_SIZE are KiB; traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE (the gfx950 correction for wide
streaming reads, MI355X_MICROARCH.md, HBM section)."""
import json
import re
import sys

HEADLINE = {  # bench.py's names of the headline step's kernels -> (kernel name as rocprofv3 prints it, grid)
    "k_da_partition2<512,8,4,true>": ("void k_da_partition2<512, 8, 4, true, unsigned short, false, false>(DaSrc, DaDomain, DaStore)", 262144),  # (last parameters: rows carry a NULL bitmap, several key columns)
    "k_da_probe_count<512,uint16_t>": ("void k_da_probe_count<512, unsigned short, false, false, false>(DaProbeArgs)", 262144),
}


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    kernels = {}
    for line in open(src):
        m = re.match(r"^(.*?) \[grid (\d+)\]\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([0-9.]+)\s*$", line)
        if m:
            name, grid, counter, calls, avg = m.group(1).strip(), int(m.group(2)), m.group(3), int(m.group(4)), float(m.group(5))
        else:  # a name longer than the summary's column lost its "[grid N]" (round 6: k_da_partition2's seventh template parameter): one shape per kernel then
            m = re.match(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([0-9.]+)\s*$", line)
            if not m or m.group(1).startswith(("kernel", "==")):
                continue
            name, grid, counter, calls, avg = m.group(1).strip(), 0, m.group(2), int(m.group(3)), float(m.group(4))
        e = kernels.setdefault("%s [grid %d]" % (name, grid), {"launches": calls})
        e[counter] = avg
    out = {"source": ("rocprofv3 --kernel-trace --pmc <counter> -- python bench.py --steps 5 --no-cpu-baseline on one MI355X, separate passes for "
                      "FETCH_SIZE and WRITE_SIZE; one average per (kernel, grid size), so the shapes of a run are not mixed (1e8-row joins: grid "
                      "262144 = 256 workgroups). FETCH_SIZE / WRITE_SIZE are KiB; traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: wide "
                      "streaming reads count half). " + note),
           "workload": {"probe_rows": 100000000, "build_rows": 100000000, "radix_bits": 11}}
    for key, (name, grid) in HEADLINE.items():
        e = kernels.get("%s [grid %d]" % (name, grid))
        if e is None:  # (a template parameter added since: match the name up to its parameter list)
            # (round 4's summaries print k_da_partition2 without its last template parameter)
            stem, cand = name.split(">(")[0], []
            while not cand and "," in stem:  # (earlier rounds' summaries print the kernel with fewer template parameters)
                stem = stem.rsplit(",", 1)[0]
                cand = [v for k, v in kernels.items() if k.startswith(stem + ">(") and k.endswith("[grid %d]" % grid)]
            if not cand:  # ... or the summary cut the name before its grid (one shape per kernel in a headline-only pass)
                cand = [v for k, v in kernels.items() if k.startswith(name.split(">(")[0] + ">(") and k.endswith("[grid 0]")]
            e = cand[0] if len(cand) == 1 else None
        if e and "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            out[key] = {"FETCH_SIZE_KiB": e["FETCH_SIZE"], "WRITE_SIZE_KiB": e["WRITE_SIZE"], "launches": e["launches"],
                        "traffic_bytes": int(round((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0, -3))}
    out["kernels_KiB_per_launch"] = {k: v for k, v in sorted(kernels.items()) if not k.startswith("__amd") and v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0) >= 1024.0}
    json.dump(out, open(dst, "w"), indent=1)
    print("%d kernels, headline: %s" % (len(out["kernels_KiB_per_launch"]), {k: out.get(k, {}).get("traffic_bytes") for k in HEADLINE}))


if __name__ == "__main__":
    main()
