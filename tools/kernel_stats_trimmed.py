#!/usr/bin/env python3
"""Per-kernel launch statistics of a rocprofv3 --kernel-trace run WITHOUT the warm-up dispatches.

    python tools/kernel_stats_trimmed.py <dir with *_kernel_trace.csv> <out.csv>

rocprofv3's own *_kernel_stats.csv averages every dispatch of a kernel, and since round 5 `gd_create` runs one compute on an
8 kb contig (it takes the first launch of every tile-path kernel out of a process's first compute): those ~10 us dispatches
pull `AverageNs` down -- a reader recomputing a roofline fraction from the file got 0.69 where the workload's launches give
0.66.  Here a dispatch counts when its grid is at least 1 % of the kernel's largest grid -- the rule
tools/traffic_from_pmc.py applies to the counter passes -- and the file says how many were left out."""
import csv
import glob
import os
import statistics
import sys

root, out = sys.argv[1], sys.argv[2]
per = {}
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Name")
        if not name or "gd::" not in name:
            continue
        grid = float(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        per.setdefault(name, []).append((grid, dur))
rows = []
tot = sum(d for v in per.values() for g, d in v if g >= 0.01 * max(x for x, _ in v))
for name, v in per.items():
    top = max(g for g, _ in v)
    keep = [d for g, d in v if g >= 0.01 * top]
    rows.append((name, len(keep), sum(keep), sum(keep) / len(keep), 100.0 * sum(keep) / tot if tot else 0.0, min(keep), max(keep),
                 statistics.pstdev(keep) if len(keep) > 1 else 0.0, len(v) - len(keep)))
rows.sort(key=lambda r: -r[2])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "WarmupDispatchesLeftOut"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), "%.3f" % r[3], "%.2f" % r[4], int(r[5]), int(r[6]), "%.3f" % r[7], r[8]])
print(open(out).read())
