#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    for k in ("gd_tile_kernel", "gd_ltile2_kernel", "gd_ltile_kernel", "gd_ckpt_kernel", "gd_prep_kernel", "gd_runs_order_kernel",
              "gd_expand_scatter_kernel", "gd_scan_kernel", "gd_depthwed", "gd_inflate_kernel", "gd_bam_walk_kernel", "gd_region", "gd_"):
        if k in name:
            return k + (name[name.index("<"):name.index(">") + 1] if "<" in name else "")
    return name[:60]


for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (%s)" % os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        if "gd" not in row["Name"]: continue
        print("  %-46s calls=%-5s total_ns=%-13s avg_ns=%-12s pct=%s" % (
            short(row["Name"]), row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"]))

agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "gd::" not in row["Kernel_Name"]: continue
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
print("== PMC (mean per dispatch)")
for k in sorted(agg):
    print(" ", k)
    for c in sorted(agg[k]):
        print("     %-24s %.6g  (n=%d)" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
