#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) per kernel."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    """gd::fast::gd_tile_fast_kernel<1>(gd::Job) -> gd_tile_fast_kernel<1>"""
    m = re.search(r"(gd_[A-Za-z0-9_]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


# the statistics over the WORKLOAD's launches when tools/kernel_stats_trimmed.py has made them (gd_create's warm-up
# dispatches pull rocprofv3's own averages down: 14 calls averaging 3.54 ms for 13 launches of 3.81), else rocprofv3's
trimmed = os.path.join(root, "kernel_stats_trimmed.csv")
stats = [trimmed] if os.path.exists(trimmed) else glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
for f in stats:
    print("== kernel stats (%s)%s" % (os.path.relpath(f, root), ": warm-up dispatches of gd_create left out" if f == trimmed else ""))
    for row in csv.DictReader(open(f)):
        if "gd" not in row["Name"]: continue
        print("  %-46s calls=%-5s total_ns=%-13s avg_ns=%-12s pct=%s%s" % (
            short(row["Name"]), row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"],
            ("  left_out=%s" % row["WarmupDispatchesLeftOut"]) if "WarmupDispatchesLeftOut" in row else ""))

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
