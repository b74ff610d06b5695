#!/usr/bin/env python3
"""A read's kernels over time, from a `rocprofv3 --kernel-trace` CSV (e.g. of `tools/scope3.py --rocprof DIR`): for every
slice of the run, how much of it each class of kernel (inflate, CRC, record walks, the copy kernel, indexing, depth) was
present on the device and how many launches of it were in flight on average -- where the device waits for bytes, and where
one class waits for another, shows as slices in which nothing (or only one thing) runs.
    python tools/timeline.py DIR/scope3_kernel_trace.csv [--slice-ms 50]"""
import argparse
import csv
import json

CLASSES = [("inflate", "gd_inflate_kernel"), ("crc", "gd_inflate_crc"), ("walk", "gd_bam_walk_kernel"), ("copy", "gd_h2d_kernel"),
           ("copy", "gd_copy_words"), ("index", "gd_index_records"), ("depth", "gd_tile"), ("depth", "gd_prep"), ("depth", "gd_runs"),
           ("runtime", "__amd_rocclr")]


def classify(name):
    for c, pat in CLASSES:
        if pat in name:
            return c
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--slice-ms", type=float, default=50.0)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    ev = []
    with open(a.trace) as fh:
        for r in csv.DictReader(fh):
            ev.append((classify(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not ev:
        raise SystemExit("no kernels in " + a.trace)
    t0 = min(e[1] for e in ev)
    t1 = max(e[2] for e in ev)
    sl = a.slice_ms * 1e6
    n = int((t1 - t0) / sl) + 1
    classes = sorted({e[0] for e in ev})
    rows = []
    for i in range(n):
        lo, hi = t0 + i * sl, t0 + (i + 1) * sl
        row = {"from_ms": round(i * a.slice_ms, 1)}
        for c in classes:
            iv = sorted((max(lo, s), min(hi, e)) for k, s, e in ev if k == c and s < hi and e > lo)
            busy, inflight, end = 0.0, 0.0, lo
            for s, e in iv:
                inflight += e - s
                if e > end:
                    busy += e - max(s, end)
                    end = e
            row[c] = {"present": round(busy / sl, 2), "in_flight": round(inflight / sl, 2)}
        rows.append(row)
    total = {c: round(sum(e - s for k, s, e in ev if k == c) / 1e6, 1) for c in classes}
    if a.json:
        print(json.dumps({"span_ms": round((t1 - t0) / 1e6, 1), "kernel_ms_by_class": total, "slices": rows}))
        return
    print("span %.1f ms; kernel time by class (ms): %s" % ((t1 - t0) / 1e6, total))
    print("%8s  " % "from ms" + "  ".join("%-14s" % c for c in classes) + "   (present | launches in flight)")
    for r in rows:
        print("%8.1f  " % r["from_ms"] + "  ".join("%4.2f | %-7.2f" % (r[c]["present"], r[c]["in_flight"]) for c in classes))


if __name__ == "__main__":
    main()
