#!/usr/bin/env python3
"""HBM traffic of gd_tile_kernel from rocprofv3 PMC passes.

    python tools/traffic_from_pmc.py gpurun_out/prof_<tag> profiles/r01_wgs_traffic.json [kernel+kernel...] [bench args] [label]

A kernel is a substring of the demangled name after "::" (e.g. "gd_tile_fast_kernel<1, true>"); `label` is what the
JSON calls it (bench.py accepts a file only for the kernel it timed: gd_stats.tile_kernel's name).

The optional third argument lists the kernels whose traffic is summed (default
gd_tile_kernel; the chunk path's roofline kernel is gd_ckpt_kernel,gd_ltile2_kernel).

Reads the counter_collection CSVs of the separate `--pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` passes made by tools/prof.sh and applies the corrections of
MI355X_MICROARCH.md (section HBM): both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so it
is doubled; WRITE_SIZE is taken as is.  Values are means per dispatch -- over the dispatches of the WORKLOAD: since
round 5 `gd_create` runs one compute on an 8 kb contig to warm the runtime's copy and fill paths, and those launches
(a grid below 1 % of the kernel's largest) are left out of the mean and counted in `dispatches_left_out`.
"""
import csv
import glob
import json
import os
import sys

root, out = sys.argv[1], sys.argv[2]
kernels = sys.argv[3].split("+") if len(sys.argv) > 3 else ["gd_tile_kernel"]
bench_args = sys.argv[4] if len(sys.argv) > 4 else ""
label = sys.argv[5] if len(sys.argv) > 5 else "+".join(kernels)
fetch_kib = write_kib = 0.0
n_disp = {}
left_out = {}
for kname in kernels:
    raw = {"FETCH_SIZE": [], "WRITE_SIZE": []}
    for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if ("::" + kname) in row["Kernel_Name"] and row["Counter_Name"] in raw:
                raw[row["Counter_Name"]].append((float(row.get("Grid_Size") or 0), float(row["Counter_Value"])))
    acc = {}
    for k, v in raw.items():
        top = max((g for g, _ in v), default=0.0)
        acc[k] = [x for g, x in v if g >= 0.01 * top]
        left_out[kname + ":" + k] = len(v) - len(acc[k])
    fetch_kib += sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])      # mean per dispatch, summed over the kernels
    write_kib += sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
    n_disp[kname] = {k: len(v) for k, v in acc.items()}
read_b = fetch_kib * 1024 * 2
write_b = write_kib * 1024
res = {
    "kernel": label, "kernels_matched": kernels,
    "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
    "dispatches": n_disp, "dispatches_left_out": left_out,
    "hbm_read_bytes_per_launch": read_b, "hbm_write_bytes_per_launch": write_b,
    "hbm_bytes_per_launch": read_b + write_b,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof.sh), "
              "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; "
              "python bench.py %s--no-cpu-baseline --no-host-stream --emulate-shards= --steps 5 --warmup 2" % (bench_args + " " if bench_args else ""),
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
