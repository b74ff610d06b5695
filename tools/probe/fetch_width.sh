#!/bin/bash
# FETCH_SIZE per load width on this device: tools/probe/fetch_width.sh  (run on the GPU box; prints the calibration table)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/fetch_width
rm -rf $out; mkdir -p $out
$R/tools/probe/fetch_width > $out/timing.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc -- $R/tools/probe/fetch_width > $out/pmc.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, os, json
out = sys.argv[1]
GiB8 = float(8 << 30)
want = {"fw_b128": GiB8, "fw_b64": GiB8, "fw_b32": GiB8, "fw_b32x4": GiB8, "fw_u16x4": GiB8, "fw_gather64": GiB8}   # bytes of LINES touched
asked = {"fw_gather64": GiB8 / 16}
acc = {}
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE":
            k = row["Kernel_Name"].split("(")[0]
            acc.setdefault(k, []).append(float(row["Counter_Value"]))
res = {}
print("%-12s %14s %14s %8s" % ("kernel", "FETCH_SIZE KiB", "bytes of lines", "ratio"))
for k in want:
    v = acc.get(k)
    if not v:
        continue
    m = sum(v) / len(v)
    res[k] = {"fetch_size_kib": m, "bytes_of_lines_touched": want[k], "bytes_asked_for": asked.get(k, want[k]),
              "reported_over_lines": m * 1024 / want[k]}
    print("%-12s %14.0f %14.0f %8.3f" % (k, m, want[k], m * 1024 / want[k]))
json.dump(res, open(os.path.join(out, "fetch_width.json"), "w"), indent=1)
PY
cat $out/timing.txt | tail -6
