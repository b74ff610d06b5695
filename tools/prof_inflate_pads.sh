#!/bin/bash
# HBM-side traffic of the inflate kernel at different occupancies (GD_OPT_INFLATE_LDS_PAD): FETCH_SIZE and WRITE_SIZE per dispatch,
# in dispatch order (tools/inflate_bench.py: one warm-up, then three dispatches per pad value).
#   tools/prof_inflate_pads.sh <lengths> <pads>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_inflate_pads
rm -rf $out; mkdir -p $out
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/$pmc -- python $R/tools/inflate_bench.py "$1" "$2" > $out/$pmc.log 2>&1
done
python3 - "$out" "$2" <<'PY'
import csv, glob, os, sys
out, pads = sys.argv[1], [int(x) for x in sys.argv[2].split(",")]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gd_inflate_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    vals = [v for _, v in rows][1:]                      # (the warm-up dispatch)
    print(c, "KiB per dispatch, by pad:", {p: [round(x) for x in vals[3 * i:3 * i + 3]] for i, p in enumerate(pads)})
PY
grep "lds pad\|kernel" $out/WRITE_SIZE.log | head -12
