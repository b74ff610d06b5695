#!/bin/bash
# The two ways from host memory into HBM (gd_push / gd_acquire + gd_commit) on chr20 and on the whole 30x genome:
# bench.py's host_stream_scope alone.  gpurun --timeout 1500 -- 'bash tools/host_stream_ab.sh r13s'
tag=${1:-hs}
mkdir -p gpurun_out
python - > gpurun_out/${tag}_host_stream.json 2> gpurun_out/${tag}_host_stream.err <<'PY'
import json, os, sys
sys.path.insert(0, ".")
import bench
opts = tuple(os.environ.get("OPTS", "").split())       # OPTS="8=16": gd_set_option pairs (8 = GD_OPT_H2D_KERNEL: the copy kernel's grid)
out = {"chr20": bench.host_stream_scope(0, 1000, 1, 4, opts=opts),
       "wgs": bench.host_stream_scope(0, 1000, 1, 4, genome=True, reps=2, opts=opts)}
print(json.dumps(out, indent=1))
PY
echo "exit $?"
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_host_stream.json"))
for k, v in d.items():
    for n, x in v["variants"].items():
        print(k, n, "ms %.1f feed_ms %.1f GB/s %.1f" % (x["ms"], x["feed_ms"], x["host_to_device_GBps"]))
PY
