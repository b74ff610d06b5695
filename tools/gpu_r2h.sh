#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for w in chr20 wgs; do
timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-stream > gpurun_out/r2h_bench_$w.json 2> gpurun_out/r2h_$w.err
done
python - <<'P'
import json
for w in ("chr20","wgs"):
    d=json.load(open('gpurun_out/r2h_bench_%s.json'%w))
    print(w, 'ms/step %.4f'%d['ms_per_step'], d['kernels_ms'], 'frac %.3f'%d['roofline']['frac'], 'value %.3e'%d['value'])
P
