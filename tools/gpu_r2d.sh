#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_normalize.py -x -q 2>&1 | tail -25 > gpurun_out/r2d_norm.log
cat gpurun_out/r2d_norm.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2d_all.log
cat gpurun_out/r2d_all.log
python bench.py --no-cpu-baseline --no-host-stream --steps 10 > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r2d_bench.json'))
print(d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'])
P
