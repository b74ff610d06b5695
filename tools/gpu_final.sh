#!/bin/bash
# last check of HEAD: the whole GPU suite, smoke, the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
{
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench (defaults)"; timeout 900 python bench.py 2>gpurun_out/final.err | tail -1 | tee gpurun_out/final_bench_wgs.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernels_ms"], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])'
} > gpurun_out/final.log 2>&1
cat gpurun_out/final.log
