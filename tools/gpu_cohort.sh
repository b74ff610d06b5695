#!/bin/bash
# sums-only output: parity tests + the cohort bench line (BASELINE.json config 4)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-c}
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["kernels_ms"], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["kernel"])'
{
echo "== pytest gpu (sums-only, depthwed)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_depthwed.py -m gpu -x -q -k "sums or depthwed" 2>&1 | tail -3
echo "== bench cohort"; timeout 900 python bench.py --workload cohort --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/${T}_cohort.err | tail -1 | tee gpurun_out/${T}_bench_cohort.json | python -c "$P"
tail -3 gpurun_out/${T}_cohort.err | grep -v amdgpu.ids
} > gpurun_out/cohort_$T.log 2>&1
cat gpurun_out/cohort_$T.log
