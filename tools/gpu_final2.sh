#!/bin/bash
# last check of HEAD: whole GPU suite, smoke, default bench line, cohort line + its kernel trace, a long soak
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
{
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench (defaults)"; timeout 900 python bench.py 2>gpurun_out/final.err | tail -1 | tee gpurun_out/final_bench_wgs.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernels_ms"], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])'
echo "== bench cohort"; timeout 900 python bench.py --workload cohort --no-cpu-baseline 2>>gpurun_out/final.err | tail -1 | tee gpurun_out/final_bench_cohort.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernels_ms"], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])'
echo "== long soak"; GOLEFT_SOAK_SEEDS=32 GOLEFT_SOAK_JOBS=60 timeout 600 python -m pytest tests/test_gpu_soak.py -q -n 6 2>&1 | grep -v amdgpu | tail -2
GOLEFT_SOAK_CLI=300 timeout 300 python -m pytest tests/test_gpu_cli_soak.py -q -n 6 2>&1 | grep -v amdgpu | tail -2
} > gpurun_out/final.log 2>&1
cat gpurun_out/final.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_cohort_trace -- python $OLDPWD/bench.py --workload cohort --samples 100 --no-cpu-baseline --steps 3 --warmup 1 > $OLDPWD/gpurun_out/prof_cohort_trace.log 2>&1 )
f=$(find gpurun_out/prof_cohort_trace -name "*kernel_stats.csv" | head -1); { head -1 $f; grep "gd::" $f; } > gpurun_out/cohort_kernel_stats.csv; cat gpurun_out/cohort_kernel_stats.csv
find gpurun_out/prof_cohort_trace -name "*.csv" -size +2M -delete
