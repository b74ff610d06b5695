#!/bin/bash
# Session-3 verification of HEAD: GPU suite, the three bench workloads, one kernel-trace pass.
mkdir -p gpurun_out
{
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== bench wgs"; timeout 600 python bench.py --verify 2>gpurun_out/g_wgs.err | tail -1 | tee gpurun_out/g_bench_wgs.json
echo "== bench cohort 200"; timeout 900 python bench.py --workload cohort --samples 200 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/g_cohort.err | tail -1 | tee gpurun_out/g_bench_cohort.json
echo "== bench ont wgs"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/g_ont.err | tail -1 | tee gpurun_out/g_bench_ont.json
echo "== rocprof kernel trace"
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_g/trace -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/prof_g/trace.log 2>&1
cd $R; python tools/pmc_summary.py gpurun_out/prof_g 2>&1 | head -12
find gpurun_out/prof_g -name "*.csv" -size +2M -delete
tail -3 gpurun_out/g_*.err
} > gpurun_out/round_g.log 2>&1
cat gpurun_out/round_g.log
