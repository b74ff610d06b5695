#!/bin/bash
# bench lines of HEAD for the docs: wgs (default, with cpu baseline), chr20 (config 2), ont, cohort-200
mkdir -p gpurun_out
{
echo "== bench wgs"; timeout 600 python bench.py --verify 2>gpurun_out/n_wgs.err | tail -1 | tee gpurun_out/n_bench_wgs.json
echo "== bench chr20"; timeout 600 python bench.py --workload chr20 --verify 2>gpurun_out/n_chr20.err | tail -1 | tee gpurun_out/n_bench_chr20.json
echo "== bench ont"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/n_ont.err | tail -1 | tee gpurun_out/n_bench_ont.json
echo "== bench cohort 200"; timeout 900 python bench.py --workload cohort --samples 200 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/n_cohort.err | tail -1 | tee gpurun_out/n_bench_cohort.json
echo "== rocprof kernel trace ont"
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_n/ont/trace -- python $R/bench.py --workload ont --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/prof_n_ont_trace.log 2>&1
cd $R; python tools/pmc_summary.py gpurun_out/prof_n/ont | head -8
find gpurun_out/prof_n -name "*.csv" -size +2M -delete
} > gpurun_out/round_n.log 2>&1
cat gpurun_out/round_n.log
