#!/bin/bash
# Session-4 verification of HEAD (chunk path included): GPU suite, wgs + ont bench lines,
# kernel-trace stats for both, FETCH/WRITE traffic passes for the ont (chunk) launch.
mkdir -p gpurun_out/prof_i
R=$PWD
{
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== bench wgs"; timeout 600 python bench.py --verify 2>gpurun_out/i_wgs.err | tail -1 | tee gpurun_out/i_bench_wgs.json
echo "== bench ont wgs"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/i_ont.err | tail -1 | tee gpurun_out/i_bench_ont.json
echo "== rocprof kernel trace wgs"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_i/wgs/trace -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/prof_i/wgs_trace.log 2>&1
echo "== rocprof kernel trace ont"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_i/ont/trace -- python $R/bench.py --workload ont --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/prof_i/ont_trace.log 2>&1
i=1
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/prof_i/ont/pmc$i -- python $R/bench.py --workload ont --no-cpu-baseline --steps 5 --warmup 2 > $R/gpurun_out/prof_i/ont_pmc$i.log 2>&1
  i=$((i+1))
done
cd $R
for w in wgs ont; do echo "-- $w"; python tools/pmc_summary.py gpurun_out/prof_i/$w 2>&1 | grep -v "^ *$" | head -40; done
find gpurun_out/prof_i -name "*.csv" -size +2M -delete
tail -3 gpurun_out/i_*.err
} > gpurun_out/round_i.log 2>&1
cat gpurun_out/round_i.log
