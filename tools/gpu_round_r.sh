#!/bin/bash
# Session-4 final evidence: on ONE box, the bench line (HIP events) and rocprofv3 --kernel-trace --stats of the
# same command, for wgs and ont; FETCH/WRITE PMC passes for both; OPT=1 A/B; seqstats + md_flags microbench.
mkdir -p gpurun_out/prof_r
R=$PWD
{
echo "== bench wgs (HIP events)"; timeout 600 python bench.py --verify 2>gpurun_out/r_wgs.err | tail -1 | tee gpurun_out/r_bench_wgs.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['host_stream_scope']['value'])"
echo "== bench ont (HIP events)"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/r_ont.err | tail -1 | tee gpurun_out/r_bench_ont.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'])"
echo "== variants OPT=1"; timeout 600 python tools/variants.py --steps 8 "-" "OPT=1" "-" "OPT=1" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for w in wgs ont; do
  extra=""; [ $w = ont ] && extra="--workload ont"
  cmd="python $R/bench.py $extra --no-cpu-baseline --no-host-stream --steps 5 --warmup 2"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r/$w/trace -- $cmd > $R/gpurun_out/prof_r/${w}_trace.log 2>&1
  i=1
  for pmc in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $R/gpurun_out/prof_r/$w/pmc$i -- $cmd > $R/gpurun_out/prof_r/${w}_pmc$i.log 2>&1
    i=$((i+1))
  done
done
cd $R
for w in wgs ont; do echo "-- rocprofv3 $w"; python tools/pmc_summary.py gpurun_out/prof_r/$w 2>&1 | grep -v "^ *$" | head -30; done
echo "== microbench seq_stats / md_flags"; timeout 300 python tools/microbench_side.py 2>&1 | tail -6
find gpurun_out/prof_r -name "*.csv" -size +2M -delete
} > gpurun_out/round_r.log 2>&1
cat gpurun_out/round_r.log
