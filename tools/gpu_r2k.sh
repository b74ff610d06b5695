#!/bin/bash
# HEAD verification: GPU suite, smoke, bench lines of every BASELINE.json config (HIP events), rocprofv3
# kernel-trace of the wgs / ont commands, the reference's own timed invocation on a synthetic chr1 BAM
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-k}
mkdir -p gpurun_out/prof_$T
R=$PWD
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["kernels_ms"], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("verified_bit_exact"))'
{
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== bench wgs"; timeout 900 python bench.py --verify 2>gpurun_out/${T}_wgs.err | tail -1 | tee gpurun_out/${T}_bench_wgs.json | python -c "$P"
echo "== bench chr20"; timeout 600 python bench.py --workload chr20 --steps 50 --verify --no-cpu-baseline --no-host-stream 2>gpurun_out/${T}_chr20.err | tail -1 | tee gpurun_out/${T}_bench_chr20.json | python -c "$P"
echo "== bench ont"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/${T}_ont.err | tail -1 | tee gpurun_out/${T}_bench_ont.json | python -c "$P"
echo "== bench cohort"; timeout 900 python bench.py --workload cohort --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/${T}_cohort.err | tail -1 | tee gpurun_out/${T}_bench_cohort.json | python -c "$P"
tail -3 gpurun_out/${T}_cohort.err | grep -v amdgpu.ids
echo "== scope iii: goleft depth --chrom chr1 -p 20 -o -w 16384 on a synthetic 30x chr1 BAM (indexcov/paper/cmp.sh:6)"
timeout 900 python tools/scope3.py --paper --name chr1 --length 249250621 2>gpurun_out/${T}_scope3.err | tail -1 | tee gpurun_out/${T}_scope3_chr1.json | cut -c1-1500
tail -3 gpurun_out/${T}_scope3.err
cd /tmp && export TMPDIR=/tmp
for w in wgs ont; do
  extra=""; [ $w = ont ] && extra="--workload ont"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T/$w/trace -- python $R/bench.py $extra --no-cpu-baseline --no-host-stream --steps 5 --warmup 2 > $R/gpurun_out/prof_$T/${w}_trace.log 2>&1
done
cd $R
for w in wgs ont; do echo "-- rocprofv3 $w"; python tools/pmc_summary.py gpurun_out/prof_$T/$w 2>&1 | grep -v "^ *$" | grep -v "PMC" | head -12; done
find gpurun_out/prof_$T -name "*.csv" -size +2M -delete
} > gpurun_out/round_$T.log 2>&1
cat gpurun_out/round_$T.log
