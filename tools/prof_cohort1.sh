#!/bin/bash
# one PMC pass (instruction mix) over the cohort bench of gd_sums_stream_kernel
cd /tmp 2>/dev/null && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_cohort1
rm -rf $out; mkdir -p $out
cmd="python $R/bench.py --workload cohort --samples 100 --no-cpu-baseline --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $out/pmc1 -- $cmd > $out/pmc1.log 2>&1
python $R/tools/pmc_summary.py $out 2>&1 | grep -A12 "gd_sums_stream"
for f in $(find $out -name "*counter_collection.csv"); do { head -1 $f; grep "gd::" $f; } > $f.tmp && mv $f.tmp $f; done
find $out -name "*.csv" -size +4M -delete
