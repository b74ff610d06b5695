#!/bin/bash
# rocprofv3 passes for one bench command (run on the GPU box through gpurun).
#   tools/prof.sh <tag> [bench args...]
# Pass 0: --kernel-trace --stats (per-kernel time).  Passes 1..4: PMC counters, each alone
# (never combined with trace domains other than --kernel-trace).
tag=$1; shift
cd /tmp 2>/dev/null && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cmd="python $R/bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 5 --warmup 2 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $cmd > $out/trace.log 2>&1
# the same statistics over the WORKLOAD's launches (gd_create's warm-up dispatches left out): what a roofline is recomputed from
python $R/tools/kernel_stats_trimmed.py $out/trace $out/kernel_stats_trimmed.csv > /dev/null 2>&1
i=1
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1
  i=$((i+1))
done
python $R/tools/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
# keep the engine's rows only (the synthetic-data generator's torch kernels dominate the raw CSVs)
for f in $(find $out -name "*counter_collection.csv"); do { head -1 $f; grep "gd::" $f; } > $f.tmp && mv $f.tmp $f; done
find $out -name "*.csv" -size +4M -delete
