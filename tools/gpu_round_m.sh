#!/bin/bash
# SQ counters of the ONT launch (chunk path)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_m
mkdir -p $out
cmd="python $R/bench.py --workload ont --no-cpu-baseline --steps 3 --warmup 1"
i=1
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1
  i=$((i+1))
done
python $R/tools/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
find $out -name "*.csv" -size +4M -delete
