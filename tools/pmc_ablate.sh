#!/bin/bash
# SQ counters of gd_tile_kernel under phase ablation (timing/instruction mix only).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for a in ${ABL:-0 12 13 10 8}; do
  out=$R/gpurun_out/pmcab_$a
  mkdir -p $out
  i=1
  for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    GOLEFT_GD_ABLATE=$a rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc$i -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $out/pmc$i.log 2>&1
    i=$((i+1))
  done
  echo "=== ablate=$a"
  python $R/tools/pmc_summary.py $out | grep -A 17 "^  gd_tile_kernel"
  rm -rf $out
done
