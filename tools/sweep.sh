#!/bin/bash
# Tile-shape sweep on the GPU box: GOLEFT_GD_TILE x GOLEFT_GD_THREADS, WGS workload.
mkdir -p gpurun_out
for cfg in "4096 256" "4096 512" "8192 256" "8192 512" "8192 1024" "16384 512" "16384 1024"; do
  set -- $cfg
  GOLEFT_GD_TILE=$1 GOLEFT_GD_THREADS=$2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${SWEEP_ARGS} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$1 NT=$2', 'value=%.3e'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], d['kernels_ms'], 'frac=%.3f'%d['roofline']['frac'])"
done
