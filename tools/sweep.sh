#!/bin/bash
# Tile-shape sweep on the GPU box: GOLEFT_GD_TILE x GOLEFT_GD_THREADS, WGS workload.
mkdir -p gpurun_out
CFGS=${CFGS:-"4096:256 4096:512 8192:256 8192:512 16384:512"}
for cfg in $CFGS; do
  T=${cfg%%:*}; NT=${cfg##*:}
  GOLEFT_GD_TILE=$T GOLEFT_GD_THREADS=$NT python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${SWEEP_ARGS} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$T NT=$NT', 'value=%.3e'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], d['kernels_ms'], 'frac=%.3f'%d['roofline']['frac'])"
done
