#!/bin/bash
# Phase ablation of gd_tile_kernel (results are wrong on purpose; timing only).
for a in 0 1 2 4 8 12 13 5 9; do
  GOLEFT_GD_ABLATE=$a python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${SWEEP_ARGS} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$a tile_ms=%.3f'%d['kernels_ms']['tile'])"
done
