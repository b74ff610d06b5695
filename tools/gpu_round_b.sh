#!/bin/bash
mkdir -p gpurun_out
{
echo "== variants"; timeout 600 python tools/variants.py --verify --steps 6 "-" "OPT=1" 2>&1 | grep variant
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
} > gpurun_out/round_b.log 2>&1
cat gpurun_out/round_b.log
