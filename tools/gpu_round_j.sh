#!/bin/bash
# v7 tile kernel bring-up: A/B against v6 on the resident WGS stream (bit-exact check), then the parity suite
mkdir -p gpurun_out
{
echo "== variants"; timeout 600 python tools/variants.py --verify --steps 8 "-" "KERNEL=v6" "-" "KERNEL=v6" 2>&1 | tail -6
echo "== variants W=250"; timeout 600 python tools/variants.py --verify --steps 4 --contigs 4 --window 250 "-" "KERNEL=v6" 2>&1 | tail -3
echo "== variants W=100"; timeout 600 python tools/variants.py --verify --steps 4 --contigs 4 --window 100 "-" "KERNEL=v6" 2>&1 | tail -3
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
} > gpurun_out/round_j.log 2>&1
cat gpurun_out/round_j.log
