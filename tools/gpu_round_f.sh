#!/bin/bash
mkdir -p gpurun_out
{
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== bench cohort 24"; timeout 900 python bench.py --workload cohort --samples 24 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1
echo "== bench cohort 200"; timeout 1500 python bench.py --workload cohort --samples 200 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
} > gpurun_out/round_f.log 2>&1
cat gpurun_out/round_f.log
