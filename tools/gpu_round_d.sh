#!/bin/bash
mkdir -p gpurun_out
{
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== bench ont-chr20"; timeout 600 python bench.py --workload ont-chr20 --steps 5 --warmup 2 --verify --no-cpu-baseline 2>&1 | tail -1
echo "== bench ont wgs"; timeout 900 python bench.py --workload ont --steps 5 --warmup 2 --verify --cpu-sample-contigs 2 2>&1 | tail -1
echo "== bench wgs forced scatter"; GOLEFT_GD_PATH=scatter timeout 600 python bench.py --steps 5 --warmup 2 --verify --no-cpu-baseline 2>&1 | tail -1
echo "== bench wgs"; timeout 600 python bench.py --verify 2>&1 | tail -1
} > gpurun_out/round_d.log 2>&1
cat gpurun_out/round_d.log
