#!/bin/bash
mkdir -p gpurun_out
{
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== bench ont wgs"; timeout 900 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms'], d.get('verified_bit_exact'))"
echo "== bench ont wgs scope wg"; GOLEFT_GD_SCOPE=wg timeout 900 python bench.py --workload ont --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms'], d.get('verified_bit_exact'))"
} > gpurun_out/round_e.log 2>&1
cat gpurun_out/round_e.log
