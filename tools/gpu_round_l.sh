#!/bin/bash
# ltile2 bring-up: parity suite, then ONT bench new vs old long-read kernel
mkdir -p gpurun_out
{
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
echo "== bench ont (ltile2)"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/l_1.err | tail -1 | tee gpurun_out/l_bench_ont.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['roofline']['frac'], d.get('verified_bit_exact'))"
echo "== bench ont (old ltile)"; GOLEFT_GD_KERNEL=v6 timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/l_2.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['roofline']['frac'], d.get('verified_bit_exact'))"
tail -n 3 gpurun_out/l_1.err gpurun_out/l_2.err
} > gpurun_out/round_l.log 2>&1
cat gpurun_out/round_l.log
