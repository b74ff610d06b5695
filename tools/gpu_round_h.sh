#!/bin/bash
# chunk path bring-up: parity suite, ONT bench on chunk vs scatter
mkdir -p gpurun_out
{
echo "== pytest gpu parity"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
echo "== bench ont chr20 chunk"; timeout 600 python bench.py --workload ont-chr20 --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/h_1.err | tail -1
echo "== bench ont wgs chunk"; timeout 900 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>gpurun_out/h_2.err | tail -1 | tee gpurun_out/h_bench_ont.json
tail -n 5 gpurun_out/h_1.err gpurun_out/h_2.err
} > gpurun_out/round_h.log 2>&1
cat gpurun_out/round_h.log
