#!/bin/bash
# ONT: tile shape variants of the chunk path
mkdir -p gpurun_out
{
for v in "" "GOLEFT_GD_TILE=8192" "GOLEFT_GD_TILE=8192 GOLEFT_GD_THREADS=512" "GOLEFT_GD_THREADS=512"; do
echo "== ont [$v]"; env $v timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['roofline']['frac'], d.get('verified_bit_exact'), d['config']['tile_positions'])"
done
for v in "" "GOLEFT_GD_TILE=8192"; do
echo "== chr20 [$v]"; env $v timeout 600 python bench.py --workload chr20 --steps 20 --warmup 3 --verify --no-cpu-baseline --no-host-stream 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['roofline']['frac'], d.get('verified_bit_exact'), d['config']['tile_positions'])"
done
} > gpurun_out/round_q.log 2>&1
cat gpurun_out/round_q.log
