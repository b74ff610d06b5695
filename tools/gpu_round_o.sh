#!/bin/bash
# prep-kernel rewrite: parity suite + wgs / chr20 / ont timings
mkdir -p gpurun_out
{
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for w in wgs chr20 ont; do
echo "== bench $w"; timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --verify --no-cpu-baseline --no-host-stream 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('verified_bit_exact'))"
done
} > gpurun_out/round_o.log 2>&1
cat gpurun_out/round_o.log
