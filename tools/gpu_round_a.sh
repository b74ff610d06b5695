#!/bin/bash
# One gpurun call: smoke, variant A/B (+ exactness), GPU test-suite, default bench.
mkdir -p gpurun_out
{
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== variants"; timeout 600 python tools/variants.py --verify --steps 6 "-" "KERNEL=v5" "OPT=1" "TILE=8192,THREADS=512" "TILE=8192,THREADS=256" "TILE=4096,THREADS=512" 2>&1 | tail -12
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -2
} > gpurun_out/round_a.log 2>&1
cat gpurun_out/round_a.log
