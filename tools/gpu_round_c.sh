#!/bin/bash
mkdir -p gpurun_out
{
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
} > gpurun_out/round_c.log 2>&1
cat gpurun_out/round_c.log
