#!/bin/bash
# packed descriptors: tests first, then the v8 / v7 A/B on the same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_packed.py -x -q -m gpu > gpurun_out/t_packed.log 2>&1; echo "packed rc=$?"; tail -15 gpurun_out/t_packed.log
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/t_wgs_v8.json 2> gpurun_out/t_wgs_v8.err; tail -c 1500 gpurun_out/t_wgs_v8.json; tail -3 gpurun_out/t_wgs_v8.err
GOLEFT_GD_KERNEL=v7 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/t_wgs_v7.json 2> gpurun_out/t_wgs_v7.err; tail -c 1500 gpurun_out/t_wgs_v7.json
