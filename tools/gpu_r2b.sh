#!/bin/bash
# round 2, call b: export block test + 2-rank strong-scaling dry run (gloo, both ranks on device 0) + N=1 line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_export.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > gpurun_out/r2b_tests.log
GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/r2b_dry2.json 2> gpurun_out/r2b_dry2.err
python bench.py --no-cpu-baseline --no-host-stream --steps 10 > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
cat gpurun_out/r2b_tests.log; tail -c 1500 gpurun_out/r2b_dry2.json; tail -5 gpurun_out/r2b_dry2.err
