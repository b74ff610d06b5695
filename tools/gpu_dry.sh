#!/bin/bash
# multi-process bench flow on the 1-GPU box: N ranks over gloo sharing device 0 (control flow only -- the
# gather there is a host-staged gloo copy and says nothing about xGMI)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for n in ${1:-2}; do
GOLEFT_BENCH_SKIP_COHORT=${SKIP:-1} GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SINGLE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
  --master-addr 127.0.0.1 --master-port 2953$n bench.py --gpus $n --steps 6 --warmup 2 > gpurun_out/dry$n.json 2> gpurun_out/dry$n.err
echo "== N=$n"; tail -1 gpurun_out/dry$n.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['scaling'], d['ms_per_step'], d['value'], d['kernels_ms'], d['split'])"; tail -3 gpurun_out/dry$n.err | cut -c1-300
done
