cd $GRAFT_REPO_ROOT
export GOLEFT_BENCH_SINGLE_DEVICE=1 GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SKIP_COHORT=1 GOLEFT_BENCH_SKIP_FILE_NGPU=1
[ -f tests/stubs/librccl_stub.so ] || hipcc -O2 -shared -fPIC -o tests/stubs/librccl_stub.so tests/stubs/rccl_stub.cpp -lrt
GOLEFT_RCCL_LIB=$PWD/tests/stubs/librccl_stub.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-host-stream > gpurun_out/tr2.json 2> gpurun_out/tr2.err; echo "exit $?"
grep '^{' gpurun_out/tr2.json | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d.get('split') or {}
print('n_gpus', d['n_gpus'], 'ms', round(d['ms_per_step'],3), 'value', d['value'], s.get('collective'), s.get('collective_verified_against_torch_gather'), s.get('collective_fallback_reason'), d.get('ranks_seen'), d.get('launcher'))"
tail -3 gpurun_out/tr2.err
