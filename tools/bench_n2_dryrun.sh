#!/bin/bash
# `python bench.py --gpus 2` on a ONE-GPU box (run through gpurun): every rank on device 0, the process group over gloo -- the
# control flow of an N > 1 run, never a measurement.  (a) with the real RCCL: two ranks on one device are refused, the ranks fall
# back to torch.distributed's gather together, the line says why and the exit status is 3; (b) with the RCCL stand-in of the
# tests (GOLEFT_RCCL_LIB): the library's own collective carries the gather -- hand-shake, id broadcast, gd_comm_init, the
# word-for-word check against torch's gather, the pipelined steps.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GOLEFT_BENCH_SINGLE_DEVICE=1 GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SKIP_COHORT=1 GOLEFT_BENCH_SKIP_FILE_NGPU=1
summ='import sys,json
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); s=d.get("split") or {}
print("  n_gpus", d["n_gpus"], "ms", round(d["ms_per_step"],3), "collective:", s.get("collective"), "| verified:", s.get("collective_verified_against_torch_gather"), "| fallback:", s.get("collective_fallback_reason"), "| library:", s.get("collective_library"))'
[ -f tests/stubs/librccl_stub.so ] || hipcc -O2 -shared -fPIC -o tests/stubs/librccl_stub.so tests/stubs/rccl_stub.cpp -lrt
echo "== (a) real RCCL, two ranks on one device"
timeout 600 python bench.py --gpus 2 --workload chr20 --steps 5 --warmup 2 --no-cpu-baseline --no-host-stream > gpurun_out/n2a.json 2> gpurun_out/n2a.err; echo "  exit status $?"
python -c "$summ" < gpurun_out/n2a.json; grep "bench.py:" gpurun_out/n2a.err | tail -2
echo "== (b) the stand-in: the library's collective between two processes"
GOLEFT_RCCL_LIB=$PWD/tests/stubs/librccl_stub.so timeout 600 python bench.py --gpus 2 --workload chr20 --steps 5 --warmup 2 --no-cpu-baseline --no-host-stream > gpurun_out/n2b.json 2> gpurun_out/n2b.err; echo "  exit status $?"
python -c "$summ" < gpurun_out/n2b.json; tail -2 gpurun_out/n2b.err
echo "== (c) three ranks, the stand-in, the genome"
GOLEFT_RCCL_LIB=$PWD/tests/stubs/librccl_stub.so timeout 900 python bench.py --gpus 3 --steps 5 --warmup 2 --no-cpu-baseline --no-host-stream > gpurun_out/n3c.json 2> gpurun_out/n3c.err; echo "  exit status $?"
python -c "$summ" < gpurun_out/n3c.json; tail -2 gpurun_out/n3c.err
