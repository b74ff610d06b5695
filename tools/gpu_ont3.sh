#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-o}
{
echo "== pytest gpu (chunk-path subset)"; timeout 900 python -m pytest tests/test_gpu_normalize.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
for v in "" "--opt 1=8192 --opt 2=512"; do
echo "== bench ont $v"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline --no-host-stream $v 2>gpurun_out/${T}_ont.err | tail -1 | tee gpurun_out/${T}_bench_ont.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('ingest_normalise_ms_rank0'), d['config'].get('ingest_checkpoint_ms_rank0'), d.get('verified_bit_exact'))"
tail -2 gpurun_out/${T}_ont.err | grep -v amdgpu.ids
done
echo "== wgs, windows-only output (the tile kernel without its store stream)"
GOLEFT_BENCH_OUTPUTS=windows timeout 600 python bench.py --no-cpu-baseline --no-host-stream 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'])"
echo "== wgs"
timeout 600 python bench.py --no-cpu-baseline --no-host-stream 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'])"
echo "== hbm"; timeout 300 python tools/hbm_bench.py 2>&1 | tail -12
} > gpurun_out/ont3_$T.log 2>&1
cat gpurun_out/ont3_$T.log
