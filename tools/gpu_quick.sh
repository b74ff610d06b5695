#!/bin/bash
# quick check of a kernel change: the parity-heavy GPU tests + wgs / chr20 bench lines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-q}
{
echo "== pytest gpu (parity subset)"; timeout 900 python -m pytest tests/test_gpu_normalize.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py tests/test_gpu_multidevice.py tests/test_gpu_export.py -m gpu -x -q 2>&1 | tail -4
echo "== bench wgs"; timeout 600 python bench.py --verify --no-cpu-baseline --no-host-stream 2>gpurun_out/${T}_wgs.err | tail -1 | tee gpurun_out/${T}_bench_wgs.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('verified_bit_exact'))"
echo "== bench chr20"; timeout 600 python bench.py --workload chr20 --steps 50 --verify --no-cpu-baseline --no-host-stream 2>gpurun_out/${T}_chr20.err | tail -1 | tee gpurun_out/${T}_bench_chr20.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('verified_bit_exact'))"
echo "== bench ont"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline --no-host-stream 2>gpurun_out/${T}_ont.err | tail -1 | tee gpurun_out/${T}_bench_ont.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('ingest_normalise_ms_rank0'), d['config'].get('ingest_checkpoint_ms_rank0'), d['config'].get('canonical_cigar_ops_rank0'), d.get('verified_bit_exact'))"
tail -3 gpurun_out/${T}_ont.err
} > gpurun_out/quick_$T.log 2>&1
cat gpurun_out/quick_$T.log; tail -3 gpurun_out/${T}_wgs.err
