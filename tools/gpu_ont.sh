#!/bin/bash
# long-read path experiments: chunk-path parity tests + ont bench variants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-o}
{
echo "== pytest gpu (chunk-path subset)"; timeout 900 python -m pytest tests/test_gpu_normalize.py tests/test_gpu_parity.py tests/test_seqstats.py tests/test_gpu_ref_fixtures.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -3
for v in "" "--opt 2=512"; do
echo "== bench ont $v"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline --no-host-stream $v 2>gpurun_out/${T}_ont.err | tail -1 | tee gpurun_out/${T}_bench_ont.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('ingest_normalise_ms_rank0'), d['config'].get('ingest_checkpoint_ms_rank0'), d['config'].get('canonical_cigar_ops_rank0'), d.get('verified_bit_exact'))"
tail -2 gpurun_out/${T}_ont.err
done
} > gpurun_out/ont_$T.log 2>&1
cat gpurun_out/ont_$T.log
