#!/bin/bash
# long-read path: tile shape variants (GD_OPT_TILE_POSITIONS = 1, GD_OPT_TILE_THREADS = 2)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-o}
{
for v in "" "--opt 1=8192" "--opt 1=8192 --opt 2=512"; do
echo "== bench ont $v"; timeout 600 python bench.py --workload ont --steps 5 --warmup 2 --verify --no-cpu-baseline --no-host-stream $v 2>gpurun_out/${T}_ont.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('verified_bit_exact'))"
tail -2 gpurun_out/${T}_ont.err | grep -v amdgpu.ids
done
} > gpurun_out/ont2_$T.log 2>&1
cat gpurun_out/ont2_$T.log
