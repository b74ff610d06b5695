#!/bin/bash
# LT2 tile-shape variants on the ONT bench, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
{
for o in "" "--opt 2=512" "--opt 1=8192 --opt 2=512"; do
  echo "== bench ont $o"; timeout 600 python bench.py --workload ont --no-cpu-baseline --steps 5 --warmup 2 $o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d.get('verified_bit_exact'))"
done
} > gpurun_out/ont_opts.log 2>&1
cat gpurun_out/ont_opts.log
