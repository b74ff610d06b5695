#!/bin/bash
# same-box comparison of experiment builds goleft_amd/libgoleft_depth_<name>.so: tools/gpu_exp.sh "<bench args>" name...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ARGS=$1; shift
A=goleft_amd/libgoleft_depth.so
cp $A /tmp/base.so
{
for v in base "$@" base; do
  [ $v = base ] && cp /tmp/base.so $A || cp goleft_amd/libgoleft_depth_$v.so $A
  echo "== $v"; timeout 600 python bench.py --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'], d.get('verified_bit_exact'))"
done
cp /tmp/base.so $A
} > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
