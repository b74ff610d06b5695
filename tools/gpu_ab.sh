#!/bin/bash
# same-box A/B of two builds of libgoleft_depth.so: goleft_amd/libgoleft_depth.so (new) vs
# goleft_amd/libgoleft_depth_prev.so (built from the previous commit); wgs bench line of each, twice.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
A=goleft_amd/libgoleft_depth.so; B=goleft_amd/libgoleft_depth_prev.so
cp $A /tmp/new.so; cp $B /tmp/prev.so
{
for rep in 1 2; do
for v in new prev; do
  cp /tmp/$v.so $A
  echo "== $v ($rep)"; timeout 600 python bench.py --no-cpu-baseline --no-host-stream ${ABARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['ms_per_step'], d['value'])"
done
done
cp /tmp/new.so $A
} > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
