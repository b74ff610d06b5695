cd $GRAFT_REPO_ROOT
for w in 1 6 1 6; do
GOLEFT_DEPTH_SO=$GRAFT_REPO_ROOT/goleft_amd/libgoleft_depth_w$w.so python bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 5 --warmup 2 --workload cohort 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves $w: step ms', round(d['ms_per_step'],2), 'kernel ms', round(d['roofline']['avg_kernel_ms'],2), 'frac', round(d['roofline']['frac'],3))"
done
