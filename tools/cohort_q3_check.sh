cd $GRAFT_REPO_ROOT
export GOLEFT_DEPTH_SO=$GRAFT_REPO_ROOT/goleft_amd/libgoleft_depth_q3.so
python -m pytest tests/test_gpu_parity.py tests/test_depthwed.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config4" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_l2_cohort_q3; rm -rf $out; mkdir -p $out
cmd="python $R/bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 5 --warmup 2 --workload cohort"
$cmd 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('q3 (no profiler): step ms', round(d['ms_per_step'],2), 'kernel ms', round(d['roofline']['avg_kernel_ms'],2), 'frac', round(d['roofline']['frac'],3))"
GOLEFT_DEPTH_SO= python $R/bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 5 --warmup 2 --workload cohort 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shipped (no profiler): step ms', round(d['ms_per_step'],2), 'kernel ms', round(d['roofline']['avg_kernel_ms'],2), 'frac', round(d['roofline']['frac'],3))"
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/pmc1 -- $cmd > $out/pmc1.log 2>&1
for f in $(find $out -name "*counter_collection.csv"); do { head -1 $f; grep "gd::" $f; } > $f.tmp && mv $f.tmp $f; done
python $R/tools/pmc_summary.py $out | grep -A3 "gd_sums_stream" | tr '\n' ' '; echo
find $out -name "*.csv" -size +2M -delete
