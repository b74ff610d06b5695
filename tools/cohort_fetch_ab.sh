#!/bin/bash
# Round 6: what the cohort kernel's over-fetch is (run through gpurun).  gd_sums_stream_kernel reads 154 GB of records and fetches 181.
#  (a) the same kernel on data WITHOUT reads of three ops (GOLEFT_SYNTH_PLAIN=2) and with `150M` only (=1): does the over-fetch go?
#  (b) builds that walk the queue of odd reads EARLIER (-DGD_SUMS_DRAIN_AT=n; the shipped kernel walks it once, at the wave's end)
# one rocprofv3 --pmc pass each: FETCH_SIZE, L2 misses, L1 -> L2 read requests; the step time is the bench's own
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {   # name, env...
  local name=$1; shift
  local out=$R/gpurun_out/prof_l2_cohort_$name; rm -rf $out; mkdir -p $out
  local cmd="python $R/bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 3 --warmup 1 --workload cohort"
  env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/pmc1 -- $cmd > $out/pmc1.log 2>&1
  grep '^{' $out/pmc1.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name: step ms', round(d['ms_per_step'],2), 'kernel ms', round(d['roofline']['avg_kernel_ms'],2), 'bytes read', d['roofline']['bytes_really_read_per_launch'])"
  for f in $(find $out -name "*counter_collection.csv"); do { head -1 $f; grep "gd::" $f; } > $f.tmp && mv $f.tmp $f; done
  python $R/tools/pmc_summary.py $out | grep -A3 "gd_sums_stream" | tr '\n' ' '; echo
  find $out -name "*.csv" -size +2M -delete
}
if [ "$1" != drains ]; then
run shipped
run plain1 GOLEFT_SYNTH_PLAIN=1
run plain2 GOLEFT_SYNTH_PLAIN=2
fi
for d in 64 32 16; do
  [ -f $R/goleft_amd/libgoleft_depth_drain$d.so ] && run drain$d GOLEFT_DEPTH_SO=$R/goleft_amd/libgoleft_depth_drain$d.so
done
