#!/bin/bash
# L2-side counters of one bench command (run on the GPU box through gpurun): who asks the fabric for what.
#   tools/prof_l2.sh <tag> [bench args...]      e.g.  tools/prof_l2.sh cohort --workload cohort
# Passes (each alone with --kernel-trace, never with other trace domains): memory-side read requests by size; L2 hits /
# misses / requests; L1 -> L2 read requests and atomics.  Summary: tools/pmc_summary.py (means per dispatch).
tag=$1; shift
cd /tmp 2>/dev/null && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_l2_$tag
rm -rf $out; mkdir -p $out
cmd="python $R/bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 3 --warmup 1 $@"
i=1
for pmc in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1
  i=$((i+1))
done
for f in $(find $out -name "*counter_collection.csv"); do { head -1 $f; grep "gd::" $f; } > $f.tmp && mv $f.tmp $f; done
find $out -name "*.csv" -size +4M -delete
python $R/tools/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
