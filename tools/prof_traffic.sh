#!/bin/bash
# the two HBM-byte passes (FETCH_SIZE, WRITE_SIZE, each alone) of one bench command: tools/prof_traffic.sh <tag> [bench args...]
tag=$1; shift
cd /tmp 2>/dev/null && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cmd="python $R/bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --steps 3 --warmup 1 $@"
i=3
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1
  i=$((i+1))
done
for f in $(find $out -name "*counter_collection.csv"); do { head -1 $f; grep "gd::" $f; } > $f.tmp && mv $f.tmp $f; done
find $out -name "*.csv" -size +4M -delete
