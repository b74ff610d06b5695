#!/bin/bash
# Round 5, GPU session 14: rocprofv3 --kernel-trace --stats of the ingest kernels as they are at HEAD (the inflate kernel with
# its loads issued first), on libdeflate level-1 short records and on level-6 records with aux tags.   tools/r12_session14.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12t}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
LEN=63025520,63025520
for v in ld1 ld6aux; do
  echo "== inflate_bench $v under rocprofv3 --kernel-trace --stats" >> $LOG
  if [ $v = ld1 ]; then E=""; else E="SYNTH_BAM_LEVEL=6 SYNTH_BAM_AUX=1"; fi
  ( cd /tmp && env $E INFLATE_BENCH_NO_ZLIB=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_$v -o x -- python $R/tools/inflate_bench.py $LEN > $O/${T}_$v.txt 2>&1 )
  grep -h "lds pad\|kernel" $O/${T}_$v.txt >> $LOG
  f=$(find $O/${T}_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 $f; grep "gd::" $f; } > $O/${T}_inflate_${v}_kernel_stats.csv && cat $O/${T}_inflate_${v}_kernel_stats.csv | cut -c1-160 >> $LOG
  find $O/${T}_$v -name "*kernel_trace.csv" -delete
done
cat $LOG
