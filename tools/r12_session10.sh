#!/bin/bash
# Round 5, GPU sessions 10-11: the two-wave inflate kernel (a decoder wave and a writer wave per 64 members,
# gd_inflate_pair.hpp of commits 886121a / fdd3ff3, removed since; GD_OPT_INFLATE_PROBE bit 2) against the one-wave kernel.  Probes: 4 two waves; 5 / 6: without the
# writer's match-source loads / block stores; 12: the writer throws the tokens away (the decoder's own pace).
#   tools/r12_session10.sh <tag> [probes] [also level 6: 0|1]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12k}; PROBES=${2:-0,4,0,4,5,6}; L6=${3:-1}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
LEN=63025520,63025520
echo "== inflate_bench, deflate level 1, probes $PROBES" >> $LOG
( cd /tmp && INFLATE_BENCH_PROBES=$PROBES timeout 900 python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate.txt 2>&1 )
grep -h "lds pad\|kernel" $O/${T}_inflate.txt >> $LOG; tail -3 $O/${T}_inflate.txt | grep -i "error\|Traceback" >> $LOG
if [ "$L6" = 1 ]; then
  echo "== the same on aux-tag records at deflate level 6" >> $LOG
  ( cd /tmp && SYNTH_BAM_LEVEL=6 SYNTH_BAM_AUX=1 INFLATE_BENCH_PROBES=$PROBES timeout 900 python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate6.txt 2>&1 )
  grep -h "lds pad\|kernel" $O/${T}_inflate6.txt >> $LOG
fi
cat $LOG
