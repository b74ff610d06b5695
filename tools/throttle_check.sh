#!/bin/bash
# Does the container's CPU quota (cpu.max) stop the file -> BED read?  The genome BAM of bench.py's bam_file_scope, the CLI three
# times, cgroup throttle counters around every run.   gpurun --timeout 1500 -- 'bash tools/throttle_check.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
# VARIANTS_STR="A=1 B=2": the runs repeated with each assignment exported in turn
read -r -a VARIANTS <<< "${VARIANTS_STR:-}"
export TMPDIR=/tmp
D=$(mktemp -d /tmp/thr.XXXX)
LENS=$(python3 -c "import sys; sys.path.insert(0,'$R'); from goleft_amd import synth; print(','.join(str(x) for x in synth.HG19_LENGTHS))")
cat /sys/fs/cgroup/cpu.max
s=$SECONDS
$R/goleft_amd/synth-bam $D/synth.bam chrS $LENS 30 20 > $D/info.json || exit 1
echo "synth-bam $((SECONDS-s)) s"; ls -l $D | head
cat $D/synth.bam > /dev/null
stat() { awk '/nr_throttled|throttled_usec|usage_usec/ {printf "%s ", $2}' /sys/fs/cgroup/cpu.stat; }
# QUEUES="4 8 16": the runs repeated with GPU_MAX_HW_QUEUES set to each (the HIP runtime's hardware queues per process)
for hwq in ${QUEUES:-default}; do
[ "$hwq" != default ] && export GPU_MAX_HW_QUEUES=$hwq && echo "== GPU_MAX_HW_QUEUES=$hwq"
for var in "${VARIANTS[@]:-}"; do
[ -n "$var" ] && echo "== $var" && export $var
for i in ${RUNS:-1 2 3}; do
  sleep 2
  a=($(stat)); t0=$(date +%s%N)
  GOLEFT_DEPTH_TIMING=1 GOLEFT_INGEST_TIMING=1 $R/goleft_amd/goleft-depth depth -w 1000 -p 0 -r $D/synth.fa --prefix $D/out $D/synth.bam 2> $D/err.$i > /dev/null &
  pid=$!
  # per-thread CPU time (utime + stime, clock ticks) of the CLI, sampled until it exits: the last sample is kept
  while [ -n "$SAMPLE" ] && kill -0 $pid 2>/dev/null; do
    for t in /proc/$pid/task/*; do awk -v t=${t##*/} '{n=split($0,f," "); print t, f[n-38]+f[n-37]}' $t/stat 2>/dev/null; done > $D/threads.tmp
    [ -s $D/threads.tmp ] && mv $D/threads.tmp $D/threads.$i
    sleep 0.1
  done
  wait $pid
  t1=$(date +%s%N); b=($(stat))
  echo "run $i: wall $(( (t1 - t0) / 1000000 )) ms, cpu $(( (${b[0]} - ${a[0]}) / 1000 )) ms, throttled ${a[1]} -> ${b[1]} periods, $(( (${b[2]} - ${a[2]}) / 1000 )) ms"
  [ -n "$SAMPLE" ] && echo "  threads at the last sample (ticks of 10 ms): $(sort -k2 -n -r $D/threads.$i | awk '{printf "%s ", $2}')"
  grep -h "^{" $D/err.$i | grep -o '"lib_wait_inflate_s[^,]*\|"lib_wait_link_s[^,]*\|"lib_read_s[^,]*\|"lib_count_walk_s[^,]*\|"read_s[^,]*'  | tr "\n" " "; echo
done
done
done
if [ -n "$PROF" ]; then     # PROF=tag: one more run under rocprofv3 --kernel-trace --stats; the ingest kernels' rows are kept
  ( cd /tmp && GOLEFT_SLOW_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -o x -- $R/goleft_amd/goleft-depth depth -w 1000 -p 0 -r $D/synth.fa --prefix $D/out $D/synth.bam > $D/prof.log 2>&1 ); tail -3 $D/prof.log
  f=$(find $D/prof -name "*kernel_stats.csv" | head -1)
  tr=$(find $D/prof -name "*kernel_trace.csv" | head -1)
  [ -n "$tr" ] && python $R/tools/timeline.py $tr --slice-ms 50 2>/dev/null | awk 'NR<3 || ($2+$5+$14)>0' | cut -c1-86,120-150 | head -60
  mkdir -p $R/gpurun_out; [ -n "$f" ] && cp $f $R/gpurun_out/${PROF}_genome_read_kernel_stats.csv && cut -d, -f1-6 $f | cut -c1-150 | head -14
fi
if [ -n "$THREADS" ]; then python $R/tools/thread_cpu.py $R/goleft_amd/goleft-depth depth -w 1000 -p 0 -r $D/synth.fa --prefix $D/out $D/synth.bam; fi
sha256sum $D/out.depth.bed $D/out.callable.bed | cut -c1-16
rm -rf $D
