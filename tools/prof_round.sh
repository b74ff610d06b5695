#!/bin/bash
# The round's profile set for the headline workload and config 2: tools/prof.sh (kernel trace + the four PMC passes), the
# traffic JSON bench.py reads, the trimmed kernel statistics and the summary, copied under profiles/ with the round's tag.
#   gpurun --timeout 2400 -- 'bash tools/prof_round.sh r13v'
T=${1:-r13v}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in wgs chr20; do
  args="--workload $w"
  bash tools/prof.sh ${T}_$w $args > gpurun_out/${T}_${w}_prof.log 2>&1
  python tools/traffic_from_pmc.py gpurun_out/prof_${T}_$w gpurun_out/${T}_${w}_traffic.json gd_tile_fast_kernel \
      "python bench.py $args --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 5 --warmup 2" "gd_tile_fast_kernel<raw>" >> gpurun_out/${T}_${w}_prof.log 2>&1
  cp gpurun_out/prof_${T}_$w/kernel_stats_trimmed.csv gpurun_out/${T}_${w}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/prof_${T}_$w/summary.txt gpurun_out/${T}_${w}_rocprofv3_summary.txt 2>/dev/null
  grep '^{' gpurun_out/prof_${T}_$w/trace.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])"
  head -5 gpurun_out/${T}_${w}_kernel_stats.csv | cut -c1-160
  python -c "import json; d=json.load(open('gpurun_out/${T}_${w}_traffic.json')); print(d['hbm_bytes_per_launch'], d['dispatches'], d['dispatches_left_out'])"
done
