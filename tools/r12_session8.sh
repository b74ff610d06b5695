#!/bin/bash
# Round 5, last GPU session: HEAD -- counter traffic of every workload's dominant kernel (separate --pmc passes), the kernel
# trace of the headline command, the whole GPU suite, the default bench line, two-rank dry runs.   tools/r12_session8.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12i}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
for w in wgs chr20 ont cohort; do
  echo "== traffic $w" >> $LOG
  bash tools/prof_traffic.sh ${T}_$w --workload $w --emulate-shards= >> $LOG 2>&1
done
python tools/traffic_from_pmc.py $O/prof_${T}_wgs $O/${T}_wgs_traffic.json "gd_tile_fast_kernel" "--workload wgs" "gd_tile_fast_kernel<raw>" >> $LOG 2>&1
python tools/traffic_from_pmc.py $O/prof_${T}_chr20 $O/${T}_chr20_traffic.json "gd_tile_fast_kernel" "--workload chr20" "gd_tile_fast_kernel<raw>" >> $LOG 2>&1
python tools/traffic_from_pmc.py $O/prof_${T}_ont $O/${T}_ont_traffic.json "gd_ltile2_kernel" "--workload ont" "gd_ltile2_kernel" >> $LOG 2>&1
python tools/traffic_from_pmc.py $O/prof_${T}_ont $O/${T}_ont_dels_raw_traffic.json "gd_dels_raw_kernel" "--workload ont" "gd_dels_raw_kernel" >> $LOG 2>&1
python tools/traffic_from_pmc.py $O/prof_${T}_cohort $O/${T}_cohort_traffic.json "gd_sums_stream_kernel" "--workload cohort" "gd_sums_stream_kernel<raw>" >> $LOG 2>&1
echo "== kernel trace of the headline command" >> $LOG
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_wgs_trace -o x -- python $R/bench.py --no-cpu-baseline --no-host-stream --bam-scope off --other-workloads= --emulate-shards= --steps 10 --warmup 3 > $O/${T}_wgs_trace.txt 2>&1 )
f=$(find $O/${T}_wgs_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { head -1 $f; grep "gd::" $f; } > $O/${T}_wgs_kernel_stats.csv && head -6 $O/${T}_wgs_kernel_stats.csv >> $LOG
find $O/${T}_wgs_trace -name "*kernel_trace.csv" -delete
echo "== pytest -m gpu (all) + smoke" >> $LOG
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -3 >> $LOG; grep -h "^FAILED\|^ERROR" $O/${T}_pytest.txt | head -20 >> $LOG
python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
echo "== two ranks on the one device over gloo, torch.distributed's gather asked for" >> $LOG
( cd $R && GOLEFT_BENCH_NATIVE_GATHER=0 GOLEFT_BENCH_SINGLE_DEVICE=1 GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SKIP_COHORT=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/${T}_n2.out 2>$O/${T}_n2.err )
grep '^{"metric' $O/${T}_n2.out | tail -1 > $O/${T}_bench_wgs_n2_gloo_dryrun_one_device.json; tail -1 $O/${T}_n2.out | cut -c1-80 >> $LOG
python3 -c "
import json; d=json.load(open('$O/${T}_bench_wgs_n2_gloo_dryrun_one_device.json')); s=d.get('split') or {}
print('  n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'collective', s.get('collective'), '| fallback:', s.get('collective_fallback_reason'), 'sum', s.get('gathered_sum_of_window_sums'))" >> $LOG 2>&1
echo "== the same with the default collective (the library's: two ranks on one device)" >> $LOG
( cd $R && GOLEFT_BENCH_SINGLE_DEVICE=1 GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SKIP_COHORT=1 timeout 300 python bench.py --gpus 2 --workload chr20 --steps 5 --warmup 2 > $O/${T}_n2native.out 2>$O/${T}_n2native.err; echo "  exit $?" >> $LOG )
grep '^{"metric' $O/${T}_n2native.out | tail -1 | python3 -c "
import json,sys; t=sys.stdin.read().strip()
if t:
    d=json.loads(t); s=d.get('split') or {}; print('  collective', s.get('collective'), '| fallback:', s.get('collective_fallback_reason'), '| verified', s.get('collective_verified_against_torch_gather'))
else: print('  no line')" >> $LOG 2>&1
grep -v "amdgpu.ids\|socket.cpp" $O/${T}_n2native.err | tail -4 >> $LOG
echo "== python bench.py (defaults)" >> $LOG
( cd $R && timeout 1500 python bench.py > $O/${T}_bench.out 2>$O/${T}_bench.err ); tail -1 $O/${T}_bench.out > $O/${T}_bench_wgs_n1.json
python3 -c "
import json; d=json.load(open('$O/${T}_bench_wgs_n1.json'))
print('  step %.3f ms value %.3e frac %.3f traffic %s first %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['first_compute']['ratio_to_warm']))
b=d['bam_file_scope']; print('  bam_file_scope', {k: b.get(k) for k in ('file','wall_s','value','outputs_identical','oracle_identical','paper_invocation_s','error')}, (b.get('device_decoder') or {}).get('all_wall_s'))
for n, v in (b.get('variants') or {}).items(): print('  variant', n, {k: v.get(k) for k in ('device_wall_s','host_wall_s','outputs_identical','oracle_identical','error')})
for n, v in (d.get('other_workloads') or {}).items(): print('  other', n, {k: v.get(k) for k in ('ms_per_step','value','seconds_in_bench','error')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('traffic'), (v.get('first_compute') or {}).get('ratio_to_warm'), v.get('kernels_ms'))
print('  emu', {n: round(v['projected_speedup'],2) for n, v in d['emulated_sharding']['by_n_gpus'].items()}, 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
" >> $LOG 2>&1
grep -v "amdgpu.ids" $O/${T}_bench.err | tail -3 >> $LOG
cat $LOG
