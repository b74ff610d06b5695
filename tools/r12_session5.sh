#!/bin/bash
# Round 5, fifth GPU session: HEAD after the prune -- the whole GPU suite, smoke, the two-rank dry run of bench.py's default
# collective (falls back together on one device), and the default bench line.   tools/r12_session5.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12e}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== pytest -m gpu (all) + smoke" >> $LOG
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -3 >> $LOG; grep -h "^FAILED\|^ERROR" $O/${T}_pytest.txt | head -20 >> $LOG
python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
echo "== two ranks on the one device over gloo (chr20: rank 1 holds no contig)" >> $LOG
( cd $R && GOLEFT_BENCH_SINGLE_DEVICE=1 GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SKIP_COHORT=1 timeout 420 python bench.py --gpus 2 --workload chr20 --steps 5 --warmup 2 2>$O/${T}_n2.err | tail -1 > $O/${T}_bench_chr20_n2_gloo_dryrun_one_device.json )
python3 -c "
import json; d=json.load(open('$O/${T}_bench_chr20_n2_gloo_dryrun_one_device.json')); s=d.get('split') or {}
print('  n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'collective', s.get('collective'), '| fallback:', s.get('collective_fallback_reason'), '| verified', s.get('collective_verified_against_torch_gather'), 'sum', s.get('gathered_sum_of_window_sums'))" >> $LOG 2>&1
grep -v "amdgpu.ids\|socket.cpp" $O/${T}_n2.err | tail -3 >> $LOG
echo "== two ranks, whole genome (every rank holds contigs)" >> $LOG
( cd $R && GOLEFT_BENCH_SINGLE_DEVICE=1 GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SKIP_COHORT=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 2>$O/${T}_n2w.err | tail -1 > $O/${T}_bench_wgs_n2_gloo_dryrun_one_device.json )
python3 -c "
import json; d=json.load(open('$O/${T}_bench_wgs_n2_gloo_dryrun_one_device.json')); s=d.get('split') or {}
print('  n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'collective', s.get('collective'), '| fallback:', s.get('collective_fallback_reason'), 'sum', s.get('gathered_sum_of_window_sums'), 'file_ngpu' , 'bam_file_scope_ngpu' in d)" >> $LOG 2>&1
grep -v "amdgpu.ids\|socket.cpp" $O/${T}_n2w.err | tail -3 >> $LOG
echo "== python bench.py (defaults)" >> $LOG
( cd $R && timeout 1200 python bench.py 2>$O/${T}_bench.err | tail -1 > $O/${T}_bench_wgs_n1.json )
python3 -c "
import json; d=json.load(open('$O/${T}_bench_wgs_n1.json'))
print('  step %.3f ms value %.3e frac %.3f first %.3f sum %s' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['first_compute']['ratio_to_warm'], d['sum_of_window_sums']))
b=d['bam_file_scope']; print('  bam_file_scope', {k: b.get(k) for k in ('file','wall_s','value','outputs_identical','oracle_identical','paper_invocation_s','error')}, (b.get('device_decoder') or {}).get('all_wall_s'))
for n, v in (b.get('variants') or {}).items(): print('  variant', n, {k: v.get(k) for k in ('device_wall_s','host_wall_s','outputs_identical','oracle_identical','error')})
for n, v in (d.get('other_workloads') or {}).items(): print('  other', n, {k: v.get(k) for k in ('ms_per_step','value','seconds_in_bench','error')}, (v.get('roofline') or {}).get('frac'), (v.get('first_compute') or {}).get('ratio_to_warm'), v.get('kernels_ms'))
print('  emu', {n: round(v['projected_speedup'],2) for n, v in d['emulated_sharding']['by_n_gpus'].items()}, 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for k in ('host_stream_scope','host_stream_scope_wgs'): print(' ', k, {n: round(v['host_to_device_GBps'],1) for n, v in d[k]['variants'].items()})
" >> $LOG 2>&1
grep -v "amdgpu.ids" $O/${T}_bench.err | tail -3 >> $LOG
cat $LOG
