#!/bin/bash
# Round 5, closing GPU session: the whole GPU suite, smoke and the default bench line at HEAD.   tools/r12_session11.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12n}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== pytest -m gpu (all) + smoke" >> $LOG
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -3 >> $LOG; grep -h "^FAILED\|^ERROR" $O/${T}_pytest.txt | head -20 >> $LOG
python -c "import __graft_entry__ as g; g.smoke()" >> $LOG 2>&1
echo "== python bench.py (defaults)" >> $LOG
( cd $R && timeout 1500 python bench.py > $O/${T}_bench.out 2>$O/${T}_bench.err ); tail -1 $O/${T}_bench.out > $O/${T}_bench_wgs_n1.json
python3 -c "
import json; d=json.load(open('$O/${T}_bench_wgs_n1.json'))
print('  step %.3f ms value %.3e frac %.3f traffic %s first %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['first_compute']['ratio_to_warm']))
b=d['bam_file_scope']; print('  bam_file_scope', {k: b.get(k) for k in ('file','wall_s','value','outputs_identical','oracle_identical','paper_invocation_s','error')}, (b.get('device_decoder') or {}).get('all_wall_s'))
for n, v in (b.get('variants') or {}).items(): print('  variant', n, {k: v.get(k) for k in ('device_wall_s','host_wall_s','outputs_identical','oracle_identical','error')})
for n, v in (d.get('other_workloads') or {}).items(): print('  other', n, {k: v.get(k) for k in ('ms_per_step','value','error')}, (v.get('roofline') or {}).get('frac'), (v.get('first_compute') or {}).get('ratio_to_warm'))
print('  emu', {n: round(v['projected_speedup'],2) for n, v in d['emulated_sharding']['by_n_gpus'].items()}, 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
" >> $LOG 2>&1
grep -v "amdgpu.ids" $O/${T}_bench.err | tail -3 >> $LOG
cat $LOG
