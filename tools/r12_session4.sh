#!/bin/bash
# Round 5, fourth GPU session: what the genome read's time is made of beside the inflate kernel -- no CRC at all, the record
# walks on the copy kernel's CUs, more / fewer inflate launches per range -- and a two-rank dry run of bench.py's default
# (native) collective with its fallback.   tools/r12_session4.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12d}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== CLI tests with the walks on the copy CUs" >> $LOG
GOLEFT_INGEST_WALK_CUS=1 timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_ref_fixtures.py tests/test_gpu_multidevice.py -m gpu -x -q > $O/${T}_pytest_walkcus.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest_walkcus.txt | tail -2 >> $LOG
GENOME=$(python3 -c "import sys; sys.path.insert(0,'$R'); from goleft_amd import synth; print(','.join(str(x) for x in synth.HG19_LENGTHS))")
echo "== genome file -> BED: default / no CRC / walks on copy CUs / 16, 4 launches per range / walks + 16" >> $LOG
python $R/tools/scope3.py --length $GENOME --name chrS --no-host --pause 8 --variants "GOLEFT_INGEST_WALK_CUS=1,GOLEFT_INGEST_BATCHES=16;GOLEFT_INGEST_BATCHES=4;GOLEFT_INGEST_BATCHES=16;GOLEFT_INGEST_WALK_CUS=1;GOLEFT_TRUST_BGZF=1" > $O/${T}_scope3_genome.json 2>$O/${T}_scope3_genome.err
python3 -c "
import json; d=json.load(open('$O/${T}_scope3_genome.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'wall_s' in v:
        print('  %s %s wall %.3f s  %.3e ref-b/s' % (k, v['env'], v['wall_s'], v['ref_bases_per_s'])); print('     ', {a: v['phases'][a] for a in sorted(v['phases']) if a.startswith('lib_') or a in ('setup_s','read_s','rows_s','decode_s','begin_s')})" >> $LOG 2>&1
tail -3 $O/${T}_scope3_genome.err >> $LOG
echo "== two ranks on the one device over gloo: the default collective (the library's) must fall back together" >> $LOG
( cd $R && GOLEFT_BENCH_SINGLE_DEVICE=1 GOLEFT_BENCH_BACKEND=gloo GOLEFT_BENCH_SKIP_COHORT=1 timeout 420 python bench.py --gpus 2 --workload chr20 --steps 5 --warmup 2 2>$O/${T}_n2.err | tail -1 > $O/${T}_bench_chr20_n2_gloo_dryrun_one_device.json )
python3 -c "
import json; d=json.load(open('$O/${T}_bench_chr20_n2_gloo_dryrun_one_device.json')); s=d.get('split') or {}
print('  n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'collective', s.get('collective'), 'fallback', s.get('collective_fallback_reason'), 'verified', s.get('collective_verified_against_torch_gather'), 'sum', s.get('gathered_sum_of_window_sums'))" >> $LOG 2>&1
tail -3 $O/${T}_n2.err >> $LOG
cat $LOG
