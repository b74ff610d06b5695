#!/bin/bash
# Round 5, GPU session 13: where gd_ingest_begin's time goes (a measurement build with marks, -DGD_INGEST_BEGIN_TIMING).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12p}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
GENOME=$(python3 -c "import sys; sys.path.insert(0,'$R'); from goleft_amd import synth; print(','.join(str(x) for x in synth.HG19_LENGTHS))")
SCOPE3_KEEP_STDERR=$O/${T}_stderr.txt timeout 900 python $R/tools/scope3.py --length $GENOME --name chrS --no-host --pause 8 --variants "LD_LIBRARY_PATH=$R/goleft_amd/variants/begintime" > $O/${T}_scope3_genome.json 2>$O/${T}_scope3_genome.err
python3 -c "
import json; d=json.load(open('$O/${T}_scope3_genome.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'wall_s' in v:
        print('  %s wall %.3f s' % (k, v['wall_s']))
        for r in v.get('all_runs') or []: print('     run', {a: round(b, 3) for a, b in r.items() if isinstance(b, float)})" >> $LOG 2>&1
grep "^\[begin\|^== " $O/${T}_stderr.txt | awk '/^== /{run=$0; n=0} /^\[begin/{ if (run ~ /variants\/begintime/ && run ~ /run 2/) print }' | head -150 >> $LOG
tail -3 $O/${T}_scope3_genome.err >> $LOG
cat $LOG
