#!/bin/bash
# Round 5, GPU session 12: the buffer sets of the ranges to come allocated by a thread while the first range is read.
#   tools/r12_session12.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12o}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== pytest: everything that reads BAM files on the device" >> $LOG
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_ref_fixtures.py tests/test_gpu_multidevice.py tests/test_gpu_bamdecode.py tests/test_gpu_shim.py tests/test_gpu_soak.py -m gpu -x -q > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -2 >> $LOG; grep -h "^FAILED\|^ERROR" $O/${T}_pytest.txt | head >> $LOG
GENOME=$(python3 -c "import sys; sys.path.insert(0,'$R'); from goleft_amd import synth; print(','.join(str(x) for x in synth.HG19_LENGTHS))")
echo "== genome file -> BED, three runs 8 s apart; the same with 1 GB parts" >> $LOG
timeout 900 python $R/tools/scope3.py --length $GENOME --name chrS --no-host --pause 8 --variants "GOLEFT_INGEST_PART_MB=1024" > $O/${T}_scope3_genome.json 2>$O/${T}_scope3_genome.err
python3 -c "
import json; d=json.load(open('$O/${T}_scope3_genome.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'wall_s' in v:
        print('  %s wall %.3f s  %.3e ref-b/s' % (k, v['wall_s'], v['ref_bases_per_s']))
        for r in v.get('all_runs') or []: print('     run', {a: round(b, 3) for a, b in r.items() if isinstance(b, float)})
        print('     ', {a: round(v['phases'][a], 3) for a in sorted(v['phases']) if a.startswith('lib_') or a in ('setup_s','read_s','rows_s','decode_s','begin_s')})" >> $LOG 2>&1
tail -3 $O/${T}_scope3_genome.err >> $LOG
cat $LOG
