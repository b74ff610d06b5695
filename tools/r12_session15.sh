#!/bin/bash
# Round 5, GPU session 15: four ranges in flight instead of three (GOLEFT_INGEST_DEPTH=4 on the build of commit b7eb1c8, whose
# library held four pending ranges; the library is back at three: the ABI refuses a fourth gd_ingest_begin).   tools/r12_session15.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12u}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
echo "== pytest: BAM files on the device, three and four ranges in flight" >> $LOG
timeout 300 python -m pytest tests/test_gpu_cli.py tests/test_gpu_ref_fixtures.py tests/test_gpu_multidevice.py -m gpu -x -q > $O/${T}_pytest3.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest3.txt | tail -1 >> $LOG
GOLEFT_INGEST_DEPTH=4 GOLEFT_INGEST_PART_KB=64 timeout 300 python -m pytest tests/test_gpu_cli.py tests/test_gpu_ref_fixtures.py tests/test_gpu_multidevice.py -m gpu -x -q > $O/${T}_pytest4.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest4.txt | tail -1 >> $LOG
GENOME=$(python3 -c "import sys; sys.path.insert(0,'$R'); from goleft_amd import synth; print(','.join(str(x) for x in synth.HG19_LENGTHS))")
echo "== genome file -> BED: three ranges in flight / four" >> $LOG
timeout 900 python $R/tools/scope3.py --length $GENOME --name chrS --no-host --pause 8 --variants "GOLEFT_INGEST_DEPTH=4" > $O/${T}_scope3_genome.json 2>$O/${T}_scope3_genome.err
python3 -c "
import json; d=json.load(open('$O/${T}_scope3_genome.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'wall_s' in v:
        print('  %s %s wall %.3f s  %.3e ref-b/s' % (k, v.get('env'), v['wall_s'], v['ref_bases_per_s']))
        for r in v.get('all_runs') or []: print('     run', {a: round(b, 3) for a, b in r.items() if isinstance(b, float)})
        print('     ', {a: round(v['phases'][a], 3) for a in sorted(v['phases']) if a.startswith('lib_') or a in ('setup_s','read_s','rows_s','decode_s','begin_s')})" >> $LOG 2>&1
tail -3 $O/${T}_scope3_genome.err >> $LOG
cat $LOG
