#!/bin/bash
# Round 5, ninth GPU session: the inflate kernel with its two loads issued above the loop's bookkeeping and the block-header
# path (GD_INFLATE_HOIST 1, the product) against the round-4 order (0), the section cycles of both from the measurement
# builds (-DGD_INFLATE_TIMING), and the genome file -> BED with either library (the variant libraries of commit 573965c:
# GD_INFLATE_HOIST was a macro then).   tools/r12_session9.sh <tag>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r12j}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
LOG=$O/${T}.log; : > $LOG
V=$R/goleft_amd/variants
echo "== pytest inflate / BAM decode (product)" >> $LOG
timeout 600 python -m pytest tests/test_gpu_bamdecode.py tests/test_gpu_ref_fixtures.py -m gpu -x -q > $O/${T}_pytest.txt 2>&1; grep -h "passed\|failed" $O/${T}_pytest.txt | tail -2 >> $LOG
LEN=63025520,63025520
for v in product hoist0 time1 time0; do
  echo "== inflate_bench $v" >> $LOG
  if [ $v = product ]; then ( cd /tmp && timeout 600 python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate_$v.txt 2>&1 )
  else ( cd /tmp && GOLEFT_DEPTH_SO=$V/libgoleft_depth_$v.so INFLATE_BENCH_NO_ZLIB=1 timeout 600 python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate_$v.txt 2>&1 ); fi
  grep -h "lds pad\|kernel\|sections\|%" $O/${T}_inflate_$v.txt >> $LOG
done
echo "== the same on aux-tag records at deflate level 6 (product, hoist0)" >> $LOG
for v in product hoist0; do
  if [ $v = product ]; then ( cd /tmp && SYNTH_BAM_LEVEL=6 SYNTH_BAM_AUX=1 INFLATE_BENCH_NO_ZLIB=1 timeout 600 python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate6_$v.txt 2>&1 )
  else ( cd /tmp && GOLEFT_DEPTH_SO=$V/libgoleft_depth_$v.so SYNTH_BAM_LEVEL=6 SYNTH_BAM_AUX=1 INFLATE_BENCH_NO_ZLIB=1 timeout 600 python $R/tools/inflate_bench.py $LEN > $O/${T}_inflate6_$v.txt 2>&1 ); fi
  echo "  $v: $(grep -h 'kernel' $O/${T}_inflate6_$v.txt | tail -1)" >> $LOG
done
GENOME=$(python3 -c "import sys; sys.path.insert(0,'$R'); from goleft_amd import synth; print(','.join(str(x) for x in synth.HG19_LENGTHS))")
echo "== genome file -> BED: product / the round-4 load order" >> $LOG
timeout 900 python $R/tools/scope3.py --length $GENOME --name chrS --no-host --pause 8 --variants "LD_LIBRARY_PATH=$V/hoist0" > $O/${T}_scope3_genome.json 2>$O/${T}_scope3_genome.err
python3 -c "
import json; d=json.load(open('$O/${T}_scope3_genome.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'wall_s' in v:
        print('  %s %s wall %.3f s  %.3e ref-b/s  all %s' % (k, v.get('env'), v['wall_s'], v['ref_bases_per_s'], v.get('all_wall_s')))" >> $LOG 2>&1
tail -3 $O/${T}_scope3_genome.err >> $LOG
cat $LOG
